"""Timeline of ONE training step from a rocprofv3 rocpd database (--kernel-trace): every dispatch between the starts of two
consecutive dispatches of the anchor kernel (default: the TSFormer encoder), in start order, with its queue, start offset, duration
and the idle gap since the previous dispatch END on the same queue; then, per queue, busy time and the union of busy intervals.
usage: python tools/prof_timeline.py <results.db> [--anchor tsformer_encoder] [--step -2] > profiles/xyz_timeline.md"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")
    return name[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "stream_id" if "stream_id" in cols else "queue_id"
    rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.{qcol}, d.queue_id from rocpd_kernel_dispatch d "
                       "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "tsformer_encoder"
    which = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -2
    starts = [r[1] for r in rows if anchor in r[0]]
    t0, t1 = starts[which], starts[which + 1] if which + 1 != 0 else rows[-1][2]
    step = [r for r in rows if t0 <= r[1] < t1]
    print(f"step of {(t1 - t0) / 1e3:.1f} us, {len(step)} dispatches ({qcol} / queue_id shown)\n")
    print("| start us | dur us | gap us | stream | kernel |")
    print("|---|---|---|---|---|")
    last_end = {}
    busy = {}
    for name, st, en, q, qq in step:
        gap = (st - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = en
        busy.setdefault(q, []).append((st, en))
        print(f"| {(st - t0) / 1e3:.1f} | {(en - st) / 1e3:.1f} | {gap:.1f} | {q}/{qq} | {short(name)} |")
    print()
    allint = sorted(i for v in busy.values() for i in v)
    union, cs, ce = 0, None, None
    for st, en in allint:
        if cs is None or st > ce:
            if cs is not None:
                union += ce - cs
            cs, ce = st, en
        else:
            ce = max(ce, en)
    union += ce - cs
    for q, v in busy.items():
        print(f"stream {q}: {len(v)} dispatches, busy {sum(e - s for s, e in v) / 1e3:.1f} us, first start {(v[0][0] - t0) / 1e3:.1f}, last end {(v[-1][1] - t0) / 1e3:.1f}")
    print(f"union of busy intervals {union / 1e3:.1f} us of {(t1 - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
