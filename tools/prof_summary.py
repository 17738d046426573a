"""Summarise a rocprofv3 rocpd database (--kernel-trace) into a per-kernel table (markdown), the same
content as `rocprofv3 --stats` kernel_stats: calls, total / average / min / max duration, share.
usage: python tools/prof_summary.py <results.db> [--after-last <kernel name substring>] > profiles/xyz.md
--after-last X: only the dispatches that start after the LAST dispatch of a kernel whose name contains X (bench.py pre-trains the
TSFormer checkpoint before its timed region: `--after-last attn_mfma_bwd` leaves the training steps only)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select s.kernel_name, d.start, d.end, d.grid_size_x*d.grid_size_y*d.grid_size_z, "
                       "d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z, s.arch_vgpr_count, s.accum_vgpr_count, "
                       "d.group_segment_size, d.private_segment_size "
                       "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id").fetchall()
    if "--after-last" in sys.argv:
        pat = sys.argv[sys.argv.index("--after-last") + 1]
        cut = max((r[2] for r in rows if pat in r[0]), default=0)
        rows = [r for r in rows if r[1] > cut]
        print(f"(dispatches after the last `{pat}` kernel only)\n")
    agg = {}
    for name, st, en, grid, wg, vg, ag, lds, scr in rows:
        a = agg.setdefault(short(name), {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "vgpr": vg, "agpr": ag, "lds": lds, "scr": scr, "wg": wg})
        dur = en - st
        a["n"] += 1
        a["tot"] += dur
        a["min"] = min(a["min"], dur)
        a["max"] = max(a["max"], dur)
    total = sum(a["tot"] for a in agg.values())
    t0 = min(r[1] for r in rows)
    t1 = max(r[2] for r in rows)
    print(f"kernel dispatches: {len(rows)}   sum of kernel time: {total / 1e6:.2f} ms   trace span: {(t1 - t0) / 1e6:.2f} ms\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | wg | vgpr | agpr | lds B | scratch B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        print(f"| {k} | {a['n']} | {a['tot'] / 1e6:.3f} | {a['tot'] / a['n'] / 1e3:.1f} | {a['min'] / 1e3:.1f} | {a['max'] / 1e3:.1f} | "
              f"{100.0 * a['tot'] / total:.1f} | {a['wg']} | {a['vgpr']} | {a['agpr']} | {a['lds']} | {a['scr']} |")


if __name__ == "__main__":
    main()
