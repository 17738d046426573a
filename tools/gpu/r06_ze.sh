#!/bin/bash
# round 6: resident waves per SIMD of every kernel of the C3 and C2 steps (SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8 x 1024))
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
scan() { name=$1; shift
  rm -rf gpurun_out/pmc_scan
  (cd /tmp && env "$@" > /dev/null 2>&1)
}
for cfg in "C3 --config TSFormer_PEMS-BAY --steps 6 --warmup 2" "C2 --no-prefetch --steps 6 --warmup 2" "C4 --no-prefetch --config STEP_PEMS07 --steps 6 --warmup 2"; do
  set -- $cfg; name=$1; shift
  rm -rf gpurun_out/pmc_scan
  (cd /tmp && STEP_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/pmc_scan -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 "$@" > /dev/null 2>&1)
  python - $name <<'PY' > gpurun_out/r06_ze_occupancy_scan_$name.txt
import sqlite3, glob, sys
for db in glob.glob('gpurun_out/pmc_scan/**/*.db', recursive=True):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("""select s.kernel_name, p.name, sum(e.value), count(distinct d.id), sum(d.end-d.start) from rocpd_pmc_event e
       join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
       join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name""").fetchall()
    ks = {}
    for k, pn, v, n, t in rows:
        ks.setdefault(k, {})[pn] = v; ks[k]['n'] = n; ks[k]['t'] = t
    out = []
    for k, v in ks.items():
        if 'SQ_WAVE_CYCLES' not in v or not v.get('GRBM_GUI_ACTIVE'): continue
        occ = v['SQ_WAVE_CYCLES'] * 4 / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)
        out.append((v['t'] / 4, k, v['n'], occ, v['SQ_WAVES'] / v['n']))      # four counters -> the dispatch rows repeat per counter
    tot = sum(o[0] for o in out)
    print(f"# config {sys.argv[1]}: kernel, dispatches, total us, share, resident waves per SIMD while it runs, waves per dispatch")
    for t, k, n, occ, w in sorted(out, reverse=True)[:40]:
        print(f"{k[:110]:110s} {n:5d} {t / 1000:9.0f} us {100 * t / tot:5.1f} %  {occ:5.2f} waves/SIMD  {w:9.0f}")
PY
done
head -32 gpurun_out/r06_ze_occupancy_scan_C3.txt; head -30 gpurun_out/r06_ze_occupancy_scan_C2.txt
