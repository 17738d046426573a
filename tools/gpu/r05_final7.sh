#!/bin/bash
# round 5: live HBM counters of the encoder launch at the PEMS07 and 4096-node sizes (bench.py's PMC child, two rocprofv3 passes each)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r05zt}
timeout 110 python bench.py --config STEP_PEMS07 --no-extras --no-cpu-baseline --steps 30 --warmup 8 2> /dev/null | grep '^{"metric"' > gpurun_out/${t}_bench_C4.json
timeout 110 python bench.py --config SYNTH_4096 --no-extras --no-cpu-baseline --steps 12 --warmup 4 2> /dev/null | grep '^{"metric"' > gpurun_out/${t}_bench_C5.json
python - <<'PY'
import json
for c in ("C4", "C5"):
    try:
        d = json.loads(open(f"gpurun_out/r05zt_bench_{c}.json").read())
        print(c, round(d["value"], 1), round(d["ms_per_step"], 3), d["roofline"]["traffic"], d["roofline"]["traffic_detail"])
    except Exception as e:
        print(c, "ERR", e)
PY
