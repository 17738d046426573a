#!/bin/bash
# round 5: LIVE HBM counters of the encoder launch at the PEMS07 size (extras on: the PMC child under rocprofv3, two passes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r05zs}
timeout 150 python bench.py --config STEP_PEMS07 --no-cpu-baseline --other-configs - --steps 20 --warmup 5 2> gpurun_out/${t}_C4.err | grep '^{"metric"' > gpurun_out/${t}_bench_C4.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05zs_bench_C4.json").read())
print(round(d["value"], 1), round(d["ms_per_step"], 3), d["roofline"]["traffic"], d["roofline"]["traffic_detail"], d["roofline"].get("ms_per_launch_alone"))
PY
