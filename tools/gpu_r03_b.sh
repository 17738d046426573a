#!/bin/bash
# round 3, GPU call B: full GPU test suite (no -x), bench with / without the auxiliary-stream leaves
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=10 -k "not dropout_on" > gpurun_out/r03b_gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03b_gpu_tests.log
for rep in 1 2; do
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 60 > gpurun_out/r03b_bench_aux_$rep.json 2> gpurun_out/r03b_bench_aux.err
  STEP_NO_AUX=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 60 > gpurun_out/r03b_bench_noaux_$rep.json 2> gpurun_out/r03b_bench_noaux.err
done
tail -3 gpurun_out/r03b_gpu_tests.log
python - <<'PY'
import json
for n in ("aux_1", "noaux_1", "aux_2", "noaux_2"):
    try:
        d = json.load(open(f"gpurun_out/r03b_bench_{n}.json")); print(n, round(d["ms_per_step"], 3), round(d["value"], 1), d["roofline"]["ms_per_launch"])
    except Exception as e: print(n, "failed", e)
PY
