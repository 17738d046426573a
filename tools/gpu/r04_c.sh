#!/bin/bash
# round 4, GPU call C: slice-resident gcn layer kernels (gwnet_slice.h): the GPU suite, then same-box A/B of the four-kernel path
# (STEP_GCN_SLICE=0) against the slice path at C2 and C1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=r04c
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 -k "not c5_4096 and not benchmark_checkpoint" > gpurun_out/${t}_gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests.log
b() { # name, env, args
  name=$1; envs=$2; shift 2
  env $envs timeout 400 python bench.py --no-extras --no-cpu-baseline --no-pmc --steps 60 --warmup 15 "$@" > gpurun_out/${t}_bench_$name.json 2> gpurun_out/${t}_bench_$name.err
}
b C2_old STEP_GCN_SLICE=0
b C2_slice STEP_GCN_SLICE=1
b C2_old2 STEP_GCN_SLICE=0
b C2_slice2 STEP_GCN_SLICE=1
b C1_old STEP_GCN_SLICE=0 --config STEP_METR-LA
b C1_slice STEP_GCN_SLICE=1 --config STEP_METR-LA
for f in gpurun_out/${t}_bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d=json.loads(line)
        print(round(d["value"],1), round(d["ms_per_step"],3), "enc", round(d["roofline"]["ms_per_launch"],3), "host", round(d["host_enqueue_ms_per_step"],2), "loss", d["config"]["final_loss"])
PY
done
tail -5 gpurun_out/${t}_gpu_tests.log
