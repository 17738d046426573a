#!/bin/bash
# round 3, GPU call G: stream priorities A/B (side chain high, backward leaves low), print the device's priority range
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03g
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" > gpurun_out/${tag}_prio_range.log 2>&1
ab() { # name, config, steps, env...
  name=$1; cfg=$2; steps=$3; shift 3
  for rep in 1 2; do
    for env in "$@"; do
      env $env timeout 300 python bench.py --config $cfg --no-extras --no-cpu-baseline --pretrain-steps 0 --steps $steps --warmup 8 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$name', '$env', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1), 'enc', round(d['roofline']['ms_per_launch'], 3))"
    done
  done
}
ab C2 STEP_PEMS04 60 "STEP_STREAM_PRIO=0" "STEP_STREAM_PRIO=1" "STEP_STREAM_PRIO=side" "STEP_STREAM_PRIO=aux" > gpurun_out/${tag}_ab_C2.log 2>&1
ab C4 STEP_PEMS07 40 "STEP_STREAM_PRIO=0" "STEP_STREAM_PRIO=1" > gpurun_out/${tag}_ab_C4.log 2>&1
timeout 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_sharded_graph_learner.py -q > gpurun_out/${tag}_tests.log 2>&1
cat gpurun_out/${tag}_prio_range.log gpurun_out/${tag}_ab_C2.log gpurun_out/${tag}_ab_C4.log; tail -2 gpurun_out/${tag}_tests.log
