#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05s}
rm -f gpurun_out/${t}_prio.log
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())" >> gpurun_out/${t}_prio.log 2>&1
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 50 --warmup 12 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| host', round(d.get('host_enqueue_ms_per_step') or 0, 2), '| queues', d.get('runtime_env'))" >> gpurun_out/${t}_prio.log 2>&1
}
run X=0 --
run STEP_PRIORITY_AUX=1 --
run STEP_PRIORITY_AUX=1 STEP_PRIORITY_PREFETCH=1 --
run STEP_PRIORITY_SIDE=-1 --
run STEP_PRIORITY_AUX=1 STEP_PRIORITY_SIDE=-1 --
run STEP_PRIORITY_AUX=1 -- --encoder-workgroups 168
run STEP_PRIORITY_AUX=1 -- --config STEP_PEMS07
cat gpurun_out/${t}_prio.log
