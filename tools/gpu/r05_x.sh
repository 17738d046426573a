#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05x}
rm -f gpurun_out/${t}_mainprio.log
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 50 --warmup 12 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| host', round(d.get('host_enqueue_ms_per_step') or 0, 2), '| queues', d.get('runtime_env'))" >> gpurun_out/${t}_mainprio.log 2>&1
}
run X=0 --
run BENCH_MAIN_PRIORITY=-1 --
run BENCH_MAIN_PRIORITY=-1 STEP_PRIORITY_SIDE=-1 --
run BENCH_MAIN_PRIORITY=-1 STEP_PRIORITY_SIDE=-1 -- --config STEP_PEMS07
run X=0 -- --config STEP_PEMS07
run STEP_NO_AUX=1 --
cat gpurun_out/${t}_mainprio.log; tail -3 gpurun_out/${t}_last.err
