#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05y}
rm -f gpurun_out/${t}_c4c5.log
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 40 --warmup 10 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| host', round(d.get('host_enqueue_ms_per_step') or 0, 2), '| queues', d.get('runtime_env'))" >> gpurun_out/${t}_c4c5.log 2>&1
}
for n in 416 448 480 512; do run X=0 -- --config STEP_PEMS07 --encoder-workgroups $n; done
run X=0 -- --config STEP_PEMS07 --encoder-workgroups 0
run X=0 -- --config SYNTH_4096 --steps 20 --warmup 5
run GPU_MAX_HW_QUEUES=4 -- --config SYNTH_4096 --steps 20 --warmup 5 --prefetch --encoder-workgroups 0
run GPU_MAX_HW_QUEUES=4 -- --config SYNTH_4096 --steps 20 --warmup 5 --prefetch --prefetch-early --encoder-workgroups 256
run GPU_MAX_HW_QUEUES=4 -- --config SYNTH_4096 --steps 20 --warmup 5 --prefetch --prefetch-early --encoder-workgroups 128
cat gpurun_out/${t}_c4c5.log
