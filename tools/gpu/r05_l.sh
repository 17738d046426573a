#!/bin/bash
# sweep of the encoder's compute-unit share with the next batch's frozen branch prefetched
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05l}
rm -f gpurun_out/${t}_sweep.log
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 50 --warmup 12 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| host', round(d.get('host_enqueue_ms_per_step') or 0, 2), '| queues', d.get('runtime_env'))" >> gpurun_out/${t}_sweep.log 2>&1
}
run X=0 --
for n in 136 144 152 160 168 176; do run GPU_MAX_HW_QUEUES=4 -- --prefetch --encoder-workgroups $n; done
run GPU_MAX_HW_QUEUES=8 -- --prefetch --encoder-workgroups 152
run GPU_MAX_HW_QUEUES=3 -- --prefetch --encoder-workgroups 152
run X=0 -- --config STEP_PEMS07
for n in 224 288 352 416; do run GPU_MAX_HW_QUEUES=4 -- --config STEP_PEMS07 --prefetch --encoder-workgroups $n; done
run X=0 -- --config SYNTH_4096 --steps 20 --warmup 5
for n in 224 320 416; do run GPU_MAX_HW_QUEUES=4 -- --config SYNTH_4096 --steps 20 --warmup 5 --prefetch --encoder-workgroups $n; done
run X=0 -- --config STEP_METR-LA
for n in 128 256; do run GPU_MAX_HW_QUEUES=4 -- --config STEP_METR-LA --prefetch --encoder-workgroups $n; done
cat gpurun_out/${t}_sweep.log
