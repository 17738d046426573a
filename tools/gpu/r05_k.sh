#!/bin/bash
# spatial split: a persistent encoder on N compute units, the frozen branch of the NEXT batch prefetched next to this batch's backward
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05k}
rm -f gpurun_out/${t}_persist_prefetch.log
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 60 --warmup 15 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| queues', d.get('runtime_env'))" >> gpurun_out/${t}_persist_prefetch.log 2>&1
}
run X=0 --
run X=0 -- --prefetch
run GPU_MAX_HW_QUEUES=4 -- --prefetch
for n in 128 160 192 224; do
  run STEP_HIP_LIB=step_amd/libstep_hip_persist$n.so -- --prefetch
  run STEP_HIP_LIB=step_amd/libstep_hip_persist$n.so GPU_MAX_HW_QUEUES=4 -- --prefetch
done
run STEP_HIP_LIB=step_amd/libstep_hip_persist192.so --
cat gpurun_out/${t}_persist_prefetch.log
