#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05v}
rm -f gpurun_out/${t}_dp_prefetch.log
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 60 --warmup 15 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
dp = d.get('data_parallel') or {}
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| host', round(d.get('host_enqueue_ms_per_step') or 0, 2), '| queues', d.get('runtime_env'), '| exposed', dp.get('per_rank_exposed_wait_ms'), '| small', (dp.get('small_collectives') or {}).get('exposed_ms_per_step'))" >> gpurun_out/${t}_dp_prefetch.log 2>&1
}
run X=0 --
run X=0 -- --force-process-group --no-shard
run GPU_MAX_HW_QUEUES=8 -- --force-process-group --no-shard
run GPU_MAX_HW_QUEUES=6 -- --force-process-group --no-shard
run X=0 -- --force-process-group
run GPU_MAX_HW_QUEUES=8 -- --force-process-group
run GPU_MAX_HW_QUEUES=8 --
run GPU_MAX_HW_QUEUES=8 -- --config STEP_PEMS07 --force-process-group
cat gpurun_out/${t}_dp_prefetch.log
