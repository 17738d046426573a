#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05j}
rm -f gpurun_out/${t}_dp_one_rank.log
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 60 --warmup 15 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
dp = d.get('data_parallel') or {}
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| queues', d.get('runtime_env'), '| exposed', dp.get('per_rank_exposed_wait_ms'), '| small', (dp.get('small_collectives') or {}).get('exposed_ms_per_step'))" >> gpurun_out/${t}_dp_one_rank.log 2>&1
}
run X=0 --
run STEP_STREAM_PROBE=0 --
run GPU_MAX_HW_QUEUES=4 --
run X=0 -- --force-process-group --collectives rccl --no-shard
run GPU_MAX_HW_QUEUES=2 -- --force-process-group --collectives rccl --no-shard
run STEP_STREAM_PROBE=0 -- --force-process-group --collectives rccl --no-shard
run X=0 -- --force-process-group --collectives torch --no-shard
run X=0 -- --force-process-group --collectives rccl
run GPU_MAX_HW_QUEUES=2 -- --force-process-group --collectives rccl
run X=0 -- --force-process-group --collectives torch
run X=0 -- --config STEP_PEMS07
run X=0 -- --config STEP_PEMS07 --force-process-group --collectives rccl
cat gpurun_out/${t}_dp_one_rank.log
