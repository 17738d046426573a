#!/bin/bash
# is the chain host-bound?  batch sweep at PEMS04 (the host's work per step does not depend on the batch)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05o}
rm -f gpurun_out/${t}_batch_sweep.log
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 50 --warmup 12 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| host', round(d.get('host_enqueue_ms_per_step') or 0, 2), '| queues', d.get('runtime_env'))" >> gpurun_out/${t}_batch_sweep.log 2>&1
}
for b in 8 4 2 1; do run X=0 -- --batch $b; done
for b in 4 2; do run X=0 -- --batch $b --no-prefetch; done
run X=0 -- --batch 8 --resident-batches
cat gpurun_out/${t}_batch_sweep.log
