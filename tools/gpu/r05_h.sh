#!/bin/bash
# the data-parallel code path on ONE rank: no process group / torch.distributed / RCCL C API, whole graph learner and time slices
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05h}
rm -f gpurun_out/${t}_dp_one_rank.log
i=0
for mode in "" "--force-process-group --collectives torch --no-shard" "--force-process-group --collectives rccl --no-shard" "--force-process-group --collectives torch" "--force-process-group --collectives rccl" "" ; do
  i=$((i+1))
  timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 60 --warmup 15 $mode > gpurun_out/${t}_dp_$i.out 2> gpurun_out/${t}_dp_$i.err
  grep '^{"metric"' gpurun_out/${t}_dp_$i.out | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
dp = d.get('data_parallel') or {}
print('[$mode]', '| ms_per_step', round(d['ms_per_step'], 3), '| host enqueue', round(d.get('host_enqueue_ms_per_step') or 0, 2), '| queues', d.get('runtime_env'), '| collectives', dp.get('collectives'), '| exposed', dp.get('per_rank_exposed_wait_ms'), '| small', dp.get('small_collectives'))" >> gpurun_out/${t}_dp_one_rank.log 2>&1
done
cat gpurun_out/${t}_dp_one_rank.log; tail -3 gpurun_out/${t}_dp_2.err
