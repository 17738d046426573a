#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=r05g
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_graphed_step.py tests/test_gpu_reference_runner_live.py tests/test_gpu_pretrain.py -m gpu -q -rP > gpurun_out/${t}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_tests.log
# data-parallel path on one rank: plain, torch.distributed, RCCL C API (whole graph learner and time slices)
for mode in "" "--force-process-group --collectives torch --no-shard" "--force-process-group --collectives rccl --no-shard" "--force-process-group --collectives torch" "--force-process-group --collectives rccl"; do
  timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 60 --warmup 15 $mode 2>gpurun_out/${t}_err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$mode', '| ms_per_step', round(d['ms_per_step'], 3), '| host enqueue', d.get('host_enqueue_ms_per_step'), '| dp', json.dumps(d.get('data_parallel', {}))[:600])" >> gpurun_out/${t}_dp_one_rank.log 2>&1
done
tail -4 gpurun_out/${t}_tests.log; cat gpurun_out/${t}_dp_one_rank.log; tail -5 gpurun_out/${t}_err.log
