#!/bin/bash
# round 3, call T: bf16 activations in the pre-training attention block, inverse scaling inside the native loss, host-side caches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03t
timeout 900 python -m pytest tests/test_gpu_pretrain.py tests/test_gpu_step.py -q -rP -m gpu > gpurun_out/${tag}_tests.log 2>&1; echo "tests exit $?"
grep -E "passed|failed|error" gpurun_out/${tag}_tests.log | tail -3
grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/${tag}_tests.log | head -20
timeout 200 python tools/bench_pt_attention.py 2>/dev/null > gpurun_out/${tag}_attention.log; cat gpurun_out/${tag}_attention.log
timeout 400 python bench.py --config TSFormer_PEMS-BAY --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_c3_bench.json; cut -c1-400 gpurun_out/${tag}_c3_bench.json
for cfg in STEP_PEMS04 STEP_METR-LA; do
timeout 300 python bench.py --config $cfg --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms_per_step', round(d['ms_per_step'], 3), 'host_enqueue_ms_per_step', round(d['host_enqueue_ms_per_step'], 3))"
done > gpurun_out/${tag}_host_enqueue.log 2>&1
cat gpurun_out/${tag}_host_enqueue.log
