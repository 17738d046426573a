#!/bin/bash
# round 3, call V: the 128 x 384 weight-gradient tile -- test, C3 A/B (STEP_GEMM_WIDE_WGRAD=0/1), C3 parity tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03v
timeout 900 python -m pytest tests/test_gpu_pretrain.py -q -rP -m gpu -k "wide or full_size or bf16_vs or runs" > gpurun_out/${tag}_tests.log 2>&1; echo "tests exit $?"
grep -E "passed|failed|wide weight|C3 full" gpurun_out/${tag}_tests.log | tail -8
for rep in 1 2; do for w in 0 1; do
STEP_GEMM_WIDE_WGRAD=$w timeout 400 python bench.py --config TSFormer_PEMS-BAY --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 STEP_GEMM_WIDE_WGRAD=$w', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done; done > gpurun_out/${tag}_wide_wgrad_ab_C3.log 2>&1
cat gpurun_out/${tag}_wide_wgrad_ab_C3.log
