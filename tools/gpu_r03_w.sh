#!/bin/bash
# round 3, call W: split-K depth of the pre-training weight-gradient GEMMs (are the atomics the cost?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03w
for t in 96 192 384 768 1536; do
STEP_GEMM_WIDE_WGRAD=0 STEP_GEMM_SPLIT_TARGET=$t timeout 400 python bench.py --config TSFormer_PEMS-BAY --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 STEP_GEMM_SPLIT_TARGET=$t', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done > gpurun_out/${tag}_split_target_ab_C3.log 2>&1
cat gpurun_out/${tag}_split_target_ab_C3.log
