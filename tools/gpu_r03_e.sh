#!/bin/bash
# round 3, GPU call E: GEMM heuristics A/B (split-K target of the auto-split GEMMs, tall-M tile threshold), C3 after the vectorised bf16 column sums
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03e
ab() { # name, config, steps, env...
  name=$1; cfg=$2; steps=$3; shift 3
  for rep in 1 2; do
    for env in "$@"; do
      env $env timeout 300 python bench.py --config $cfg --no-extras --no-cpu-baseline --pretrain-steps 0 --steps $steps --warmup 8 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$name', '$env', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
    done
  done
}
ab C2 STEP_PEMS04 60 "X=1" "STEP_GEMM_SPLIT_TARGET=384" "STEP_GEMM_SPLIT_TARGET=256" "STEP_GEMM_SPLIT_TARGET=1536" > gpurun_out/${tag}_ab_C2_split.log 2>&1
ab C4 STEP_PEMS07 40 "X=1" "STEP_GEMM_TALL_M=512" "STEP_GEMM_SPLIT_TARGET=384" > gpurun_out/${tag}_ab_C4.log 2>&1
timeout 300 python -m pytest tests/test_gpu_pretrain.py -q -k "ffn" > gpurun_out/${tag}_tests.log 2>&1
timeout 300 python bench.py --config TSFormer_PEMS-BAY --steps 30 --warmup 8 > gpurun_out/${tag}_bench_C3.json 2> gpurun_out/${tag}_bench_C3.err
cat gpurun_out/${tag}_ab_C2_split.log gpurun_out/${tag}_ab_C4.log; tail -2 gpurun_out/${tag}_tests.log; head -c 300 gpurun_out/${tag}_bench_C3.json
