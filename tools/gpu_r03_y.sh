#!/bin/bash
# round 3, call Y: fc-forward split-K workspace in the graph learner, bias gradients out of the LayerNorm backward: all GPU tests, A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03y
timeout 1800 python -m pytest tests -q -m gpu -rP > gpurun_out/${tag}_gpu_tests_full.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/${tag}_gpu_tests_full.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/${tag}_gpu_tests_full.log | head
ab() { # name config steps
for rep in 1 2; do for w in 0 1; do
STEP_GEMM_SPLITK_WS=$w timeout 400 python bench.py --config $2 --steps $3 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 STEP_GEMM_SPLITK_WS=$w', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done; done
}
{ ab C2 STEP_PEMS04 60; ab C4 STEP_PEMS07 40; ab C5 SYNTH_4096 20; ab C3 TSFormer_PEMS-BAY 20; } > gpurun_out/${tag}_splitk_ws_ab.log 2>&1
cat gpurun_out/${tag}_splitk_ws_ab.log
