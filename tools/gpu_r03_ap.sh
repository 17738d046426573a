#!/bin/bash
# round 3, call AP: m-fast block order of the d_a2 GEMM (the row tiles of one column tile adjacent): tests, same-box A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03ap
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_dgl_conv.py tests/test_gpu_step.py -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/${tag}_tests.log | tail -2
ab() { # name config steps
for rep in 1 2; do for w in 0 1; do
STEP_GEMM_M_FAST=$w timeout 400 python bench.py --config $2 --steps $3 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 STEP_GEMM_M_FAST=$w', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done; done
}
{ ab C2 STEP_PEMS04 60; ab C4 STEP_PEMS07 40; ab C5 SYNTH_4096 20; } > gpurun_out/${tag}_m_fast_ab.log 2>&1
cat gpurun_out/${tag}_m_fast_ab.log
