#!/bin/bash
# round 6: second version of the pre-training attention kernels: parity tests, then A/B against the first version
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_pretrain.py -x -q -k "attention or matrix_core" 2>&1 | tail -5 > gpurun_out/r06_u_tests.log
L=gpurun_out/r06_u_attention_ab.log; : > $L
STEP_PT_ATTN_V1=1 python tools/bench_pt_attention.py 2>&1 | sed 's/^/v1 /' >> $L
python tools/bench_pt_attention.py 2>&1 | sed 's/^/v2 /' >> $L
cat gpurun_out/r06_u_tests.log $L
