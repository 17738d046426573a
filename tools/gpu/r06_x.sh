#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_pretrain.py -x -q 2>&1 | tail -5 > gpurun_out/r06_x_tests.log
L=gpurun_out/r06_x_attention_ab.log; : > $L
STEP_PT_ATTN_V1=1 python tools/bench_pt_attention.py 2>&1 | grep -v amdgpu.ids | sed 's/^/v1 /' >> $L
python tools/bench_pt_attention.py 2>&1 | grep -v amdgpu.ids | sed 's/^/v2 /' >> $L
F="--no-extras --no-cpu-baseline --no-pmc --config TSFormer_PEMS-BAY"
for rep in 1 2; do
for v in 1 0; do
STEP_PT_ATTN_V1=$v python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('C3 v1=$v',round(d['value'],1),round(d['ms_per_step'],4))" >> $L
done; done
cat gpurun_out/r06_x_tests.log $L
