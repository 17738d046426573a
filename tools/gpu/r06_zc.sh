#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_pretrain.py -x -q 2>&1 | tail -3 > gpurun_out/r06_zc_tests.log
cat gpurun_out/r06_zc_tests.log
L=gpurun_out/r06_zc_rows_linear_waves.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc --config TSFormer_PEMS-BAY"
for rep in 1 2; do
for v in 8 12; do
STEP_PT_LIN_WAVES=$v python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('C3 rows_linear waves $v',round(d['value'],1),round(d['ms_per_step'],4))" >> $L
done; done
cat $L
