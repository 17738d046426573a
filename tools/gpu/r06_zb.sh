#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pretrain.py -x -q -k "two_sequences or pool_drawn" 2>&1 | tail -5 > gpurun_out/r06_zb_tests.log
cat gpurun_out/r06_zb_tests.log
L=gpurun_out/r06_zb_C4.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc --config STEP_PEMS07"
run() { name=$1; shift; env "$@" python bench.py $F ${EXTRA} 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4),round(d['roofline']['ms_per_launch'],3))" >> $L; }
for rep in 1 2; do
EXTRA="--encoder-workgroups 416" run "C4 one sequence per workgroup, n=416" STEP_ENC_NSEQ=1
EXTRA="" run "C4 two sequences, default n=256" STEP_ENC_NSEQ=2
EXTRA="--encoder-workgroups 160" run "C4 two sequences, n=160" STEP_ENC_NSEQ=2
EXTRA="--encoder-workgroups 192" run "C4 two sequences, n=192" STEP_ENC_NSEQ=2
done
cat $L
