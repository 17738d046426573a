#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_zn_C4_knobs.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc --config STEP_PEMS07"
run() { name=$1; shift; env "$@" python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4))" >> $L; }
for rep in 1 2; do
run "default" X=1
run "TALL_M=768" STEP_GEMM_TALL_M=768
run "SPLIT_TARGET=512" STEP_GEMM_SPLIT_TARGET=512
run "SPLIT_TARGET=1024" STEP_GEMM_SPLIT_TARGET=1024
run "ADJ_PIECES=0" STEP_ADJ_PIECES=0
done
cat $L
