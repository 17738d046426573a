#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_zf_fused_tile.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc"
run() { name=$1; shift; env "$@" python bench.py $F ${EXTRA} 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4))" >> $L; }
for rep in 1 2; do
EXTRA="" run "C2 128" STEP_GEMM_FUSED_TILE=128
EXTRA="" run "C2 64" STEP_GEMM_FUSED_TILE=64
EXTRA="--config STEP_PEMS07" run "C4 128" STEP_GEMM_FUSED_TILE=128
EXTRA="--config STEP_PEMS07" run "C4 64" STEP_GEMM_FUSED_TILE=64
EXTRA="--config SYNTH_4096" run "C5 128" STEP_GEMM_FUSED_TILE=128
EXTRA="--config SYNTH_4096" run "C5 64" STEP_GEMM_FUSED_TILE=64
done
cat $L
