#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_zm_hop_xt.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc"
run() { name=$1; shift; env "$@" python bench.py $F ${EXTRA} 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4))" >> $L; }
for rep in 1 2; do
EXTRA="--config STEP_PEMS07" run "C4 default (1024)" X=1
EXTRA="--config STEP_PEMS07" run "C4 xT from 512" STEP_HOP_XT_MIN_N=512
EXTRA="" run "C2 default" X=1
EXTRA="" run "C2 xT from 256" STEP_HOP_XT_MIN_N=256
done
cat $L
