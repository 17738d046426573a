#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_zo_C3_knobs.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc --config TSFormer_PEMS-BAY"
run() { name=$1; shift; env "$@" python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4))" >> $L; }
for rep in 1 2; do
run "default" X=1
run "FFN_FWD_WAVES=4" STEP_FFN_FWD_WAVES=4
run "LN_BWD_BLOCKS=2048" STEP_LN_BWD_BLOCKS=2048
run "LN_BWD_BLOCKS=1024" STEP_LN_BWD_BLOCKS=1024
done
cat $L
