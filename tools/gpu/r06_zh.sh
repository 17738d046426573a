#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_dgl_conv.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3 > gpurun_out/r06_zh_tests.log
cat gpurun_out/r06_zh_tests.log
L=gpurun_out/r06_zh_conv1_fwd.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc"
run() { name=$1; lib=$2; shift 2; STEP_HIP_LIB=$PWD/step_amd/$lib python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4))" >> $L; }
for rep in 1 2; do
run "C2 old" libstep_hip_oldconv.so
run "C2 new" libstep_hip.so
run "C4 old" libstep_hip_oldconv.so --config STEP_PEMS07
run "C4 new" libstep_hip.so --config STEP_PEMS07
run "C5 old" libstep_hip_oldconv.so --config SYNTH_4096
run "C5 new" libstep_hip.so --config SYNTH_4096
done
cat $L
