#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_gemm_bf16.py tests/test_gpu_step.py -x -q 2>&1 | tail -3 > gpurun_out/r06_zd_tests.log
cat gpurun_out/r06_zd_tests.log
L=gpurun_out/r06_zd_gemm_two_stages.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc"
run() { name=$1; lib=$2; shift 2; STEP_HIP_LIB=$PWD/step_amd/$lib python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4),round(d['roofline']['ms_per_launch'],3))" >> $L; }
for rep in 1 2; do
run "C2 one stage" libstep_hip_pf1.so
run "C2 two stages" libstep_hip.so
run "C2 inline one stage" libstep_hip_pf1.so --no-prefetch
run "C2 inline two stages" libstep_hip.so --no-prefetch
run "C4 one stage" libstep_hip_pf1.so --config STEP_PEMS07
run "C4 two stages" libstep_hip.so --config STEP_PEMS07
done
run "C5 one stage" libstep_hip_pf1.so --config SYNTH_4096
run "C5 two stages" libstep_hip.so --config SYNTH_4096
run "C1 one stage" libstep_hip_pf1.so --config STEP_METR-LA
run "C1 two stages" libstep_hip.so --config STEP_METR-LA
cat $L
