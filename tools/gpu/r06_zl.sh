#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_zl_encoder_ln_two_fma.log; : > $L
python -m pytest tests/test_gpu_kernels.py tests/test_encoder_no_scratch.py -x -q 2>&1 | tail -2 >> $L
F="--no-cpu-baseline --no-pmc --no-runner-figure --other-configs STEP_PEMS07"
for rep in 1 2; do for lib in libstep_hip_ln0.so libstep_hip.so; do
STEP_HIP_LIB=$PWD/step_amd/$lib python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('$lib C2',round(d['value'],1),round(d['ms_per_step'],4),'encoder in step',round(r['ms_per_launch'],3),'alone',round(r['ms_per_launch_alone'],3),'whole chip in step',round(r['whole_chip_in_step']['ms_per_launch'],3),'C4',round(d['other_configs']['STEP_PEMS07']['ms_per_step'],3))" >> $L
done; done
cat $L
