#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_zk_ln_bwd_prefetch.log; : > $L
python -m pytest tests/test_gpu_pretrain.py -x -q 2>&1 | tail -2 >> $L
for lib in libstep_hip_oldln.so libstep_hip.so; do
  STEP_HIP_LIB=$PWD/step_amd/$lib python tools/bench_pt_ln.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib /" >> $L
  for b in 2048 4096; do STEP_LN_BWD_BLOCKS=$b STEP_HIP_LIB=$PWD/step_amd/$lib python tools/bench_pt_ln.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$lib /" >> $L; done
done
F="--no-extras --no-cpu-baseline --no-pmc --config TSFormer_PEMS-BAY"
for rep in 1 2; do for lib in libstep_hip_oldln.so libstep_hip.so; do
STEP_HIP_LIB=$PWD/step_amd/$lib python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('C3 $lib',round(d['value'],1),round(d['ms_per_step'],4))" >> $L
done; done
cat $L
