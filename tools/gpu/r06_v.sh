#!/bin/bash
# round 6: ablations of the second-version attention kernels (WRONG results: timing only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_v_attn2_ablations.log; : > $L
python tools/bench_pt_attention.py 2>&1 | grep -v amdgpu.ids | sed 's/^/full /' >> $L
for v in 1 2 3 48 51 67 115; do
  STEP_HIP_LIB=$PWD/step_amd/libstep_hip_a2abl$v.so python tools/bench_pt_attention.py 2>&1 | grep -v amdgpu.ids | sed "s/^/abl$v /" >> $L
done
cat $L
