#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in libstep_hip.so libstep_hip_abl1.so libstep_hip_abl2.so libstep_hip_abl4.so libstep_hip_abl7.so; do
  STEP_HIP_LIB=step_amd/$lib timeout 200 python tools/bench_pt_attention.py 2>/dev/null
done > gpurun_out/r03l_attention_ablations.log 2>&1
cat gpurun_out/r03l_attention_ablations.log
