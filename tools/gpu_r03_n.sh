#!/bin/bash
# round 3, call N: ablations of the re-written pre-training attention kernels + the 64-bit-product Philox
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pretrain.py -q -m gpu -x 2>&1 | tail -2
for lib in libstep_hip.so libstep_hip_mul64.so libstep_hip_abl2.so libstep_hip_abl4.so libstep_hip_abl6.so; do
  STEP_HIP_LIB=step_amd/$lib timeout 200 python tools/bench_pt_attention.py 2>/dev/null
done > gpurun_out/r03n_attention_ablations.log 2>&1
cat gpurun_out/r03n_attention_ablations.log
