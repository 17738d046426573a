#!/bin/bash
# round 4, call ac: what the weight-fragment stream of the feed-forward row kernels costs (FF_ABLATE=32: no fragment DMA; results wrong, timing only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04ac}
for v in default abl32 default abl32; do
  lib=step_amd/libstep_hip_$v.so; [ $v = default ] && lib=step_amd/libstep_hip.so
  STEP_HIP_LIB=$lib timeout 200 python tools/bench_pt_ffn.py 2>&1 | grep -E "forward" | grep -v projections
done > gpurun_out/${t}_ffn_no_weight_stream.log
cat gpurun_out/${t}_ffn_no_weight_stream.log
