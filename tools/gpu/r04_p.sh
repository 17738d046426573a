#!/bin/bash
# round 4, call p: the fused feed-forward kernels -- their test with the reference's ReLU decisions taken on the operands the kernels see,
# and where their time goes (FF_ABLATE variants: 1 no matrix-core chains, 2 no row loads, 4 no stores)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04p}
timeout 600 python -m pytest tests/test_gpu_pretrain.py -q -rP -k "fused_feed_forward_block or dropout_runs" > gpurun_out/${t}_ffn_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_ffn_tests.log
grep -E "passed|failed|fused feed-forward|rc |Error|error" gpurun_out/${t}_ffn_tests.log | tail -12
for v in default abl1 abl2 abl4 abl6 abl7; do
  lib=step_amd/libstep_hip_$v.so; [ $v = default ] && lib=step_amd/libstep_hip.so
  STEP_HIP_LIB=$lib timeout 300 python tools/bench_pt_ffn.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/${t}_ffn_ablations.log
cat gpurun_out/${t}_ffn_ablations.log
