#!/bin/bash
# round 4, call q: row kernels with LDS-staged (coalesced) tile loads / stores; where the weight-gradient kernels' time goes; C3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04q}
timeout 600 python -m pytest tests/test_gpu_pretrain.py -q -rP -k "fused_feed_forward_block or dropout_runs or full_size" > gpurun_out/${t}_ffn_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_ffn_tests.log
grep -E "passed|failed|fused feed-forward|C3 full|rc |Error|error" gpurun_out/${t}_ffn_tests.log | tail -12
for v in default abl8 abl16 abl24; do
  lib=step_amd/libstep_hip_$v.so; [ $v = default ] && lib=step_amd/libstep_hip.so
  STEP_HIP_LIB=$lib timeout 300 python tools/bench_pt_ffn.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/${t}_ffn_ablations.log
cat gpurun_out/${t}_ffn_ablations.log
timeout 600 python bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 15 --warmup 5 > gpurun_out/${t}_bench_C3.json 2> gpurun_out/${t}_bench_C3.err
python -c "
import json
d=json.loads(open('gpurun_out/${t}_bench_C3.json').read().strip().splitlines()[-1]); print('C3', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/${t}_bench_C3.err
