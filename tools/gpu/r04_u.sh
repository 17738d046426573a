#!/bin/bash
# round 4, call u: LayerNorm-backward block cap A/B, the one-launch layer pack, C3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04u}
timeout 900 python -m pytest tests/test_gpu_pretrain.py -q -rP > gpurun_out/${t}_pretrain_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_pretrain_tests.log
grep -E "passed|failed|rc |Error" gpurun_out/${t}_pretrain_tests.log | tail -5
for b in 256 512 1024 2048 4096 8192; do STEP_LN_BWD_BLOCKS=$b timeout 120 python tools/bench_pt_ln.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${t}_layernorm_backward_blocks.log
cat gpurun_out/${t}_layernorm_backward_blocks.log
for b in 512 1024 4096; do
STEP_LN_BWD_BLOCKS=$b timeout 600 python bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 15 --warmup 5 > gpurun_out/${t}_bench_C3_b$b.json 2> gpurun_out/${t}_bench_C3.err
python -c "
import json
d=json.loads(open('gpurun_out/${t}_bench_C3_b$b.json').read().strip().splitlines()[-1]); print('C3 blocks=$b', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/${t}_bench_C3.err
done
