#!/bin/bash
# round 4, call ab: the weight ring of the feed-forward row kernels synchronised by LDS counters instead of workgroup barriers (STEP_FFN_RING_FLAGS=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04ab}
STEP_FFN_RING_FLAGS=1 timeout 600 python -m pytest tests/test_gpu_pretrain.py -q -rP -k "fused_feed_forward or full_size or layernorm_output or dropout_runs" > gpurun_out/${t}_flags_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_flags_tests.log
grep -E "passed|failed|rc |fused feed-forward|C3 full|Error" gpurun_out/${t}_flags_tests.log | tail -8
for f in 0 1 0 1; do echo "STEP_FFN_RING_FLAGS=$f"; STEP_FFN_RING_FLAGS=$f timeout 200 python tools/bench_pt_ffn.py 2>&1 | grep -E "forward|pack" | grep -v projections; done > gpurun_out/${t}_ring_flags_ab.log
cat gpurun_out/${t}_ring_flags_ab.log
for f in 0 1; do
STEP_FFN_RING_FLAGS=$f timeout 600 python bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 15 --warmup 5 > gpurun_out/${t}_bench_C3_flags$f.json 2> gpurun_out/${t}_bench_C3.err
python -c "
import json
d=json.loads(open('gpurun_out/${t}_bench_C3_flags$f.json').read().strip().splitlines()[-1]); print('C3 flags=$f', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/${t}_bench_C3.err
done
