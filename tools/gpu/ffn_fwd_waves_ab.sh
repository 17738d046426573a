#!/bin/bash
# feed-forward forward row kernel: workgroups of four waves, two per compute unit (STEP_FFN_FWD_WAVES=4), against eight waves, one per unit
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04ah}
STEP_FFN_FWD_WAVES=4 timeout 600 python -m pytest tests/test_gpu_pretrain.py -q -rP -k "fused_feed_forward or full_size or layernorm_output" > gpurun_out/${t}_fwd4_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_fwd4_tests.log
grep -E "passed|failed|rc " gpurun_out/${t}_fwd4_tests.log | tail -3
for w in 8 4 8 4; do echo "STEP_FFN_FWD_WAVES=$w"; STEP_FFN_FWD_WAVES=$w timeout 200 python tools/bench_pt_ffn.py 2>&1 | grep -E "forward" | grep -v projections; done > gpurun_out/${t}_ffn_fwd_waves_ab.log
cat gpurun_out/${t}_ffn_fwd_waves_ab.log
for w in 8 4; do
STEP_FFN_FWD_WAVES=$w timeout 600 python bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 15 --warmup 5 > gpurun_out/${t}_bench_C3_w$w.json 2> gpurun_out/${t}_bench_C3.err
python -c "
import json
d=json.loads(open('gpurun_out/${t}_bench_C3_w$w.json').read().strip().splitlines()[-1]); print('C3 waves=$w', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/${t}_bench_C3.err
done
