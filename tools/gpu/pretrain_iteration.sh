#!/bin/bash
# one iteration on the pre-training step (config C3): its GPU tests, the micro-benchmarks of its kernels, the bench line and the kernel table
# usage (through gpurun): bash tools/gpu/pretrain_iteration.sh <tag>  ->  gpurun_out/<tag>_*
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04r}
timeout 900 python -m pytest tests/test_gpu_pretrain.py -q -rP > gpurun_out/${t}_pretrain_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_pretrain_tests.log
grep -E "passed|failed|pool-drawn|C3 full|rc |Error" gpurun_out/${t}_pretrain_tests.log | tail -12
timeout 300 python tools/bench_pt_ffn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${t}_ffn_bench.log; cat gpurun_out/${t}_ffn_bench.log
(ATTN_POOL=1 timeout 300 python tools/bench_pt_attention.py; ATTN_POOL=0 timeout 300 python tools/bench_pt_attention.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/${t}_attention_bench.log; cat gpurun_out/${t}_attention_bench.log
timeout 600 python bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 15 --warmup 5 > gpurun_out/${t}_bench_C3.json 2> gpurun_out/${t}_bench_C3.err
python -c "
import json
d=json.loads(open('gpurun_out/${t}_bench_C3.json').read().strip().splitlines()[-1]); print('C3', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/${t}_bench_C3.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${t}_C3 -o p -- python $GRAFT_REPO_ROOT/bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 15 --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${t}_prof_C3.err)
db=$(find gpurun_out/prof_${t}_C3 -name '*.db' | head -1)
python tools/prof_summary.py $db > gpurun_out/${t}_C3_pretrain_train_step.md; rm -rf gpurun_out/prof_${t}_C3
head -30 gpurun_out/${t}_C3_pretrain_train_step.md | cut -c1-140
