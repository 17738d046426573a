#!/bin/bash
# round 4, call o: the fused feed-forward kernels of the pre-training step (csrc/pretrain_fused.hip): their tests, the pre-training tests
# with them on the path, C3 with and without them, kernel table of C3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04o}
timeout 900 python -m pytest tests/test_gpu_pretrain.py -q -rP -x > gpurun_out/${t}_pretrain_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_pretrain_tests.log
grep -E "passed|failed|fused feed-forward|C3 full size|rc " gpurun_out/${t}_pretrain_tests.log | tail -12
for f in 0 1; do
  STEP_PT_FUSED_FFN=$f timeout 600 python bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 15 --warmup 5 > gpurun_out/${t}_bench_C3_fused$f.json 2> gpurun_out/${t}_bench_C3_fused$f.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/${t}_bench_C3_fused$f.json').read().strip().splitlines()[-1]); print('fused=$f', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/${t}_bench_C3_fused$f.err
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${t}_C3 -o p -- python $GRAFT_REPO_ROOT/bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 15 --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${t}_prof_C3.err)
db=$(find gpurun_out/prof_${t}_C3 -name '*.db' | head -1)
python tools/prof_summary.py $db > gpurun_out/${t}_C3_pretrain_train_step.md; rm -rf gpurun_out/prof_${t}_C3
head -24 gpurun_out/${t}_C3_pretrain_train_step.md | cut -c1-150
