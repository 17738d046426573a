#!/bin/bash
# round 3, call U: kernel table of the C3 pre-training step with bf16 activations around the attention
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03u
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_C3 -o p -- python $GRAFT_REPO_ROOT/bench.py --config TSFormer_PEMS-BAY --steps 13 --warmup 3 --no-extras --no-cpu-baseline --no-pmc > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_C3.err)
db=$(find gpurun_out/prof_${tag}_C3 -name '*.db' | head -1)
python tools/prof_summary.py $db > gpurun_out/${tag}_C3_pretrain_train_step.md
rm -rf gpurun_out/prof_${tag}_C3
head -45 gpurun_out/${tag}_C3_pretrain_train_step.md | cut -c1-200
