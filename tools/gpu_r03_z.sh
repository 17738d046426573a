#!/bin/bash
# round 3, call Z: C3 kernel table and C2 step timeline at the current HEAD
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03z
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_C3 -o p -- python $GRAFT_REPO_ROOT/bench.py --config TSFormer_PEMS-BAY --steps 13 --warmup 3 --no-extras --no-cpu-baseline --no-pmc > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_C3.err)
db=$(find gpurun_out/prof_${tag}_C3 -name '*.db' | head -1)
python tools/prof_summary.py $db > gpurun_out/${tag}_C3_pretrain_train_step.md
rm -rf gpurun_out/prof_${tag}_C3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag} -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err)
db=$(find gpurun_out/prof_${tag} -name '*.db' | head -1)
python tools/prof_timeline.py $db > gpurun_out/${tag}_C2_step_timeline.md 2> gpurun_out/${tag}_timeline.err
rm -rf gpurun_out/prof_${tag}
head -30 gpurun_out/${tag}_C3_pretrain_train_step.md | cut -c1-150; tail -5 gpurun_out/${tag}_C2_step_timeline.md
