#!/bin/bash
# kernel timeline of the prefetch + persistent-encoder step at PEMS04
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05m}
(cd /tmp && GPU_MAX_HW_QUEUES=4 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${t} -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --no-loader-figure --steps 12 --warmup 4 --prefetch --encoder-workgroups 160 > $GRAFT_REPO_ROOT/gpurun_out/${t}_prof.out 2> $GRAFT_REPO_ROOT/gpurun_out/${t}_prof.err)
db=$(find gpurun_out/prof_${t} -name '*.db' | head -1)
python tools/prof_timeline.py $db --anchor adam_clip > gpurun_out/${t}_C2_step_timeline_prefetch.md 2> gpurun_out/${t}_tl.err
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step_prefetch.md 2>> gpurun_out/${t}_tl.err
rm -rf gpurun_out/prof_${t}
tail -8 gpurun_out/${t}_C2_step_timeline_prefetch.md; head -30 gpurun_out/${t}_C2_train_step_prefetch.md
