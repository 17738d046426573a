#!/bin/bash
# round 3, call S: per-dispatch timeline of one C4 (PEMS07) training step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03s
for cfg in STEP_PEMS07; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag} -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 12 --warmup 4 --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err)
db=$(find gpurun_out/prof_${tag} -name '*.db' | head -1)
python tools/prof_timeline.py $db > gpurun_out/${tag}_${cfg}_step_timeline.md 2> gpurun_out/${tag}_timeline.err
rm -rf gpurun_out/prof_${tag}
tail -6 gpurun_out/${tag}_${cfg}_step_timeline.md
done
