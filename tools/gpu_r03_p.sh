#!/bin/bash
# round 3, call P: per-dispatch timeline of one C2 training step (which stream is waiting for what after the encoder)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03p
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag} -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err)
db=$(find gpurun_out/prof_${tag} -name '*.db' | head -1)
python tools/prof_timeline.py $db > gpurun_out/${tag}_C2_step_timeline.md 2> gpurun_out/${tag}_timeline.err
sqlite3 $db "pragma table_info(rocpd_kernel_dispatch)" > gpurun_out/${tag}_schema.txt 2>&1 || python -c "
import sqlite3,sys; db=sqlite3.connect('$db'); print([r[1] for r in db.execute('pragma table_info(rocpd_kernel_dispatch)')])" > gpurun_out/${tag}_schema.txt
rm -rf gpurun_out/prof_${tag}
tail -8 gpurun_out/${tag}_C2_step_timeline.md; cat gpurun_out/${tag}_timeline.err | tail -5
