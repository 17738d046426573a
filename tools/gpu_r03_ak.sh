#!/bin/bash
# round 3, call AK: edge column pass on the auxiliary stream, fewer loss-reduce blocks, fc_his queued before the join with the second stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03ak
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_full_size.py tests/test_gpu_sharded_graph_learner.py tests/test_gpu_training_parity.py tests/test_gpu_runner_golden.py tests/test_gpu_dgl_conv.py -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/${tag}_tests.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/${tag}_tests.log | head
for cfg in STEP_PEMS04 STEP_PEMS07 SYNTH_4096; do for rep in 1 2; do
timeout 400 python bench.py --config $cfg --steps 40 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done; done > gpurun_out/${tag}_bench.log 2>&1
cat gpurun_out/${tag}_bench.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag} -o p -- python $GRAFT_REPO_ROOT/bench.py --config STEP_PEMS04 --steps 12 --warmup 4 --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err)
db=$(find gpurun_out/prof_${tag} -name '*.db' | head -1)
python tools/prof_timeline.py $db > gpurun_out/${tag}_C2_step_timeline.md 2> gpurun_out/${tag}_timeline.err
rm -rf gpurun_out/prof_${tag}
grep -n "row_dot" gpurun_out/${tag}_C2_step_timeline.md | head -2
