#!/bin/bash
# round 3, call AL: after reverting the edge column move: tests of the touched paths, C2 / C4 / C5 with and without the leaf stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03al
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_full_size.py tests/test_gpu_runner_golden.py -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; echo "tests exit $?"
grep -E "passed|failed" gpurun_out/${tag}_tests.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/${tag}_tests.log | head
ab() { # name config steps
for rep in 1 2; do for w in 0 1; do
STEP_LEAF_STREAM=$w timeout 400 python bench.py --config $2 --steps $3 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 STEP_LEAF_STREAM=$w', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done; done
}
{ ab C2 STEP_PEMS04 60; ab C4 STEP_PEMS07 40; ab C5 SYNTH_4096 20; } > gpurun_out/${tag}_leaf_stream_ab.log 2>&1
cat gpurun_out/${tag}_leaf_stream_ab.log
