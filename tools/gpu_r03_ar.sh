#!/bin/bash
# round 3, call AR: GPU_MAX_HW_QUEUES (hardware queues the runtime multiplexes streams onto; default 4) at C2 / C4
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do for q in 4 8 2; do for cfg in STEP_PEMS04 STEP_PEMS07; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --config $cfg --steps 40 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg GPU_MAX_HW_QUEUES=$q', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done; done; done > gpurun_out/r03ar_hw_queues_ab.log 2>&1
cat gpurun_out/r03ar_hw_queues_ab.log
