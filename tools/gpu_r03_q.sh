#!/bin/bash
# round 3, call Q: is the C2 step bound by the host?  host time to enqueue a step next to the device time of the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in STEP_PEMS04 STEP_METR-LA; do
timeout 300 python bench.py --config $cfg --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'ms_per_step', round(d['ms_per_step'], 3), 'host_enqueue_ms_per_step', round(d['host_enqueue_ms_per_step'], 3))"
done > gpurun_out/r03q_host_enqueue.log 2>&1
cat gpurun_out/r03q_host_enqueue.log
