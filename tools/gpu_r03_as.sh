#!/bin/bash
# round 3, call AS: GPU_MAX_HW_QUEUES = 1 / 2 / 3 at C2, 2 vs 4 at C5 and C3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
{ for q in 1 2 3 2 1; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 40 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STEP_PEMS04 GPU_MAX_HW_QUEUES=$q', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done
for cfg in SYNTH_4096 TSFormer_PEMS-BAY; do for q in 4 2; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --config $cfg --steps 15 --warmup 5 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 0 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg GPU_MAX_HW_QUEUES=$q', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
done; done; } > gpurun_out/r03as_hw_queues_ab.log 2>&1
cat gpurun_out/r03as_hw_queues_ab.log
