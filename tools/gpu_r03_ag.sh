#!/bin/bash
# round 3, call AG: why is C4 slower inside the default run's other_configs (6.87 ms) than alone (6.08 ms)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03ag
one() { python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('$1', 'ms_per_step', round(d['ms_per_step'], 3), 'enc in-step', round(r['ms_per_launch'], 3), 'fallback', r['fallback_units_per_launch'], 'host', round(d['host_enqueue_ms_per_step'], 3))"; }
timeout 400 python bench.py --config STEP_PEMS07 --steps 20 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 0 2>/dev/null | one "C4 alone random-init"
timeout 400 python bench.py --config STEP_PEMS07 --steps 20 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 100 2>/dev/null | one "C4 alone pretrain-100"
timeout 400 python bench.py --config STEP_PEMS07 --steps 20 --warmup 8 --no-pmc --no-extras --no-cpu-baseline --pretrain-steps 300 2>/dev/null | one "C4 alone pretrain-300"
timeout 600 python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-loader-figure --other-configs STEP_PEMS07 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 then C4 as other config:', d['ms_per_step'], d['other_configs'])"
