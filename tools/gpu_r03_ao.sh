#!/bin/bash
# round 3, call AO: the remaining bench modes at the final code: C1 (STEP_METR-LA), exact-f32 mode of C2, validation forward of C2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03ao
timeout 400 python bench.py --config STEP_METR-LA --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > gpurun_out/${tag}_bench_STEP_METR-LA.json 2>/dev/null
timeout 400 python bench.py --matmul f32 --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-pmc > gpurun_out/${tag}_bench_C2_f32mode.json 2>/dev/null
timeout 400 python bench.py --forward-only --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-pmc > gpurun_out/${tag}_bench_C2_validation_forward.json 2>/dev/null
for f in STEP_METR-LA C2_f32mode C2_validation_forward; do python -c "
import sys, json; d = json.loads(open('gpurun_out/${tag}_bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'], 1), 'windows/s', round(d['ms_per_step'], 3), 'ms', 'host', round(d.get('host_enqueue_ms_per_step', 0), 3))"; done
