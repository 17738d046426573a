#!/bin/bash
# round 3, call AH: C4 as an "other config" of the default run with a fresh allocator cache; host time and device allocations in its timed region
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 30 --warmup 8 --no-pmc --no-cpu-baseline --no-loader-figure --other-configs STEP_PEMS07,SYNTH_4096 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', d['ms_per_step']); [print(k, {kk: v.get(kk) for kk in ('ms_per_step', 'encoder_ms_per_launch', 'host_enqueue_ms_per_step', 'device_allocs_in_timed_region')}) for k, v in d['other_configs'].items()]"
