#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05u}
python tools/host_sections.py STEP_PEMS04 60 > gpurun_out/${t}_host_sections_C2.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/${t}_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests_full.log
rm -f gpurun_out/${t}_bench.log
run() { # env..., then -- then args
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs timeout 600 python bench.py --no-extras --no-pmc --no-cpu-baseline --no-loader-figure --steps 60 --warmup 15 "$@" 2> gpurun_out/${t}_last.err | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
print('[$envs | $*]', 'ms_per_step', round(d['ms_per_step'], 3), '| enc in-step', round(d['roofline']['ms_per_launch'], 3), '| host', round(d.get('host_enqueue_ms_per_step') or 0, 2), '| queues', d.get('runtime_env'))" >> gpurun_out/${t}_bench.log 2>&1
}
run X=0 --
run X=0 -- --config STEP_METR-LA
run X=0 -- --batch 4
run X=0 -- --config STEP_PEMS07
cat gpurun_out/${t}_host_sections_C2.log | grep -v amdgpu; tail -4 gpurun_out/${t}_gpu_tests_full.log; cat gpurun_out/${t}_bench.log
