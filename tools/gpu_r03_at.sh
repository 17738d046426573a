#!/bin/bash
# round 3, call AT: the default bench line with the two-hardware-queue setting inside bench.py (headline + extras, no other configs / CPU baseline)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --other-configs - > gpurun_out/r03at_bench.json 2> gpurun_out/r03at_bench.err; echo "rc $?"
python -c "
import json; d = json.loads(open('gpurun_out/r03at_bench.json').read().strip().splitlines()[-1]); r = d['roofline']
print(d['value'], d['ms_per_step'], d['step_ms'], d['runtime_env'], 'enc', r['ms_per_launch'], r['ms_per_launch_alone'], 'traffic', r['traffic']['hbm_bytes_per_launch'], r['traffic']['static'])"
