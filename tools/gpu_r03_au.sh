#!/bin/bash
# round 3, call AU: the default bench line with its other configs under the two-queue setting (no CPU baseline: 2.3 GPU-minutes left)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 115 python bench.py --no-cpu-baseline > gpurun_out/r03au_bench.json 2> gpurun_out/r03au_bench.err; echo "rc $?"
python -c "
import json; d = json.loads(open('gpurun_out/r03au_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['runtime_env']); [print(k, v.get('ms_per_step'), v.get('value')) for k, v in d['other_configs'].items()]"
