#!/bin/bash
# same-box A/B of whole training steps: tools/ab_bench.sh "<bench args>" lib1.so lib2.so ...   (two interleaved repetitions)
args=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    STEP_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --no-loader-figure 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$lib', d['config'].get('workload'), 'ms_per_step', round(d['ms_per_step'], 3), d['unit'], round(d['value'], 1))"
  done
done
