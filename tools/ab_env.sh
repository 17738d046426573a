#!/bin/bash
# same-box A/B of whole training steps under different environments: tools/ab_env.sh "<bench args>" "ENV1=.." "ENV2=.." ...  (use "X=0" for the default)
args=$1; shift
for rep in 1 2; do
  for e in "$@"; do
    env $e python bench.py $args --no-cpu-baseline --no-loader-figure 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$e', d['config'].get('workload')[:12], 'ms_per_step', round(d['ms_per_step'], 3), d['unit'], round(d['value'], 1))"
  done
done
