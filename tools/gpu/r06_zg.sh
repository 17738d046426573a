#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_zg_short_run.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc"
run() { name=$1; shift; python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4),d.get('step_ms'),round(d['host_enqueue_ms_per_step'],3),round(d['roofline']['ms_per_launch'],3))" >> $L; }
for rep in 1 2 3; do
run "20/5" --steps 20 --warmup 5
run "20/20" --steps 20 --warmup 20
run "100/20" --steps 100 --warmup 20
done
cat $L
