#!/bin/bash
# round 6: encoder share at round-count boundaries (2456 sequences at C2: n = 164 -> 15 rounds per workgroup, 160 -> 16; C4: 1766 two-sequence units)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_zj_split_rounds.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc"
run() { name=$1; shift; python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4),round(d['roofline']['ms_per_launch'],3))" >> $L; }
for rep in 1 2; do
for n in 154 160 164 166 176; do run "C2 n=$n" --encoder-workgroups $n; done
done
for rep in 1 2; do
for n in 236 256 272 296; do run "C4 n=$n" --config STEP_PEMS07 --encoder-workgroups $n; done
done
cat $L
