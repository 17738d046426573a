#!/bin/bash
# round 6: PEMS07 with the two-sequence encoder: share of the chip for the prefetched encoder, announcement early / late
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_za_C4_split_sweep.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc --config STEP_PEMS07 --steps 60 --warmup 15"
run() { name=$1; shift; python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4),round(d['roofline']['ms_per_launch'],3))" >> $L; }
for n in 192 224 256 288 320 352 384 416 448; do run "late n=$n" --encoder-workgroups $n; done
for n in 224 256 288 320 352 384; do run "early n=$n" --encoder-workgroups $n --prefetch-early; done
run "inline" --no-prefetch
cat $L
