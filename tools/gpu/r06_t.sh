#!/bin/bash
# round 6: encoder on all compute units with a 2-slot weight ring (105 KB of LDS: the chain's small kernels can share its compute units), chain kernels at wave priority 3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/r06_t_coresident_sweep.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc --steps 60 --warmup 15"
run() { name=$1; lib=$2; shift 2; STEP_HIP_LIB=$PWD/step_amd/$lib python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4),round(d['roofline']['ms_per_launch'],3))" >> $L; }
run "base n=160" libstep_hip.so
for n in 160 192 224 256; do run "ring2 n=$n" libstep_hip_r2.so --encoder-workgroups $n; done
for n in 160 192 224 256; do run "ring2+prio n=$n" libstep_hip_r2p.so --encoder-workgroups $n; done
for n in 160 176; do run "prio n=$n" libstep_hip_p.so --encoder-workgroups $n; done
run "base n=160 again" libstep_hip.so
cat $L
