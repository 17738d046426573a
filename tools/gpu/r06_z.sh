#!/bin/bash
# round 6: two sequences per encoder workgroup at P = 168: parity tests, occupancy, A/B of the steps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder_range_guard.py tests/test_encoder_no_scratch.py -x -q 2>&1 | tail -4 > gpurun_out/r06_z2_tests.log
cat gpurun_out/r06_z2_tests.log
L=gpurun_out/r06_z2_two_sequences_ab.log; : > $L
F="--no-extras --no-cpu-baseline --no-pmc"
run() { name=$1; shift; env "$@" python bench.py $F ${EXTRA} 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$name',round(d['value'],1),round(d['ms_per_step'],4),round(d['roofline']['ms_per_launch'],3),d['roofline'].get('ms_per_launch_alone'))" >> $L; }
for rep in 1 2; do
EXTRA="--config STEP_PEMS07" run "C4 one sequence per workgroup" STEP_ENC_NSEQ=1
EXTRA="--config STEP_PEMS07" run "C4 two sequences" STEP_ENC_NSEQ=2
done
EXTRA="--config STEP_PEMS07 --no-prefetch" run "C4 inline, one" STEP_ENC_NSEQ=1
EXTRA="--config STEP_PEMS07 --no-prefetch" run "C4 inline, two" STEP_ENC_NSEQ=2
EXTRA="--config STEP_METR-LA" run "C1 one" STEP_ENC_NSEQ=1
EXTRA="--config STEP_METR-LA" run "C1 two" STEP_ENC_NSEQ=2
EXTRA="--config SYNTH_4096" run "C5 one" STEP_ENC_NSEQ=1
EXTRA="--config SYNTH_4096" run "C5 two" STEP_ENC_NSEQ=2
cat $L
