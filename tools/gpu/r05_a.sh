#!/bin/bash
# round 5, call 1: where the encoder's time goes (phase stamps, matrix / vector ablations, counters) + first A/B of the fragment-prefetch variants,
# the whole-step A/B and the overlap policy at the large graphs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=r05a
rocm-smi --showclocks 2>/dev/null | head -12 > gpurun_out/${t}_box.log; nproc >> gpurun_out/${t}_box.log
cd scratch_ab
L="default=./libenc_default.so ffn=./libenc_ffn.so ffnq3=./libenc_ffnq3.so ffnq1=./libenc_ffnq1.so q3=./libenc_q3.so prio2p=./libenc_prio2p.so timing=./libenc_timing.so timingp=./libenc_timingp.so"
ENC_AB_OUT=../gpurun_out timeout 600 ./enc_ab $L > ../gpurun_out/${t}_enc_ab_p336.log 2>&1
timeout 300 ./enc_ab default=./libenc_default.so ablV=./libenc_ablV.so ablM=./libenc_ablM.so abl32=./libenc_abl32.so abl64=./libenc_abl64.so > ../gpurun_out/${t}_enc_ablations_p336.log 2>&1
ENC_AB_P=168 ENC_AB_S=3532 ENC_AB_OUT=../gpurun_out timeout 600 ./enc_ab default=./libenc_default.so ffn=./libenc_ffn.so ffnq3=./libenc_ffnq3.so timing=./libenc_timing.so timingp=./libenc_timingp.so > ../gpurun_out/${t}_enc_ab_p168.log 2>&1
cd ..
for f in gpurun_out/enc_timing_*.bin; do python tools/enc_phase_table.py $f; done > gpurun_out/${t}_encoder_phase_table.md 2>&1
# counters of the current default kernel and of the prefetch variant
timeout 600 bash tools/pmc_enc_ab.sh default mem > gpurun_out/${t}_pmc_default.log 2>&1
cp gpurun_out/pmc_ab_default_summary.txt gpurun_out/${t}_pmc_default_summary.txt
timeout 400 bash tools/pmc_enc_ab.sh ffnq3 > gpurun_out/${t}_pmc_ffnq3.log 2>&1
cp gpurun_out/pmc_ab_ffnq3_summary.txt gpurun_out/${t}_pmc_ffnq3_summary.txt
rm -rf gpurun_out/pmc_ab_*/
# whole step, same box
timeout 900 bash tools/ab_bench.sh "--no-extras --no-pmc --steps 60 --warmup 15" step_amd/libstep_hip.so step_amd/libstep_hip_ffnq3.so > gpurun_out/${t}_bench_ab_C2.log 2>&1
timeout 900 bash tools/ab_env.sh "--config STEP_PEMS07 --no-extras --no-pmc --steps 40 --warmup 10" X=0 STEP_NO_OVERLAP=1 > gpurun_out/${t}_overlap_C4.log 2>&1
timeout 900 bash tools/ab_env.sh "--config SYNTH_4096 --no-extras --no-pmc --steps 20 --warmup 5" X=0 STEP_NO_OVERLAP=1 > gpurun_out/${t}_overlap_C5.log 2>&1
tail -n 12 gpurun_out/${t}_enc_ab_p336.log; cat gpurun_out/${t}_bench_ab_C2.log gpurun_out/${t}_overlap_C4.log gpurun_out/${t}_overlap_C5.log
