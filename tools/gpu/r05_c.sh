#!/bin/bash
# round 5, call 3: progress-based wave priorities, DMA pieces from the first-dispatched waves
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=r05c
cd scratch_ab
L="default=./libenc_default.so pp=./libenc_pp.so fo=./libenc_fo.so ppfo=./libenc_ppfo.so all=./libenc_all.so allil=./libenc_allil.so ppt=./libenc_ppt.so"
ENC_AB_OUT=../gpurun_out timeout 600 ./enc_ab $L > ../gpurun_out/${t}_enc_ab_p336.log 2>&1
ENC_AB_P=168 ENC_AB_S=3532 timeout 600 ./enc_ab default=./libenc_default.so pp=./libenc_pp.so ppfo=./libenc_ppfo.so all=./libenc_all.so > ../gpurun_out/${t}_enc_ab_p168.log 2>&1
cd ..
for f in gpurun_out/enc_timing_ppt_*.bin; do python tools/enc_phase_table.py $f; done > gpurun_out/${t}_encoder_phase_table_pp.md 2>&1
grep "bench-like\|DROPOUT" gpurun_out/${t}_enc_ab_p336.log gpurun_out/${t}_enc_ab_p168.log
