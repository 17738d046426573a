#!/bin/bash
# round 5, call 2: what one SIMD does with the key-tile loop's instruction mix (tools/simd_probe.cpp), and the interleaved key-tile step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=r05b
cd scratch_ab
timeout 300 ./simd_probe > ../gpurun_out/${t}_simd_probe.log 2>&1
L="default=./libenc_default.so il=./libenc_il.so ilp=./libenc_ilp.so sp2=./libenc_sp2.so sp2il=./libenc_sp2il.so ilt=./libenc_ilt.so"
ENC_AB_OUT=../gpurun_out timeout 600 ./enc_ab $L > ../gpurun_out/${t}_enc_ab_p336.log 2>&1
ENC_AB_P=168 ENC_AB_S=3532 timeout 600 ./enc_ab default=./libenc_default.so il=./libenc_il.so ilp=./libenc_ilp.so > ../gpurun_out/${t}_enc_ab_p168.log 2>&1
cd ..
for f in gpurun_out/enc_timing_ilt_*.bin; do python tools/enc_phase_table.py $f; done > gpurun_out/${t}_encoder_phase_table_il.md 2>&1
cat gpurun_out/${t}_simd_probe.log; grep "bench-like" gpurun_out/${t}_enc_ab_p336.log gpurun_out/${t}_enc_ab_p168.log
