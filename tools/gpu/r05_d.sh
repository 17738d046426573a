#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=r05d
cd scratch_ab
L="default=./libenc_default.so p0=./libenc_p0.so all=./libenc_all.so p2=./libenc_p2.so p3=./libenc_p3.so p4=./libenc_p4.so"
timeout 600 ./enc_ab $L > ../gpurun_out/${t}_enc_ab_p336.log 2>&1
ENC_AB_P=168 ENC_AB_S=3532 timeout 600 ./enc_ab $L > ../gpurun_out/${t}_enc_ab_p168.log 2>&1
cd ..
grep "bench-like" gpurun_out/${t}_enc_ab_p336.log gpurun_out/${t}_enc_ab_p168.log
