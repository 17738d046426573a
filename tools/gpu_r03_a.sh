#!/bin/bash
# round 3, GPU call A: full GPU test suite, encoder A/B (variants + pool size), the new bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=15 -k "not dropout_on" > gpurun_out/r03a_gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03a_gpu_tests.log
(cd scratch_ab && timeout 300 ./enc_ab default=./libenc_default.so sumsc=./libenc_sumsc.so resmfma=./libenc_resmfma.so prio=./libenc_prio.so old=./libenc_old.so > ../gpurun_out/r03a_enc_ab_pool20.log 2>&1)
(cd scratch_ab && ENC_AB_POOL_LOG2=18 timeout 300 ./enc_ab default=./libenc_default.so old=./libenc_old.so > ../gpurun_out/r03a_enc_ab_pool18.log 2>&1)
timeout 900 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
echo "bench rc $?" >> gpurun_out/r03a_bench.err
tail -3 gpurun_out/r03a_gpu_tests.log; grep -c . gpurun_out/r03a_enc_ab_pool20.log; head -c 600 gpurun_out/r03a_bench.json
