#!/bin/bash
# round 3, GPU call K: clean full GPU test log at the final commit + SQ / HBM counters of the encoder launch (torch-free harness)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/r03k_gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03k_gpu_tests.log
timeout 600 bash tools/pmc_enc_ab.sh default mem > gpurun_out/r03k_pmc.log 2>&1
rm -rf gpurun_out/pmc_ab_default_sq* gpurun_out/pmc_ab_default_mem*
tail -3 gpurun_out/r03k_gpu_tests.log; tail -5 gpurun_out/r03k_pmc.log | cut -c1-200
