#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_runner_golden.py -q -rP > gpurun_out/r03h_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03h_tests.log
timeout 300 python bench.py --config TSFormer_PEMS-BAY --steps 30 --warmup 8 > gpurun_out/r03h_bench_C3.json 2>/dev/null
tail -3 gpurun_out/r03h_tests.log; head -c 300 gpurun_out/r03h_bench_C3.json
