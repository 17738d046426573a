#!/bin/bash
# round 3, call M: the pre-training attention kernels after vectorised fills, staged stores and the 16-bit dropout stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pretrain.py -q -rP -m gpu > gpurun_out/r03m_pretrain_tests.log 2>&1; echo "tests exit $?"
tail -5 gpurun_out/r03m_pretrain_tests.log
timeout 200 python tools/bench_pt_attention.py 2>/dev/null > gpurun_out/r03m_attention.log; cat gpurun_out/r03m_attention.log
timeout 400 python bench.py --config TSFormer_PEMS-BAY --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03m_c3_bench.json; cat gpurun_out/r03m_c3_bench.json
