#!/bin/bash
# round 5: two ranks on one device through the RCCL stand-in (tests/fake_rccl) first, then the whole GPU suite at the final library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r05zw}
timeout 600 python -m pytest tests/test_gpu_comm_two_ranks.py tests/test_gpu_comm.py -m gpu -q -rP > gpurun_out/${t}_comm_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_comm_tests.log
timeout 1200 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/${t}_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests_full.log
tail -40 gpurun_out/${t}_comm_tests.log | cut -c1-400; tail -4 gpurun_out/${t}_gpu_tests_full.log
