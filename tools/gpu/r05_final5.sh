#!/bin/bash
# round 5: after the rank-identical clip norm of the time-sliced path -- the two-rank stand-in test three times (it was the flaky one), the
# time-slice tests with their new replica-identity assertions, the optimizer / communicator tests, and the two-rank bench flow over gloo
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r05zv}
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_comm_two_ranks.py -m gpu -q -rP > gpurun_out/${t}_two_ranks_$i.log 2>&1; echo "pytest rc $?" >> gpurun_out/${t}_two_ranks_$i.log; done
timeout 600 python -m pytest tests/test_gpu_sharded_graph_learner.py tests/test_gpu_comm.py tests/test_gpu_step.py tests/test_gpu_graphed_step.py -m gpu -q -x > gpurun_out/${t}_related_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_related_tests.log
WORLD=2 timeout 300 bash tools/bench_dpN_single_device.sh --no-extras --no-pmc --steps 8 --warmup 3 > gpurun_out/${t}_dp2_gloo_single_device.json 2> gpurun_out/${t}_dp2.err
for i in 1 2 3; do grep -h "time slices step\|pytest rc\|passed\|failed\|differs" gpurun_out/${t}_two_ranks_$i.log | cut -c1-260 | tail -8; done
tail -4 gpurun_out/${t}_related_tests.log; head -c 250 gpurun_out/${t}_dp2_gloo_single_device.json
