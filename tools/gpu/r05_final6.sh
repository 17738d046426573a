#!/bin/bash
# round 5, at the final commit: smoke(), the two-rank stand-in test, the live reference-runner test, the driver's bench command form
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r05zu}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${t}_smoke.log 2>&1
timeout 200 python -m pytest tests/test_gpu_comm_two_ranks.py tests/test_gpu_reference_runner_live.py -m gpu -q > gpurun_out/${t}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_tests.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-pmc > gpurun_out/${t}_bench_20_steps.json 2> /dev/null
tail -1 gpurun_out/${t}_smoke.log; tail -3 gpurun_out/${t}_tests.log; head -c 260 gpurun_out/${t}_bench_20_steps.json
