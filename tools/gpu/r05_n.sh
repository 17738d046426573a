#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05n}
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err
echo "bench rc $?" >> gpurun_out/${t}_bench.err
timeout 600 python -m pytest tests/test_gpu_step.py tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/${t}_tests.log 2>&1
tail -3 gpurun_out/${t}_tests.log; tail -3 gpurun_out/${t}_bench.err; cut -c1-400 gpurun_out/${t}_bench.json
