#!/bin/bash
# evidence run: smoke(), the whole GPU suite, the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
t=${1:-r05e}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${t}_smoke.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/${t}_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests_full.log
timeout 900 python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err
echo "bench rc $?" >> gpurun_out/${t}_bench.err
tail -3 gpurun_out/${t}_gpu_tests_full.log; cut -c1-300 gpurun_out/${t}_bench.json; tail -2 gpurun_out/${t}_smoke.log
