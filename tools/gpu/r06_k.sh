#!/bin/bash
# round 6: the whole GPU suite, smoke() and the default bench line (the driver's command) at the current commit -> gpurun_out/<tag>_*
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t=${1:-r06_k}
python -m pytest tests -m gpu -q -rP --durations=8 > gpurun_out/${t}_gpu_tests_full.log 2>&1; echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests_full.log; tail -3 gpurun_out/${t}_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${t}_smoke.log 2>&1; tail -1 gpurun_out/${t}_smoke.log
python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err; echo "bench rc $?" >> gpurun_out/${t}_bench.err; head -c 300 gpurun_out/${t}_bench.json
