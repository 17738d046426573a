#!/bin/bash
# the evidence run without the kernel tables of the STEP configs (unchanged since `r04_final.sh r04zz`): smoke(), the whole GPU suite, the default
# bench line, the kernel table of config C3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04zz2}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${t}_smoke.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/${t}_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests_full.log
timeout 900 python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err
echo "bench rc $?" >> gpurun_out/${t}_bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${t}_C3 -o p -- python $GRAFT_REPO_ROOT/bench.py --config TSFormer_PEMS-BAY --no-extras --no-cpu-baseline --no-pmc --steps 12 --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${t}_prof_C3.err)
db=$(find gpurun_out/prof_${t}_C3 -name '*.db' | head -1)
python tools/prof_summary.py $db > gpurun_out/${t}_C3_pretrain_train_step.md; rm -rf gpurun_out/prof_${t}_C3
tail -2 gpurun_out/${t}_gpu_tests_full.log; cut -c1-200 gpurun_out/${t}_bench.json
