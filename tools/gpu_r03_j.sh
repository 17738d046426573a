#!/bin/bash
# round 3, GPU call J: per-kernel tables without stream overlap (accurate durations) for C2 / C4 / C5 + the C2 roofline table; smoke();
# the time-slice tests after the band fix; the final default bench line at HEAD
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03j
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
timeout 600 python -m pytest tests/test_gpu_sharded_graph_learner.py -q -rP > gpurun_out/${tag}_sharded_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_sharded_tests.log
prof() { # name, filter-kernel, bench args...
  name=$1; shift
  (cd /tmp && STEP_NO_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --pretrain-steps 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_$name.err)
  db=$(find gpurun_out/prof_${tag}_$name -name '*.db' | head -1)
  python tools/prof_summary.py $db > gpurun_out/${tag}_${name}_train_step_no_overlap.md
  rm -rf gpurun_out/prof_${tag}_$name
}
prof C2 --steps 20 --warmup 3
prof C4 --config STEP_PEMS07 --steps 15 --warmup 3
prof C5 --config SYNTH_4096 --steps 8 --warmup 2
python tools/roofline_table.py gpurun_out/${tag}_C2_train_step_no_overlap.md > gpurun_out/${tag}_roofline_table.md 2> gpurun_out/${tag}_roofline.err
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc $?" >> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_smoke.log | tail -2; tail -3 gpurun_out/${tag}_sharded_tests.log; head -c 300 gpurun_out/${tag}_bench.json; echo; head -12 gpurun_out/${tag}_roofline_table.md
