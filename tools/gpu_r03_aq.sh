#!/bin/bash
# round 3, GPU call AQ (evidence at HEAD): smoke(), the full GPU test log, the default bench line (driver's command), kernel tables of C2
# (with the streams overlapping, training steps only) and of C2 / C4 / C5 without overlap, the C2 roofline table, the C2 step timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03aq
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/${tag}_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_gpu_tests_full.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc $?" >> gpurun_out/${tag}_bench.err
prof() { # name, env, bench args...
  name=$1; envs=$2; shift 2
  (cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_$name.err)
  db=$(find gpurun_out/prof_${tag}_$name -name '*.db' | head -1)
}
prof C2 STEP_NO_OVERLAP=1 --steps 20 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${tag}_C2_train_step_no_overlap.md; rm -rf gpurun_out/prof_${tag}_C2
python tools/roofline_table.py gpurun_out/${tag}_C2_train_step_no_overlap.md > gpurun_out/${tag}_roofline_table.md 2> gpurun_out/${tag}_roofline.err
prof C4 STEP_NO_OVERLAP=1 --config STEP_PEMS07 --steps 15 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${tag}_C4_train_step_no_overlap.md; rm -rf gpurun_out/prof_${tag}_C4
prof C2o X=1 --steps 20 --warmup 4
python tools/prof_summary.py $db > gpurun_out/${tag}_C2_train_step.md
python tools/prof_timeline.py $db > gpurun_out/${tag}_C2_step_timeline.md; rm -rf gpurun_out/prof_${tag}_C2o
prof C4o X=1 --config STEP_PEMS07 --steps 12 --warmup 4
python tools/prof_timeline.py $db > gpurun_out/${tag}_C4_step_timeline.md; rm -rf gpurun_out/prof_${tag}_C4o
tail -2 gpurun_out/${tag}_smoke.log; tail -3 gpurun_out/${tag}_gpu_tests_full.log; head -c 400 gpurun_out/${tag}_bench.json; echo; tail -2 gpurun_out/${tag}_bench.err
