#!/bin/bash
# round 3, GPU call F (evidence run): full GPU test suite with the tests' figures, default bench line, kernel table of the headline
# command (training steps only), the other configs as their own bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-r03f}
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/${tag}_gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_gpu_tests.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc $?" >> gpurun_out/${tag}_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_C2 -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 40 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_C2_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_C2.err)
db=$(find gpurun_out/prof_${tag}_C2 -name '*.db' | head -1)
python tools/prof_summary.py $db --after-last attn_mfma_bwd > gpurun_out/${tag}_C2_train_step.md
rm -rf gpurun_out/prof_${tag}_C2
for cfg in STEP_METR-LA STEP_PEMS07 SYNTH_4096 TSFormer_PEMS-BAY; do
  timeout 400 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/${tag}_bench_$cfg.json 2> gpurun_out/${tag}_bench_$cfg.err
done
timeout 300 python bench.py --forward-only --no-extras --no-cpu-baseline --steps 60 > gpurun_out/${tag}_bench_C2_validation_forward.json 2> /dev/null
timeout 300 python bench.py --matmul f32 --no-extras --no-cpu-baseline --steps 60 > gpurun_out/${tag}_bench_C2_f32mode.json 2> /dev/null
tail -3 gpurun_out/${tag}_gpu_tests.log; head -c 250 gpurun_out/${tag}_bench.json; echo; head -8 gpurun_out/${tag}_C2_train_step.md
for f in gpurun_out/${tag}_bench_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), round(d['ms_per_step'],3))"; done
