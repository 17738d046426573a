#!/bin/bash
# round 3, GPU call D: pre-training / sharding tests after the fused feed-forward and norm-slot changes; same-box A/B of the auxiliary
# stream and the split adjacency gradient at C4 / C5; C3 bench + kernel table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03d
timeout 900 python -m pytest tests -m gpu -q -rP -k "pretrain or sharded or attention or ffn" > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_tests.log
ab() { # name, config, steps
  for rep in 1 2; do
    for env in "X=1" "STEP_NO_AUX=1" "STEP_ADJ_PIECES=0"; do
      env $env timeout 300 python bench.py --config $2 --no-extras --no-cpu-baseline --steps $3 --warmup 8 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$1', '$env', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1))"
    done
  done
}
ab C4 STEP_PEMS07 40 > gpurun_out/${tag}_ab_C4.log 2>&1
ab C5 SYNTH_4096 20 > gpurun_out/${tag}_ab_C5.log 2>&1
timeout 300 python bench.py --config TSFormer_PEMS-BAY --steps 30 --warmup 8 > gpurun_out/${tag}_bench_C3.json 2> gpurun_out/${tag}_bench_C3.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_C3 -o p -- python $GRAFT_REPO_ROOT/bench.py --config TSFormer_PEMS-BAY --steps 13 --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_C3.err)
db=$(find gpurun_out/prof_${tag}_C3 -name '*.db' | head -1)
python tools/prof_summary.py $db > gpurun_out/${tag}_C3_pretrain_train_step.md
rm -rf gpurun_out/prof_${tag}_C3
tail -3 gpurun_out/${tag}_tests.log; cat gpurun_out/${tag}_ab_C4.log gpurun_out/${tag}_ab_C5.log; head -c 400 gpurun_out/${tag}_bench_C3.json
