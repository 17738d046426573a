#!/bin/bash
# round 6, second session: the N = 64 horizon-12 test as a three-run mean; kernel + roofline tables of C2 (no overlap; as scheduled + timeline), C4, C3 at the final code
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r06_zx}
python -m pytest tests/test_gpu_training_parity.py -q -rP -k "test_h12_mae_parity and not pems04 and not dropout" > gpurun_out/${t}_n1_three_run_mean.log 2>&1; grep -E "^N1|passed|failed" gpurun_out/${t}_n1_three_run_mean.log | cut -c1-400
prof() { # name, env, bench args...
  name=$1; envs=$2; shift 2
  (cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${t}_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${t}_prof_$name.err)
  db=$(find gpurun_out/prof_${t}_$name -name '*.db' | head -1)
}
prof C2 STEP_NO_OVERLAP=1 --no-prefetch --steps 20 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C2
prof C4 STEP_NO_OVERLAP=1 --no-prefetch --config STEP_PEMS07 --steps 15 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C4_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C4
prof C3 X=1 --config TSFormer_PEMS-BAY --steps 12 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C3_pretrain_train_step.md; rm -rf gpurun_out/prof_${t}_C3
prof C2o X=1 --steps 20 --warmup 4
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step.md
python tools/prof_timeline.py $db --anchor adam_clip > gpurun_out/${t}_C2_step_timeline.md; rm -rf gpurun_out/prof_${t}_C2o
for c in C2:STEP_PEMS04 C4:STEP_PEMS07; do
  python tools/roofline_table.py gpurun_out/${t}_${c%%:*}_train_step_no_overlap.md --config ${c##*:} --json gpurun_out/${t}_kernel_roofline.json > gpurun_out/${t}_${c%%:*}_roofline_table.md 2>> gpurun_out/${t}_roofline.err
done
python tools/roofline_table.py gpurun_out/${t}_C3_pretrain_train_step.md --config TSFormer_PEMS-BAY --json gpurun_out/${t}_kernel_roofline.json > gpurun_out/${t}_C3_roofline_table.md 2>> gpurun_out/${t}_roofline.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/${t}_bench_20_steps.json 2> /dev/null
grep "Sum of kernel" gpurun_out/${t}_C*_roofline_table.md; head -c 250 gpurun_out/${t}_bench_20_steps.json; echo; grep "attn2\|tsformer_encoder" gpurun_out/${t}_C*_t*.md | cut -c1-200
