#!/bin/bash
# round 6, evidence at the final code: kernel tables (streams not overlapping, frozen branch inline) of C2 / C4 / C5 / C1 with their roofline
# tables, the C2 step as scheduled (prefetch + persistent encoder + kNN stream): kernel table + timeline, the runner-driven loop, the one-rank
# RCCL line.  (smoke(), the whole GPU suite and the default bench line: tools/gpu/r06_k -- scratch_ab/r06_k.sh -- of the same commit.)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r06_z}
prof() { # name, env, bench args...
  name=$1; envs=$2; shift 2
  (cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${t}_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${t}_prof_$name.err)
  db=$(find gpurun_out/prof_${t}_$name -name '*.db' | head -1)
}
prof C2 STEP_NO_OVERLAP=1 --no-prefetch --steps 20 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C2
prof C4 STEP_NO_OVERLAP=1 --no-prefetch --config STEP_PEMS07 --steps 15 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C4_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C4
prof C5 STEP_NO_OVERLAP=1 --no-prefetch --config SYNTH_4096 --steps 10 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C5_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C5
prof C1 STEP_NO_OVERLAP=1 --no-prefetch --config STEP_METR-LA --steps 20 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C1_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C1
prof C2o X=1 --steps 20 --warmup 4
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step.md
python tools/prof_timeline.py $db --anchor adam_clip > gpurun_out/${t}_C2_step_timeline.md; rm -rf gpurun_out/prof_${t}_C2o
for c in C1:STEP_METR-LA C2:STEP_PEMS04 C4:STEP_PEMS07 C5:SYNTH_4096; do
  python tools/roofline_table.py gpurun_out/${t}_${c%%:*}_train_step_no_overlap.md --config ${c##*:} --json gpurun_out/${t}_kernel_roofline.json > gpurun_out/${t}_${c%%:*}_roofline_table.md 2>> gpurun_out/${t}_roofline.err
done
for v in "--runner native --dataset device --loss native" "--runner native --dataset device" "--runner reference --dataset host --iters 24" "--runner native --dataset host --iters 24"; do
  python tools/runner_feed_bench.py $v 2>/dev/null | tail -1 >> gpurun_out/${t}_runner_feed.jsonl
done
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc --force-process-group --steps 40 --warmup 10 > gpurun_out/${t}_bench_rccl_one_rank.json 2> gpurun_out/${t}_bench_rccl_one_rank.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/${t}_bench_20_steps.json 2> /dev/null
grep -c . gpurun_out/${t}_runner_feed.jsonl; head -c 300 gpurun_out/${t}_bench_rccl_one_rank.json; echo; grep "Sum of kernel" gpurun_out/${t}_C*_roofline_table.md
