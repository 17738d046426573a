#!/bin/bash
# round 4, GPU call B: encoder A/B (last-key-tile shortcut) at P=168 and P=336; encoder tests; kernel tables (no overlap) of C2 / C4 / C5
# at the current code with their roofline tables; kernel trace of the one-rank RCCL step.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=r04b
( cd scratch_ab
  ENC_AB_P=168 ENC_AB_S=3532 timeout 300 ./enc_ab notail=./libenc_notail.so tail=./libenc_default.so > ../gpurun_out/${t}_enc_ab_p168_s3532.log 2>&1
  timeout 300 ./enc_ab notail=./libenc_notail.so tail=./libenc_default.so > ../gpurun_out/${t}_enc_ab_p336.log 2>&1 )
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "encoder" > gpurun_out/${t}_encoder_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_encoder_tests.log
prof() { # name, env, bench args...
  name=$1; envs=$2; shift 2
  (cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${t}_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${t}_prof_$name.err)
  db=$(find gpurun_out/prof_${t}_$name -name '*.db' | head -1)
}
prof C2 STEP_NO_OVERLAP=1 --steps 20 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C2
prof C4 STEP_NO_OVERLAP=1 --config STEP_PEMS07 --steps 15 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C4_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C4
prof C5 STEP_NO_OVERLAP=1 --config SYNTH_4096 --steps 10 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C5_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C5
prof C1 STEP_NO_OVERLAP=1 --config STEP_METR-LA --steps 20 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C1_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C1
prof rccl1 X=1 --steps 12 --warmup 4 --force-process-group --no-shard
python tools/prof_summary.py $db > gpurun_out/${t}_C2_rccl1_noshard_train_step.md
python tools/prof_timeline.py $db > gpurun_out/${t}_C2_rccl1_noshard_timeline.md; rm -rf gpurun_out/prof_${t}_rccl1
prof C2o X=1 --steps 12 --warmup 4
python tools/prof_timeline.py $db > gpurun_out/${t}_C2_step_timeline.md; rm -rf gpurun_out/prof_${t}_C2o
tail -2 gpurun_out/${t}_encoder_tests.log
grep -h "median" gpurun_out/${t}_enc_ab_*.log | cut -c1-160
