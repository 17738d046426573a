#!/bin/bash
# round 5, second evidence run (after the encoder's scratch spills were removed and the communicator self-check was added): smoke(), the whole
# GPU suite, the default bench line, the one-rank RCCL lines, the encoder alone against the previous build and round 4's with its counters,
# the bench flow of two data-parallel ranks on one device over gloo (functional), kernel tables of C2 / C4 and the scheduled C2 step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r05zy}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${t}_smoke.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/${t}_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests_full.log
timeout 900 python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err
echo "bench rc $?" >> gpurun_out/${t}_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/${t}_bench_20_steps.json 2> /dev/null
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-pmc --force-process-group --steps 40 --warmup 10 2> gpurun_out/${t}_bench_rccl_one_rank.err | grep '^{"metric"' > gpurun_out/${t}_bench_rccl_one_rank.json
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-pmc --force-process-group --no-shard --steps 40 --warmup 10 2> /dev/null | grep '^{"metric"' > gpurun_out/${t}_bench_rccl_one_rank_noshard.json
(cd scratch_ab && timeout 300 ./enc_ab default=./libenc_default.so prev=./libenc_prev.so r04=./libenc_r04.so > ../gpurun_out/${t}_enc_ab_p336.log 2>&1; ENC_AB_P=168 ENC_AB_S=3532 timeout 300 ./enc_ab default=./libenc_default.so prev=./libenc_prev.so r04=./libenc_r04.so > ../gpurun_out/${t}_enc_ab_p168.log 2>&1)
timeout 600 bash tools/pmc_enc_ab.sh default mem > gpurun_out/${t}_pmc_default.log 2>&1
cp gpurun_out/pmc_ab_default_summary.txt gpurun_out/${t}_encoder_pmc_summary.txt; cp gpurun_out/encoder_pmc.json gpurun_out/${t}_encoder_pmc.json; rm -rf gpurun_out/pmc_ab_*/
WORLD=2 timeout 600 bash tools/bench_dpN_single_device.sh --no-extras --no-pmc --steps 8 --warmup 3 > gpurun_out/${t}_dp2_gloo_single_device.json 2> gpurun_out/${t}_dp2.err
cat gpurun_out/dp_rank0.err gpurun_out/dp_rank1.err 2>/dev/null | tail -20 >> gpurun_out/${t}_dp2.err
WORLD=2 MASTER_PORT=29577 timeout 600 bash tools/bench_dpN_single_device.sh --config TSFormer_PEMS-BAY --no-extras --no-pmc --steps 6 --warmup 2 > gpurun_out/${t}_dp2_gloo_C3_single_device.json 2>> gpurun_out/${t}_dp2.err
prof() { # name, env, bench args...
  name=$1; envs=$2; shift 2
  (cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${t}_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --no-pmc --pretrain-steps 0 "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${t}_prof_$name.err)
  db=$(find gpurun_out/prof_${t}_$name -name '*.db' | head -1)
}
prof C2 STEP_NO_OVERLAP=1 --no-prefetch --steps 20 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C2
prof C4 STEP_NO_OVERLAP=1 --no-prefetch --config STEP_PEMS07 --steps 15 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C4_train_step_no_overlap.md; rm -rf gpurun_out/prof_${t}_C4
prof C2o X=1 --steps 20 --warmup 4
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step.md
python tools/prof_timeline.py $db --anchor adam_clip > gpurun_out/${t}_C2_step_timeline.md; rm -rf gpurun_out/prof_${t}_C2o
tail -2 gpurun_out/${t}_smoke.log; tail -3 gpurun_out/${t}_gpu_tests_full.log; head -c 300 gpurun_out/${t}_bench.json; echo; tail -2 gpurun_out/${t}_bench.err; grep -h "bench-like data dropout 0.1" gpurun_out/${t}_enc_ab_p*.log | cut -c1-120; head -c 300 gpurun_out/${t}_dp2_gloo_single_device.json
