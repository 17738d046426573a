#!/bin/bash
# round 5, evidence at the final code: smoke(), the whole GPU suite, the default bench line (the driver's command), the one-rank RCCL line,
# the encoder alone (harness) with its counters, kernel tables (streams not overlapping, frozen branch inline) of C2 / C4 / C5 / C1 / C3 with
# their roofline tables, and the C2 step as scheduled (prefetch + persistent encoder): kernel table + timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r05zz}
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${t}_smoke.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/${t}_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests_full.log
timeout 1200 python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err
echo "bench rc $?" >> gpurun_out/${t}_bench.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/${t}_bench_20_steps.json 2> /dev/null
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc --force-process-group --steps 40 --warmup 10 > gpurun_out/${t}_bench_rccl_one_rank.json 2> gpurun_out/${t}_bench_rccl_one_rank.err
timeout 600 python bench.py --no-extras --no-cpu-baseline --no-pmc --force-process-group --no-shard --steps 40 --warmup 10 > gpurun_out/${t}_bench_rccl_one_rank_noshard.json 2> /dev/null
timeout 600 python bench.py --config STEP_METR-LA --no-extras --no-pmc > gpurun_out/${t}_bench_C1.json 2> /dev/null
(cd scratch_ab && timeout 300 ./enc_ab default=./libenc_default.so r04=./libenc_r04.so > ../gpurun_out/${t}_enc_ab_p336.log 2>&1; ENC_AB_P=168 ENC_AB_S=3532 timeout 300 ./enc_ab default=./libenc_default.so r04=./libenc_r04.so > ../gpurun_out/${t}_enc_ab_p168.log 2>&1)
timeout 600 bash tools/pmc_enc_ab.sh default mem > gpurun_out/${t}_pmc_default.log 2>&1
cp gpurun_out/pmc_ab_default_summary.txt gpurun_out/${t}_encoder_pmc_summary.txt; cp gpurun_out/encoder_pmc.json gpurun_out/${t}_encoder_pmc.json; rm -rf gpurun_out/pmc_ab_*/
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
prof C3 X=1 --config TSFormer_PEMS-BAY --steps 12 --warmup 3
python tools/prof_summary.py $db > gpurun_out/${t}_C3_pretrain_train_step.md; rm -rf gpurun_out/prof_${t}_C3
prof C2o X=1 --steps 20 --warmup 4
python tools/prof_summary.py $db > gpurun_out/${t}_C2_train_step.md
python tools/prof_timeline.py $db --anchor adam_clip > gpurun_out/${t}_C2_step_timeline.md; rm -rf gpurun_out/prof_${t}_C2o
for c in C1:STEP_METR-LA C2:STEP_PEMS04 C4:STEP_PEMS07 C5:SYNTH_4096; do
  python tools/roofline_table.py gpurun_out/${t}_${c%%:*}_train_step_no_overlap.md --config ${c##*:} --json gpurun_out/${t}_kernel_roofline.json > gpurun_out/${t}_${c%%:*}_roofline_table.md 2>> gpurun_out/${t}_roofline.err
done
python tools/roofline_table.py gpurun_out/${t}_C3_pretrain_train_step.md --config TSFormer_PEMS-BAY --json gpurun_out/${t}_kernel_roofline.json > gpurun_out/${t}_C3_roofline_table.md 2>> gpurun_out/${t}_roofline.err
tail -2 gpurun_out/${t}_smoke.log; tail -3 gpurun_out/${t}_gpu_tests_full.log; head -c 400 gpurun_out/${t}_bench.json; echo; tail -2 gpurun_out/${t}_bench.err
