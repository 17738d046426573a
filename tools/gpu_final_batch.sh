#!/bin/bash
# Final evidence batch of a round (run on the GPU box through gpurun): the full GPU test suite, smoke(), then everything
# tools/gpu_profiles_r02.sh collects plus the other configs' bench lines and the kernel tables without stream overlap.
#   usage: tools/gpu_final_batch.sh <tag>
tag=${1:-r02_final}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -s > gpurun_out/${tag}_gpu_tests_full.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/${tag}_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/gpu_profiles_r02.sh $tag
prof() { # name, env, bench args...
  name=$1; shift; envs=$1; shift
  env $envs rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_$name -o p -- python bench.py --no-cpu-baseline --no-loader-figure "$@" > /dev/null 2> gpurun_out/${tag}_prof_$name.err
  db=$(find gpurun_out/prof_${tag}_$name -name '*.db' | head -1)
  python tools/prof_summary.py $db > gpurun_out/${tag}_${name}.md
  rm -rf gpurun_out/prof_${tag}_$name
}
prof C2_train_step_no_overlap STEP_NO_OVERLAP=1 --steps 20 --warmup 3
prof C5_train_step_no_overlap STEP_NO_OVERLAP=1 --config SYNTH_4096 --steps 8 --warmup 2
prof C3_pretrain_train_step X=1 --config TSFormer_PEMS-BAY --steps 10 --warmup 3
python bench.py --config STEP_METR-LA --no-cpu-baseline > gpurun_out/${tag}_bench_C1.json 2>/dev/null
python bench.py --config TSFormer_PEMS-BAY --no-cpu-baseline > gpurun_out/${tag}_bench_C3.json 2>/dev/null
python bench.py --forward-only --no-cpu-baseline > gpurun_out/${tag}_bench_C2_validation_forward.json 2>/dev/null
python bench.py --matmul f32 --no-cpu-baseline > gpurun_out/${tag}_bench_C2_f32mode.json 2>/dev/null
ls gpurun_out | grep -c ${tag}
