#!/bin/bash
# Round-2 measurement batch (run on the GPU box through gpurun): default bench line, rocprofv3 kernel tables of C2 / C4 / C5,
# HBM-traffic counters of the encoder.  Everything lands in gpurun_out/ (copy what is to be kept into profiles/).
tag=${1:-r02_f}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench_C2.json 2> gpurun_out/${tag}_bench_C2.err
prof() { # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_$name -o p -- python bench.py --no-cpu-baseline --no-loader-figure "$@" > gpurun_out/${tag}_bench_${name}_profiled.json 2> gpurun_out/${tag}_prof_$name.err
  db=$(find gpurun_out/prof_${tag}_$name -name '*.db' | head -1)
  python tools/prof_summary.py $db > gpurun_out/${tag}_${name}_train_step.md
  rm -rf gpurun_out/prof_${tag}_$name
}
prof C2 --steps 20 --warmup 3
prof C4 --config STEP_PEMS07 --steps 15 --warmup 3
prof C5 --config SYNTH_4096 --steps 8 --warmup 2
python bench.py --config STEP_PEMS07 --no-cpu-baseline > gpurun_out/${tag}_bench_C4.json 2> gpurun_out/${tag}_bench_C4.err
python bench.py --config SYNTH_4096 --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/${tag}_bench_C5.json 2> gpurun_out/${tag}_bench_C5.err
tools/pmc_enc_ab.sh default mem > gpurun_out/${tag}_pmc.log 2>&1
rm -rf gpurun_out/pmc_ab_default_sq* gpurun_out/pmc_ab_default_mem*
