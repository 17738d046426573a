#!/bin/bash
# round 3, call AN: the data-parallel bench flow at the final code (2 and 4 ranks over gloo on ONE device: functional check of the time
# slices, the 2 MB all-reduce after the auxiliary-stream join, the small sums, the norm slot), C2 and the C3 pre-training path
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03an
WORLD=2 timeout 600 tools/bench_dpN_single_device.sh --steps 12 --warmup 4 > gpurun_out/${tag}_dp2_gloo.json 2> gpurun_out/${tag}_dp2_gloo.err; echo "dp2 rc $?"
WORLD=4 timeout 600 tools/bench_dpN_single_device.sh --steps 8 --warmup 3 > gpurun_out/${tag}_dp4_gloo.json 2> gpurun_out/${tag}_dp4_gloo.err; echo "dp4 rc $?"
WORLD=2 timeout 600 tools/bench_dpN_single_device.sh --config TSFormer_PEMS-BAY --steps 6 --warmup 2 > gpurun_out/${tag}_dp2_gloo_C3.json 2> gpurun_out/${tag}_dp2_gloo_C3.err; echo "dp2 C3 rc $?"
for f in dp2_gloo dp4_gloo dp2_gloo_C3; do tail -1 gpurun_out/${tag}_$f.json | cut -c1-260; tail -2 gpurun_out/${tag}_$f.err | cut -c1-200; done
