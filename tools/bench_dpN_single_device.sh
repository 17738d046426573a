#!/bin/bash
# Functional check of the --gpus N path of bench.py on a ONE-GPU box: N ranks, all on device 0, gloo group (the driver's RCCL runs are
# the performance measurement; this proves the exchange steps of the data-parallel path end to end).
# Usage: WORLD=8 tools/bench_dpN_single_device.sh [bench args]      (default WORLD=2)
W=${WORLD:-2}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29533} WORLD_SIZE=$W LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
pids=""
for r in $(seq 1 $((W - 1))); do
  RANK=$r python bench.py --gpus $W --backend gloo --no-cpu-baseline --no-loader-figure "$@" > /dev/null 2> gpurun_out/dp_rank$r.err &
  pids="$pids $!"
done
RANK=0 python bench.py --gpus $W --backend gloo --no-cpu-baseline --no-loader-figure "$@" 2> gpurun_out/dp_rank0.err | grep '^{'
rc=${PIPESTATUS[0]}
for p in $pids; do wait $p || rc=$?; done
exit $rc
