#!/bin/bash
# Functional check of the --gpus 2 path of bench.py on a ONE-GPU box: two ranks, both on device 0, gloo group (the driver's RCCL runs are
# the performance measurement; this proves the exchange steps of the data-parallel path end to end).  Usage: tools/bench_dp2_single_device.sh [bench args]
export MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29533} WORLD_SIZE=2 LOCAL_RANK=0 HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=1 python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-loader-figure "$@" > /dev/null 2> gpurun_out/dp2_rank1.err &
pid=$!
RANK=0 python bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-loader-figure "$@" 2> gpurun_out/dp2_rank0.err
rc=$?
wait $pid || rc=$?
exit $rc
