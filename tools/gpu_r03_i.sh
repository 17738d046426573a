#!/bin/bash
# round 3, GPU call I: persistent encoder launch (one workgroup per compute unit looping over sequences; 240 / 224 leave compute units to
# the second stream) -- kernel alone in the torch-free harness, whole step with STEP_HIP_LIB variants
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=r03i
(cd scratch_ab && timeout 300 ./enc_ab default=./libenc_default.so persist256=./libenc_persist256.so > ../gpurun_out/${tag}_enc_ab.log 2>&1)
for rep in 1 2; do
  for lib in libstep_hip.so libstep_hip_persist256.so libstep_hip_persist240.so libstep_hip_persist224.so; do
    STEP_HIP_LIB=step_amd/$lib timeout 300 python bench.py --no-extras --no-cpu-baseline --pretrain-steps 0 --steps 60 --warmup 8 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('$lib', 'ms_per_step', round(d['ms_per_step'], 3), round(d['value'], 1), 'enc', round(d['roofline']['ms_per_launch'], 3))"
  done
done > gpurun_out/${tag}_ab_C2.log 2>&1
grep -E "median|DROPOUT|f16 :" gpurun_out/${tag}_enc_ab.log | cut -c1-170; cat gpurun_out/${tag}_ab_C2.log
