#!/bin/bash
# round 4, GPU call A: box facts; encoder A/B harness at P=168 (parked one-workgroup vs unparked two-workgroup dispatch) and P=336
# (interleaved matrix-instruction order); the new parity tests (C5 at N=4096, the oracle's own encoder at C2/C4, the timed encoder on
# the bench checkpoint); whole GPU suite; step benches C4/C1/C5 with either encoder dispatch; the one-rank RCCL run.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=r04a
( free -g; nproc; rocm-smi --showmeminfo vram | head -8 ) > gpurun_out/${t}_box.log 2>&1
( cd scratch_ab
  ENC_AB_P=168 ENC_AB_S=3532 timeout 300 ./enc_ab park=./libenc_default.so@STEP_ENC_PARK=1 dual=./libenc_default.so@STEP_ENC_PARK=0 ilpark=./libenc_il.so@STEP_ENC_PARK=1 > ../gpurun_out/${t}_enc_ab_p168_s3532.log 2>&1
  ENC_AB_P=168 ENC_AB_S=4096 timeout 300 ./enc_ab park=./libenc_default.so@STEP_ENC_PARK=1 dual=./libenc_default.so@STEP_ENC_PARK=0 > ../gpurun_out/${t}_enc_ab_p168_s4096.log 2>&1
  timeout 300 ./enc_ab base=./libenc_default.so basepark=./libenc_default.so@STEP_ENC_PARK=1 ilpark=./libenc_il.so@STEP_ENC_PARK=1 > ../gpurun_out/${t}_enc_ab_p336.log 2>&1 )
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_kernels.py -m gpu -q -rP --durations=12 -k "c5_4096 or oracle_own_encoder or benchmark_checkpoint" > gpurun_out/${t}_new_parity_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_new_parity_tests.log
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 -k "not c5_4096 and not oracle_own_encoder and not benchmark_checkpoint" > gpurun_out/${t}_gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests.log
b() { # name, env, args
  name=$1; envs=$2; shift 2
  env $envs timeout 400 python bench.py --no-extras --no-cpu-baseline --no-pmc --steps 40 --warmup 10 "$@" > gpurun_out/${t}_bench_$name.json 2> gpurun_out/${t}_bench_$name.err
}
b C4_park STEP_ENC_PARK=1 --config STEP_PEMS07
b C4_dual STEP_ENC_PARK=0 --config STEP_PEMS07
b C1_park STEP_ENC_PARK=1 --config STEP_METR-LA
b C1_dual STEP_ENC_PARK=0 --config STEP_METR-LA
b C5_park STEP_ENC_PARK=1 --config SYNTH_4096 --steps 20
b C5_dual STEP_ENC_PARK=0 --config SYNTH_4096 --steps 20
b C2 X=1
b C2_rccl1 X=1 --force-process-group
b C2_rccl1_q2 GPU_MAX_HW_QUEUES=2 --force-process-group
b C2_rccl1_noshard X=1 --force-process-group --no-shard
for f in gpurun_out/${t}_bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(round(d["value"],1), round(d["ms_per_step"],3), "enc", round(d["roofline"]["ms_per_launch"],3), "host", round(d["host_enqueue_ms_per_step"],2), d.get("data_parallel",{}).get("small_collectives"), d.get("data_parallel",{}).get("per_rank_exposed_wait_ms"))
except Exception as e:
    print("ERR", e)
PY
done
tail -3 gpurun_out/${t}_new_parity_tests.log; tail -3 gpurun_out/${t}_gpu_tests.log
grep -h "median" gpurun_out/${t}_enc_ab_*.log | cut -c1-200
