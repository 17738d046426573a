#!/bin/bash
# round 5, last call: the encoder alone (this build / the build before the scratch spills were removed / round 4's kernel), the forecast
# origins through pinned vs pageable host memory (same box, alternated), and the default bench line at the final code
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r05zx}
(cd scratch_ab && timeout 300 ./enc_ab default=./libenc_default.so before=./libenc_before.so r04=./libenc_r04.so > ../gpurun_out/${t}_enc_ab_p336.log 2>&1; ENC_AB_P=168 ENC_AB_S=3532 timeout 300 ./enc_ab default=./libenc_default.so before=./libenc_before.so r04=./libenc_r04.so > ../gpurun_out/${t}_enc_ab_p168.log 2>&1)
for rep in 1 2; do
  for v in "" "--pageable-origins"; do
    for c in STEP_PEMS04 STEP_PEMS07; do
      if [ $rep == 2 ] && [ $c == STEP_PEMS07 ]; then continue; fi
      echo "[$c $v]" >> gpurun_out/${t}_origins_ab.log
      timeout 300 python bench.py --config $c --no-extras --no-cpu-baseline --no-pmc --steps 60 --warmup 15 $v 2> /dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'windows/s', round(d['ms_per_step'],3), 'ms/step', d['step_ms'], 'host enqueue', round(d['host_enqueue_ms_per_step'],3))" >> gpurun_out/${t}_origins_ab.log
    done
  done
done
timeout 900 python bench.py > gpurun_out/${t}_bench.json 2> gpurun_out/${t}_bench.err
echo "bench rc $?" >> gpurun_out/${t}_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/${t}_bench_20_steps.json 2> /dev/null
grep -h "bench-like data dropout 0.1" gpurun_out/${t}_enc_ab_p*.log | cut -c1-110; cat gpurun_out/${t}_origins_ab.log; head -c 300 gpurun_out/${t}_bench.json; echo
