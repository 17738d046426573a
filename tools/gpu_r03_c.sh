#!/bin/bash
# round 3, GPU call C: full GPU test suite with the tests' printed figures, the default bench line, a 2-rank functional run of the
# data-parallel bench path on one device (gloo), the rocprofv3 kernel table of the headline command, encoder HBM counters of C4 / C5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-r03c}
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/${tag}_gpu_tests.log 2>&1
echo "pytest rc $?" >> gpurun_out/${tag}_gpu_tests.log
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc $?" >> gpurun_out/${tag}_bench.err
WORLD=2 timeout 600 tools/bench_dpN_single_device.sh --steps 20 --warmup 5 > gpurun_out/${tag}_dp2_gloo.json 2> gpurun_out/${tag}_dp2_gloo.err
echo "dp2 rc $?" >> gpurun_out/${tag}_dp2_gloo.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_C2 -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_C2_profiled.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_C2.err)
db=$(find gpurun_out/prof_${tag}_C2 -name '*.db' | head -1)
python tools/prof_summary.py $db > gpurun_out/${tag}_C2_train_step.md
rm -rf gpurun_out/prof_${tag}_C2
timeout 600 python - > gpurun_out/${tag}_encoder_pmc_other_configs.json 2> gpurun_out/${tag}_pmc_other.err <<'PY'
import json, bench
out = {}
for name in ("STEP_PEMS07", "SYNTH_4096"):
    cfg = bench.CONFIGS[name]
    t = bench.live_pmc_traffic(name, cfg["B"], None)
    fl = bench.encoder_flops(cfg, cfg["B"])
    out[name] = {"traffic": t, "algorithmic_flop_per_launch": fl, "compulsory_bytes": cfg["B"] * cfg["N"] * (cfg["L"] * 4 + cfg["L"] // 12 * 96 * 2)}
print(json.dumps(out))
PY
tail -3 gpurun_out/${tag}_gpu_tests.log; head -c 300 gpurun_out/${tag}_bench.json; echo; head -c 300 gpurun_out/${tag}_dp2_gloo.json; echo; head -5 gpurun_out/${tag}_C2_train_step.md
