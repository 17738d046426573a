#!/bin/bash
# round 4, call n: (1) which GPU_MAX_HW_QUEUES settings crash a GraphedTrainStep replay (the runtime reads the variable at initialisation;
# the probe removes it from os.environ afterwards so that the constructor's guard does not fire); (2) the whole GPU suite again.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
t=${1:-r04n}
cat > /tmp/probe.py <<'PY'
import os, sys, faulthandler
faulthandler.enable()
import torch
torch.cuda.init(); torch.zeros(1, device="cuda")
q = os.environ.pop("GPU_MAX_HW_QUEUES", None)
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tests.test_gpu_graphed_step import _setup
from tests.test_gpu_step import inputs_of
from step_amd import GraphedTrainStep
g, model, opt = _setup("bf16", dropout=True)
mean, std = [float(x) for x in g["meta.scaler"]]
step = GraphedTrainStep(model, opt, inputs_of(g), scaler=(mean, std), epoch=1, warmup=2)
for _ in range(5):
    loss = step()
torch.cuda.synchronize()
print(f"queues={q}: 5 replays ok, loss {float(loss):.4f}", flush=True)
PY
for q in 1 2 3 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python /tmp/probe.py 2>&1 | grep -v amdgpu.ids | grep -E "queues=|Segmentation|Error|error" | head -3
  echo "GPU_MAX_HW_QUEUES=$q rc ${PIPESTATUS[0]}"
done > gpurun_out/${t}_graph_replay_hw_queues.log 2>&1
timeout 300 python /tmp/probe.py 2>&1 | grep -E "queues=|Segmentation" >> gpurun_out/${t}_graph_replay_hw_queues.log
timeout 1800 python -m pytest tests -m gpu -q -rP --durations=10 > gpurun_out/${t}_gpu_tests_full.log 2>&1
echo "pytest rc $?" >> gpurun_out/${t}_gpu_tests_full.log
tail -5 gpurun_out/${t}_gpu_tests_full.log; cat gpurun_out/${t}_graph_replay_hw_queues.log
