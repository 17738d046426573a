"""SURVEY.md 8(f) row 2 on the device: two / four data-parallel ranks (gloo group, all on cuda:0) with the graph learner in time slices
against the same two ranks with the whole graph learner, and against one process evaluating both batches (tests/shard_worker.py)."""
import os
import subprocess
import sys

import pytest

from tests.test_abi_and_host import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,storage", [(2, "bf16"), (4, "bf16"), (2, "f32")])
def test_time_sliced_graph_learner_matches_unsharded_data_parallel(world, storage):
    """storage = "f32": the same comparison with STEP_DGL_F32_STORAGE=1 (f32 conv activations, bf16 GEMM operands) and tighter bands --
    separates the slice / halo logic from the bf16 rounding of the channels-last rows."""
    script = os.path.join(ROOT, "tests", "shard_worker.py")
    for attempt in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if storage == "f32":
            env["STEP_DGL_F32_STORAGE"] = "1"
        procs = [subprocess.Popen([sys.executable, script, ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                 for r in range(world)]
        outs = [p.communicate(timeout=600)[0].decode() for p in procs]
        if all(p.returncode == 0 for p in procs):
            break
        if attempt == 0 and not any("AssertionError" in o or "Error" in o for o in outs):
            continue
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o[-4000:]
    print("\n".join(o[-1500:] for o in outs))
