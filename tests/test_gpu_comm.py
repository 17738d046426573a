"""The step's data-parallel collectives on RCCL's C API (csrc/comm.cpp, step_amd/comm.py) on ONE device: a communicator of one rank
(a gpurun box has one GPU, and RCCL refuses two ranks on the same device) -- the library is loaded, the communicator is built from a
unique id, every entry point runs on real streams, and a training step whose exchange goes through it gives the gradients of the step
without a process group.  The N > 1 arithmetic of the same host code runs over gloo in tests/test_abi_and_host.py and
tests/test_gpu_sharded_graph_learner.py; the bytes on xGMI are the driver's 8-GPU run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("gloo", rank=0, world_size=1)          # carries nothing here: one rank
from step_amd import comm as C, _lib
from tests.helpers import load_golden, rel_l2
from tests.test_gpu_step import build_native, inputs_of
from oracle import step_oracle as O
assert C.available()
nc = C.NativeComm()
assert nc.world == 1 and nc.rank == 0 and nc.version >= 20000, nc.version
x = torch.randn(1 << 20, device="cuda"); x0 = x.clone()
nc.allreduce_(x); nc.allreduce_(x, average=True)
d = torch.randn(48, device="cuda", dtype=torch.float64); d0 = d.clone()
nc.allreduce_(d)
b = torch.arange(1000, device="cuda", dtype=torch.int64); b0 = b.clone()
nc.broadcast_(b)
side = torch.cuda.Stream()
with torch.cuda.stream(side):                                   # any stream: the calls take the current one
    nc.grad_allreduce_begin(x[:4096]); nc.grad_allreduce_begin(x[4096:]); nc.grad_allreduce_join()
torch.cuda.synchronize()
assert torch.equal(x, x0) and torch.equal(d, d0) and torch.equal(b, b0)
# error path: a bad handle is rejected with a message, nothing is launched
rc = _lib.lib().step_comm_allreduce(None, _lib.ptr(x), 4, 0, 0, None)
assert rc == 1 and b"communicator" in _lib.lib().step_last_error()

def grads(mode, shard, collectives, name):
    g = load_golden(name)
    m = build_native(g); m.train(); m.backend.dropout = 0.0; m.tsformer.dropout_p = 0.0
    m.matmul_precision = mode
    m._noise_override = g["in.u"]
    if collectives is not None:
        m.enable_native_data_parallel(single_rank_collectives=True, shard_graph_learner=shard, collectives=collectives)
        assert (m._comm is not None) == (collectives == "rccl")
    hist, longh, fut = inputs_of(g)
    pred, theta, knn, coef = m(history_data=hist, long_history_data=longh, future_data=None, batch_seen=0, epoch=1)
    O.step_loss(pred[..., :1] * 150 + 200, fut[..., :1] * 150 + 200, theta, knn, coef).backward()
    torch.cuda.synchronize()
    if shard:
        m.discrete_graph_learning.gather_fc_weight()
    return m._flat_grad.clone().cpu(), float(pred.sum())

for mode, shard in (("f32", False), ("bf16", False), ("bf16", True)):
    name = "step_small" if shard else "step_tiny"          # (a time slice needs >= 128 conv2 columns: 182 there)
    base, pb = grads(mode, False, None, name)
    for coll in ("rccl", "torch"):
        got, pg = grads(mode, shard, coll, name)
        if shard:          # one slice = the whole series, but the layout carries the spare norm slot: compare through the views' total norm
            e = abs(float(got.norm()) - float(base.norm())) / float(base.norm())
        else:
            e = rel_l2(got, base)
        print(f"one-rank step, {mode}{' time-sliced' if shard else ''}, collectives={coll}: gradient vs no process group {e:.1e}")
        assert pg == pb and e < (2e-3 if shard else 1e-6), (mode, shard, coll, e)
nc.close()
dist.destroy_process_group()
print("COMM-OK")
"""


def test_rccl_c_api_collectives_on_one_rank(tmp_path):
    script = tmp_path / "comm_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    print(out[-3000:])
    assert p.returncode == 0 and "COMM-OK" in out, out[-3000:]
