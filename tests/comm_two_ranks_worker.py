"""Worker of tests/test_gpu_comm_two_ranks.py: one process per data-parallel rank, all on cuda:0, a gloo group for the rendezvous, and
$STEP_RCCL_LIB pointing at tests/fake_rccl (a librccl stand-in that moves data between processes sharing one GPU).  What runs here with
nranks > 1 for the first time: NativeComm's unique-id exchange and self-check, the overlapped mean all-reduce of the flat gradient
(step_grad_allreduce_begin / _join on the probed side stream), and the stream-ordered f32 / f64 sums of the time-sliced graph learner --
each compared with the SAME step whose collectives go through torch.distributed (gloo), which tests/shard_worker.py validates on its own."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from oracle import step_oracle as O                      # noqa: E402  (loss only: test infrastructure)
from step_amd import comm as C                           # noqa: E402
from step_amd.optim import FusedAdamClip                 # noqa: E402
from tests import train_problem as TPb                   # noqa: E402
from tests.helpers import rel_l2                         # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)

# ---- the communicator by itself
assert C.available()
nc = C.NativeComm()                                      # runs self_check(): mean (w + 1) / 2, sum w (w + 1) / 2, broadcast from rank 0
assert nc.world == world and nc.rank == rank and nc.version == 29901, (nc.world, nc.rank, nc.version)       # 29901: the stand-in answered
x = torch.arange(3 << 20, device=dev, dtype=torch.float32) * (rank + 1)            # 12 MB: two staging rounds of the stand-in
nc.allreduce_(x)
want = torch.arange(3 << 20, device=dev, dtype=torch.float32) * (world * (world + 1) / 2)
assert torch.equal(x, want)
g = torch.full((70000,), float(rank), device=dev)
side = torch.cuda.Stream()
nc.use_side_stream(side)
g.mul_(2.0)                                              # queued on the compute stream BEFORE begin: the reduction must see it
nc.grad_allreduce_begin(g[:4096]); nc.grad_allreduce_begin(g[4096:])
nc.grad_allreduce_join()
g.add_(1.0)                                              # queued AFTER join: must see the reduced values
torch.cuda.synchronize()
assert bool((g == (world - 1) + 1.0).all()), g[:4]       # mean of 2 r over the ranks = world - 1
b = torch.full((1000,), rank + 5, device=dev, dtype=torch.int64)
nc.broadcast_(b, root=world - 1)
assert bool((b == world + 4).all())
nc.close()
print(f"rank {rank}: communicator of {world} ranks ok", flush=True)

# ---- the training step: collectives through the communicator vs through torch.distributed, same weights, same batches
N, L, T_train, B = 48, 288, 700, 2
prob = TPb.Problem(N, L, T_train, n_train=32, n_eval=4)


def build(shard, collectives):
    m = TPb.build_native(N, L, T_train, prob.series, k=5, seed=0).to(dev)
    m.train()
    m.matmul_precision = "bf16"
    m.backend.dropout = 0.0
    m.tsformer.dropout_p = 0.0
    m.enable_native_data_parallel(shard_graph_learner=shard, collectives=collectives)
    assert (m._comm is not None) == (collectives == "rccl")
    return m


def step(m, it):
    ts = prob.schedule(2, B, seed=50 + rank)[it]
    hist, longh, fut = prob.batch(ts)
    m._noise_override = torch.rand(B, N * N, 2, generator=torch.Generator().manual_seed(1000 * rank + it))
    pred, theta, knn, coef = m(history_data=hist.to(dev), long_history_data=longh.to(dev), future_data=None, batch_seen=it, epoch=1)
    loss = O.step_loss(O.rescale(pred[..., [0]], prob.mean, prob.std), O.rescale(fut.to(dev)[..., [0]], prob.mean, prob.std), theta, knn, coef)
    loss.backward()
    torch.cuda.synchronize()
    return pred.detach().clone(), float(loss.detach())


for shard in (False, True):
    mt, mn = build(shard, "torch"), build(shard, "rccl")
    ot = FusedAdamClip(mt, lr=2e-3, weight_decay=1e-5, eps=1e-8, max_norm=3.0)
    on = FusedAdamClip(mn, lr=2e-3, weight_decay=1e-5, eps=1e-8, max_norm=3.0)
    for it in range(2):
        ot.zero_grad(); on.zero_grad()
        pt, lt = step(mt, it)
        pn, ln = step(mn, it)
        ft, fn = mt._flat_grad, mn._flat_grad
        e = {"pred": rel_l2(pn.cpu(), pt.cpu()), "loss": abs(ln - lt) / abs(lt), "grad": rel_l2(fn.cpu(), ft.cpu())}
        # other ranks' gradients really arrived: the reduced buffer differs from what this rank alone would have produced
        ot.step(); on.step()
        e["grad_norm"] = abs(float(ot.grad_norm) - float(on.grad_norm)) / float(ot.grad_norm)
        print(f"rank {rank} {'time slices' if shard else 'whole graph learner'} step {it}: stand-in communicator vs torch.distributed {e}", flush=True)
        # same arithmetic on both transports ((a + b) / 2 either way); what is left is the order of the kernels' own f32 atomics, and from the
        # second step on Adam's +-lr steps on round-off-sized gradients (tests/shard_worker.py, part III)
        tol = 2e-5 if it == 0 else 2e-2
        assert e["pred"] < tol and e["loss"] < tol and e["grad"] < (1e-3 if it == 0 else 5e-2) and e["grad_norm"] < tol * 10, e      # step 0 measured 3.0 .. 4.5e-5
    # every rank ends with bit-identical replicated parameters (same reduced gradient, same clip factor, same Adam step)
    for pname in ("backend.nodevec1", "backend.end_conv_2.weight", "discrete_graph_learning.conv1.weight"):
        t = dict(mn.named_parameters())[pname].detach().double().cpu()
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        assert all(torch.equal(every[0], q) for q in every), f"{pname} differs between the ranks ({'time slices' if shard else 'whole graph learner'})"
    mn._comm.close(); mn._comm = None
dist.destroy_process_group()
print("rank", rank, "ok", flush=True)
