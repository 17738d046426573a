"""Worker of tests/test_gpu_sharded_graph_learner.py (one process per data-parallel rank, gloo group, every rank on cuda:0).

Three modules with identical initial weights train on per-rank batches:
  A  data parallel, whole graph learner on every rank (flat-gradient all-reduce)
  B  data parallel, graph learner in time slices (fc.weight sharded, five small exchanges per step)
  C  no process group: this rank evaluates BOTH ranks' batches one after the other and averages the gradients
and must agree to summation order: forward outputs, loss, every gradient, and the parameters after two fused clip+Adam steps."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from oracle import step_oracle as O                      # noqa: E402  (loss only: test infrastructure)
from step_amd.optim import FusedAdamClip                 # noqa: E402
from tests import train_problem as TPb                   # noqa: E402
from tests.helpers import rel_l2                         # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
N, L, T_train, B, STEPS = 48, 288, 700, 2, 2
prob = TPb.Problem(N, L, T_train, n_train=32, n_eval=4)


def build(mode):
    m = TPb.build_native(N, L, T_train, prob.series, k=5, seed=0).to(dev)
    m.train()
    m.matmul_precision = "bf16"
    m.backend.dropout = 0.0
    m.tsformer.dropout_p = 0.0
    if mode in "AB":
        m.enable_native_data_parallel(shard_graph_learner=mode == "B")
    return m


def batch_of(r, it):
    ts = prob.schedule(STEPS, B, seed=50 + r)[it]
    hist, longh, fut = prob.batch(ts)
    u = torch.rand(B, N * N, 2, generator=torch.Generator().manual_seed(1000 * r + it))
    return hist.to(dev), longh.to(dev), fut.to(dev), u


def step(m, r, it, backward=True):
    hist, longh, fut, u = batch_of(r, it)
    m._noise_override = u
    pred, theta, knn, coef = m(history_data=hist, long_history_data=longh, future_data=None, batch_seen=it, epoch=1)
    loss = O.step_loss(O.rescale(pred[..., [0]], prob.mean, prob.std), O.rescale(fut[..., [0]], prob.mean, prob.std), theta, knn, coef)
    if backward:
        loss.backward()
    return pred.detach(), theta.detach(), float(loss.detach()), m._last["sampled_adj"].clone(), m._last["g"].clone()


A, Bm, C = build("A"), build("B"), build("C")
sh = Bm.discrete_graph_learning._shard
assert sh is not None and sh["world"] == world and A.discrete_graph_learning._shard is None
T2 = T_train - 18
optA = FusedAdamClip(A, lr=2e-3, weight_decay=1e-5, eps=1e-8, max_norm=3.0)
optB = FusedAdamClip(Bm, lr=2e-3, weight_decay=1e-5, eps=1e-8, max_norm=3.0)
layA, layB = A._grad_layout(), Bm._grad_layout()
foA, fnA, _ = layA["items"]["dgl.fc_w"]
foB, fnB, _ = layB["items"]["dgl.fc_w"]
assert foA == foB and fnB == 100 * 16 * (sh["b"] - sh["a"]) and fnA == 100 * 16 * T2
def compare(it, data_rank, tight):
    """one forward/backward of A and B on the batch of `data_rank` (None: this rank's own); returns the error report"""
    optA.zero_grad(); optB.zero_grad()
    r = rank if data_rank is None else data_rank
    pa, ta, la, adja, ga = step(A, r, it)
    pb, tb, lb, adjb, gb = step(Bm, r, it)
    e = {"g": rel_l2(gb.cpu(), ga.cpu()), "pred": rel_l2(pb.cpu(), pa.cpu()), "theta": float((tb - ta).abs().max()), "loss": abs(lb - la) / abs(la),
         "flips": int((adja != adjb).sum())}
    fa, fb = A._flat_grad, Bm._flat_grad
    e["grad_rest"] = rel_l2(fb[:foB].cpu(), fa[:foA].cpu())
    ga_fc = fa[foA:foA + fnA].view(100, 16, T2)[:, :, sh["a"]:sh["b"]].reshape(-1)
    e["grad_fc_slice"] = rel_l2(fb[foB:foB + fnB].cpu(), ga_fc.cpu())
    # per tensor, skipping those whose gradient is round-off only: the graph-convolution bias feeds (through the residual add) a
    # training-mode BatchNorm, whose input gradient sums to zero over every channel -- the exact gradient of be.gconv_b.* is 0 and
    # what any implementation produces for it is the cancellation noise of that sum (it varies from run to run with the order of
    # the atomics: 1e-3 .. 1e-1 relative between two passes over the SAME inputs)
    floor = 1e-4 * float(fa[:foA].abs().max())
    worst = max((rel_l2(fb[o:o + n].cpu(), fa[o:o + n].cpu()), k) for k, (o, n, _) in layA["items"].items() if k != "dgl.fc_w"
                and not k.startswith("be.gconv_b.") and float(fa[o:o + n].abs().max()) > floor)
    e["worst_tensor"] = (round(worst[0], 6), worst[1])
    return e, fa


# I. the same batch on every rank: the averaged gradient of g equals each rank's own, so the bf16 operand roundings coincide and the
#    time-sliced backward must reproduce the whole one to f32 summation order
e, _ = compare(0, 0, True)
print(f"rank {rank} same batch on all ranks: {e}", flush=True)
assert e["flips"] == 0 and e["g"] < 1e-4 and e["pred"] < 1e-4 and e["theta"] < 1e-4 and e["loss"] < 1e-5, e
# worst tensor: the conv1 / conv2 bias and weight gradients are sums over bf16-stored rows (2^-9 per element) with heavy cancellation,
# and the slices round them at different points than the unsharded pass; measured 7e-4 .. 2.2e-3 over builds and world sizes
assert e["grad_rest"] < 2e-4 and e["grad_fc_slice"] < 8e-4 and e["worst_tensor"][0] < 6e-3, e          # fc slice: measured 0.5 .. 5.2e-4
if os.environ.get("STEP_DGL_F32_STORAGE") == "1":
    # f32 storage of the conv activations (ADVICE round 2 asked whether the residue is the bf16 rounding of the stored rows): it is
    # not -- measured grad_rest 8.6e-6 (22x tighter than with bf16 rows) but fc slice 5.2e-4 and worst tensor 2.8e-3 (conv1_w), the same
    # as with bf16 storage: what differs between the sliced and the whole backward is where the bf16 OPERAND roundings of the fc /
    # conv contractions fall relative to the slice sums, not the halo logic (which grad_rest, at summation-order level, checks)
    assert e["grad_rest"] < 5e-5, ("f32 storage", e)

# II. per-rank batches: A rounds each rank's d(fc output) to bf16 and averages the products, B averages first and rounds once --
#     two equally valid bf16 roundings (2^-9 per element), amplified a little by the cancellations of the BatchNorm backward
e, fa = compare(0, None, False)
acc = None
for r in range(world):                     # C: both ranks' batches on this process, gradients averaged by hand = A's all-reduce
    C.zero_grad(set_to_none=True)
    step(C, r, 0)
    acc = C._flat_grad.clone() if acc is None else acc + C._flat_grad
e["dp_vs_sequential"] = rel_l2(fa.cpu(), (acc / world).cpu())
optA.step(); optB.step()
e["grad_norm"] = abs(float(optA.grad_norm) - float(optB.grad_norm)) / float(optA.grad_norm)
print(f"rank {rank} per-rank batches: {e}", flush=True)
assert e["flips"] == 0 and e["g"] < 1e-4 and e["pred"] < 1e-4 and e["theta"] < 1e-4 and e["loss"] < 1e-5, e
assert e["grad_rest"] < 1e-3 and e["grad_fc_slice"] < 1e-2 and e["worst_tensor"][0] < 3e-2, e
assert e["dp_vs_sequential"] < 1e-4 and e["grad_norm"] < 1e-4, e

# III. after the fused clip+Adam step (Adam's first step moves every element by ~lr whatever the gradient's size, so elements whose
#      gradient is round-off differ by up to 2 lr): gather B's fc slices back into fc.weight, compare whole state_dicts and the next loss
Bm.discrete_graph_learning.gather_fc_weight()
sa, sb = A.state_dict(), Bm.state_dict()
assert set(sa) == set(sb)
# (the graph-convolution biases are left out of the per-tensor figure: their exact gradient is zero -- see compare() -- so Adam's first
#  step moves each of their elements by +-lr on the sign of round-off noise, independently in the two models; dmax below still bounds them)
noise_driven = lambda k: k.startswith("backend.gconv.") and k.endswith(".mlp.mlp.bias")
worst = max((rel_l2(sb[k].float().cpu(), sa[k].float().cpu()), k) for k in sa
            if sa[k].is_floating_point() and sa[k].numel() > 1 and not noise_driven(k))
dmax = max(float((sb[k].float() - sa[k].float()).abs().max()) for k in sa if sa[k].is_floating_point() and not k.endswith("running_var"))
print(f"rank {rank}: parameters after the step, worst rel-L2 sharded vs unsharded {worst}, largest element difference {dmax:.2e} (lr 2e-3)", flush=True)
assert worst[0] < 2e-2 and dmax <= 2.05 * 2e-3, (worst, dmax)
e, _ = compare(1, None, False)
print(f"rank {rank} second step: {e}", flush=True)
assert e["loss"] < 1e-2 and e["g"] < 2e-2, e
# data-parallel replicas stay bit-identical (what DistributedDataParallel guarantees): same reduced gradient, same clip factor, same Adam
# step on every rank -- for the time slices this needs the gradient norm formed identically everywhere (step.py: _total_sumsq)
for name, m in (("whole graph learner", A), ("time slices", Bm)):
    for pname in ("backend.nodevec1", "backend.end_conv_2.weight", "discrete_graph_learning.conv1.weight"):
        t = dict(m.named_parameters())[pname].detach().double().cpu()
        every = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        assert all(torch.equal(every[0], x) for x in every), f"{name}: {pname} differs between the ranks after two optimizer steps"
# every rank holds the same gathered fc.weight
w = Bm.discrete_graph_learning.fc.weight.detach().double().cpu()
both = [torch.empty_like(w) for _ in range(world)]
dist.all_gather(both, w)
assert all(torch.equal(both[0], x) for x in both)
dist.destroy_process_group()
print("rank", rank, "ok", flush=True)
