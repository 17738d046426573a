"""Full-size parity at the shapes of BASELINE.json's configs: one training step of the native module, dropout off, against
the CPU oracle fed the device encoder's hidden states and the same Gumbel noise -- prediction, edge probabilities, loss and
every gradient.  C2 STEP_PEMS04 (N=307, L=4032 -> P=336, T=13 599, two windows), C4 STEP_PEMS07 (N=883 -- not a multiple of 8,
so the padded-pitch adjacency stacks and the unaligned GEMM fallbacks are on the path --, P=168, T=16 513, one window) and a
synthetic N=2048 graph (the N^2 terms of C5: edge MLP, Gram + top-k, diffusion hops; the train series is shortened to 2500
steps so that the oracle's [N^2, 100] edge tensor and its autograd copies fit the host).  Few windows keep the oracle
(torch CPU fp32) at seconds to a minute; nothing in the native path depends on B beyond the batch loops.

Round 4 adds (parity gaps):
* SYNTH_4096 -- BASELINE config 5 ITSELF (N = 4096, L = 2016, the full 16 513-step train series, B = 1), which the reference cannot
  construct (two 275 GB one-hot matrices, discrete_graph_learning.py:88-89): the oracle evaluates the edge MLP in receiver-row blocks
  that are recomputed in the backward pass (`edge_row_chunk`), everything else is the same restatement that is pinned to the
  reference at N = 20 / 37 and compared with the device at N = 307 / 883 / 2048 here;
* the oracle's OWN fp32 encoder at C2 and C4 (no device hidden states injected): hidden-state rel-L2 <= 1e-2 and Jaccard of the
  kNN prior >= 0.98 -- SURVEY 8c's numbers for the 16-bit path --, plus the count of ones of an all-zero-history window."""
import os
import numpy as np
import pytest
import torch

import bench as Bn
from oracle import step_oracle as O
from tests.helpers import rel_l2, max_abs
from tests.test_gpu_step import ref_name

pytestmark = pytest.mark.gpu


CASES = {
    "STEP_PEMS04": (dict(Bn.CONFIGS["STEP_PEMS04"]), 2),
    "STEP_PEMS07": (dict(Bn.CONFIGS["STEP_PEMS07"]), 1),
    "SYNTH_2048": (dict(N=2048, L=288 * 7, T_train=2500, T_all=2016 + 2500, B=1, k=10), 1),
}


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("case", ["STEP_PEMS04", "STEP_PEMS07", "SYNTH_2048"])
def test_full_size_training_step_parity(case, mode):
    """mode "f32": exact contractions downstream of the encoder (tight); mode "bf16": what bench.py times -- hops, DGL conv and
    fc on the bf16 matrix cores (tolerances of bf16 operand rounding, measured values in DESIGN.md section 2)."""
    tight = mode == "f32"
    cfg, B = CASES[case]
    N, L, Ttr, k = cfg["N"], cfg["L"], cfg["T_train"], cfg["k"]
    data = Bn.synth_series(cfg["T_all"], N)
    model = Bn.make_model(cfg, data)
    sd = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    model = model.cuda()
    model.train()
    model.matmul_precision = mode
    model.backend.dropout = 0.0
    model.tsformer.dropout_p = 0.0
    gen = torch.Generator().manual_seed(5)
    u = torch.rand(B, N * N, 2, generator=gen)
    model._noise_override = u
    d = torch.from_numpy(data)
    ts = [L + 17, L + 17 + 301][:B]
    hist = torch.stack([d[a - 12:a] for a in ts]); fut = torch.stack([d[a:a + 12] for a in ts]); longh = torch.stack([d[a - L:a] for a in ts])
    mean, std = 200.0, 150.0
    pred, theta, knn, coef = model(history_data=hist.cuda(), long_history_data=longh.cuda(), future_data=None, batch_seen=0, epoch=1)
    loss = O.step_loss(O.rescale(pred[..., [0]], mean, std), O.rescale(fut[..., [0]].cuda(), mean, std), theta, knn, coef)
    loss.backward()
    torch.cuda.synchronize()

    P = L // 12
    hid = model._last["hidden_bf16"].float().cpu().view(B, N, P, 96)
    last = model._last["hidden_last"].cpu().view(B, N, 96)
    p = {}
    for kk, v in sd.items():
        v = v.clone()
        if v.is_floating_point() and not kk.startswith("tsformer.") and "running_" not in kk:
            v.requires_grad_(True)
        p[kk] = v
    torch.set_num_threads(min(32, torch.get_num_threads()))
    o_pred, o_theta, o_knn, o_coef = O.step_forward(hist, longh[..., [0]], d[:Ttr, :, 0], p, u, k, 1, training=True, hidden=hid, hidden_last=last)
    o_loss = O.step_loss(O.rescale(o_pred, mean, std), O.rescale(fut[..., [0]], mean, std), o_theta, o_knn, o_coef)
    o_loss.backward()

    e_pred = rel_l2(pred.detach().cpu(), o_pred)
    print(f"full size {case} [{mode}]: pred rel-L2", e_pred, "theta max-abs", max_abs(theta.detach().cpu(), o_theta), "loss", float(loss), float(o_loss))
    assert max_abs(theta.detach().cpu(), o_theta) < (2e-5 if tight else 5e-3)
    assert float(loss) == pytest.approx(float(o_loss), rel=2e-3 if tight else 5e-3)
    dk = (knn.cpu() != o_knn).sum().item()
    print(f"full size {case} [{mode}]: kNN entries differing from oracle(device hidden):", dk, "of", knn.numel())
    assert dk <= 4 * B
    errs = {}
    for kname, t in dict(model._trainable()).items():
        og = p[ref_name(kname)].grad
        assert og is not None and t.grad is not None, kname
        if float(og.abs().max()) < 1e-4:          # analytically zero (a bias in front of a train-mode BatchNorm): round-off only
            assert max_abs(t.grad.cpu(), og) < (2e-4 if tight else 5e-3), kname
            continue
        errs[kname] = rel_l2(t.grad.cpu(), og)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print(f"full size {case} [{mode}]: worst gradient rel-L2:", [(a, round(b, 5)) for a, b in worst])
    if tight:
        assert max(errs.values()) < 2e-2, worst
    else:
        # bf16 mode, per tensor.  Every GraphWaveNet weight, the fc weight (98 % of the gradient bytes) and the edge MLP: within 5 %.  The
        # graph learner's small tensors in front of / behind a train-mode BatchNorm (conv / fc biases, BatchNorm affines, conv weights) and the
        # adaptive adjacency's node embeddings are sums whose terms cancel almost completely in exact arithmetic -- fc_b, for one, is minus the
        # sum of a zero-sum vector over the rows the ReLU switched off -- so the 2^-9 operand rounding of the FORWARD contractions (which moves
        # BatchNorm statistics and flips ReLU masks near zero) shows up amplified: 6-13 % measured.  tools/precision_split.py (round 6,
        # profiles/r06_d_precision_split.log, r06_e_precision_split_f32_storage.log) runs the step with bf16 operands in one half at a time
        # against the exact-f32 step: the graph learner's own contractions alone give conv2_b 10.6 %, conv1_b 9.3 %, bn1_b 8.9 %, bn2_b 8.3 %,
        # fc_b 7.3 %; with f32-STORED activations and bf16 operands 9.3 / 5.5 / 8.9 / 6.8 / 6.3 % -- it is the operand format of the products,
        # not sums over bf16-stored values (round 5's guess): nothing short of wider operands (float16 activations, split bf16) removes it.
        loose = ("dgl.conv1_", "dgl.conv2_", "dgl.bn1_", "dgl.bn2_", "dgl.bn3_", "dgl.fc_b", "be.nodevec")
        for kname, e in errs.items():
            bound = 0.16 if kname.startswith(loose) else 0.05
            assert e < bound, (kname, e, bound, worst)
    num = sum(float(((dict(model._trainable())[a].grad.cpu() - p[ref_name(a)].grad) ** 2).sum()) for a in errs)
    den = sum(float((p[ref_name(a)].grad ** 2).sum()) for a in errs)
    print(f"full size {case} [{mode}]: whole-gradient rel-L2", (num / den) ** 0.5)
    assert (num / den) ** 0.5 < (5e-3 if tight else 5e-2)
    assert e_pred < (2e-3 if tight else 1e-2)


def _oracle_params(sd):
    p = {}
    for kk, v in sd.items():
        v = v.clone()
        if v.is_floating_point() and not kk.startswith("tsformer.") and "running_" not in kk:
            v.requires_grad_(True)
        p[kk] = v
    return p


def _host_ram_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


_C5_ORACLE = {}


def _c5_problem():
    """One oracle evaluation (forward + backward) of config C5 shared by the two matmul modes: the device's hidden states of the
    f32-mode run are injected (the encoder is the same kernel in both modes, dropout off: bit-identical hidden states)."""
    cfg = dict(Bn.CONFIGS["SYNTH_4096"])
    data = Bn.synth_series(cfg["T_all"], cfg["N"])
    model = Bn.make_model(cfg, data)
    sd = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    return cfg, data, model, sd


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_full_size_c5_4096_nodes_parity(mode):
    """BASELINE config 5 at its own size.  With >= 96 GB of free host memory the oracle back-propagates through the whole graph
    learner (conv activations of 2.2 + 4.3 GB and their autograd copies); otherwise the global feature g is computed without
    autograd and the gradients checked are those of everything downstream of g (GraphWaveNet, edge MLP)."""
    tight = mode == "f32"
    if "cfg" not in _C5_ORACLE:
        _C5_ORACLE["cfg"], _C5_ORACLE["data"], _C5_ORACLE["model"], _C5_ORACLE["sd"] = _c5_problem()
    cfg, data, sd = _C5_ORACLE["cfg"], _C5_ORACLE["data"], _C5_ORACLE["sd"]
    N, L, Ttr, k, B = cfg["N"], cfg["L"], cfg["T_train"], cfg["k"], 1
    model = _C5_ORACLE["model"].cuda()
    model.train()
    model.matmul_precision = mode
    model.backend.dropout = 0.0
    model.tsformer.dropout_p = 0.0
    gen = torch.Generator().manual_seed(5)
    u = torch.rand(B, N * N, 2, generator=gen)
    model._noise_override = u
    d = torch.from_numpy(data)
    a = L + 17
    hist, fut, longh = d[a - 12:a][None], d[a:a + 12][None], d[a - L:a][None]
    mean, std = 200.0, 150.0
    model.zero_grad()
    pred, theta, knn, coef = model(history_data=hist.cuda(), long_history_data=longh.cuda(), future_data=None, batch_seen=0, epoch=1)
    loss = O.step_loss(O.rescale(pred[..., [0]], mean, std), O.rescale(fut[..., [0]].cuda(), mean, std), theta, knn, coef)
    loss.backward()
    torch.cuda.synchronize()
    P = L // 12
    if "out" not in _C5_ORACLE:
        hid = model._last["hidden_bf16"].float().cpu().view(B, N, P, 96)
        last = model._last["hidden_last"].cpu().view(B, N, 96)
        p = _oracle_params(sd)
        torch.set_num_threads(min(32, torch.get_num_threads()))
        full = _host_ram_gb() >= 96.0
        g = None
        if not full:
            with torch.no_grad():
                g = O.dgl_global_feature(d[:Ttr, :, 0], p, training=True)
        o_pred, o_theta, o_knn, o_coef = O.step_forward(hist, longh[..., [0]], d[:Ttr, :, 0], p, u, k, 1, training=True, hidden=hid,
                                                        hidden_last=last, edge_row_chunk=128, g=g)
        o_loss = O.step_loss(O.rescale(o_pred, mean, std), O.rescale(fut[..., [0]], mean, std), o_theta, o_knn, o_coef)
        o_loss.backward()
        _C5_ORACLE["out"] = (p, o_pred.detach(), o_theta.detach(), o_knn, float(o_loss), full)
    p, o_pred, o_theta, o_knn, o_loss, full = _C5_ORACLE["out"]
    e_pred = rel_l2(pred.detach().cpu(), o_pred)
    e_theta = max_abs(theta.detach().cpu(), o_theta)
    dk = (knn.cpu() != o_knn).sum().item()
    print(f"C5 N=4096 [{mode}] (oracle autograd through the global branch: {full}): pred rel-L2 {e_pred:.3e}, theta max-abs {e_theta:.3e}, "
          f"loss {float(loss):.6f} vs {o_loss:.6f}, kNN entries differing {dk} of {knn.numel()} ({int(knn.sum().item())} ones)")
    assert e_theta < (2e-5 if tight else 5e-3)
    assert float(loss) == pytest.approx(o_loss, rel=2e-3 if tight else 5e-3)
    assert dk <= 8
    errs, num, den = {}, 0.0, 0.0
    for kname, t in dict(model._trainable()).items():
        og = p[ref_name(kname)].grad
        if og is None:
            assert not full and kname.startswith("dgl.") and not kname.startswith(("dgl.fc_out", "dgl.fc_cat")), kname
            continue
        assert t.grad is not None, kname
        if float(og.abs().max()) < 1e-4:
            assert max_abs(t.grad.cpu(), og) < (2e-4 if tight else 5e-3), kname
            continue
        errs[kname] = rel_l2(t.grad.cpu(), og)
        num += float(((t.grad.cpu() - og) ** 2).sum())
        den += float((og ** 2).sum())
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    whole = (num / den) ** 0.5
    print(f"C5 N=4096 [{mode}]: {len(errs)} gradients compared, whole-gradient rel-L2 {whole:.3e}, worst:", [(a_, round(b_, 5)) for a_, b_ in worst])
    if tight:
        assert max(errs.values()) < 2e-2, worst
    else:
        # bf16 mode, per tensor.  Every GraphWaveNet weight, the fc weight (98 % of the gradient bytes) and the edge MLP: within 5 %.  The
        # graph learner's small tensors in front of / behind a train-mode BatchNorm (conv / fc biases, BatchNorm affines, conv weights) and the
        # adaptive adjacency's node embeddings are sums whose terms cancel almost completely in exact arithmetic -- fc_b, for one, is minus the
        # sum of a zero-sum vector over the rows the ReLU switched off -- so the 2^-9 operand rounding of the FORWARD contractions (which moves
        # BatchNorm statistics and flips ReLU masks near zero) shows up amplified: 6-13 % measured.  tools/precision_split.py (round 6,
        # profiles/r06_d_precision_split.log, r06_e_precision_split_f32_storage.log) runs the step with bf16 operands in one half at a time
        # against the exact-f32 step: the graph learner's own contractions alone give conv2_b 10.6 %, conv1_b 9.3 %, bn1_b 8.9 %, bn2_b 8.3 %,
        # fc_b 7.3 %; with f32-STORED activations and bf16 operands 9.3 / 5.5 / 8.9 / 6.8 / 6.3 % -- it is the operand format of the products,
        # not sums over bf16-stored values (round 5's guess): nothing short of wider operands (float16 activations, split bf16) removes it.
        loose = ("dgl.conv1_", "dgl.conv2_", "dgl.bn1_", "dgl.bn2_", "dgl.bn3_", "dgl.fc_b", "be.nodevec")
        for kname, e in errs.items():
            bound = 0.16 if kname.startswith(loose) else 0.05
            assert e < bound, (kname, e, bound, worst)
    assert whole < (5e-3 if tight else 5e-2)
    assert e_pred < (2e-3 if tight else 1e-2)
    if mode == "bf16":
        _C5_ORACLE.clear()


def _jaccard(a, b):
    a, b = a.bool(), b.bool()
    return float((a & b).sum()) / max(float((a | b).sum()), 1.0)


@pytest.mark.parametrize("case", ["STEP_PEMS04", "STEP_PEMS07"])
def test_full_size_hidden_and_prior_graph_vs_oracle_own_encoder(case):
    """SURVEY 8c's bounds for the 16-bit path, end to end at full size and dropout off: the oracle runs its OWN fp32 TSFormer on the
    same windows (tsformer.py:71-105, transformer_layers.py:13-20) -- nothing of the device is injected -- and selects its own
    top-(k N) cosine entries (discrete_graph_learning.py:91-111, similarity.py:6-16).  hidden rel-L2 <= 1e-2, adj_knn Jaccard >=
    0.98.  Then an all-zero long history (39 % of PEMS04's training windows, forecasting_dataset.py:66-67): every node has the
    same hidden state, every cosine ties, and only the COUNT of ones is defined up to the number of diagonal slots the tie order
    hands out (each selected diagonal entry is cleared afterwards, :165-166)."""
    cfg, B = CASES[case]
    N, L, k = cfg["N"], cfg["L"], cfg["k"]
    data = Bn.synth_series(cfg["T_all"], N)
    model = Bn.make_model(cfg, data)
    sd = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    model = model.cuda()
    model.train()
    model.matmul_precision = "bf16"
    model.backend.dropout = 0.0
    model.tsformer.dropout_p = 0.0
    d = torch.from_numpy(data)
    ts = [L + 17, L + 17 + 301][:B]
    hist = torch.stack([d[a - 12:a] for a in ts]); longh = torch.stack([d[a - L:a] for a in ts])
    P = L // 12
    torch.set_num_threads(min(32, torch.get_num_threads()))
    p = {kk: v for kk, v in sd.items()}
    with torch.no_grad():
        pred, theta, knn, coef = model(history_data=hist.cuda(), long_history_data=longh.cuda(), future_data=None, batch_seen=0, epoch=1)
        hid = model._last["hidden_bf16"].float().cpu().view(B, N, P, 96)
        o_hid = O.tsformer_encode(longh[..., 0], p).reshape(B, N, P, 96)
        o_knn, _ = O.cosine_knn_graph(o_hid.reshape(B, N, -1), k * N)
    e_hid = rel_l2(hid, o_hid)
    jac = [_jaccard(knn[b].cpu(), o_knn[b]) for b in range(B)]
    print(f"{case}: hidden rel-L2 vs the oracle's own fp32 encoder {e_hid:.3e}; adj_knn Jaccard per window {[round(j, 4) for j in jac]} "
          f"({int(o_knn.sum().item())} ones in the oracle's graphs)")
    assert e_hid < 1e-2
    assert min(jac) >= 0.98
    # ---- zero-history window
    zl = torch.zeros(1, L, N, 3)
    with torch.no_grad():
        _, _, zknn, _ = model(history_data=hist[:1].cuda(), long_history_data=zl.cuda(), future_data=None, batch_seen=0, epoch=1)
        zo_hid = O.tsformer_encode(zl[..., 0], p).reshape(1, N, -1)
        zo_knn, zsim = O.cosine_knn_graph(zo_hid, k * N)
    c_dev, c_or = int(zknn.sum().item()), int(zo_knn.sum().item())
    print(f"{case}: all-zero history: ones in adj_knn device {c_dev}, oracle {c_or} (k N = {k * N}; oracle cosines in "
          f"[{float(zsim.min()):.7f}, {float(zsim.max()):.7f}])")
    assert k * N - N <= c_dev <= k * N and k * N - N <= c_or <= k * N
    assert abs(c_dev - c_or) <= N
