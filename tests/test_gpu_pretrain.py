"""GPU parity of the native TSFormer pre-training step (forward + every parameter gradient) against the
reference's own outputs (tests/golden/tsformer_pretrain_tiny.npz) and the oracle."""
import pytest
import torch

from oracle import step_oracle as O
from tests.helpers import load_golden, rel_l2, max_abs

pytestmark = pytest.mark.gpu


def _model(g, L):
    from step_amd import TSFormer
    m = TSFormer(12, 1, 96, 4, 4, 0.1, L / 12, 0.75, 4, 1, mode="pre-train")
    sd = {k[len("param."):]: v for k, v in g.items() if k.startswith("param.")}
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def test_pretrain_matches_reference():
    g = load_golden("tsformer_pretrain_tiny")
    x = g["in.x"]
    B, L, N, _ = x.shape
    model = _model(g, L)
    model.train()
    model.dropout_p = 0.0
    um, mk = g["in.unmasked"].tolist(), g["in.masked"].tolist()
    model.mask.forward = lambda: (um, mk)
    recon, label = model(history_data=x.cuda(), future_data=None, batch_seen=0, epoch=1)
    assert recon.shape == g["out.recon"].shape
    assert max_abs(label.cpu(), g["out.label"]) == 0.0
    e = max_abs(recon.detach().cpu(), g["out.recon"])
    print("recon max abs err vs reference", e)
    assert e < 2e-4
    loss = O.masked_mae(recon * 150.0 + 200.0, label * 150.0 + 200.0, 0.0)
    assert float(loss) == pytest.approx(float(g["out.loss"]), rel=1e-4)
    loss.backward()
    torch.cuda.synchronize()
    worst, n = 0.0, 0
    for name, prm in model.named_parameters():
        want = g.get("grad." + name)
        if want is None:
            continue
        assert prm.grad is not None, name
        if float(want.abs().max()) < 1e-5:
            assert max_abs(prm.grad.cpu(), want) < 1e-4, name
            continue
        err = rel_l2(prm.grad.cpu(), want)
        worst = max(worst, err)
        assert err < 5e-3, (name, err)
        n += 1
    print("checked", n, "gradients, worst rel-L2", worst)
    assert n > 50


def test_pretrain_dropout_runs_and_is_replayable():
    g = load_golden("tsformer_pretrain_tiny")
    x = g["in.x"].cuda()
    model = _model(g, x.shape[1])
    model.train()
    um, mk = g["in.unmasked"].tolist(), g["in.masked"].tolist()
    model.mask.forward = lambda: (um, mk)
    outs = []
    for _ in range(2):
        torch.manual_seed(7)
        model._seed_ctr2 = 0
        model.zero_grad()
        recon, label = model(history_data=x, future_data=None, batch_seen=0, epoch=1)
        loss = O.masked_mae(recon, label, 0.0)
        loss.backward()
        assert torch.isfinite(loss)
        outs.append((recon.detach().clone(), model.output_layer.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert rel_l2(outs[0][1].cpu(), outs[1][1].cpu()) < 1e-4
    # and dropout actually perturbs the result
    model.dropout_p = 0.0
    clean, _ = model(history_data=x, future_data=None, batch_seen=0, epoch=1)
    assert rel_l2(outs[0][0].cpu(), clean.detach().cpu()) > 1e-3


def test_pretrain_bf16_mode_close_to_f32_mode():
    """matmul_precision="bf16": the linear layers of the pre-training step on bf16 operands (f32 accumulate); attention, LayerNorm
    and the reductions stay f32.  Reconstruction and gradients against the exact-f32 mode of the same module."""
    g = load_golden("tsformer_pretrain_tiny")
    x = g["in.x"].cuda()
    um, mk = g["in.unmasked"].tolist(), g["in.masked"].tolist()
    res = {}
    for mode in ("f32", "bf16"):
        model = _model(g, x.shape[1])
        model.train()
        model.dropout_p = 0.0
        model.matmul_precision = mode
        model.mask.forward = lambda: (um, mk)
        recon, label = model(history_data=x, future_data=None, batch_seen=0, epoch=1)
        loss = O.masked_mae(recon * 150.0 + 200.0, label * 150.0 + 200.0, 0.0)
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (recon.detach().cpu(), float(loss), {n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None})
    e = rel_l2(res["bf16"][0], res["f32"][0])
    num = sum(float(((res["bf16"][2][n] - res["f32"][2][n]) ** 2).sum()) for n in res["f32"][2])
    den = sum(float((res["f32"][2][n] ** 2).sum()) for n in res["f32"][2])
    print("pre-train bf16 vs f32 mode: recon rel-L2", e, "loss", res["bf16"][1], res["f32"][1], "whole-gradient rel-L2", (num / den) ** 0.5)
    assert e < 2e-2
    assert res["bf16"][1] == pytest.approx(res["f32"][1], rel=5e-3)
    assert (num / den) ** 0.5 < 0.15         # masked-MAE gradients flip sign where reconstruction ~ label (8 % measured)


@pytest.mark.parametrize("T,p", [(42, 0.1), (168, 0.1), (40, 0.0), (77, 0.25), (336, 0.1)])
def test_matrix_core_attention_matches_f32_attention(T, p):
    """step_pt_attention_{fwd,bwd}_bf16 (bf16 operands on the matrix cores, what the pre-training module uses in bf16 mode) against
    the exact-f32 kernels: same row statistics, same dropout stream (identical keep masks for identical seed / site), outputs and
    gradients within bf16 operand rounding.  T = 336 exercises the backward's fall-back (LDS), T = 77 an unaligned mask stream."""
    from step_amd import _lib as L
    S = 6
    gen = torch.Generator().manual_seed(T)
    qkv = (torch.randn(S, T, 288, generator=gen) * 1.5).cuda()
    dout = torch.randn(S, T, 96, generator=gen).cuda()
    seed, site = 0x1234_5678_9ABC, 7
    st = L.stream()
    res = {}
    for tag in ("", "_bf16"):
        out = torch.empty(S, T, 96, device="cuda")
        stats = torch.empty(S * 4 * T, 2, device="cuda")
        dqkv = torch.zeros(S, T, 288, device="cuda")
        if tag:
            kb = torch.zeros(S * 4 * T * ((T + 31) // 32), dtype=torch.int32, device="cuda") if T != 77 else None      # (T = 77: the backward regenerates the masks)
            L.call("step_pt_attention_fwd" + tag, L.ptr(qkv), S, T, p, seed, site, L.ptr(out), L.ptr(stats), L.ptr(kb), st)
            L.call("step_pt_attention_bwd" + tag, L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(stats), S, T, p, seed, site, L.ptr(dqkv), L.ptr(kb), st)
        else:
            L.call("step_pt_attention_fwd" + tag, L.ptr(qkv), S, T, p, seed, site, L.ptr(out), L.ptr(stats), st)
            L.call("step_pt_attention_bwd" + tag, L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(stats), S, T, p, seed, site, L.ptr(dqkv), st)
        torch.cuda.synchronize()
        res[tag] = (out.cpu(), stats.cpu(), dqkv.cpu())
    (o0, s0, g0), (o1, s1, g1) = res[""], res["_bf16"]
    e_out, e_g = rel_l2(o1, o0), rel_l2(g1, g0)
    e_m = float((s1[:, 0] - s0[:, 0]).abs().max())
    e_l = float(((s1[:, 1] - s0[:, 1]).abs() / s0[:, 1]).max())
    parts = {k: rel_l2(g1[..., a:a + 96], g0[..., a:a + 96]) for k, a in (("dq", 0), ("dk", 96), ("dv", 192))}
    print(f"T={T} p={p}: out rel-L2 {e_out:.2e}, dqkv {e_g:.2e} {parts}, row max abs {e_m:.2e}, row sum rel {e_l:.2e}")
    assert e_out < 1.5e-2 and e_g < 2.5e-2 and max(parts.values()) < 3e-2
    assert e_m < 0.1 and e_l < 3e-2
    assert torch.isfinite(g1).all() and torch.isfinite(o1).all()
