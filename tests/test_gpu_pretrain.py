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
