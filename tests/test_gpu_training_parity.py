"""Training-trajectory and horizon-12 MAE parity (BASELINE.json metric: "... horizon-12 MAE parity"):
K optimizer steps of the native STEP module versus K steps of the CPU oracle on the same windows, the same
Gumbel noise and the same torch.optim.Adam + clip_grad_norm_, dropout off.  The frozen TSFormer's hidden
states are constant over training, so the oracle is fed the device encoder's states (tight comparison) and,
separately, its own fp32 states (the bf16-encoder effect on the forecast error)."""
import numpy as np
import pytest
import torch

from oracle import step_oracle as O
from tests.helpers import load_golden, params_of, rel_l2
from tests.test_gpu_step import build_native, inputs_of, ref_name

pytestmark = pytest.mark.gpu
K_STEPS = 8


def _h12_mae(pred, fut, mean, std):
    return float(O.masked_mae(O.rescale(pred[:, 11], mean, std), O.rescale(fut[:, 11, :, [0]], mean, std), 0.0))


def _oracle_run(g, hidden, hidden_last, noises):
    N, L, T, B, k, epoch, tr = [int(x) for x in g["meta"]]
    mean, std = [float(x) for x in g["meta.scaler"]]
    p = params_of(g)
    train = [v for kk, v in p.items() if v.requires_grad]
    opt = torch.optim.Adam(train, lr=2e-3, weight_decay=1e-5, eps=1e-8)
    losses = []
    for it in range(K_STEPS):
        opt.zero_grad(set_to_none=True)
        stats = {}
        pred, theta, knn, coef = O.step_forward(g["in.hist"], g["in.long_hist0"].unsqueeze(-1), g["in.node_feats"], p, noises[it], k, 1,
                                                training=True, stats=stats, hidden=hidden, hidden_last=hidden_last)
        loss = O.step_loss(O.rescale(pred, mean, std), O.rescale(g["in.future"][..., [0]], mean, std), theta, knn, coef)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([q for q in train if q.grad is not None], 3.0)
        opt.step()
        losses.append(float(loss))
    with torch.no_grad():
        pred, _, _, _ = O.step_forward(g["in.hist"], g["in.long_hist0"].unsqueeze(-1), g["in.node_feats"], p, noises[0], k, 1,
                                       training=True, hidden=hidden, hidden_last=hidden_last)
    return losses, _h12_mae(pred, g["in.future"], mean, std)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["step_tiny", "step_small"])
def test_k_step_trajectory_and_h12_mae_parity(name, mode):
    """mode "f32": every contraction outside the TSFormer exact (tight); mode "bf16" (what bench.py times): diffusion hops,
    DGL conv2 and fc on the bf16 matrix cores -- the trajectory must stay within a few % of the fp32 oracle's."""
    g = load_golden(name)
    N, L, T, B, k, epoch, tr = [int(x) for x in g["meta"]]
    mean, std = [float(x) for x in g["meta.scaler"]]
    gen = torch.Generator().manual_seed(11)
    noises = [torch.rand(B, N * N, 2, generator=gen) for _ in range(K_STEPS)]
    model = build_native(g)
    model.train()
    model.matmul_precision = mode
    model.backend.dropout = 0.0
    model.tsformer.dropout_p = 0.0
    hist, long_hist, fut = inputs_of(g)
    params = [q for q in model.parameters() if q.requires_grad]
    opt = torch.optim.Adam(params, lr=2e-3, weight_decay=1e-5, eps=1e-8)
    losses = []
    for it in range(K_STEPS):
        model._noise_override = noises[it]
        opt.zero_grad(set_to_none=True)
        pred, theta, knn, coef = model(history_data=hist, long_history_data=long_hist, future_data=None, batch_seen=it, epoch=1)
        loss = O.step_loss(O.rescale(pred[..., [0]], mean, std), O.rescale(fut[..., [0]], mean, std), theta, knn, coef)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 3.0)
        opt.step()
        losses.append(float(loss))
    model._noise_override = noises[0]
    with torch.no_grad():
        pred, _, _, _ = model(history_data=hist, long_history_data=long_hist, future_data=None, batch_seen=0, epoch=1)
    h12 = _h12_mae(pred.cpu(), g["in.future"], mean, std)
    hid = model._last["hidden_bf16"].float().cpu().view(B, N, L // 12, 96)
    last = model._last["hidden_last"].cpu().view(B, N, 96)
    o_losses, o_h12 = _oracle_run(g, hid, last, noises)
    print(name, mode, "native losses", [round(x, 3) for x in losses])
    print(name, "oracle losses", [round(x, 3) for x in o_losses])
    print(name, "H12 MAE native", h12, "oracle(device hidden)", o_h12)
    # Adam turns round-off sized gradient differences into O(lr) parameter differences (sign-like updates), so two
    # correct implementations drift apart step by step; the first steps must agree tightly, the later ones to a few %
    # Under Adam the trajectories of two correct implementations separate step by step: round-off sized gradient
    # differences (split-K atomics make even two runs of this module differ in the last bits) become O(lr) parameter
    # differences on near-zero gradients.  Tight on the first steps, a band afterwards; the single-step gradient parity
    # (test_gpu_step.py) and the full-size mode experiment (tools/mode_parity.py) are the sharp checks.
    # Scale of the bands: the CPU oracle alone, fed states perturbed at the fp32 round-off level (1e-7), moves its 8-step losses by
    # up to 2.3 % and its horizon-12 MAE by up to 3.5 % on step_small (tools/trajectory_sensitivity.py,
    # profiles/r01_y_trajectory_sensitivity.txt); the late-step bands are ~3x that.
    if mode == "f32":
        assert losses[:3] == pytest.approx(o_losses[:3], rel=2e-3)
        assert losses == pytest.approx(o_losses, rel=0.10)
        assert h12 == pytest.approx(o_h12, rel=0.10)         # horizon-12 MAE after K steps (0.001-1.5 % observed)
    else:
        assert losses[:2] == pytest.approx(o_losses[:2], rel=5e-3)      # before / after one update
        assert losses[:3] == pytest.approx(o_losses[:3], rel=4e-2)
        assert losses == pytest.approx(o_losses, rel=0.15)
        if name != "step_tiny":                              # 40 series: the single-horizon MAE is sample noise there
            assert h12 == pytest.approx(o_h12, rel=0.12)     # 0.7-2.3 % observed
    f_losses, f_h12 = _oracle_run(g, None, None, noises)     # oracle with its own fp32 TSFormer
    print(name, "oracle(fp32 hidden) losses", [round(x, 3) for x in f_losses], "H12 MAE", f_h12)
    assert losses[:3] == pytest.approx(f_losses[:3], rel=2e-2)
    assert losses == pytest.approx(f_losses, rel=0.12)      # two correct runs whose inputs differ by the bf16 encoder error (1-2 %)
    # the single-horizon MAE of B*N <= 111 series after 8 chaotic Adam steps is dominated by sample noise
    # (it moves by tens of % between two fp32 runs that differ by 1 % in one input); reported, not asserted
