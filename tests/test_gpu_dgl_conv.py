"""GPU test of the DGL global-feature stage (conv1 -> bn1 -> conv2 -> bn2 -> fc -> bn3) through the C ABI:
matmul mode bf16 (matrix-core conv2 forward / dgrad / wgrad, bf16 fc) against the exact-f32 mode and the CPU oracle."""
import ctypes

import pytest
import torch

from oracle import step_oracle as O
from tests.helpers import load_golden, params_of, rel_l2, max_abs

pytestmark = pytest.mark.gpu


def _run(L, series_nt, tensors, grads_like, dg, bf16, fill):
    N, T = series_nt.shape
    params = {k: v.clone() for k, v in tensors.items()}       # the struct holds raw pointers: keep the tensors alive
    p = fill(params, bf16)
    saved = torch.empty(L.lib().step_dgl_global_saved_floats(N, T), device="cuda")
    work = torch.empty(L.lib().step_dgl_global_work_floats(N, T, 0), device="cuda")
    g = torch.empty(N, 100, device="cuda")
    L.call("step_dgl_global_forward", L.ptr(series_nt), N, T, ctypes.byref(p), 1, 0.1, L.ptr(saved), L.ptr(work), L.ptr(g), L.stream())
    grads = {k: torch.zeros_like(v) for k, v in grads_like.items()}
    gs = fill(grads, bf16)
    work = torch.empty(L.lib().step_dgl_global_work_floats(N, T, 1), device="cuda")
    L.call("step_dgl_global_backward", L.ptr(series_nt), N, T, ctypes.byref(p), L.ptr(saved), L.ptr(dg), L.ptr(work), ctypes.byref(gs),
           L.stream())
    torch.cuda.synchronize()
    del params
    return g, grads


@pytest.mark.parametrize("N,T", [(307, 300), (150, 2100), (64, 4000)])
def test_dgl_global_bf16_mode_vs_f32_mode(N, T):
    from step_amd import _lib as L
    from step_amd.step_arch.discrete_graph_learning import fill_dgl_struct
    gen = torch.Generator().manual_seed(N * 1000 + T)
    # z-scored traffic-like series: a daily wave with a per-node phase plus noise
    tt = torch.arange(T, dtype=torch.float32)
    series = (torch.sin(2 * 3.14159265 * tt[None, :] / 288.0 + 6.28 * torch.rand(N, 1, generator=gen))
              + 0.3 * torch.randn(N, T, generator=gen)).cuda()
    K = 16 * (T - 18)
    t = {"conv1_w": torch.randn(8, 1, 10, generator=gen) * 0.3, "conv1_b": torch.randn(8, generator=gen) * 0.1,
         "conv2_w": torch.randn(16, 8, 10, generator=gen) * 0.1, "conv2_b": torch.randn(16, generator=gen) * 0.1,
         "fc_w": torch.randn(100, K, generator=gen) * (1.0 / K ** 0.5), "fc_b": torch.randn(100, generator=gen) * 0.1,
         "bn1_w": torch.rand(8, generator=gen) + 0.5, "bn1_b": torch.randn(8, generator=gen) * 0.1,
         "bn2_w": torch.rand(16, generator=gen) + 0.5, "bn2_b": torch.randn(16, generator=gen) * 0.1,
         "bn3_w": torch.rand(100, generator=gen) + 0.5, "bn3_b": torch.randn(100, generator=gen) * 0.1,
         "bn1_rm": torch.zeros(8), "bn1_rv": torch.ones(8), "bn2_rm": torch.zeros(16), "bn2_rv": torch.ones(16),
         "bn3_rm": torch.zeros(100), "bn3_rv": torch.ones(100),
         "fc_out_w": torch.zeros(100, 200), "fc_out_b": torch.zeros(100), "fc_cat_w": torch.zeros(2, 100), "fc_cat_b": torch.zeros(2)}
    t = {k: v.cuda().contiguous() for k, v in t.items()}
    trainable = {k: v for k, v in t.items() if not (k.endswith("_rm") or k.endswith("_rv"))}
    dg = torch.randn(N, 100, generator=gen).cuda()
    g32, gr32 = _run(L, series, t, trainable, dg, False, fill_dgl_struct)
    g16, gr16 = _run(L, series, t, trainable, dg, True, fill_dgl_struct)
    # the exact-f32 mode against the CPU oracle (float64 autograd)
    pre = "discrete_graph_learning."
    po = {}
    for k, v in trainable.items():
        base, kind = k.rsplit("_", 1)
        po[pre + base + (".weight" if kind == "w" else ".bias")] = v.cpu().double().requires_grad_(True)
    go = O.dgl_global_feature(series.cpu().double().t(), po)
    (go * dg.cpu().double()).sum().backward()
    assert rel_l2(g32.cpu(), go.detach()) < 1e-4
    for k in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "bn1_w", "bn2_w", "fc_w"):
        base, kind = k.rsplit("_", 1)
        eo = rel_l2(gr32[k].cpu(), po[pre + base + (".weight" if kind == "w" else ".bias")].grad)
        assert eo < 2e-3, (k, eo)
    e = rel_l2(g16.cpu(), g32.cpu())
    print(f"N={N} T={T}: global feature rel-L2 bf16 vs f32 mode {e:.2e}")
    assert e < 2e-2
    errs = {k: rel_l2(gr16[k].cpu(), gr32[k].cpu())
            for k in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "bn1_w", "bn1_b", "bn2_w", "bn2_b", "bn3_w", "bn3_b", "fc_w", "fc_b")}
    print("   grad rel-L2 bf16 vs f32 mode:", {k: f"{v:.1e}" for k, v in errs.items()})
    # bf16 operand rounding (2^-9 per element) is amplified by the three BatchNorm-after-ReLU backward passes: every bias-like
    # gradient is a masked part of a sum that cancels exactly (sum_all k(dy - m1 - xhat m2) = 0), so a 5e-3 forward
    # perturbation shows up as 5-20 % on those; the weight gradients stay below 10 %.  (A wrong lane map gives O(1).)
    for k in ("conv2_w", "conv1_w", "fc_w"):
        assert errs[k] < 0.12, errs
    assert max(errs.values()) < 0.3, errs


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("N,T", [(150, 2100), (64, 4000)])
def test_dgl_fused_batchnorm_backward_matches_three_pass(N, T, bf16, monkeypatch):
    """The BatchNorm backward passes are fused away (dgl.hip: the two per-channel sums come out of the weight-gradient
    contractions -- exactly, see conv2_wgrad_finish_kernel / bn2_fused_coef_kernel --, the transform rides in the epilogue of the
    kernel that produces the incoming gradient).  STEP_DGL_LEGACY_BN=1 keeps the three-pass form (reduce, finalize, apply):
    same gradients up to summation order in exact-f32 mode.  In bf16 mode the default additionally stores the conv activations and
    their gradients as channels-last bf16 rows (dgl_conv_mfma.hip, STEP_DGL_F32_STORAGE=1 keeps f32): one more bf16 rounding of values
    that the matrix cores round anyway."""
    from step_amd import _lib as L
    from step_amd.step_arch.discrete_graph_learning import fill_dgl_struct
    gen = torch.Generator().manual_seed(N + T)
    tt = torch.arange(T, dtype=torch.float32)
    series = (torch.sin(2 * 3.14159265 * tt[None, :] / 288.0 + 6.28 * torch.rand(N, 1, generator=gen)) + 0.3 * torch.randn(N, T, generator=gen)).cuda()
    K = 16 * (T - 18)
    t = {"conv1_w": torch.randn(8, 1, 10, generator=gen) * 0.3, "conv1_b": torch.randn(8, generator=gen) * 0.1,
         "conv2_w": torch.randn(16, 8, 10, generator=gen) * 0.1, "conv2_b": torch.randn(16, generator=gen) * 0.1,
         "fc_w": torch.randn(100, K, generator=gen) * (1.0 / K ** 0.5), "fc_b": torch.randn(100, generator=gen) * 0.1,
         "bn1_w": torch.rand(8, generator=gen) + 0.5, "bn1_b": torch.randn(8, generator=gen) * 0.1,
         "bn2_w": torch.rand(16, generator=gen) + 0.5, "bn2_b": torch.randn(16, generator=gen) * 0.1,
         "bn3_w": torch.rand(100, generator=gen) + 0.5, "bn3_b": torch.randn(100, generator=gen) * 0.1,
         "bn1_rm": torch.zeros(8), "bn1_rv": torch.ones(8), "bn2_rm": torch.zeros(16), "bn2_rv": torch.ones(16),
         "bn3_rm": torch.zeros(100), "bn3_rv": torch.ones(100),
         "fc_out_w": torch.zeros(100, 200), "fc_out_b": torch.zeros(100), "fc_cat_w": torch.zeros(2, 100), "fc_cat_b": torch.zeros(2)}
    t = {k: v.cuda().contiguous() for k, v in t.items()}
    trainable = {k: v for k, v in t.items() if not (k.endswith("_rm") or k.endswith("_rv"))}
    dg = torch.randn(N, 100, generator=gen).cuda()
    monkeypatch.setenv("STEP_DGL_LEGACY_BN", "1")
    monkeypatch.setenv("STEP_DGL_F32_STORAGE", "1")       # bf16 mode: the reference run also keeps the [N][C][T] f32 activations
    _, legacy = _run(L, series, t, trainable, dg, bf16, fill_dgl_struct)
    monkeypatch.setenv("STEP_DGL_LEGACY_BN", "0")
    monkeypatch.setenv("STEP_DGL_F32_STORAGE", "0")       # ... against the default: channels-last bf16 rows, everything fused
    _, fused = _run(L, series, t, trainable, dg, bf16, fill_dgl_struct)
    errs = {k: rel_l2(fused[k].cpu(), legacy[k].cpu()) for k in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "bn1_w", "bn1_b", "bn2_w", "bn2_b", "fc_w")}
    print(f"N={N} T={T} bf16={bf16}: fused vs three-pass BatchNorm backward, gradient rel-L2:", {k: f"{v:.1e}" for k, v in errs.items()})
    if not bf16:
        assert max(errs.values()) < 5e-4, errs
    else:
        # two bf16 realisations of gradients that are small differences of large sums (each is 5-10 % from the exact-f32 mode,
        # test_dgl_global_bf16_mode_vs_f32_mode); at full size the whole gradient of the step agrees to 7e-3 (test_gpu_full_size.py)
        for k in ("conv1_w", "conv2_w", "fc_w", "bn1_w", "bn2_w"):
            assert errs[k] < 0.25, errs
        assert max(errs.values()) < 0.4, errs
