"""Where the error of the training-mode (dropout on) encoder comes from when the weights are much sharper than an
initialisation: the operand-format model (tests/enc_rounding_model.py) against the fp64 oracle, CPU only.

VERDICT round 2, weak #1: on the A/B harness's stress weights (all matrices x 3, i.e. attention scores x 9) the kernel's hidden
states are 3e-3 from the oracle without dropout but 3e-2 with the same dropout masks replayed.  The suspects were two details of
the DROP branch (residual re-added from its 16-bit copy; f32 denominator against bf16-rounded numerator).  This test pins the
actual cause: the model reproduces both numbers WITHOUT either detail, and nearly all of the dropout-on error enters through the
16-bit roundings on the SCORE path (x, Wq, Wk, q, k).  With scores in the hundreds, a float16 rounding of q or k moves a score by
|s| 2^-12 ~ 0.1 and the winner-take-all softmax turns that into O(10 %) probability changes; dropout removes winners and moves
the states by 68 % here, so the perturbed network sits on far more of those razor edges than the clean one.  Any implementation
with 16-bit Q/K operands has this error; the kernel is held to the model on the GPU (tests/test_gpu_kernels.py)."""
import numpy as np
import torch

from tests import enc_rounding_model as RM


def _problem(S=3, P=336, factor=3.0):
    from step_amd.step_arch.tsformer import TSFormer
    torch.manual_seed(0)
    m = TSFormer(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, num_token=P, mask_ratio=0.75,
                 encoder_depth=4, decoder_depth=1, mode="forecasting")
    sd = RM.sharpened({k: v.detach().clone() for k, v in m.state_dict().items()}, factor)
    rng = np.random.default_rng(0)
    t = np.arange(12 * P)
    x = np.stack([np.sin(2 * np.pi * t / 288 + rng.uniform(0, 6)) * rng.uniform(0.5, 1.5) + 0.3 * np.sin(2 * np.pi * t / 2016)
                  + 0.25 * rng.standard_normal(12 * P) for _ in range(S)])
    g = torch.Generator().manual_seed(5)
    b = lambda *shape: (torch.rand(*shape, generator=g) < 0.9).double()
    masks = {"pos": b(S, P, 96), "layers": [{"attn": b(S, 4, P, P), "drop1": b(S, P, 96), "ffn": b(S, P, 384), "drop2": b(S, P, 96)}
                                            for _ in range(4)]}
    return torch.from_numpy(x), sd, masks


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def test_dropout_on_error_on_sharp_weights_is_score_path_conditioning():
    x, sd, masks = _problem()
    exact0 = RM.encode(x, sd, None, None)
    exact1 = RM.encode(x, sd, None, None, drop=masks, keep=0.9)
    e0 = _rel(RM.encode(x, sd), exact0)
    e1 = _rel(RM.encode(x, sd, drop=masks, keep=0.9), exact1)
    score = _rel(RM.encode(x, sd, drop=masks, keep=0.9, sites=RM.SCORE_PATH), exact1)
    rest = _rel(RM.encode(x, sd, drop=masks, keep=0.9, sites=[s for s in RM.SITES if s not in RM.SCORE_PATH]), exact1)
    pert = _rel(exact1, exact0)
    print(f"operand-format model, weights x 3: dropout off {e0:.2e}, dropout on {e1:.2e} (score-path roundings alone {score:.2e}, all other "
          f"roundings alone {rest:.2e}); dropout moves the states by {pert:.2f}")
    assert 1e-3 < e0 < 8e-3                     # the kernel measures 3.4e-3 (profiles/r02_ah_encoder_no_scratch_ab.log)
    assert 1.2e-2 < e1 < 6e-2                   # the kernel measures 2.9e-2
    assert score > 0.7 * e1 and rest < 0.4 * e1
    assert pert > 0.5


def test_same_model_meets_the_bound_on_initialisation_scale_weights():
    """the same model at the scale of an initialised / lightly trained TSFormer: both modes well inside the 1e-2 bound"""
    x, sd, masks = _problem(factor=1.0)
    exact0 = RM.encode(x, sd, None, None)
    exact1 = RM.encode(x, sd, None, None, drop=masks, keep=0.9)
    e0, e1 = _rel(RM.encode(x, sd), exact0), _rel(RM.encode(x, sd, drop=masks, keep=0.9), exact1)
    print(f"operand-format model, plain weights: dropout off {e0:.2e}, on {e1:.2e}")
    assert e0 < 5e-3 and e1 < 5e-3
