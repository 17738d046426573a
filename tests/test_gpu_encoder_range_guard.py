"""Range guard of the fused encoder's float16 operand path.  The reference computes the TSFormer in fp32
(step/step_arch/tsformer/transformer_layers.py:13-20), so a checkpoint may hold activations or weights beyond float16's 65 504; the
native encoder must not turn those into silent inf / NaN: it falls back to its bfloat16 fragments (fp32's exponent range)."""
import warnings

import numpy as np
import pytest
import torch

from oracle import step_oracle as O
from tests import train_problem as TPb
from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
FFN1 = "encoder.transformer_encoder.layers.1.linear1.weight"


def _tsformer(P=40, seed=0):
    from step_amd.step_arch.tsformer import TSFormer
    targs, _ = TPb.model_args(8, P * 12)
    torch.manual_seed(seed)
    m = TSFormer(**targs).cuda().eval()
    rng = np.random.default_rng(3)
    series = torch.tensor(rng.normal(size=(6, P * 12)), dtype=torch.float32).cuda()
    return m, series


def _oracle_hidden(m, series):
    sd = {"tsformer." + k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    return O.tsformer_encode(series.cpu().T[None].contiguous(), sd)[0]


def test_weights_float16_cannot_hold_are_packed_as_bfloat16():
    m, series = _tsformer()
    with torch.no_grad():
        m.state_dict()[FFN1][0, 0] = 1.0e5
    with pytest.warns(UserWarning, match="beyond float16"):
        out = m.encode_series(series, want_f32=True)
    assert m.encoder_operand == "f16" and m.encoder_operand_in_use == "bf16" and m.range_fallbacks == 1
    assert bool(torch.isfinite(out["hidden_f32"]).all())
    assert rel_l2(out["hidden_f32"].cpu(), _oracle_hidden(m, series)) < 2.5e-2          # the bfloat16 path's bound (test_gpu_kernels.py)


def test_activation_overflow_is_caught_on_the_first_launch_and_rerun_on_bfloat16():
    m, series = _tsformer()
    with torch.no_grad():
        m.state_dict()[FFN1].mul_(3.0e5)              # weights up to ~3e4 (float16 holds them), hidden layer of the FFN ~1e5 and more
    assert float(m.state_dict()[FFN1].abs().max()) < 65504.0
    want = _oracle_hidden(m, series)
    assert bool(torch.isfinite(want).all())
    # without the guard: float16 packs inf, the launch writes NaN -- what "silent" looked like
    m.range_guard = False
    raw = m.encode_series(series, want_f32=True)["hidden_f32"]
    assert not bool(torch.isfinite(raw).all())
    m.range_guard = True
    with pytest.warns(UserWarning, match="float16 operand overflow"):
        out = m.encode_series(series, want_f32=True)
    assert m.encoder_operand_in_use == "bf16" and m.range_fallbacks == 1
    assert bool(torch.isfinite(out["hidden_f32"]).all()) and bool(torch.isfinite(out["sqnorm"]).all())
    assert rel_l2(out["hidden_f32"].cpu(), want) < 2.5e-2
    # the same launch made on bfloat16 fragments from the start
    m2, _ = _tsformer()
    m2.load_state_dict(m.state_dict())
    m2.encoder_operand = "bf16"
    ref = m2.encode_series(series, want_f32=True)["hidden_f32"]
    assert torch.equal(out["hidden_f32"], ref)
    # and nothing more is re-run or warned about afterwards
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        again = m.encode_series(series, want_f32=True)["hidden_f32"]
    assert torch.equal(again, ref) and m.range_fallbacks == 1


def test_overflow_after_the_checked_launches_is_found_by_the_poll_without_a_synchronize():
    m, series = _tsformer()
    m.range_check_launches, m.range_poll_every = 0, 1
    with torch.no_grad():
        m.state_dict()[FFN1].mul_(3.0e5)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        first = m.encode_series(series, want_f32=True)["hidden_f32"]          # overflows; only the poll's copy is queued behind it
    assert m.encoder_operand_in_use == "f16"
    torch.cuda.synchronize()
    assert not bool(torch.isfinite(first).all())
    with pytest.warns(UserWarning, match="periodic poll"):
        second = m.encode_series(series, want_f32=True)["hidden_f32"]
    assert m.encoder_operand_in_use == "bf16" and bool(torch.isfinite(second).all())


def test_well_scaled_weights_stay_on_float16_and_new_weights_are_judged_afresh():
    m, series = _tsformer()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for _ in range(4):
            out = m.encode_series(series, want_f32=True)
    assert m.encoder_operand_in_use == "f16" and m.range_fallbacks == 0
    assert rel_l2(out["hidden_f32"].cpu(), _oracle_hidden(m, series)) < 7e-3
    with torch.no_grad():
        m.state_dict()[FFN1].mul_(3.0e5)
    m._plist = None
    with pytest.warns(UserWarning):
        m.encode_series(series)
    assert m.encoder_operand_in_use == "bf16"
    good, _ = _tsformer()
    m.load_state_dict(good.state_dict())                 # weights that fit again: float16 again
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m.encode_series(series)
    assert m.encoder_operand_in_use == "f16"
