"""The operand-fragment index maps of the fused feed-forward kernels (csrc/pretrain_fused.hip), executed on the host with a model of the
matrix instruction (tests/ffn_fused_host.py): forward, input gradient and the three parameter gradients of
f2 = W2 . keep(relu(W1 . h1 + b1)) / (1 - p) + b2 against plain matrix algebra, with and without a keep-mask pool, R not a multiple of 32.
What this pins: the chain k-slot map both operands of a product have to share, the transposed roles in the weight-gradient kernels
(hidden unit as the lane), and that the lane-mask form of the pool words (forward, backward-data) and the word-per-unit form (weight
gradients) address the same bit."""
import numpy as np
import pytest

from tests import ffn_fused_host as F


@pytest.mark.parametrize("drop", [False, True])
def test_fragment_maps_compute_the_feed_forward_block(drop):
    rng = np.random.default_rng(3)
    R = 32 * 2 + 13
    x = rng.standard_normal((R, 96))
    w1 = rng.standard_normal((384, 96)) * 0.15
    b1 = rng.standard_normal(384) * 0.1
    w2 = rng.standard_normal((96, 384)) * 0.1
    b2 = rng.standard_normal(96) * 0.1
    df2 = rng.standard_normal((R, 96))
    dh1_in = rng.standard_normal((R, 96))
    pool, seed, site, p = None, 0x1234_5678_9ABC, 34, 0.0
    keep = np.ones((R, 384), dtype=bool)
    if drop:
        p = 0.25
        pool = rng.integers(0, 1 << 63, size=1024, dtype=np.uint64) | (rng.integers(0, 2, size=1024, dtype=np.uint64) << np.uint64(63))
        keep = F.keep_matrix(pool, seed, site, R)
        assert 0.4 < keep.mean() < 0.6
    ik = 1.0 / (1.0 - p)
    pre = x @ w1.T + b1
    hid = np.maximum(pre, 0) * keep * ik
    want_f2 = hid @ w2.T + b2
    dhid = (df2 @ w2) * (pre > 0) * keep * ik
    want_dh1 = dh1_in + dhid @ w1
    want_dw1, want_db1, want_dw2 = dhid.T @ x, dhid.sum(0), df2.T @ hid
    got_f2 = F.forward(x, w1, b1, w2, b2, pool, seed, site, ik)
    got_dh1 = F.backward_data(df2, x, w1, b1, w2, dh1_in, pool, seed, site, ik)
    got_dw1, got_db1, got_dw2 = F.backward_weights(df2, x, w1, b1, w2, pool, seed, site, ik)
    for name, a, b in (("f2", got_f2, want_f2), ("dh1", got_dh1, want_dh1), ("dw1", got_dw1, want_dw1), ("db1", got_db1, want_db1),
                       ("dw2", got_dw2, want_dw2)):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-9), name
