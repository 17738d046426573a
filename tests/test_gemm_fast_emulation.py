"""CPU check of the index arithmetic of the staged GEMM kernel (csrc/gemm_bf16.hip) through tests/emu_gemm_fast.py: the three
operand loaders, k / row tails, slot remaps with the two-level batch (the diffusion hop), per-k affine with a period that is
not a multiple of the load width (the DGL fc), split-K, the all-ones bias-gradient column, the column-block affine, and the
k-permutation of the exact-f32 variant."""
import numpy as np
import pytest

from tests import emu_gemm_fast as E

rng = np.random.default_rng(0)


def test_hop_shape_bf16_stack_times_slot_strided_activations():
    Bn, Nn, T, S = 2, 37, 2, 7
    N8 = (Nn + 7) // 8 * 8
    P = rng.random((3, Bn, Nn, Nn)) / Nn
    PT = np.zeros((3, Bn, Nn, N8)); PT[..., :Nn] = E.rbf16(P.transpose(0, 1, 3, 2)); PT[..., Nn:] = 7.0      # pad must be masked
    cat = rng.standard_normal((Bn, Nn, T, S * 32))
    out = cat.copy().reshape(-1)
    g = E.Desc(M=Nn, N=T * 32, K=Nn, sam=N8, sak=1, sbk=T * S * 32, sbn=1, ldc=T * S * 32, scn=1, batch=3 * Bn, batch0=Bn, sab=Nn * N8,
               sab1=Bn * Nn * N8, sbb=Nn * T * S * 32, sbb1=64, scb=Nn * T * S * 32, scb1=64, b_off=32, c_off=64, b_nblk=32, b_nstride=S * 32,
               c_nblk=32, c_nstride=S * 32)
    E.gemm_fast(g, PT.reshape(-1), cat.reshape(-1), out, E.KC_BF16, E.MC_F32)
    got = out.reshape(cat.shape)
    for s in range(3):
        want = np.einsum("bvw,bvtc->bwtc", E.rbf16(P[s]), E.rbf16(cat[..., 32 + 64 * s:64 + 64 * s]))
        assert np.abs(got[..., 64 + 64 * s:96 + 64 * s] - want).max() < 1e-9, s
    assert np.array_equal(got[..., :64], cat[..., :64])


@pytest.mark.parametrize("f32c", [False, True])
def test_adjacency_gradient_shape_k_remap_both_sides(f32c):
    Nn, T, S = 21, 3, 7
    cat = rng.standard_normal((Nn, T, S * 32))
    dP = rng.standard_normal((Nn, Nn))
    C = dP.copy().reshape(-1)
    g = E.Desc(M=Nn, N=Nn, K=T * 32, sam=T * S * 32, sak=1, sbk=1, sbn=T * S * 32, ldc=Nn, scn=1, a_off=32, b_off=64, a_kblk=32,
               a_kstride=S * 32, b_kblk=32, b_kstride=S * 32, accumulate=1)
    E.gemm_fast(g, cat.reshape(-1), cat.reshape(-1), C, E.KC_F32, E.KC_F32, bf16=not f32c, f32c=f32c)
    r = (lambda t: t) if f32c else E.rbf16
    want = dP + np.einsum("vtc,wtc->vw", r(cat[:, :, 32:64]), r(cat[:, :, 64:96]))
    assert np.abs(C.reshape(Nn, Nn) - want).max() < 1e-9


def test_fc_shape_affine_with_odd_period_split_k_and_ragged_k():
    M, N, period, nch = 19, 12, 67, 5
    K = period * nch                       # 335: not a multiple of 4 or 64; the row pitch is padded
    Kp = (K + 3) // 4 * 4
    A = np.full((M, Kp), 9.0); A[:, :K] = rng.standard_normal((M, K))
    W = np.full((N, Kp), 9.0); W[:, :K] = rng.standard_normal((N, K))
    sc, sh = rng.standard_normal(nch), rng.standard_normal(nch)
    C = np.zeros(M * N)
    g = E.Desc(M=M, N=N, K=K, sam=Kp, sak=1, sbk=1, sbn=Kp, ldc=N, scn=1, a_kscale=sc, a_kshift=sh, a_kperiod=period, accumulate=2, splitk=3)
    E.gemm_fast(g, A.reshape(-1), W.reshape(-1), C, E.KC_F32, E.KC_F32)
    An = A[:, :K].reshape(M, nch, period) * sc[None, :, None] + sh[None, :, None]
    want = E.rbf16(An.reshape(M, K)) @ E.rbf16(W[:, :K]).T
    assert np.abs(C.reshape(M, N) - want).max() < 1e-9


@pytest.mark.parametrize("f32c", [False, True])
def test_weight_gradient_shape_with_bias_column_and_column_affine(f32c):
    M, N, K, period = 32, 64, 150, 16            # N is a full tile: the all-ones column opens a second tile
    dY = rng.standard_normal((K, M)); X = rng.standard_normal((K, N))
    db = np.full(M, 0.5)
    sc, sh, mv = rng.standard_normal(N // period), rng.standard_normal(N // period), rng.standard_normal(M)
    C0 = rng.standard_normal((M, N))
    C = C0.copy().reshape(-1)
    g = E.Desc(M=M, N=N, K=K, sam=1, sak=M, sbk=N, sbn=1, ldc=N, scn=1, accumulate=1, a_rowsum=db, c_nscale=sc, c_nshift=sh, c_mvec=mv,
               c_nperiod=period)
    E.gemm_fast(g, dY.reshape(-1), X.reshape(-1), C, E.MC_F32, E.MC_F32, bf16=not f32c, f32c=f32c)
    r = (lambda t: t) if f32c else E.rbf16
    raw = r(dY).T @ r(X)
    ch = np.arange(N) // period
    want = C0 + raw * sc[ch][None, :] + mv[:, None] * sh[ch][None, :]
    assert np.abs(C.reshape(M, N) - want).max() < 1e-9
    assert np.abs(db - (0.5 + r(dY).sum(0))).max() < 1e-9
