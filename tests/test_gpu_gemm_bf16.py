"""GPU tests of the bf16-matrix-core variant of step_gemm (StepGemm.compute_bf16 = 1).

The kernel rounds both operands to bf16 (round-to-nearest-even, after the optional per-k affine) while staging
them in LDS and accumulates in f32, so the expected value is the fp64 product of the bf16-rounded operands:
the tolerance stays at accumulation-order level.  Same layout / remap / batching cases as the f32 tests."""
import pytest
import torch

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from step_amd import _lib
    _lib.lib()
    return _lib


def r16(x):
    return x.to(torch.bfloat16).double()


@pytest.mark.parametrize("M,N,K,ta,tb", [(70, 50, 33, False, False), (307, 307, 384, True, False),
                                         (129, 257, 1000, False, True), (33, 100, 2912, True, True),
                                         (500, 32, 32, False, False), (32, 224, 5000, True, False),
                                         (256, 128, 2048, False, True), (128, 64, 1100, True, False), (64, 256, 1536, False, False),
                                         (260, 132, 2052, True, True), (1024, 512, 640, True, False), (1024, 640, 512, False, True)])
def test_gemm_bf16_layouts(L, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    want = (r16(A).T if ta else r16(A)) @ (r16(B).T if tb else r16(B))
    Ad, Bd = A.cuda(), B.cuda()
    C = torch.full((M, N), float("nan"), device="cuda")
    sam, sak = (1, M) if ta else (K, 1)
    sbk, sbn = (1, K) if tb else (N, 1)
    L.gemm(Ad, Bd, C, M, N, K, sam, sak, sbk, sbn, N, compute_bf16=True)
    assert rel_l2(C.cpu(), want) < 2e-6
    # against the unrounded product: bf16 operand rounding only (2^-9 relative per operand, random signs)
    full = (A.T if ta else A).double() @ (B.T if tb else B).double()
    assert rel_l2(C.cpu(), full) < 6e-3
    bias = torch.randn(N, generator=g)
    C2 = torch.ones((M, N), device="cuda")
    L.gemm(Ad, Bd, C2, M, N, K, sam, sak, sbk, sbn, N, accumulate=1, bias=bias.cuda(), relu=True, alpha=0.5, compute_bf16=True)
    want2 = torch.relu(0.5 * want + 1.0 + bias.double())
    assert rel_l2(C2.cpu(), want2) < 2e-6
    C3 = torch.zeros((M, N), device="cuda")
    L.gemm(Ad, Bd, C3, M, N, K, sam, sak, sbk, sbn, N, accumulate=2, splitk=4, compute_bf16=True)
    assert rel_l2(C3.cpu(), want) < 2e-6
    C4 = torch.zeros((M, N), device="cuda")
    L.gemm(Ad, Bd, C4, M, N, K, sam, sak, sbk, sbn, N, accumulate=2, splitk=-1, compute_bf16=True)
    assert rel_l2(C4.cpu(), want) < 2e-6


def test_gemm_bf16_inputs_already_bf16(L):
    g = torch.Generator().manual_seed(3)
    Bn, N, F = 3, 45, 700
    H = torch.randn(Bn, N, F, generator=g).to(torch.bfloat16)
    want = H.double() @ H.double().transpose(1, 2)
    Hd = H.cuda()
    C = torch.zeros(Bn, N, N, device="cuda")
    L.gemm(Hd, Hd, C, N, N, F, F, 1, 1, F, N, batch=Bn, sab=N * F, sbb=N * F, scb=N * N, accumulate=2, splitk=3, compute_bf16=True)
    assert rel_l2(C.cpu(), want) < 2e-6


def test_gemm_bf16_slot_remap_and_kscale(L):
    g = torch.Generator().manual_seed(9)
    Nn, T, S = 21, 5, 7
    cat = torch.randn(Nn, T, S * 32, generator=g)
    P = torch.rand(Nn, Nn, generator=g)
    catd, Pd = cat.cuda(), P.cuda()
    L.gemm(Pd, catd, catd, Nn, T * 32, Nn, 1, Nn, T * S * 32, 1, T * S * 32, b_off=32, c_off=96,
           b_n=(32, S * 32), c_n=(32, S * 32), compute_bf16=True)
    want = torch.einsum("vw,vtc->wtc", r16(P), r16(cat[:, :, 32:64]))
    got = catd.cpu()
    assert rel_l2(got[:, :, 96:128], want) < 2e-6
    assert torch.equal(got[:, :, :96], cat[:, :, :96]) and torch.equal(got[:, :, 128:], cat[:, :, 128:])
    dP = torch.empty(Nn, Nn, device="cuda")
    L.gemm(catd, catd, dP, Nn, Nn, T * 32, T * S * 32, 1, 1, T * S * 32, Nn, a_off=32, b_off=64,
           a_k=(32, S * 32), b_k=(32, S * 32), compute_bf16=True)
    want = torch.einsum("vtc,wtc->vw", r16(cat[:, :, 32:64]), r16(cat[:, :, 64:96]))
    assert rel_l2(dP.cpu(), want) < 2e-6
    A = torch.randn(19, 6 * 50, generator=g)
    B = torch.randn(6 * 50, 10, generator=g)
    sc, sh = torch.randn(6, generator=g), torch.randn(6, generator=g)
    C = torch.empty(19, 10, device="cuda")
    L.gemm(A.cuda(), B.cuda(), C, 19, 10, 300, 300, 1, 10, 1, 10, a_kscale=sc.cuda(), a_kshift=sh.cuda(), a_kperiod=50,
           compute_bf16=True)
    An = torch.addcmul(sh[None, :, None].expand(19, 6, 50), A.reshape(19, 6, 50), sc[None, :, None])     # fused multiply-add in f32
    An2 = A.reshape(19, 6, 50) * sc[None, :, None] + sh[None, :, None]
    e = min(rel_l2(C.cpu(), r16(An).reshape(19, 300) @ r16(B)), rel_l2(C.cpu(), r16(An2).reshape(19, 300) @ r16(B)))
    assert e < 3e-4          # an fma/no-fma difference can flip single bf16 roundings


def test_gemm_bf16_two_level_batch(L):
    g = torch.Generator().manual_seed(21)
    Bn, Nn, T, S = 2, 37, 3, 7
    P = torch.rand(3, Bn, Nn, Nn, generator=g)
    cat = torch.randn(Bn, Nn, T, S * 32, generator=g)
    Pd, catd = P.cuda(), cat.cuda()
    L.gemm(Pd, catd, catd, Nn, T * 32, Nn, 1, Nn, T * S * 32, 1, T * S * 32, batch=3 * Bn, batch0=Bn, sab=Nn * Nn, sab1=Bn * Nn * Nn,
           sbb=Nn * T * S * 32, sbb1=64, scb=Nn * T * S * 32, scb1=64, b_off=32, c_off=64, b_n=(32, S * 32), c_n=(32, S * 32),
           compute_bf16=True)
    got = catd.cpu()
    for s in range(3):
        want = torch.einsum("bvw,bvtc->bwtc", r16(P[s]), r16(cat[..., 32 + 64 * s:64 + 64 * s]))
        assert rel_l2(got[..., 64 + 64 * s:96 + 64 * s], want) < 2e-6, s
    assert torch.equal(got[..., :64], cat[..., :64])
    out = torch.zeros(Bn, Nn, T, S * 32, device="cuda")
    L.gemm(Pd, catd, out, Nn, T * 32, Nn, 1, Nn, T * S * 32, 1, T * S * 32, batch=3 * Bn, batch0=Bn, sab=Nn * Nn, sab1=Bn * Nn * Nn,
           sbb=Nn * T * S * 32, sbb1=64, scb=Nn * T * S * 32, scb1=0, b_off=32, b_n=(32, S * 32), c_n=(32, S * 32), accumulate=2,
           compute_bf16=True)
    want = sum(torch.einsum("bvw,bvtc->bwtc", r16(P[s]), r16(got[..., 32 + 64 * s:64 + 64 * s])) for s in range(3))
    assert rel_l2(out.cpu()[..., :32], want) < 2e-6
    dP = torch.zeros(3, Bn, Nn, Nn, device="cuda")
    L.gemm(catd, catd, dP, Nn, Nn, T * 32, T * S * 32, 1, 1, T * S * 32, Nn, batch=3 * Bn, batch0=Bn, sab=Nn * T * S * 32, sab1=64,
           sbb=Nn * T * S * 32, sbb1=64, scb=Nn * Nn, scb1=Bn * Nn * Nn, a_off=32, b_off=64, a_k=(32, S * 32), b_k=(32, S * 32),
           accumulate=1, compute_bf16=True)
    for s in range(3):
        want = torch.einsum("bvtc,bwtc->bvw", r16(got[..., 32 + 64 * s:64 + 64 * s]), r16(got[..., 64 + 64 * s:96 + 64 * s]))
        assert rel_l2(dP.cpu()[s], want) < 2e-6, s


def test_gemm_bf16_fast_path_hop_shapes(L):
    """The production shapes of the fast path: bf16 k-contiguous support stack with a zero-padded pitch (K = 307 is not a
    multiple of 8), n-contiguous slot-strided activations, two-level batch, ragged M/N/K tails."""
    g = torch.Generator().manual_seed(5)
    Bn, Nn, T, S = 2, 307, 3, 7
    N8 = (Nn + 7) // 8 * 8
    P = torch.rand(3, Bn, Nn, Nn, generator=g) / Nn
    PT16 = torch.zeros(3, Bn, Nn, N8, dtype=torch.bfloat16)
    PT16[..., :Nn] = P.transpose(2, 3).to(torch.bfloat16)
    PT16[..., Nn:] = float("nan")          # the kernel must mask k >= K itself
    cat = torch.randn(Bn, Nn, T, S * 32, generator=g)
    Pd, catd = PT16.cuda(), cat.cuda()
    L.gemm(Pd, catd, catd, Nn, T * 32, Nn, N8, 1, T * S * 32, 1, T * S * 32, batch=3 * Bn, batch0=Bn, sab=Nn * N8, sab1=Bn * Nn * N8,
           sbb=Nn * T * S * 32, sbb1=64, scb=Nn * T * S * 32, scb1=64, b_off=32, c_off=64, b_n=(32, S * 32), c_n=(32, S * 32),
           compute_bf16=True)
    got = catd.cpu()
    for s in range(3):
        want = torch.einsum("bvw,bvtc->bwtc", r16(P[s]), r16(cat[..., 32 + 64 * s:64 + 64 * s]))
        assert rel_l2(got[..., 64 + 64 * s:96 + 64 * s], want) < 2e-6, s
    assert torch.equal(got[..., :64], cat[..., :64])
    # adjoint: dP[v][w] = sum_(t,c) x[v][t][c] dy[w][t][c]  (both operands k-contiguous with a slot remap along k)
    dP = torch.zeros(3, Bn, Nn, Nn, device="cuda")
    L.gemm(catd, catd, dP, Nn, Nn, T * 32, T * S * 32, 1, 1, T * S * 32, Nn, batch=3 * Bn, batch0=Bn, sab=Nn * T * S * 32, sab1=64,
           sbb=Nn * T * S * 32, sbb1=64, scb=Nn * Nn, scb1=Bn * Nn * Nn, a_off=32, b_off=64, a_k=(32, S * 32), b_k=(32, S * 32),
           accumulate=1, compute_bf16=True)
    for s in range(3):
        want = torch.einsum("bvtc,bwtc->bvw", r16(got[..., 32 + 64 * s:64 + 64 * s]), r16(got[..., 64 + 64 * s:96 + 64 * s]))
        assert rel_l2(dP.cpu()[s], want) < 2e-6, s


@pytest.mark.parametrize("K,period", [(300, 50), (13 * 47, 47), (4 * 33, 33)])
def test_gemm_bf16_fast_path_kscale_and_ktail(L, K, period):
    """Per-channel affine along k whose period is not a multiple of the 4-wide loads (the DGL fc: period 13581), padded pitch."""
    g = torch.Generator().manual_seed(K)
    M, N = 75, 12
    Kp = (K + 3) // 4 * 4
    A = torch.full((M, Kp), float("nan"))
    A[:, :K] = torch.randn(M, K, generator=g)
    Bm = torch.full((N, Kp), float("nan"))
    Bm[:, :K] = torch.randn(N, K, generator=g)
    nch = K // period
    sc, sh = torch.randn(nch, generator=g), torch.randn(nch, generator=g)
    C = torch.empty(M, N, device="cuda")
    L.gemm(A.cuda(), Bm.cuda(), C, M, N, K, Kp, 1, 1, Kp, N, a_kscale=sc.cuda(), a_kshift=sh.cuda(), a_kperiod=period, compute_bf16=True)
    An = A[:, :K].reshape(M, nch, period) * sc[None, :, None] + sh[None, :, None]
    want = r16(An.reshape(M, K)) @ r16(Bm[:, :K]).T
    assert rel_l2(C.cpu(), want) < 3e-4
    # same without the affine, split-K
    C2 = torch.zeros(M, N, device="cuda")
    L.gemm(A.cuda(), Bm.cuda(), C2, M, N, K, Kp, 1, 1, Kp, N, accumulate=2, splitk=2, compute_bf16=True)
    assert rel_l2(C2.cpu(), r16(A[:, :K]) @ r16(Bm[:, :K]).T) < 2e-6
