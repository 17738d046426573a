"""Keep-mask pool of the fused encoder: how often two (sequence, layer) units of one launch read the SAME masks.

The reference draws independent Bernoulli bits at every dropout site (positional_encoding.py:32, transformer_layers.py:10).  The
kernel reads 64-bit lane masks from a pool of Philox bits refilled every step; a (sequence, layer) unit reads a contiguous window
at a hashed WORD offset (csrc/tsformer_device.h).  Two windows that overlap at different offsets pair up bits of different
(token, feature / key) positions, which is as harmless as any other reuse of an i.i.d. bit stream; only EQUAL offsets give two
units the same noise at the same positions, and that is what could bias the cosine kNN graph between nodes.  The CPU test counts
those coincidences for the PEMS04 launch; the GPU test measures the cross-sequence correlation of the realised hidden-state noise."""
import numpy as np
import pytest
import torch

from tests import enc_dropout_host as DH


def _bases(seed64, S, depth, words):
    s32 = DH.seed32(seed64)
    return np.array([[DH.chunk_base(s32, s, l, words) for l in range(depth + 1)] for s in range(S)], dtype=np.int64)


def test_pool_chunk_coincidences_at_pems04_launch_size():
    """S = 8 x 307 sequences, 4 layers + the positional site = 12 280 chunks of 10 912 words in the default 2^18-word pool: about
    290 of the 7.5e7 chunk pairs of a launch coincide (4600 with the 16-word aligned offsets of round 2), and of the 376 000 node
    pairs the kNN graph compares, about 7 share the masks of ONE of their five dropout layers -- never of two."""
    from step_amd.step_arch.tsformer import TSFormer
    S, depth, B, N = 2456, 4, 8, 307
    words = TSFormer(12, 1, 96, 4, 4, 0.1, 336, 0.75, 4, 1, mode="forecasting").dropout_pool_words
    assert words == 1 << 18
    chunk = DH.DropLayout(11).words
    assert chunk == 10912 and 2 * chunk <= words
    same, same_sample, twice = [], [], 0
    for seed in (1, 0xC0FFEE1234567, 0x9E3779B97F4A7C15, 77, 2 ** 62 + 5):
        b = _bases(seed, S, depth, words)
        flat = np.sort(b.reshape(-1))
        same.append(int((np.diff(flat) == 0).sum()))
        # pairs of nodes of ONE sample (what the kNN graph compares) that share a whole layer's masks
        cnt = 0
        for smp in range(B):
            blk = b[smp * N:(smp + 1) * N]                       # [N, depth + 1]
            for l in range(depth + 1):
                u, c = np.unique(blk[:, l], return_counts=True)
                cnt += int((c * (c - 1) // 2).sum())
            # two nodes sharing the masks of two different layers at once
            for i in range(depth + 1):
                for j in range(i + 1, depth + 1):
                    pr = blk[:, i] * words + blk[:, j]
                    u, c = np.unique(pr, return_counts=True)
                    twice += int((c > 1).sum())
        same_sample.append(cnt)
    pairs = 12280 * 12279 / 2
    print(f"equal chunk offsets per launch (of {pairs:.3g} pairs; expectation {pairs / words:.0f}): {same}; same layer within one sample "
          f"(of {8 * 5 * 307 * 306 // 2} node pairs; expectation {8 * 5 * 307 * 306 / 2 / words:.2f}): {same_sample}")
    assert max(same) < 1.25 * pairs / words + 30                 # Poisson(288)
    assert max(same_sample) <= 20                                # Poisson(7.2) per launch
    assert twice == 0


@pytest.mark.gpu
def test_realised_dropout_noise_is_uncorrelated_across_sequences():
    """256 sequences with IDENTICAL input (any two units reading the same masks would produce identical states): after removing
    the common shift (the bias of dropout through the non-linear layers), the hidden-state noise of different sequences must be
    uncorrelated -- no pair above 0.05, mean |correlation| at the 1 / sqrt(dimension) level -- and the kNN-relevant statistic,
    the cosine between noisy states of different sequences, must have the spread independent masks give."""
    import step_amd._lib as L
    from step_amd import tsformer_pack as TP
    from tests.helpers import load_golden, params_of
    from tests.test_gpu_kernels import _encode
    g = load_golden("step_tiny")
    p = params_of(g, requires_grad=False)
    sd = {k[len("tsformer."):]: v for k, v in p.items() if k.startswith("tsformer.")}
    P, S = 336, 256
    packed = TP.pack_tsformer(sd, P, operand="f16")
    rng = np.random.default_rng(3)
    row = torch.tensor(np.sin(2 * np.pi * np.arange(P * 12) / 288.0) + 0.5 * rng.standard_normal(P * 12), dtype=torch.float32)
    x = row[None].repeat(S, 1).contiguous().cuda()
    clean, _, _, _ = _encode(L, x, packed, f16=1)
    assert torch.equal(clean[0], clean[-1])
    words = 1 << 18
    pk = packed.cuda()

    def noisy(seed):
        pool = torch.empty(words + 16, dtype=torch.int64, device="cuda")
        L.call("step_dropout_pool_fill", L.ptr(pool), words, 0.1, seed * 7919 + 13, L.stream())
        hid = torch.empty(S, P, 96, device="cuda")
        L.call("step_tsformer_encode", L.ptr(x), S, P * 12, L.ptr(pk), pk.numel(), 4, L.ENC_F16, None, L.ptr(hid), None, None, 0.1,
               L.ptr(pool), words, seed, None, L.stream())
        torch.cuda.synchronize()
        n = (hid - clean).reshape(S, -1).double()
        n = n - n.mean(0, keepdim=True)                          # the shift every sequence shares (bias of dropout through the non-linear layers)
        return hid, n / n.norm(dim=1, keepdim=True)
    hid_a, na = noisy(0x1234567)
    _, nb = noisy(0x7654321)
    same = (na @ na.T).cpu().numpy()[~np.eye(S, dtype=bool)]     # pairs of sequences of ONE launch (one pool)
    null = (na @ nb.T).cpu().numpy().reshape(-1)                 # pairs from two launches: independent pools by construction
    rms = lambda v: float(np.sqrt((v ** 2).mean()))
    q = lambda v: float(np.quantile(np.abs(v), 0.999))
    print(f"correlation of the dropout noise between sequences with identical input ({S} sequences, {na.shape[1]} values each): one launch rms "
          f"{rms(same):.5f}, 99.9 % quantile {q(same):.4f}, max {np.abs(same).max():.4f}; independent pools (null) rms {rms(null):.5f}, "
          f"99.9 % {q(null):.4f}, max {np.abs(null).max():.4f}")
    assert rms(same) < 1.15 * rms(null) + 1e-3 and q(same) < 1.25 * q(null) + 2e-3
    assert np.abs(same).max() < 0.5                               # a pair with identical masks would sit at 1.0
    assert len({hid_a[i].cpu().numpy().tobytes() for i in range(S)}) == S          # no two sequences got the same masks
