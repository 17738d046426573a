"""Host mirror of the fused encoder's dropout keep-masks (TEST INFRASTRUCTURE).

The kernel (csrc/tsformer_encoder.hip, layout in csrc/tsformer_device.h) takes its keep-masks as 64-bit LANE masks from a pool
of Bernoulli bits: bit l of a word is lane l's keep flag for one accumulator register.  This module turns a pool into the dense
0/1 masks of the reference's dropout sites (positional_encoding.py:32 and the four sites of torch.nn.TransformerEncoderLayer),
so that the oracle can replay exactly the realisation the kernel used, and restates the pool generator
(step_dropout_pool_fill: Philox4x32-10) in numpy.

Accumulator lane map (v_mfma_f32_32x32x16): lane = 32 h + c, register i  <->  row (i & 3) + 8 (i >> 2) + 4 h, column c.
"""
import numpy as np

M32 = 0xFFFFFFFF


def mix32(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & M32
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & M32
    x ^= x >> 16
    return x


def seed32(seed64):
    """step_tsformer_encode folds its 64-bit seed argument."""
    return (seed64 ^ (seed64 >> 32)) & M32


class DropLayout:
    def __init__(self, nkt):
        self.nkt = nkt
        self.ffn = 4 * nkt * nkt * 16
        self.d1 = self.ffn + nkt * 12 * 16
        self.d2 = self.d1 + nkt * 48
        self.words = self.d2 + nkt * 48


def chunk_base(s32, seq, layer, pool_words):
    return mix32(s32 + seq * 0x9E3779B1 + (layer + 1) * 0x632BE5AB) & (pool_words - 1)


_I = np.arange(16)
ROW = np.stack([(_I & 3) + 8 * (_I >> 2) + 4 * h for h in (0, 1)])        # [half, reg] -> accumulator row


def _tile_bits(pool, start):
    """16 consecutive words (wrapping) -> keep bits [reg 16, half 2, column 32]."""
    n = pool.shape[0]
    w = pool[(start + _I) & (n - 1)].astype(np.uint64)
    bits = (w[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)
    return bits.reshape(16, 2, 32).astype(np.float32)


def _tile_dense(pool, start):
    """-> [row 32, column 32] of the accumulator tile whose 16 mask words start at `start`."""
    b = _tile_bits(pool, start)
    out = np.empty((32, 32), dtype=np.float32)
    for h in (0, 1):
        out[ROW[h]] = b[:, h, :]
    return out


def encoder_masks(pool, seed64, S, P, depth=4, heads=4):
    """pool: uint64 [words] (power of two).  Returns the dense keep-masks (1 = keep) the kernel applies for
    sequences 0..S-1: dict(pos [S,P,96], layers=[dict(attn [S,heads,P(query),P(key)], drop1 [S,P,96], ffn [S,P,384],
    drop2 [S,P,96])])."""
    pool = np.asarray(pool).view(np.uint64)
    W = pool.shape[0]
    nkt = (P + 31) // 32
    dl = DropLayout(nkt)
    s32 = seed32(seed64)
    Pp = nkt * 32

    def feature_site(cb, base):          # [Pp, 96]: tile (wave, t) rows = features t*32.., columns = tokens wave*32..
        m = np.empty((Pp, 96), dtype=np.float32)
        for w in range(nkt):
            for t in range(3):
                m[w * 32:(w + 1) * 32, t * 32:(t + 1) * 32] = _tile_dense(pool, cb + base + (w * 3 + t) * 16).T
        return m[:P]

    out = {"pos": np.empty((S, P, 96), dtype=np.float32), "layers": []}
    for s in range(S):
        out["pos"][s] = feature_site(chunk_base(s32, s, depth, W), dl.d1)
    for layer in range(depth):
        L = {"attn": np.empty((S, heads, P, P), dtype=np.float32), "drop1": np.empty((S, P, 96), dtype=np.float32),
             "ffn": np.empty((S, P, 384), dtype=np.float32), "drop2": np.empty((S, P, 96), dtype=np.float32)}
        for s in range(S):
            cb = chunk_base(s32, s, layer, W)
            for hd in range(heads):
                a = np.empty((Pp, Pp), dtype=np.float32)            # [query, key]
                for w in range(nkt):
                    for kt in range(nkt):
                        # score tile S^T: rows = keys of tile kt, columns = queries of tile w
                        a[w * 32:(w + 1) * 32, kt * 32:(kt + 1) * 32] = _tile_dense(pool, cb + ((hd * nkt + w) * nkt + kt) * 16).T
                L["attn"][s, hd] = a[:P, :P]
            f = np.empty((Pp, 384), dtype=np.float32)
            for w in range(nkt):
                for ch in range(12):
                    f[w * 32:(w + 1) * 32, ch * 32:(ch + 1) * 32] = _tile_dense(pool, cb + dl.ffn + (w * 12 + ch) * 16).T
            L["ffn"][s] = f[:P]
            L["drop1"][s] = feature_site(cb, dl.d1)
            L["drop2"][s] = feature_site(cb, dl.d2)
        out["layers"].append(L)
    return out


# ----------------------------------------------------------------------------- pool generator (Philox4x32-10)
def philox4x32(c, k0, k1):
    """c: uint32 [n, 4] counters; returns uint32 [n, 4].  Same rounds and constants as csrc/common.h philox4x32."""
    c0, c1, c2, c3 = [c[:, i].astype(np.uint64) for i in range(4)]
    k0 = np.uint64(k0 & M32)
    k1 = np.uint64(k1 & M32)
    A, B = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    m = np.uint64(M32)
    for _ in range(10):
        p0 = A * c0
        p1 = B * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & m
        hi1, lo1 = p1 >> np.uint64(32), p1 & m
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & m
        k1 = (k1 + np.uint64(0xBB67AE85)) & m
    return np.stack([c0, c1, c2, c3], 1).astype(np.uint32)


def pool_fill(words, dropout_p, seed64):
    """numpy restatement of step_dropout_pool_fill: uint64 [words]."""
    thresh = int(float(np.float32(dropout_p)) * 4294967296.0) & M32
    w = np.arange(words, dtype=np.uint64)
    out = np.zeros(words, dtype=np.uint64)
    for q in range(16):                                  # lanes 4q .. 4q+3 share one Philox call
        c = np.stack([w & np.uint64(M32), w >> np.uint64(32), np.full(words, q, np.uint64), np.full(words, 0x5EEDD80F, np.uint64)], 1)
        r = philox4x32(c.astype(np.uint32), seed64 & M32, (seed64 >> 32) & M32)
        for j in range(4):
            keep = (r[:, j].astype(np.uint64) >= np.uint64(thresh)).astype(np.uint64)
            out |= keep << np.uint64(4 * q + j)
    return out
