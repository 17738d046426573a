"""Lane-level numpy mirror of csrc/pretrain_fused.hip: the operand-fragment index maps of the fused feed-forward kernels, executed with a
model of v_mfma_f32_32x32x16 (lane = 32 h + r; operand slot j of lane-half h pairs with the same slot of the other operand; result
register e of lane (h, n) = row (e & 3) + 8 (e >> 2) + 4 h, column n).  Used by tests/test_pretrain_fused_maps.py (CPU) to check that the
maps compute the feed-forward block's forward, input gradient and weight gradients, and by tests/test_gpu_pretrain.py to rebuild the
keep decisions of the device pool on the host."""
import numpy as np

M32 = 0xFFFFFFFF
LANE = np.arange(64)
R_, H_ = LANE % 32, LANE // 32
J8 = np.arange(8)
E16 = np.arange(16)


def chain_f(s, h, j):
    return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)


def row16(e, h):
    return (e & 3) + 8 * (e >> 2) + 4 * h


def mfma(a, b, c):
    """a, b: [64, 8] operand registers, c: [64, 16] accumulators -> c + A.B in accumulator layout."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for h in (0, 1):
        A[:, 8 * h:8 * h + 8] = a[32 * h:32 * h + 32]
        B[8 * h:8 * h + 8, :] = b[32 * h:32 * h + 32].T
    D = A @ B
    out = c.copy()
    for h in (0, 1):
        out[32 * h:32 * h + 32] += D[row16(E16, h)].T          # lane n of half h, register e <- D[row16(e, h)][n]
    return out


def mix32(x):
    x &= M32
    x ^= x >> 16; x = (x * 0x85EBCA6B) & M32
    x ^= x >> 13; x = (x * 0xC2B2AE35) & M32
    x ^= x >> 16
    return x


def mask_base(seed64, site, tile32, pool_words):
    seed = (seed64 ^ (seed64 >> 32)) & M32
    return mix32(seed + (tile32 & M32) * 0x9E3779B1 + (site + 1) * 0x632BE5AB) & (pool_words - 1)


def keep_matrix(pool, seed64, site, R):
    """[R, 384] booleans: the keep decision of (row, hidden unit) -- word c * 16 + i of the row tile's window, bit 32 h + r <-> row r of
    the tile, unit 32 c + (i & 3) + 8 (i >> 2) + 4 h.  pool: uint64 numpy array of a power-of-two number of words."""
    words = pool.shape[0]
    keep = np.zeros((R, 384), dtype=bool)
    u = np.arange(384)
    c, ul = u // 32, u % 32
    i = (ul & 3) + 4 * (ul >> 3)
    hp = (ul >> 2) & 1
    for tile in range((R + 31) // 32):
        base = mask_base(seed64, site, tile, words)
        w = pool[(base + c * 16 + i) & (words - 1)]                      # [384] the word of every unit
        n = min(32, R - 32 * tile)
        sh = (32 * hp[None, :] + np.arange(n)[:, None]).astype(np.uint64)
        keep[32 * tile:32 * tile + n] = ((w[None, :] >> sh) & np.uint64(1)).astype(bool)
    return keep


# ------------------------------------------------------------------------------------------------ fragments (mirror of ffn_pack_kernel)
def a_frag(w, row0, k0, s):
    """A-operand fragment [64, 8]: lane (h, r) slot j -> w[row0 + r, k0 + F(s, h, j)]"""
    return w[(row0 + R_)[:, None], k0 + chain_f(s, H_[:, None], J8[None, :])]


def a_frag_T(w, k_row0, s, col0):
    """lane (h, r) slot j -> w[k_row0 + F(s, h, j), col0 + r]   (rows of w are the contracted index)"""
    return w[k_row0 + chain_f(s, H_[:, None], J8[None, :]), (col0 + R_)[:, None]]


def pack(w1, w2):
    """per chunk c: W1 fragments [6], W2 fragments [3][2], W2^T fragments [6], W1^T fragments [3][2]"""
    out = []
    for c in range(12):
        w1a = [a_frag(w1, 32 * c, 32 * (ks >> 1), ks & 1) for ks in range(6)]
        w2c = [[a_frag(w2, 32 * t, 32 * c, s) for s in range(2)] for t in range(3)]
        w2t = [a_frag_T(w2, 32 * (ks >> 1), ks & 1, 32 * c) for ks in range(6)]
        w1t = [[a_frag_T(w1, 32 * c, s, 32 * t) for s in range(2)] for t in range(3)]
        out.append((w1a, w2c, w2t, w1t))
    return out


def rows_T(x, row0):
    """a 32-row tile in the transposed register layout (what tile_put + tile_frags deliver, before the bf16 pack): [3][64, 16],
    a[t][lane, 4 q + i] = x[row0 + r, 32 t + 8 q + 4 h + i]"""
    R = x.shape[0]
    a = np.zeros((3, 64, 16))
    for t in range(3):
        for e in range(16):
            rows = row0 + R_
            ok = rows < R
            a[t][ok, e] = x[rows[ok], 32 * t + row16(e, H_[ok])]
    return a


def pack_T(a):
    """the six operand fragments of a tile (tile_frags): b[2 t + s][lane, j] = a[t][lane, 8 s + j]"""
    return [a[t][:, 8 * s:8 * s + 8] for t in range(3) for s in range(2)]


def y_frags(x, row0):
    """rows as the contracted index: [t][s][64, 8], lane (h, r) slot j -> x[row0 + F(s, h, j), 32 t + r]"""
    R = x.shape[0]
    out = []
    for t in range(3):
        fs = []
        for s in range(2):
            rows = row0 + chain_f(s, H_[:, None], J8[None, :])
            v = np.where(rows < R, x[np.minimum(rows, R - 1), (32 * t + R_)[:, None]], 0.0)
            fs.append(v)
        out.append(fs)
    return out


def lane_masks(pool, base, c, words):
    """the 16 words of chunk c as [64, 16] booleans in the transposed layout (lane = row of the tile + 32 h, register i)"""
    m = np.zeros((64, 16), dtype=bool)
    for i in range(16):
        w = int(pool[(base + c * 16 + i) & (words - 1)])
        m[:, i] = [(w >> l) & 1 for l in range(64)]
    return m


# ------------------------------------------------------------------------------------------------ the kernels, one 32-row tile at a time
def forward(x, w1, b1, w2, b2, pool=None, seed=0, site=0, inv_keep=1.0):
    R = x.shape[0]
    pk = pack(w1, w2)
    out = np.zeros((R, 96))
    for tile in range((R + 31) // 32):
        xb = pack_T(rows_T(x, 32 * tile))
        acc = np.zeros((3, 64, 16))
        base = mask_base(seed, site, tile, pool.shape[0]) if pool is not None else 0
        for c in range(12):
            w1a, w2c, _, _ = pk[c]
            hh = b1[32 * c + row16(E16[None, :], H_[:, None])].astype(float)
            for ks in range(6):
                hh = mfma(w1a[ks], xb[ks], hh)
            if pool is not None:
                hh = np.where(lane_masks(pool, base, c, pool.shape[0]), hh, 0.0)
            hh = np.maximum(hh, 0.0)
            hb = [hh[:, :8], hh[:, 8:]]
            for t in range(3):
                acc[t] = mfma(w2c[t][0], hb[0], acc[t])
                acc[t] = mfma(w2c[t][1], hb[1], acc[t])
        for t in range(3):
            for e in range(16):
                rows = 32 * tile + R_
                ok = rows < R
                out[rows[ok], 32 * t + row16(e, H_[ok])] = acc[t][ok, e] * inv_keep + b2[32 * t + row16(e, H_[ok])]
    return out


def backward_data(df2, x, w1, b1, w2, dh1, pool=None, seed=0, site=0, inv_keep=1.0):
    R = x.shape[0]
    pk = pack(w1, w2)
    out = dh1.copy()
    for tile in range((R + 31) // 32):
        xb, db = pack_T(rows_T(x, 32 * tile)), pack_T(rows_T(df2, 32 * tile))
        acc = rows_T(dh1, 32 * tile)
        base = mask_base(seed, site, tile, pool.shape[0]) if pool is not None else 0
        for c in range(12):
            w1a, _, w2t, w1t = pk[c]
            hh = b1[32 * c + row16(E16[None, :], H_[:, None])].astype(float)
            dd = np.zeros((64, 16))
            for ks in range(6):
                hh = mfma(w1a[ks], xb[ks], hh)
                dd = mfma(w2t[ks], db[ks], dd)
            open_ = hh > 0
            if pool is not None:
                open_ &= lane_masks(pool, base, c, pool.shape[0])
            dd = np.where(open_, dd * inv_keep, 0.0)
            for t in range(3):
                acc[t] = mfma(w1t[t][0], dd[:, :8], acc[t])
                acc[t] = mfma(w1t[t][1], dd[:, 8:], acc[t])
        for t in range(3):
            for e in range(16):
                rows = 32 * tile + R_
                ok = rows < R
                out[rows[ok], 32 * t + row16(e, H_[ok])] = acc[t][ok, e]
    return out


def backward_weights(df2, x, w1, b1, w2, pool=None, seed=0, site=0, inv_keep=1.0):
    """-> dw1 [384, 96], db1 [384], dw2 [96, 384] (wave = chunk of 32 hidden units, lane = unit)"""
    R = x.shape[0]
    pk = pack(w1, w2)
    dw1, db1, dw2 = np.zeros((384, 96)), np.zeros(384), np.zeros((96, 384))
    u = R_
    widx = (u & 3) + 4 * (u >> 3)
    hp = (u >> 2) & 1
    for c in range(12):
        w1a, _, w2t, _ = pk[c]
        acc2, acc1, dbias = np.zeros((3, 64, 16)), np.zeros((3, 64, 16)), np.zeros(64)
        for tile in range((R + 31) // 32):
            xh, xd = pack_T(rows_T(x, 32 * tile)), pack_T(rows_T(df2, 32 * tile))
            yh, yd = y_frags(x, 32 * tile), y_frags(df2, 32 * tile)
            hh = np.repeat(b1[32 * c + u][:, None], 16, axis=1).astype(float)
            dd = np.zeros((64, 16))
            for ks in range(6):
                hh = mfma(xh[ks], w1a[ks], hh)
                dd = mfma(xd[ks], w2t[ks], dd)
            open_ = hh > 0
            if pool is not None:
                words = pool.shape[0]
                base = mask_base(seed, site, tile, words)
                word = pool[(base + c * 16 + widx) & (words - 1)] >> (32 * hp).astype(np.uint64)
                bits = (word & np.uint64(M32)) >> (4 * H_).astype(np.uint64)
                open_ &= ((bits[:, None] >> ((E16 & 3) + 8 * (E16 >> 2)).astype(np.uint64)[None, :]) & np.uint64(1)).astype(bool)
            hid = np.where(open_, hh, 0.0)
            dh = np.where(open_, dd, 0.0)
            dbias += dh.sum(1)
            for t in range(3):
                acc2[t] = mfma(yd[t][0], hid[:, :8], acc2[t])
                acc2[t] = mfma(yd[t][1], hid[:, 8:], acc2[t])
                acc1[t] = mfma(dh[:, :8], yh[t][0], acc1[t])
                acc1[t] = mfma(dh[:, 8:], yh[t][1], acc1[t])
        for t in range(3):
            for e in range(16):
                dw2[32 * t + row16(e, H_), 32 * c + u] += acc2[t][:, e] * inv_keep
                dw1[32 * c + row16(e, H_), 32 * t + u] += acc1[t][:, e] * inv_keep
        db1[32 * c + np.arange(32)] += (dbias[:32] + dbias[32:]) * inv_keep
    return dw1, db1, dw2
