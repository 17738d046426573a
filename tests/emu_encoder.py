"""Lane-level numpy emulation of csrc/tsformer_encoder.hip (TEST INFRASTRUCTURE).

Replays the kernel's data flow -- operand fragments read from the packed weight buffer at the
byte offsets of csrc/tsformer_layout.h, MFMA 32x32x16 evaluated in (lane, slot) space, LDS
fragment exchange between waves -- so that packing and index arithmetic can be checked on a
machine without a GPU.  ``round_bf16=False`` keeps every operand in float64, which must agree
with the oracle to round-off; ``True`` mimics the kernel's bf16 operand rounding.

``OPERAND`` (module global, default bfloat16) is the 16-bit operand type: tools/encoder_precision_study.py switches it to
float16 to price the same data flow with fp16 operand fragments (same MFMA rate, 3 more mantissa bits).
"""
import numpy as np
import torch

from step_amd import tsformer_pack as TP
from tests import enc_dropout_host as DH

THR, BIAS = 100.0, 60.0          # TSF_THR / TSF_BIAS of csrc/tsformer_encoder.hip
LIMIT = 2.0 ** 110               # TSF_LIMIT: largest softmax denominator the fast schedule accepts
STATS = {"reshifts": 0, "tiles": 0, "redone": 0}

LANES = np.arange(64)
H = LANES // 32
C = LANES % 32
ROW = np.stack([(np.arange(16) & 3) + 8 * (np.arange(16) >> 2) + 4 * h for h in (0, 1)])   # [2,16]


OPERAND = torch.bfloat16
PEAK = {"abs": 0.0}                       # largest operand magnitude seen by pack_half (range check for a float16 variant)


def bf16_round(a, dt=None):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dt or OPERAND).to(torch.float64).numpy()


def round_to_operand(x):
    """csrc/tsformer_device.h round_to_operand: nearest value of the operand type (float16 clamps at +-60000)."""
    x = np.asarray(x, dtype=np.float64)
    if OPERAND == torch.float16:
        x = np.clip(x, -60000.0, 60000.0)
    return bf16_round(x)


class Buf:
    def __init__(self, packed):
        self.raw = packed.numpy().tobytes()

    def f32(self, off, n):
        return np.frombuffer(self.raw, dtype=np.float32, count=n, offset=off).astype(np.float64)

    def frag(self, off, idx):
        b = np.frombuffer(self.raw, dtype=np.int16, count=512, offset=off + idx * 1024)
        return torch.from_numpy(b.copy()).view(OPERAND).to(torch.float64).numpy().reshape(64, 8)


def mfma(a, b, c):
    """a, b: [64, 8] operand slots; c: [64, 16] accumulator.  D[lane(h,col)][i] += sum_{h',j}
    a[32 h' + row(i,h)][j] * b[32 h' + col][j]."""
    d = c.copy()
    for l in range(64):
        h, col = l // 32, l % 32
        for i in range(16):
            row = ROW[h, i]
            d[l, i] += a[row] @ b[col] + a[32 + row] @ b[32 + col]
    return d


def mfma_fast(a, b, c):
    full = np.einsum("hrj,hcj->rc", a.reshape(2, 32, 8), b.reshape(2, 32, 8))        # [row, col]
    d = c.copy()
    for h in (0, 1):
        d[32 * h:32 * h + 32, :] += full[ROW[h]][:, :].T                              # lane col, reg i
    return d


def pack_half(v, s, rnd, dt=None):
    x = v[:, 8 * s:8 * s + 8]
    PEAK["abs"] = max(PEAK["abs"], float(np.abs(x[np.isfinite(x)]).max(initial=0.0)))
    return bf16_round(x, dt) if rnd else x.copy()


def lane_bits(pool, start):
    """keep bits of the 16 mask words at `start`: [64 lanes, 16 registers]."""
    b = DH._tile_bits(pool, start)                       # [reg, half, column]
    return b.transpose(1, 2, 0).reshape(64, 16).astype(np.float64)


def layer_norm(acc, g, b):
    """acc: [3, 64, 16]; g, b: [2, 48]."""
    tot = acc.sum(axis=(0, 2))
    tot = tot + tot[LANES ^ 32]
    mean = tot / 96.0
    d = acc - mean[None, :, None]
    q = (d * d).sum(axis=(0, 2))
    q = q + q[LANES ^ 32]
    rstd = 1.0 / np.sqrt(q / 96.0 + 1e-5)
    gg = g[H].reshape(64, 3, 16).transpose(1, 0, 2)
    bb = b[H].reshape(64, 3, 16).transpose(1, 0, 2)
    return d * rstd[None, :, None] * gg + bb


def encode_sequence(series, packed, P, depth, round_bf16=True, drop=None, always_reshift=False):
    """series: [L] float; returns hidden [P, 96] (float64).
    drop: None or dict(pool=uint64 array, seed=64-bit seed argument of step_tsformer_encode, seq=sequence index, keep=1-p):
    training-mode dropout with the kernel's keep-mask words."""
    B = Buf(packed)
    rnd = round_bf16
    nkt = (P + 31) // 32
    BF = torch.bfloat16
    thr = -BIAS if always_reshift else THR
    if drop is not None:
        pool = np.asarray(drop["pool"]).view(np.uint64)
        keep = float(drop["keep"])
        dl = DH.DropLayout(nkt)
        s32 = DH.seed32(int(drop["seed"]))
        cbase = lambda layer: DH.chunk_base(s32, int(drop["seq"]), layer, pool.shape[0])
    else:
        keep = 1.0
    LB = TP.layer_bytes()
    G_WPE = TP.HDR
    G_BPE = G_WPE + 2 * 48 * 12 * 4
    G_NG = G_BPE + 2 * 48 * 4
    G_NB = G_NG + 2 * 48 * 4
    L0 = TP.LAYER0
    POS = L0 + depth * LB
    wsec = B.f32(G_WPE, 3 * 6 * 64).reshape(3, 6, 2, 32)                         # A operands of the f32 MFMA: (t, s, k = lane / 32, row = lane % 32)
    wfull = wsec.transpose(0, 3, 1, 2).reshape(96, 12)                           # W_pe [feature 32 t + row][input 2 s + k]
    wpe = np.stack([wfull[(32 * np.arange(3)[:, None] + ROW[h][None, :]).reshape(-1)] for h in (0, 1)])     # [2, 48 (accumulator order), 12]
    bpe = B.f32(G_BPE, 96).reshape(2, 48)
    xT = []
    for w in range(nkt):
        tok = w * 32 + C
        ok = tok < P
        tokc = np.where(ok, tok, 0)
        xin = np.where(ok[:, None], series[tokc[:, None] * 12 + np.arange(12)[None, :]], 0.0)     # [64,12]
        pos = np.stack([B.f32(POS + (int(tokc[l]) * 2 + int(H[l])) * 48 * 4, 48) for l in range(64)])
        e = (np.einsum("lqj,lj->lq", wpe[H], xin) + pos).reshape(64, 3, 16).transpose(1, 0, 2).copy()   # [3,64,16]; b_pe rides in the table
        sc = np.sqrt(96.0)
        if drop is not None:
            for t in range(3):
                e[t] = e[t] * lane_bits(pool, cbase(depth) + dl.d1 + (w * 3 + t) * 16)
            sc = sc / keep
        xT.append(e * sc)
    for layer in range(depth):
        base = L0 + layer * LB
        hblk = lambda hd: base + hd * TP.BLOCK                       # head stage block
        fblk = lambda j: base + (4 + j) * TP.BLOCK                   # ffn stage block
        tailf = lambda blk, off, n: B.f32(blk + TP.TAIL + off * 4, n)
        xb = [[pack_half(xT[w][t], s, rnd) for t in range(3) for s in range(2)] for w in range(nkt)]
        bo = tailf(hblk(0), 64, 96).reshape(2, 48)
        acc = [(xT[w] if drop is None else 0.0) + bo[H].reshape(64, 3, 16).transpose(1, 0, 2) for w in range(nkt)]
        xop = lambda frs: np.stack([np.concatenate([frs[2 * t], frs[2 * t + 1]], axis=1) for t in range(3)])   # operand copy -> [3,64,16]
        skip_fast = [False] * nkt
        for hd in range(4):
            kf, vf, qb = {}, {}, []
            bq = tailf(hblk(hd), 0, 32).reshape(2, 16)
            bv = tailf(hblk(hd), 32, 32)
            for w in range(nkt):
                q = np.zeros((64, 16)); kk = np.zeros((64, 16)); vv = np.zeros((64, 16))
                for ks in range(6):
                    q = mfma_fast(B.frag(hblk(hd), ks), xb[w][ks], q)
                    kk = mfma_fast(B.frag(hblk(hd), 6 + ks), xb[w][ks], kk)
                    vv = mfma_fast(xb[w][ks], B.frag(hblk(hd), 12 + ks), vv)
                q = q + bq[H]
                vv = vv + bv[C][:, None]
                qb.append([pack_half(q, 0, rnd), pack_half(q, 1, rnd)])
                for s in range(2):
                    kf[(w, s)] = pack_half(kk, s, rnd)
                    vf[(w, s)] = pack_half(vv, s, rnd, BF)           # V and P are bfloat16 in both operand modes
            for w in range(nkt):
                def scores(kt):
                    s_ = mfma_fast(kf[(kt, 0)], qb[w][0], np.zeros((64, 16)))
                    s_ = mfma_fast(kf[(kt, 1)], qb[w][1], s_)
                    key = kt * 32 + ROW[H]                                # [64,16]
                    return np.where(key >= P, s_ - 30000.0, s_)          # slot 26: (is_padding | -30000)
                # one pass, online softmax: `shift` rides through the score MFMA (slot 25), see the kernel header
                def key_loop(fast):
                    """fast: the shift stays where the first key tile puts it (no per-tile maximum); else the re-shifting loop."""
                    shift = np.zeros(64)
                    ov = np.zeros((64, 16))
                    lsum = np.zeros(64)
                    for kt in range(nkt):
                        sc = scores(kt) - shift[:, None]
                        tmax = sc.max(axis=1)
                        STATS["tiles"] += 1
                        if kt == 0 or (not fast and bool((tmax > thr).any())):
                            STATS["reshifts"] += kt > 0
                            t = np.maximum(tmax, tmax[LANES ^ 32])
                            upd = (t > thr) | (kt == 0)
                            ns = round_to_operand(shift + t + BIAS) if rnd else shift + t + BIAS
                            d = np.where(upd, ns - shift, 0.0)
                            shift = np.where(upd, ns, shift)
                            a = np.ones(64) if kt == 0 else np.exp2(-d)
                            ov = ov * a[:, None]
                            lsum = lsum * a
                            sc = sc - d[:, None]
                        with np.errstate(over="ignore", invalid="ignore"):
                            p = np.exp2(sc)
                            p = np.where(p > 3.4028234e38, np.inf, p)                  # f32 range: overflows where the kernel does
                            if drop is not None:
                                lsum = lsum + p.sum(axis=1)
                                p = p * lane_bits(pool, cbase(layer) + ((hd * nkt + w) * nkt + kt) * 16)
                            ov = mfma_fast(vf[(kt, 0)], pack_half(p, 0, rnd, BF), ov)
                            ov = mfma_fast(vf[(kt, 1)], pack_half(p, 1, rnd, BF), ov)
                    return ov, lsum
                denom = lambda ov, lsum: ov[C, 12] if drop is None else (lsum + lsum[LANES ^ 32]) * keep
                redo = True
                if not always_reshift and not skip_fast[w]:
                    ov, lsum = key_loop(True)
                    redo = not bool((denom(ov, lsum) < LIMIT).all())        # NaN and inf fail the comparison, as in the kernel
                    STATS["redone"] += redo
                    skip_fast[w] = redo                                     # per wave, sticky for the rest of the layer
                if redo:
                    ov, lsum = key_loop(False)
                den = ov[C, 12] if drop is None else (lsum + lsum[LANES ^ 32]) * keep
                ov = ov / den[:, None]
                ob = [pack_half(ov, 0, rnd), pack_half(ov, 1, rnd)]
                for t in range(3):
                    for s in range(2):
                        acc[w][t] = mfma_fast(B.frag(hblk(hd), 18 + t * 2 + s), ob[s], acc[w][t])
        g1, b1n = tailf(hblk(3), 64, 96).reshape(2, 48), tailf(hblk(3), 160, 96).reshape(2, 48)
        g2, b2n = tailf(fblk(5), 64, 96).reshape(2, 48), tailf(fblk(5), 160, 96).reshape(2, 48)
        b2 = tailf(fblk(0), 64, 96).reshape(2, 48)
        for w in range(nkt):
            if drop is not None:           # dropout1 on (attention output + b_o), residual from its 16-bit operand copy
                m1 = np.stack([lane_bits(pool, cbase(layer) + dl.d1 + (w * 3 + t) * 16) for t in range(3)])
                acc[w] = np.where(m1 > 0, acc[w] / keep, 0.0) + xop(xb[w])
            x1 = layer_norm(acc[w], g1, b1n)
            xbw = [pack_half(x1[t], s, rnd) for t in range(3) for s in range(2)]
            b2l = b2[H].reshape(64, 3, 16).transpose(1, 0, 2)
            a2 = x1 + b2l if drop is None else b2l * keep
            for ch in range(12):
                j, cc = ch // 2, ch % 2
                b1 = tailf(fblk(j), cc * 32, 32).reshape(2, 16)
                hh = np.zeros((64, 16))
                for ks in range(6):
                    hh = mfma_fast(B.frag(fblk(j), cc * 12 + ks), xbw[ks], hh)
                hh = np.maximum(hh + b1[H], 0.0)
                if drop is not None:
                    hh = hh * lane_bits(pool, cbase(layer) + dl.ffn + (w * 12 + ch) * 16)
                hb = [pack_half(hh, 0, rnd), pack_half(hh, 1, rnd)]
                for t in range(3):
                    for s in range(2):
                        a2[t] = mfma_fast(B.frag(fblk(j), cc * 12 + 6 + t * 2 + s), hb[s], a2[t])
            if drop is not None:           # FFN-dropout and dropout2 survivor scales in one multiply, residual from the operand copy
                m2 = np.stack([lane_bits(pool, cbase(layer) + dl.d2 + (w * 3 + t) * 16) for t in range(3)])
                a2 = np.where(m2 > 0, a2 / (keep * keep), 0.0) + xop(xbw)
            xT[w] = layer_norm(a2, g2, b2n)
    ng, nb = B.f32(G_NG, 96).reshape(2, 48), B.f32(G_NB, 96).reshape(2, 48)
    hidden = np.zeros((P, 96))
    for w in range(nkt):
        y = layer_norm(xT[w], ng, nb)
        for l in range(64):
            tok = w * 32 + C[l]
            if tok >= P:
                continue
            for t in range(3):
                hidden[tok, t * 32 + ROW[H[l]]] = y[t, l]
    return hidden
