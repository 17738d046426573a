"""Element-level numpy emulation of gemm_fast_kernel (csrc/gemm_bf16.hip) -- TEST INFRASTRUCTURE.

Replays, thread by thread, which global element every operand loader fetches, how stage_fix masks / transforms it, where it
lands in the LDS tile, which LDS words a lane feeds to the matrix cores in every k step (bf16: 8 consecutive k at
ks*16 + 8*(lane>>5); exact-f32 variant: the 16-byte chunk 2*ks + (lane>>5), one element per v_mfma_f32_32x32x2_f32) and where
the accumulator registers are stored -- including the slot remaps, the two-level batch, the per-k affine, split-K, the
all-ones column (a_rowsum) and the column-block affine.  Operands are float64; ``bf16`` rounds them like the kernel does.
"""
import numpy as np
import torch

KC_F32, KC_BF16, MC_F32 = 0, 1, 2
FBK = 64


def rbf16(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def remap(i, blk, stride):
    return i if not blk else (i // blk) * stride + (i % blk)


class Desc:
    def __init__(self, **kw):
        d = dict(batch=1, sab=0, sbb=0, scb=0, alpha=1.0, accumulate=0, bias=None, relu=False, splitk=1, a_kblk=0, a_kstride=0,
                 b_kblk=0, b_kstride=0, b_nblk=0, b_nstride=0, c_nblk=0, c_nstride=0, a_kscale=None, a_kshift=None, a_kperiod=0,
                 batch0=0, sab1=0, sbb1=0, scb1=0, a_rowsum=None, c_nscale=None, c_nshift=None, c_mvec=None, c_nperiod=0,
                 a_off=0, b_off=0, c_off=0)
        d.update(kw)
        self.__dict__.update(d)


def stage(mode, BR, flat, base, row0, rows, srow, sk, k0, kend, kblk, kstride, rblk, rstride, kscale, kshift, kperiod, ones_row, bf16):
    """One operand tile [BR][FBK] as the kernel stages it (stage_load + stage_fix + stage_store)."""
    lds = np.zeros((BR, FBK))
    nthreads = 256
    if mode in (KC_F32, KC_BF16):
        vec = 4 if mode == KC_F32 else 8
        per_row = FBK // vec
        for e in range(BR * per_row):
            k = k0 + (e % per_row) * vec
            row = row0 + e // per_row
            ok = row < rows and k < kend
            off = base + (remap(row, rblk, rstride) * srow + remap(k, kblk, kstride) if ok else 0)
            v = flat[off:off + vec].astype(np.float64).copy()               # one 16-byte load
            for i in range(vec):
                kk = k + i
                val = v[i]
                if mode == KC_F32 and kscale is not None:
                    c0 = k0 // kperiod
                    c1 = c0 + 1 if (c0 + 1) * kperiod < kend else c0
                    kb = (c0 + 1) * kperiod
                    val = val * kscale[c0] + kshift[c0] if kk < kb else val * kscale[c1] + kshift[c1]
                if kk >= kend:
                    val = 0.0
                elif row >= rows:
                    val = 1.0 if row == ones_row else 0.0
                lds[e // per_row, (e % per_row) * vec + i] = val
    else:
        for e in range(BR * FBK // 16):
            row = row0 + (e % (BR // 4)) * 4
            kb = k0 + (e // (BR // 4)) * 4
            for j in range(4):
                ok = row < rows and kb + j < kend
                off = base + (remap(row, rblk, rstride) * srow + remap(kb + j, kblk, kstride) * sk if ok else 0)
                v = flat[off:off + 4].astype(np.float64).copy()
                for i in range(4):
                    kok = kb + j < kend
                    val = 0.0
                    if kok:
                        val = 1.0 if row + i == ones_row else (v[i] if row < rows else 0.0)
                    lds[(e % (BR // 4)) * 4 + i, (e // (BR // 4)) * 4 + j] = val
    return rbf16(lds) if bf16 else lds


def tile_product(As, Bs, BM, BN, f32c):
    """What the four waves accumulate from one staged k tile, through the lane maps of the two MFMA shapes."""
    out = np.zeros((BM, BN))
    TM, TN = BM // 64, BN // 64
    for wave in range(4):
        wr, wc = wave >> 1, wave & 1
        for i in range(TM):
            for j in range(TN):
                ra, rb = wr * TM * 32 + i * 32, wc * TN * 32 + j * 32
                acc = np.zeros((32, 32))
                if f32c:
                    for ks in range(FBK // 8):
                        for t in range(4):                       # MFMA #t of this step: k = 0 from lanes h = 0, k = 1 from h = 1
                            for h in (0, 1):
                                col = (2 * ks + h) * 4 + t
                                acc += np.outer(As[ra:ra + 32, col], Bs[rb:rb + 32, col])
                else:
                    for ks in range(FBK // 16):
                        for h in (0, 1):
                            sl = slice(ks * 16 + h * 8, ks * 16 + h * 8 + 8)
                            acc += As[ra:ra + 32, sl] @ Bs[rb:rb + 32, sl].T
                out[ra:ra + 32, rb:rb + 32] = acc
    return out


def gemm_fast(g, A, B, C, amode, bmode, BM=64, BN=64, bf16=True, f32c=False):
    """A, B, C: flat float64 arrays (element strides as in StepGemm).  Mutates and returns C (and g.a_rowsum)."""
    ksteps = -(-g.K // FBK)
    per = -(-ksteps // g.splitk)
    ones_row = g.N if (g.a_rowsum is not None and bmode != KC_BF16) else -1
    ncols = g.N + (1 if g.a_rowsum is not None else 0)
    for z in range(g.batch * g.splitk):
        zb, zs = z // g.splitk, z % g.splitk
        i0, i1 = (zb % g.batch0, zb // g.batch0) if g.batch0 else (zb, 0)
        abase = g.a_off + i0 * g.sab + i1 * g.sab1
        bbase = g.b_off + i0 * g.sbb + i1 * g.sbb1
        cbase = g.c_off + i0 * g.scb + i1 * g.scb1
        kbeg, kend = zs * per * FBK, min(g.K, (zs + 1) * per * FBK)
        for m0 in range(0, g.M, BM):
            for n0 in range(0, ncols, BN):
                acc = np.zeros((BM, BN))
                for k0 in range(kbeg, kend, FBK):
                    As = stage(amode, BM, A, abase, m0, g.M, g.sam, g.sak, k0, kend, g.a_kblk, g.a_kstride, 0, 0, g.a_kscale, g.a_kshift,
                               g.a_kperiod, -1, bf16 and not f32c)
                    Bs = stage(bmode, BN, B, bbase, n0, g.N, g.sbn, g.sbk, k0, kend, g.b_kblk, g.b_kstride, g.b_nblk, g.b_nstride, None,
                               None, 1, ones_row, bf16 and not f32c)
                    acc += tile_product(As, Bs, BM, BN, f32c)
                for rm in range(BM):
                    gm = m0 + rm
                    if gm >= g.M:
                        continue
                    for cn in range(BN):
                        gn = n0 + cn
                        if gn == ones_row:
                            g.a_rowsum[gm] += g.alpha * acc[rm, cn]
                        if gn >= g.N:
                            continue
                        v = g.alpha * acc[rm, cn]
                        if g.c_nscale is not None:
                            c = gn // g.c_nperiod
                            v = v * g.c_nscale[c] + g.c_nshift[c] * g.c_mvec[gm]
                        ni = remap(gn, g.c_nblk, g.c_nstride)
                        idx = cbase + gm * g.ldc + ni * g.scn
                        if g.accumulate == 2:
                            C[idx] += v
                        else:
                            if g.accumulate == 1:
                                v += C[idx]
                            if g.bias is not None:
                                v += g.bias[gn]
                            C[idx] = max(v, 0.0) if g.relu else v
    return C
