"""Lane-level numpy emulation of csrc/dgl_conv_mfma.hip (TEST INFRASTRUCTURE).

Replays the operand addressing of the three conv2 kernels and the conv1 weight-gradient kernel -- which LDS row / element a
lane (n = l & 15, group q = l >> 4) of v_mfma_f32_16x16x32_bf16 reads for every k step, and where its four accumulator
registers land -- so the index arithmetic can be checked on a machine without a GPU.  The MFMA itself is evaluated from the
documented lane maps: A[m = l&15][k = 8 (l>>4) + j], B[k = 8 (l>>4) + j][n = l&15], D[m = 4 (l>>4) + e][n = l&15].
"""
import numpy as np
import torch

KW, CI, CO = 10, 8, 16
LANES = np.arange(64)
LN, Q = LANES & 15, LANES >> 4


def bf16(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def mfma16(a, b, c):
    """a, b: [64, 8] operand slots of the 64 lanes; c: [64, 4].  Returns D in the same lane layout."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        A[LN[l], 8 * Q[l]:8 * Q[l] + 8] = a[l]
        B[8 * Q[l]:8 * Q[l] + 8, LN[l]] = b[l]
    D = A @ B
    d = c.copy()
    for l in range(64):
        for e in range(4):
            d[l, e] += D[4 * Q[l] + e, LN[l]]
    return d


def conv2_fwd(a1, w, b, sc, sh, TT=64):
    """a1 [8][T1] one node; returns a2 [16][T2] = relu(b + conv(bn1(a1)))  -- conv2_fwd_mfma_kernel, one node."""
    T1 = a1.shape[1]; T2 = T1 - (KW - 1)
    out = np.zeros((CO, T2))
    wf = np.zeros((3, 64, 8))
    for s in range(3):
        for l in range(64):
            kk = 4 * s + Q[l]
            if kk < KW:
                wf[s, l] = bf16(w[LN[l], :, kk])
    for t0 in range(0, T2, TT):
        xs = np.zeros((TT + 16, CI))                                  # LDS rows [t][8 ci]
        for tt in range(TT + 16):
            t = t0 + tt
            if t < T1:
                xs[tt] = bf16(a1[:, t] * sc + sh)
        for tb in range(0, TT, 16):
            if t0 + tb >= T2:
                break
            acc = np.zeros((64, 4))
            for s in range(3):
                bfr = np.stack([xs[tb + LN[l] + 4 * s + Q[l]] for l in range(64)])
                acc = mfma16(wf[s], bfr, acc)
            for l in range(64):
                t = t0 + tb + LN[l]
                if t < T2:
                    for e in range(4):
                        out[4 * Q[l] + e, t] = max(acc[l, e] + b[4 * Q[l] + e], 0.0)
    return out


def conv2_dgrad(dz, w, T1, TT=64):
    """dz [16][T2] -> d_a1 [8][T1]  -- conv2_dgrad_mfma_kernel, one node."""
    T2 = T1 - (KW - 1)
    out = np.zeros((CI, T1))
    wf = np.zeros((5, 64, 8))
    for s in range(5):
        for l in range(64):
            kk, c0 = 2 * s + (Q[l] >> 1), 8 * (Q[l] & 1)
            if LN[l] < CI:
                wf[s, l] = bf16(w[c0:c0 + 8, LN[l], kk])
    for t0 in range(0, T1, TT):
        zs = np.zeros((TT + 16, CO))                                  # LDS rows [t - (t0 - 9)][16 co]
        for tt in range(TT + 16):
            t = t0 - (KW - 1) + tt
            if 0 <= t < T2:
                zs[tt] = bf16(dz[:, t])
        for tb in range(0, TT, 16):
            if t0 + tb >= T1:
                break
            acc = np.zeros((64, 4))
            for s in range(5):
                bfr = np.stack([zs[tb + LN[l] + (KW - 1) - (2 * s + (Q[l] >> 1))][8 * (Q[l] & 1):8 * (Q[l] & 1) + 8] for l in range(64)])
                acc = mfma16(wf[s], bfr, acc)
            for l in range(64):
                t = t0 + tb + LN[l]
                if t < T1 and Q[l] < 2:
                    for e in range(4):
                        out[4 * Q[l] + e, t] = acc[l, e]
    return out


def conv2_wgrad(dz, a1, sc, sh, SC=64):
    """dz [16][T2], a1 [8][T1] -> (dw [16][8][10], db [16])  -- conv2_wgrad_mfma_kernel, one node, one wave's work serialised."""
    T1 = a1.shape[1]; T2 = T1 - (KW - 1)
    acc = [np.zeros((64, 4)) for _ in range(5)]
    db = np.zeros(CO)
    for tp in range(0, T2, SC):
        zs = np.zeros((CO, SC + 8)); x0 = np.zeros((CI, SC + 16)); x1 = np.zeros((CI, SC + 16))
        for tt in range(SC):
            if tp + tt < T2:
                zs[:, tt] = bf16(dz[:, tp + tt]); db += dz[:, tp + tt]
        for tt in range(SC + 10):
            tx = tp + tt
            hb = bf16(a1[:, tx] * sc + sh) if tx < T1 else np.zeros(CI)
            x0[:, tt] = hb
            if tt > 0:
                x1[:, tt - 1] = hb
        for tb0 in range(0, SC, 32):                                   # the (wave, u) pairs of the kernel
            a = np.stack([zs[LN[l], tb0 + 8 * Q[l]:tb0 + 8 * Q[l] + 8] for l in range(64)])
            for j in range(5):
                bfr = np.zeros((64, 8))
                for l in range(64):
                    kk, ci = 2 * j + (LN[l] >> 3), LN[l] & 7
                    row = x1[ci] if kk & 1 else x0[ci]
                    off = tb0 + 8 * Q[l] + (kk & ~1)
                    bfr[l] = row[off:off + 8]
                acc[j] = mfma16(a, bfr, acc[j])
    dw = np.zeros((CO, CI, KW))
    for j in range(5):
        for l in range(64):
            kk, ci = 2 * j + (LN[l] >> 3), LN[l] & 7
            for e in range(4):
                dw[4 * Q[l] + e, ci, kk] = acc[j][l, e]
    return dw, db


def conv1_wgrad(dz, x, SC=64):
    """dz [8][T1], x [T] -> (dw [8][10], db [8])  -- conv1_wgrad_mfma_kernel, one node."""
    T = x.shape[0]; T1 = T - (KW - 1)
    acc = np.zeros((64, 4)); db = np.zeros(8)
    for tp in range(0, T1, SC):
        zs = np.zeros((8, SC + 8)); x0 = np.zeros(SC + 16); x1 = np.zeros(SC + 16)
        for tt in range(SC):
            if tp + tt < T1:
                zs[:, tt] = bf16(dz[:, tp + tt]); db += dz[:, tp + tt]
        for tt in range(SC + 10):
            hb = bf16(x[tp + tt]) if tp + tt < T else 0.0
            x0[tt] = hb
            if tt > 0:
                x1[tt - 1] = hb
        for tb0 in range(0, SC, 32):
            a = np.zeros((64, 8)); bfr = np.zeros((64, 8))
            for l in range(64):
                tb = tb0 + 8 * Q[l]
                if LN[l] < 8:
                    a[l] = zs[LN[l] & 7, tb:tb + 8]
                if LN[l] < KW:
                    kk = LN[l]
                    row = x1 if kk & 1 else x0
                    bfr[l] = row[tb + (kk & ~1):tb + (kk & ~1) + 8]
            acc = mfma16(a, bfr, acc)
    dw = np.zeros((8, KW))
    for l in range(64):
        if Q[l] < 2 and LN[l] < KW:
            for e in range(4):
                dw[4 * Q[l] + e, LN[l]] = acc[l, e]
    return dw, db
