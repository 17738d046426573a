"""Pack a TSFormer ``state_dict`` into the operand-fragment buffer read by the fused encoder
kernel (``csrc/tsformer_encoder.hip``; byte layout in ``csrc/tsformer_layout.h``).

The kernel keeps activations transposed in MFMA accumulators and relies on the k-slot map
``F(s, h, j) = 16 s + 8 (j >> 2) + 4 h + (j & 3)`` of ``v_mfma_f32_32x32x16_bf16``: slot ``j``
of lane-half ``h`` in k-step ``s`` of a 32-wide block holds input feature ``F``.  Packing is a
host-side, once-per-checkpoint operation (the TSFormer is frozen in the forecasting stage,
reference ``step/step_arch/step.py:27-35``), done with plain torch CPU indexing.
"""
import math

import numpy as np
import torch

D, HEADS, HDIM, FFN, PATCH, FRAG = 96, 4, 24, 384, 12, 1024
MAGIC = 0x54534633
HDR = 64

_h = np.arange(64) // 32
_r = np.arange(64) % 32
_j = np.arange(8)


def _F(s):
    """[64 lanes, 8 slots] -> feature index inside a 32-wide block for k-step s."""
    return 16 * s + 8 * (_j[None, :] >> 2) + 4 * _h[:, None] + (_j[None, :] & 3)


def _rows16():
    """[2 halves, 16 regs] -> accumulator row inside a 32-row tile."""
    i = np.arange(16)
    return np.stack([(i & 3) + 8 * (i >> 2) + 4 * h for h in (0, 1)])


def _frag(w, row0, k0, s):
    """Operand fragment [64, 8] of a zero-padded matrix w: lane (h, r) slot j -> w[row0 + r, k0 + F(s,h,j)]."""
    return w[(row0 + _r)[:, None], k0 + _F(s)]


def _lane_vec96(v):
    """feature vector [96] -> [2, 48] in accumulator-register order (tile-major)."""
    rows = _rows16()                                   # [2,16]
    idx = np.stack([np.concatenate([t * 32 + rows[h] for t in range(3)]) for h in (0, 1)])
    return v[idx]


def _lane_vec32(v):
    return v[_rows16()]                                # [2,16]


BLOCK = 25 * FRAG
STAGES = 10
TAIL = 24 * FRAG
LAYER0 = (HDR + 2 * 48 * 12 * 4 + 3 * 2 * 48 * 4 + 1023) // 1024 * 1024


def layer_bytes():
    return STAGES * BLOCK


def total_bytes(depth, P):
    return LAYER0 + depth * layer_bytes() + P * 2 * 48 * 4


OPERAND_DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16}


def pack_tsformer(sd, P, depth=4, prefix="", enc="encoder", operand="bf16"):
    """sd: mapping name -> tensor (reference TSFormer state_dict keys, optionally prefixed).
    Returns a uint8 CPU tensor of ``total_bytes(depth, P)`` bytes (layout: csrc/tsformer_layout.h).
    ``operand``: 16-bit type of the MFMA operand fragments, "bf16" or "f16" (same layout; header word 3 records it and the
    kernel launch must be told the same, ``step_tsformer_encode(operand_f16=...)``)."""
    odt = OPERAND_DTYPES[operand]
    g = lambda k: sd[prefix + k].detach().to(torch.float32).cpu().numpy()
    out = bytearray()

    def f32_bytes(a):
        return np.ascontiguousarray(a, dtype=np.float32).tobytes()

    def bf16_bytes(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(odt)
        return t.view(torch.int16).numpy().tobytes()

    hdr = np.zeros(HDR // 4, dtype=np.int32)
    hdr[0], hdr[1], hdr[2], hdr[3] = MAGIC, P, depth, int(operand == "f16")
    out.extend(hdr.tobytes())
    wpe = g("patch_embedding.input_embedding.weight")[:, 0, :, 0]        # [96, 12]
    lane = np.arange(64)
    out.extend(f32_bytes(np.stack([[wpe[32 * t + lane % 32, 2 * s + lane // 32] for s in range(6)] for t in range(3)])))   # [3,6,64]
    out.extend(f32_bytes(_lane_vec96(g("patch_embedding.input_embedding.bias"))))
    out.extend(f32_bytes(_lane_vec96(g(enc + "_norm.weight"))))
    out.extend(f32_bytes(_lane_vec96(g(enc + "_norm.bias"))))
    out.extend(b"\0" * (LAYER0 - len(out)))
    qscale = math.log2(math.e) / math.sqrt(HDIM)
    for l in range(depth):
        lp = f"{enc}.transformer_encoder.layers.{l}."
        win, bin_ = g(lp + "self_attn.in_proj_weight"), g(lp + "self_attn.in_proj_bias")
        wo, bo = g(lp + "self_attn.out_proj.weight"), g(lp + "self_attn.out_proj.bias")
        w1, b1 = g(lp + "linear1.weight"), g(lp + "linear1.bias")
        w2, b2 = g(lp + "linear2.weight"), g(lp + "linear2.bias")
        assert len(out) == LAYER0 + l * layer_bytes()

        def head_w(which, hd, scale=1.0):
            w = np.zeros((32, D), dtype=np.float32)
            w[:HDIM] = win[which * D + hd * HDIM: which * D + (hd + 1) * HDIM] * scale
            return w
        wo_pad = np.zeros((D, HEADS, 32), dtype=np.float32)
        wo_pad[:, :, :HDIM] = wo.reshape(D, HEADS, HDIM)
        bq = np.zeros((HEADS, 32), dtype=np.float32)
        bq[:, :HDIM] = bin_[:D].reshape(HEADS, HDIM) * qscale
        bv = np.zeros((HEADS, 32), dtype=np.float32)
        bv[:, :HDIM] = bin_[2 * D:].reshape(HEADS, HDIM)
        bv[:, HDIM] = 1.0                        # all-ones V row -> softmax denominator
        for hd in range(HEADS):                                        # ---- head stage blocks
            blk = bytearray()
            for which, scale in ((0, qscale), (1, 1.0), (2, 1.0)):    # Wq, Wk, Wv
                w = head_w(which, hd, scale)
                for ks in range(6):
                    blk.extend(bf16_bytes(_frag(w, 0, (ks // 2) * 32, ks % 2)))
            for t in range(3):                                         # Wo
                for s in range(2):
                    blk.extend(bf16_bytes(_frag(wo_pad[:, hd, :], t * 32, 0, s)))
            tail = np.zeros(256, dtype=np.float32)
            tail[0:32] = _lane_vec32(bq[hd]).reshape(-1)
            tail[32:64] = bv[hd]
            if hd == 0:
                tail[64:160] = _lane_vec96(bo).reshape(-1)
            if hd == HEADS - 1:
                tail[64:160] = _lane_vec96(g(lp + "norm1.weight")).reshape(-1)
                tail[160:256] = _lane_vec96(g(lp + "norm1.bias")).reshape(-1)
            blk.extend(tail.tobytes())
            assert len(blk) == BLOCK
            out.extend(blk)
        for j in range(6):                                             # ---- ffn stage blocks
            blk = bytearray()
            for cc in range(2):
                ch = 2 * j + cc
                for ks in range(6):
                    blk.extend(bf16_bytes(_frag(w1, ch * 32, (ks // 2) * 32, ks % 2)))
                for t in range(3):
                    for s in range(2):
                        blk.extend(bf16_bytes(_frag(w2, t * 32, ch * 32, s)))
            tail = np.zeros(256, dtype=np.float32)
            for cc in range(2):
                ch = 2 * j + cc
                tail[cc * 32:(cc + 1) * 32] = _lane_vec32(b1[ch * 32:(ch + 1) * 32]).reshape(-1)
            if j == 0:
                tail[64:160] = _lane_vec96(b2).reshape(-1)
            if j == 5:
                tail[64:160] = _lane_vec96(g(lp + "norm2.weight")).reshape(-1)
                tail[160:256] = _lane_vec96(g(lp + "norm2.bias")).reshape(-1)
            blk.extend(tail.tobytes())
            assert len(blk) == BLOCK
            out.extend(blk)
    pos = g("positional_encoding.position_embedding")[:P] + g("patch_embedding.input_embedding.bias")[None, :]     # [P, 96], b_pe folded in
    out.extend(f32_bytes(np.stack([_lane_vec96(pos[p]) for p in range(P)])))         # [P,2,48]
    assert len(out) == total_bytes(depth, P), (len(out), total_bytes(depth, P))
    return torch.from_numpy(np.frombuffer(bytes(out), dtype=np.uint8).copy())
