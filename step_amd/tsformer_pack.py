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
MAGIC = 0x54534631
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


def layer_bytes():
    o = 24 * 4 * FRAG + 72 * 2 * FRAG
    o += 4 * 2 * 16 * 4 + 4 * 32 * 4 + 3 * 2 * 48 * 4 + 12 * 2 * 16 * 4 + 3 * 2 * 48 * 4
    return o


def total_bytes(depth, P):
    return HDR + 2 * 48 * 12 * 4 + 3 * 2 * 48 * 4 + depth * layer_bytes() + P * 2 * 48 * 4


def pack_tsformer(sd, P, depth=4, prefix="", enc="encoder"):
    """sd: mapping name -> tensor (reference TSFormer state_dict keys, optionally prefixed).
    Returns a uint8 CPU tensor of ``total_bytes(depth, P)`` bytes."""
    g = lambda k: sd[prefix + k].detach().to(torch.float32).cpu().numpy()
    f32_parts, out = [], bytearray()

    def put_f32(a):
        out.extend(np.ascontiguousarray(a, dtype=np.float32).tobytes())

    def put_bf16(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16)
        out.extend(t.view(torch.int16).numpy().tobytes())

    hdr = np.zeros(HDR // 4, dtype=np.int32)
    hdr[0], hdr[1], hdr[2] = MAGIC, P, depth
    out.extend(hdr.tobytes())
    wpe = g("patch_embedding.input_embedding.weight")[:, 0, :, 0]        # [96, 12]
    put_f32(np.stack([wpe[_lane_vec96(np.arange(96))[h]] for h in (0, 1)]))   # [2,48,12]
    put_f32(_lane_vec96(g("patch_embedding.input_embedding.bias")))
    put_f32(_lane_vec96(g(enc + "_norm.weight")))
    put_f32(_lane_vec96(g(enc + "_norm.bias")))
    qscale = math.log2(math.e) / math.sqrt(HDIM)
    for l in range(depth):
        lp = f"{enc}.transformer_encoder.layers.{l}."
        win, bin_ = g(lp + "self_attn.in_proj_weight"), g(lp + "self_attn.in_proj_bias")
        wo, bo = g(lp + "self_attn.out_proj.weight"), g(lp + "self_attn.out_proj.bias")
        w1, b1 = g(lp + "linear1.weight"), g(lp + "linear1.bias")
        w2, b2 = g(lp + "linear2.weight"), g(lp + "linear2.bias")
        assert len(out) == HDR + 2 * 48 * 12 * 4 + 3 * 2 * 48 * 4 + l * layer_bytes()

        def head_w(which, hd, scale=1.0):
            w = np.zeros((32, D), dtype=np.float32)
            w[:HDIM] = win[which * D + hd * HDIM: which * D + (hd + 1) * HDIM] * scale
            return w
        for which, scale in ((0, qscale), (1, 1.0), (2, 1.0)):        # WQ, WK, WV
            for hd in range(HEADS):
                w = head_w(which, hd, scale)
                for ks in range(6):
                    put_bf16(_frag(w, 0, (ks // 2) * 32, ks % 2))
        wo_pad = np.zeros((D, HEADS, 32), dtype=np.float32)
        wo_pad[:, :, :HDIM] = wo.reshape(D, HEADS, HDIM)
        for hd in range(HEADS):                                        # WO
            for t in range(3):
                for s in range(2):
                    put_bf16(_frag(wo_pad[:, hd, :], t * 32, 0, s))
        for ch in range(12):                                           # W1
            for ks in range(6):
                put_bf16(_frag(w1, ch * 32, (ks // 2) * 32, ks % 2))
        for ch in range(12):                                           # W2
            for t in range(3):
                for s in range(2):
                    put_bf16(_frag(w2, t * 32, ch * 32, s))
        bq = np.zeros((HEADS, 32), dtype=np.float32)
        bq[:, :HDIM] = bin_[:D].reshape(HEADS, HDIM) * qscale
        put_f32(np.stack([_lane_vec32(bq[hd]) for hd in range(HEADS)]))          # [4,2,16]
        bv = np.zeros((HEADS, 32), dtype=np.float32)
        bv[:, :HDIM] = bin_[2 * D:].reshape(HEADS, HDIM)
        bv[:, HDIM] = 1.0                        # all-ones V row -> softmax denominator
        put_f32(bv)
        put_f32(_lane_vec96(bo))
        put_f32(_lane_vec96(g(lp + "norm1.weight")))
        put_f32(_lane_vec96(g(lp + "norm1.bias")))
        put_f32(np.stack([_lane_vec32(b1[ch * 32:(ch + 1) * 32]) for ch in range(12)]))   # [12,2,16]
        put_f32(_lane_vec96(b2))
        put_f32(_lane_vec96(g(lp + "norm2.weight")))
        put_f32(_lane_vec96(g(lp + "norm2.bias")))
    pos = g("positional_encoding.position_embedding")[:P]              # [P, 96]
    put_f32(np.stack([_lane_vec96(pos[p]) for p in range(P)]))         # [P,2,48]
    assert len(out) == total_bytes(depth, P), (len(out), total_bytes(depth, P))
    return torch.frombuffer(bytes(out), dtype=torch.uint8).clone()
