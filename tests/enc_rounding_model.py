"""Operand-format model of the fused encoder (TEST INFRASTRUCTURE): the oracle's TSFormer forward in float64 with every tensor
that the kernel feeds to the matrix cores rounded to the kernel's operand format at the point where the kernel rounds it --
x (layer input and FFN input), the Q/K/V/O/FFN weights, q, k, o and the FFN hidden units in the 16-bit operand type (float16 or
bfloat16), the attention probabilities and V in bfloat16 -- and nothing else (accumulation, softmax, LayerNorm, residuals exact).
It is NOT the lane-level emulation (tests/emu_encoder.py, which replays the kernel's data flow): it answers "how far from the
fp32 reference is ANY implementation with these operand formats", which is the yardstick for inputs where 16-bit score operands
are ill-conditioned (sharply peaked attention: an absolute score error of |s| 2^-11 is a relative probability error of the same
size times ln 2).  ``sites`` restricts the rounding to a subset (for attributing the error to a site)."""
import math

import torch

SITES = ("x_attn", "Wq", "Wk", "Wv", "q", "k", "v", "P", "o", "Wo", "x_ffn", "W1", "ffn_hidden", "W2")
SCORE_PATH = ("x_attn", "Wq", "Wk", "q", "k")


def _ln(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + 1e-5) * w + b


def encode(series, sd, operand=torch.float16, pv=torch.bfloat16, drop=None, keep=1.0, sites=SITES, depth=4, pre=""):
    """series [S, L] -> hidden [S, P, 96] (float64).  sd: TSFormer state dict (keys without the 'tsformer.' prefix unless ``pre``);
    drop: None or the dense keep-masks in the oracle's format (tsformer_encode); operand / pv: torch dtypes or None (exact)."""
    sd = {k: v.double() for k, v in sd.items()}
    sites = set(sites)

    def rq(name, x, dt):
        return x if (dt is None or name not in sites) else x.float().to(dt).double()
    S, L = series.shape
    P = L // 12
    w = sd[pre + "patch_embedding.input_embedding.weight"][:, 0, :, 0]
    h = series.double().reshape(S, P, 12) @ w.T + sd[pre + "patch_embedding.input_embedding.bias"] + \
        sd[pre + "positional_encoding.position_embedding"][:P]
    if drop is not None:
        h = h * drop["pos"].double() / keep
    h = h * math.sqrt(96)
    sc = math.log2(math.e) / math.sqrt(24)
    for i in range(depth):
        pr = f"{pre}encoder.transformer_encoder.layers.{i}."
        dm = None if drop is None else {k: v.double() for k, v in drop["layers"][i].items()}
        hb = rq("x_attn", h, operand)
        Wi, bi = sd[pr + "self_attn.in_proj_weight"], sd[pr + "self_attn.in_proj_bias"]
        q = hb @ rq("Wq", Wi[:96] * sc, operand).T + bi[:96] * sc          # the kernel folds log2(e)/sqrt(dh) into Wq, bq
        k = hb @ rq("Wk", Wi[96:192], operand).T                          # (key bias cancels in the softmax)
        v = hb @ rq("Wv", Wi[192:], operand).T + bi[192:]
        q = rq("q", q, operand).reshape(S, P, 4, 24).transpose(1, 2)
        k = rq("k", k, operand).reshape(S, P, 4, 24).transpose(1, 2)
        v = rq("v", v, pv).reshape(S, P, 4, 24).transpose(1, 2)
        s = q @ k.transpose(-1, -2)
        p_ = torch.exp2(s - s.amax(-1, keepdim=True))
        pb = rq("P", p_, pv)
        if dm is not None:
            o = (pb * dm["attn"]) @ v / (p_.sum(-1, keepdim=True) * keep)
        else:
            o = pb @ v / pb.sum(-1, keepdim=True)
        o = rq("o", o.transpose(1, 2).reshape(S, P, 96), operand)
        o = o @ rq("Wo", sd[pr + "self_attn.out_proj.weight"], operand).T + sd[pr + "self_attn.out_proj.bias"]
        h = _ln(h + (o * dm["drop1"] / keep if dm is not None else o), sd[pr + "norm1.weight"], sd[pr + "norm1.bias"])
        f = rq("x_ffn", h, operand) @ rq("W1", sd[pr + "linear1.weight"], operand).T + sd[pr + "linear1.bias"]
        if dm is not None:
            f = f * dm["ffn"] / keep
        f = torch.relu(rq("ffn_hidden", f, operand)) @ rq("W2", sd[pr + "linear2.weight"], operand).T + sd[pr + "linear2.bias"]
        h = _ln(h + (f * dm["drop2"] / keep if dm is not None else f), sd[pr + "norm2.weight"], sd[pr + "norm2.bias"])
    return _ln(h, sd[pre + "encoder_norm.weight"], sd[pre + "encoder_norm.bias"])


def sharpened(sd, factor=3.0, bias_noise=0.1, seed=1):
    """tools/enc_ab_prepare.py's stress weights: every matrix x factor (scores x factor^2), noisy biases."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if v.ndim >= 2 and "position" not in k and "mask_token" not in k:
            out[k] = v * factor
        elif k.endswith("bias"):
            out[k] = v + bias_noise * torch.randn(v.shape, generator=g)
        else:
            out[k] = v
    return out
