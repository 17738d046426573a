"""TSFormer with the reference's module surface, computed by the fused HIP encoder.

Mirrors ``step/step_arch/tsformer/tsformer.py:21-191`` of the reference: same constructor
arguments, same ``forward`` keywords, and a ``state_dict`` whose keys and shapes are identical
(``encoder.transformer_encoder.layers.{i}.self_attn.in_proj_weight`` ...), so reference
checkpoints load unchanged.  The sub-modules below only *hold* parameters; arithmetic happens in
``libstep_hip`` (``step_tsformer_encode``).  There is no PyTorch fallback: CPU tensors raise.
"""
import math
import os

import torch
from torch import nn

from .. import _lib
from ..tsformer_pack import pack_tsformer


class _SelfAttnParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names (in_proj_weight, in_proj_bias, out_proj.*)."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _EncoderLayerParams(nn.Module):
    """Parameter holder with nn.TransformerEncoderLayer's names."""

    def __init__(self, d, ffn):
        super().__init__()
        self.self_attn = _SelfAttnParams(d)
        self.linear1 = nn.Linear(d, ffn)
        self.linear2 = nn.Linear(ffn, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class _EncoderStack(nn.Module):
    def __init__(self, d, ffn, depth):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayerParams(d, ffn) for _ in range(depth)])


class TransformerLayers(nn.Module):
    """Holder named like the reference's TransformerLayers (transformer_layers.py:6-11)."""

    def __init__(self, hidden_dim, nlayers, mlp_ratio, num_heads=4, dropout=0.1):
        super().__init__()
        self.d_model = hidden_dim
        self.transformer_encoder = _EncoderStack(hidden_dim, hidden_dim * mlp_ratio, nlayers)


class PatchEmbedding(nn.Module):
    def __init__(self, patch_size, in_channel, embed_dim):
        super().__init__()
        self.len_patch = patch_size
        self.input_embedding = nn.Conv2d(in_channel, embed_dim, kernel_size=(patch_size, 1), stride=(patch_size, 1))


class PositionalEncoding(nn.Module):
    def __init__(self, hidden_dim, dropout=0.1, max_len=1000):
        super().__init__()
        self.p = dropout
        self.position_embedding = nn.Parameter(torch.empty(max_len, hidden_dim))


class MaskGenerator(nn.Module):
    """Host-side mask draw, same procedure as the reference (tsformer/mask.py:15-28)."""

    def __init__(self, num_tokens, mask_ratio):
        super().__init__()
        self.num_tokens = num_tokens
        self.mask_ratio = mask_ratio
        self.sort = True

    def forward(self):
        import random
        mask = list(range(int(self.num_tokens)))
        random.shuffle(mask)
        mask_len = int(self.num_tokens * self.mask_ratio)
        self.masked_tokens = sorted(mask[:mask_len])
        self.unmasked_tokens = sorted(mask[mask_len:])
        return self.unmasked_tokens, self.masked_tokens


SQRT_D = 9.797958971132712


def _empty(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


MC_ATTENTION_MAX_TOKENS = 352        # step_pt_attention_{fwd,bwd}_bf16: the backward's LDS footprint (include/step_hip.h)
_BF16 = False        # operand precision of the pre-training GEMMs, set from TSFormer.matmul_precision by _PretrainFunction


_SPLITK_WS = {}


def _splitk_ws(dev):
    """Scratch for the partial tiles of the split-K weight-gradient GEMMs (StepGemm.splitk_ws): the staged path aims at 768
    workgroups of 128 x 128 tiles, i.e. at most 768 * 128 * 128 floats of partial results (50 MB), whatever the shape."""
    t = _SPLITK_WS.get(dev)
    if t is None:
        t = _SPLITK_WS[dev] = torch.empty(768 * 128 * 128, device=dev, dtype=torch.float32)
    return t


_FFN_WS = {}
PT_POOL_WORDS = 1 << 18        # keep-mask pool of the fused feed-forward blocks (2 MB: resident in every XCD's L2, like the forecasting encoder's; refilled
                               # every step; a 32-row tile reads 192 words of it at a hashed offset)


def _ffn_ws(dev, R):
    """scratch of step_pt_ffn_fused_bwd_weights / step_pt_proj_wgrad: the per-workgroup partial gradients (at most 256 x 74 112 floats = 76 MB)"""
    n = max(_lib.lib().step_pt_ffn_wgrad_ws_floats(int(R)), _lib.lib().step_pt_proj_wgrad_ws_floats(int(R)))
    t = _FFN_WS.get(dev)
    if t is None or t.numel() < n:
        t = _FFN_WS[dev] = torch.empty(n, device=dev, dtype=torch.float32)
    return t


def _linear_fwd(x, w, b, relu=False):
    """y[R,N] = x[R,K] @ w[N,K]^T + b   (step_gemm: exact-f32 matrix cores, or bf16 operands in the bf16 mode)."""
    R, K = x.shape
    N = w.shape[0]
    y = _empty(R, N, like=x)
    _lib.gemm(x, w, y, R, N, K, K, 1, 1, K, N, bias=b, relu=relu, compute_bf16=_BF16)
    return y


def _linear_bwd(dy, x, w, dw, db, dx=None, accumulate_dx=False):
    """dw[N,K] += dy^T x ; db[N] += colsum(dy) (as the all-ones column of the same GEMM) ; dx[R,K] (=|+=) dy @ w."""
    R, N = dy.shape
    K = x.shape[1]
    _lib.gemm(dy, x, dw, N, K, R, 1, N, K, 1, K, accumulate=2, splitk=-1, a_rowsum=db, compute_bf16=_BF16, splitk_ws=_splitk_ws(dy.device))
    if dx is not None:
        _lib.gemm(dy, w, dx, R, K, N, N, 1, K, 1, K, accumulate=1 if accumulate_dx else 0, compute_bf16=_BF16)
    return dx


class _PretrainFunction(torch.autograd.Function):
    """TSFormer pre-training forward + hand-written backward on libstep_hip (reference tsformer.py:71-136).
    Inputs: (model, series [S, L], unmasked idx (int32 cuda), masked idx, *parameters in model._pt_names order).
    Output: reconstruction of every token, [S, P, 12]."""

    @staticmethod
    def forward(ctx, model, series, um, mk, *params):
        global _BF16
        _BF16 = model.matmul_precision == "bf16"
        L = _lib
        st = L.stream()
        P_ = dict(zip(model._pt_names, params))
        S, Lh = series.shape
        P = Lh // 12
        Pu, Pm = um.numel(), mk.numel()
        p = model.dropout_p if model.training else 0.0
        seed = model._next_seed()
        saved = {"seed": seed, "p": p}
        fz = None
        if _BF16 and model.fused_ffn:
            # feed-forward blocks without a stored hidden layer (csrc/pretrain_fused.hip): operand fragments of this step's weights per layer,
            # and the step's keep-mask pool
            fz = {"packs": model._ffn_packs(series.device), "pool": None, "words": 0,
                  "proj": model._proj_packs(series.device) if model.fused_proj else None, "ln": model.fused_proj and model.fused_ln}
            if p > 0:
                fz["pool"], fz["words"] = model._pt_pool(series.device, p, seed)
        saved["fz"] = fz
        pos = P_["positional_encoding.position_embedding"]
        # patch embedding + positional embedding (+dropout), gather the unmasked tokens, scale by sqrt(d)
        x = _empty(S * Pu, 96, like=series)
        saved["embed_unmasked"] = model.fused_embed
        if model.fused_embed:
            # only the unmasked tokens are embedded (the others' embeddings are dead: the decoder puts mask_token + position there)
            L.call("step_pt_embed_unmasked_fwd", L.ptr(series), L.ptr(um), L.ptr(P_["patch_embedding.input_embedding.weight"]),
                   L.ptr(P_["patch_embedding.input_embedding.bias"]), L.ptr(pos), S, Lh, Pu, p, seed, 100, L.ptr(x), st)
        else:
            patches = series.view(S * P, 12)
            e0 = _linear_fwd(patches, P_["patch_embedding.input_embedding.weight"].view(96, 12), P_["patch_embedding.input_embedding.bias"])
            L.call("step_pt_add_rows", L.ptr(e0), S, P, L.ptr(pos), None, st)
            if p > 0:
                L.call("step_pt_dropout", L.ptr(e0), L.ptr(e0), e0.numel(), p, seed, 100, st)
            L.call("step_pt_token_gather", L.ptr(e0), S, P, L.ptr(um), Pu, SQRT_D, L.ptr(x), st)
            del e0
        layers = []
        for l in range(model.encoder_depth):
            x, sv = _PretrainFunction._layer_fwd(x, S, Pu, P_, f"encoder.transformer_encoder.layers.{l}.", p, seed, 16 * l, fz)
            layers.append(sv)
        y = _empty(S * Pu, 96, like=series)
        st_enc = _empty(S * Pu, 2, like=series)
        L.call("step_pt_add_layernorm_fwd", L.ptr(x), None, S * Pu, 0.0, 0, 0, L.ptr(P_["encoder_norm.weight"]), L.ptr(P_["encoder_norm.bias"]), None,
               L.ptr(y), L.ptr(st_enc), st)
        z = _linear_fwd(y, P_["enc_2_dec_emb.weight"], P_["enc_2_dec_emb.bias"])
        d0 = _empty(S * P, 96, like=series)
        L.call("step_pt_dec_input", L.ptr(z), L.ptr(P_["mask_token"]), L.ptr(pos), L.ptr(mk), S, P, Pu, p, seed, 101, L.ptr(d0), st)
        dec_layers = []
        d = d0
        for l in range(model.decoder_depth):
            d, sv = _PretrainFunction._layer_fwd(d, S, P, P_, f"decoder.transformer_encoder.layers.{l}.", p, seed, 16 * (8 + l), fz)
            dec_layers.append(sv)
        d2 = _empty(S * P, 96, like=series)
        st_dec = _empty(S * P, 2, like=series)
        L.call("step_pt_add_layernorm_fwd", L.ptr(d), None, S * P, 0.0, 0, 0, L.ptr(P_["decoder_norm.weight"]), L.ptr(P_["decoder_norm.bias"]), None,
               L.ptr(d2), L.ptr(st_dec), st)
        r = _linear_fwd(d2, P_["output_layer.weight"], P_["output_layer.bias"])
        saved.update(dict(series=series, um=um, mk=mk, layers=layers, dec_layers=dec_layers, x_enc_out=x, st_enc=st_enc, y=y,
                          d_out=d, st_dec=st_dec, d2=d2, dims=(S, P, Pu, Pm)))
        ctx.saved = saved
        ctx.model = model
        return r.view(S, P, 12)

    @staticmethod
    def _layer_fwd(x, S, T, P_, pre, p, seed, site, fz=None):
        L = _lib
        st = L.stream()
        R = S * T
        stats = _empty(S * 4 * T, 2, like=x)
        kb = None
        mc = _BF16 and T <= MC_ATTENTION_MAX_TOKENS
        if mc:
            # matrix-core attention with bf16 activations: qkv leaves its GEMM as bf16 (the type its only readers -- the attention
            # kernels -- would round it to), the attention output likewise (read by the out-projection and its weight gradient as a
            # matrix-core operand); the keep decisions of the probability dropout are handed to the backward as bit masks
            qkv = torch.empty(R, 288, device=x.device, dtype=torch.bfloat16)
            pj = fz["proj"][pre] if fz is not None and fz["proj"] is not None else None
            if pj is not None:
                # this step's weights of the layer as operand fragments: the four projections and the feed-forward block, one launch
                L.call("step_pt_layer_pack", L.ptr(P_[pre + "self_attn.in_proj_weight"]), L.ptr(P_[pre + "self_attn.in_proj_bias"]),
                       L.ptr(P_[pre + "self_attn.out_proj.weight"]), L.ptr(P_[pre + "self_attn.out_proj.bias"]), L.ptr(P_[pre + "linear1.weight"]),
                       L.ptr(P_[pre + "linear1.bias"]), L.ptr(P_[pre + "linear2.weight"]), L.ptr(P_[pre + "linear2.bias"]), L.ptr(fz["packs"][pre]),
                       L.ptr(pj[0]), L.ptr(pj[1]), L.ptr(pj[2]), L.ptr(pj[3]), st)
                L.call("step_pt_rows_linear", L.ptr(x), 0, R, L.ptr(pj[0]), 1, 3, L.ptr(qkv), 1, 0, st)
            else:
                L.call("step_pt_linear_bf16out", L.ptr(x), L.ptr(P_[pre + "self_attn.in_proj_weight"]), 1, 96, L.ptr(P_[pre + "self_attn.in_proj_bias"]),
                       R, 288, 96, L.ptr(qkv), st)
            a = torch.empty(R, 96, device=x.device, dtype=torch.bfloat16)
            kb = torch.empty(S * 4 * T * ((T + 31) // 32), dtype=torch.int32, device=x.device) if p > 0 else None
            # (with the fused path's pool the keep words come from it: no Philox in the kernel; the backward reads `kb` either way)
            pool, words = (fz["pool"], fz["words"]) if fz is not None else (None, 0)
            L.call("step_pt_attention_fwd_bf16", L.ptr(qkv), S, T, p, seed, site, L.ptr(a), L.ptr(stats), L.ptr(kb), L.ptr(pool), words, st)
        else:
            qkv = _linear_fwd(x, P_[pre + "self_attn.in_proj_weight"], P_[pre + "self_attn.in_proj_bias"])
            a = _empty(R, 96, like=x)
            L.call("step_pt_attention_fwd", L.ptr(qkv), S, T, p, seed, site, L.ptr(a), L.ptr(stats), st)
        h1pre, h1, st1 = _empty(R, 96, like=x), _empty(R, 96, like=x), _empty(R, 2, like=x)
        fuse_ln = mc and pj is not None and fz["ln"]
        if fuse_ln:
            # out-projection with the residual add, dropout and LayerNorm 1 as its output stage
            L.call("step_pt_rows_linear_ln", L.ptr(a), R, L.ptr(pj[1]), L.ptr(x), p, seed, site + 1, L.ptr(P_[pre + "norm1.weight"]),
                   L.ptr(P_[pre + "norm1.bias"]), L.ptr(h1pre), L.ptr(h1), L.ptr(st1), st)
        else:
            if mc and pj is not None:
                o = _empty(R, 96, like=x)
                L.call("step_pt_rows_linear", L.ptr(a), 1, R, L.ptr(pj[1]), 1, 1, L.ptr(o), 0, 0, st)
            else:
                o = _linear_fwd(a, P_[pre + "self_attn.out_proj.weight"], P_[pre + "self_attn.out_proj.bias"])
            # residual add (+ dropout of the branch) and LayerNorm in one pass
            L.call("step_pt_add_layernorm_fwd", L.ptr(x), L.ptr(o), R, p, seed, site + 1, L.ptr(P_[pre + "norm1.weight"]), L.ptr(P_[pre + "norm1.bias"]),
                   L.ptr(h1pre), L.ptr(h1), L.ptr(st1), st)
        f2 = None
        if fz is not None:
            pk = fz["packs"][pre]
            if not (mc and pj is not None):          # (otherwise written by step_pt_layer_pack above)
                L.call("step_pt_ffn_pack", L.ptr(P_[pre + "linear1.weight"]), L.ptr(P_[pre + "linear1.bias"]), L.ptr(P_[pre + "linear2.weight"]),
                       L.ptr(P_[pre + "linear2.bias"]), L.ptr(pk), st)
            f1 = f1d = None
            if not fuse_ln:
                f2 = _empty(R, 96, like=x)
                L.call("step_pt_ffn_fused_fwd", L.ptr(h1), R, L.ptr(pk), p, L.ptr(fz["pool"]), fz["words"], seed, site + 2, L.ptr(f2), st)
        elif _BF16:
            # ReLU and dropout in the epilogue of the first linear layer, the hidden layer stored once, as bf16 (the f32 path writes
            # relu(.) and its dropped copy: 2 x 1.3 GB per decoder layer at config C3); same Philox stream as step_pt_dropout
            f1 = None
            f1d = torch.empty(R, 384, device=x.device, dtype=torch.bfloat16)
            L.call("step_pt_ffn_hidden_fwd", L.ptr(h1), L.ptr(P_[pre + "linear1.weight"]), L.ptr(P_[pre + "linear1.bias"]), R, p, seed,
                   site + 2, L.ptr(f1d), st)
        else:
            f1 = _linear_fwd(h1, P_[pre + "linear1.weight"], P_[pre + "linear1.bias"], relu=True)
            f1d = f1
            if p > 0:
                f1d = _empty(R, 384, like=x)
                L.call("step_pt_dropout", L.ptr(f1), L.ptr(f1d), f1.numel(), p, seed, site + 2, st)
        h2pre, h2, st2 = _empty(R, 96, like=x), _empty(R, 96, like=x), _empty(R, 2, like=x)
        if fuse_ln:
            # feed-forward block with the residual add, dropout and LayerNorm 2 as its output stage
            L.call("step_pt_ffn_fused_fwd_ln", L.ptr(h1), R, L.ptr(fz["packs"][pre]), p, L.ptr(fz["pool"]), fz["words"], seed, site + 2, site + 3,
                   L.ptr(P_[pre + "norm2.weight"]), L.ptr(P_[pre + "norm2.bias"]), L.ptr(h2pre), L.ptr(h2), L.ptr(st2), st)
        else:
            if f2 is None:
                f2 = _linear_fwd(f1d, P_[pre + "linear2.weight"], P_[pre + "linear2.bias"])
            L.call("step_pt_add_layernorm_fwd", L.ptr(h1), L.ptr(f2), R, p, seed, site + 3, L.ptr(P_[pre + "norm2.weight"]), L.ptr(P_[pre + "norm2.bias"]),
                   L.ptr(h2pre), L.ptr(h2), L.ptr(st2), st)
        return h2, dict(x=x, qkv=qkv, a=a, stats=stats, keepbits=kb, h1pre=h1pre, st1=st1, h1=h1, f1=f1, f1d=f1d, h2pre=h2pre, st2=st2, pre=pre,
                        site=site, T=T, fz=fz)

    @staticmethod
    def _layer_bwd(dh2, sv, S, P_, G, p, seed):
        """dh2: gradient w.r.t. the layer output [R,96] (consumed); returns gradient w.r.t. the layer input."""
        L = _lib
        st = L.stream()
        pre, site, T = sv["pre"], sv["site"], sv["T"]
        R = S * T
        # LayerNorm backward and the dropout of the gradient that continues into the branch, one pass
        dh2pre = _empty(R, 96, like=dh2)
        df2 = _empty(R, 96, like=dh2) if p > 0 else None
        L.call("step_pt_layernorm_bwd_dropout", L.ptr(dh2), L.ptr(sv["h2pre"]), R, L.ptr(P_[pre + "norm2.weight"]), L.ptr(sv["st2"]), L.ptr(dh2pre),
               L.ptr(df2), p, seed, site + 3, L.ptr(G[pre + "norm2.weight"]), L.ptr(G[pre + "norm2.bias"]),
               L.ptr(G[pre + "linear2.bias"]) if _BF16 else None, st)          # (bf16 mode: db2 = colsum(df2) rides along)
        if df2 is None:
            df2 = dh2pre
        dh1 = dh2pre if p > 0 else dh2pre.clone()          # residual branch of H2pre = H1 + dropout(F2)
        if sv["fz"] is not None:
            # fused feed-forward block: the hidden layer and its gradient are recomputed inside the kernels (db2 = colsum(df2) came out of the
            # LayerNorm backward above)
            fz = sv["fz"]
            pk = fz["packs"][pre]
            L.call("step_pt_ffn_fused_bwd_data", L.ptr(df2), L.ptr(sv["h1"]), R, L.ptr(pk), p, L.ptr(fz["pool"]), fz["words"], seed, site + 2,
                   L.ptr(dh1), st)
            L.call("step_pt_ffn_fused_bwd_weights", L.ptr(df2), L.ptr(sv["h1"]), R, L.ptr(pk), L.ptr(P_[pre + "linear1.bias"]), p, L.ptr(fz["pool"]),
                   fz["words"], seed, site + 2, L.ptr(_ffn_ws(dh2.device, R)), L.ptr(G[pre + "linear1.weight"]), L.ptr(G[pre + "linear1.bias"]),
                   L.ptr(G[pre + "linear2.weight"]), st)
        elif _BF16:
            # bf16 hidden layer (see _layer_fwd): its gradient is masked in the GEMM epilogue and stored as bf16 as well
            hid = sv["f1d"]
            w1, w2 = P_[pre + "linear1.weight"], P_[pre + "linear2.weight"]
            # dW2[o, j] += sum_r df2[r, o] hid[r, j]  (db2 = colsum(df2) came out of the LayerNorm backward above)
            _lib.gemm(df2, hid, G[pre + "linear2.weight"], 96, 384, R, 1, 96, 384, 1, 384, accumulate=2, splitk=-1, compute_bf16=True, splitk_ws=_splitk_ws(dh2.device))
            dhid = torch.empty(R, 384, device=dh2.device, dtype=torch.bfloat16)
            L.call("step_pt_ffn_hidden_bwd", L.ptr(df2), L.ptr(w2), L.ptr(hid), R, p, L.ptr(dhid), st)
            # dW1[j, i] += sum_r dhid[r, j] h1[r, i]: computed as its transpose (A = h1 with i contiguous, B = dhid with j contiguous)
            _lib.gemm(sv["h1"], dhid, G[pre + "linear1.weight"], 96, 384, R, 1, 96, 384, 1, 1, scn=96, accumulate=2, splitk=-1, compute_bf16=True, splitk_ws=_splitk_ws(dh2.device))
            L.call("step_pt_colsum_bf16", L.ptr(dhid), R, 384, L.ptr(G[pre + "linear1.bias"]), st)
            # dh1[r, i] += sum_j dhid[r, j] w1[j, i]
            _lib.gemm(dhid, w1, dh1, R, 96, 384, 384, 1, 96, 1, 96, accumulate=1, compute_bf16=True)
        else:
            df1d = _empty(R, 384, like=dh2)
            _linear_bwd(df2, sv["f1d"], P_[pre + "linear2.weight"], G[pre + "linear2.weight"], G[pre + "linear2.bias"], df1d)
            L.call("step_pt_dropout_relu_mask", L.ptr(df1d), L.ptr(sv["f1"]), R * 384, p, seed, site + 2, st)      # dropout and ReLU backward in one pass
            _linear_bwd(df1d, sv["h1"], P_[pre + "linear1.weight"], G[pre + "linear1.weight"], G[pre + "linear1.bias"], dh1, accumulate_dx=True)
        dh1pre = _empty(R, 96, like=dh2)
        do = _empty(R, 96, like=dh2) if p > 0 else None
        L.call("step_pt_layernorm_bwd_dropout", L.ptr(dh1), L.ptr(sv["h1pre"]), R, L.ptr(P_[pre + "norm1.weight"]), L.ptr(sv["st1"]), L.ptr(dh1pre),
               L.ptr(do), p, seed, site + 1, L.ptr(G[pre + "norm1.weight"]), L.ptr(G[pre + "norm1.bias"]),
               L.ptr(G[pre + "self_attn.out_proj.bias"]) if sv["qkv"].dtype == torch.bfloat16 else None, st)       # (dbo = colsum(do))
        if do is None:
            do = dh1pre
        dx = dh1pre if p > 0 else dh1pre.clone()            # residual branch of H1pre = X + dropout(O)
        if sv["qkv"].dtype == torch.bfloat16:
            # bf16 activations (see _layer_fwd): the gradients that are only read as matrix-core operands are stored as bf16 too
            wo, wi = P_[pre + "self_attn.out_proj.weight"], P_[pre + "self_attn.in_proj_weight"]
            pj = sv["fz"]["proj"][pre] if sv["fz"] is not None and sv["fz"]["proj"] is not None else None
            if pj is None:
                # dWo[j, i] += sum_r do[r, j] a[r, i]  (dbo = colsum(do) came out of the LayerNorm backward above)
                _lib.gemm(do, sv["a"], G[pre + "self_attn.out_proj.weight"], 96, 96, R, 1, 96, 96, 1, 96, accumulate=2, splitk=-1, compute_bf16=True, splitk_ws=_splitk_ws(dh2.device))
            da = torch.empty(R, 96, device=dh2.device, dtype=torch.bfloat16)
            if pj is not None:
                L.call("step_pt_rows_linear", L.ptr(do), 0, R, L.ptr(pj[2]), 1, 1, L.ptr(da), 1, 0, st)                 # da = do @ Wo
            else:
                L.call("step_pt_linear_bf16out", L.ptr(do), L.ptr(wo), 96, 1, None, R, 96, 96, L.ptr(da), st)          # da = do @ Wo
            dqkv = torch.empty(R, 288, device=dh2.device, dtype=torch.bfloat16)
            L.call("step_pt_attention_bwd_bf16", L.ptr(sv["qkv"]), L.ptr(sv["a"]), L.ptr(da), L.ptr(sv["stats"]), S, T, p, seed, site, L.ptr(dqkv),
                   L.ptr(sv["keepbits"]), st)
            if pj is not None:
                # dWi += dqkv^T x, dbi += colsum(dqkv), dWo += do^T a: one pass over the four row tensors
                L.call("step_pt_proj_wgrad", L.ptr(sv["x"]), L.ptr(dqkv), L.ptr(do), L.ptr(sv["a"]), R, L.ptr(_ffn_ws(dh2.device, R)),
                       L.ptr(G[pre + "self_attn.in_proj_weight"]), L.ptr(G[pre + "self_attn.in_proj_bias"]), L.ptr(G[pre + "self_attn.out_proj.weight"]), st)
            else:
                # dWi[j, i] += sum_r dqkv[r, j] x[r, i], computed as its transpose (A = x with i contiguous, B = dqkv with j contiguous)
                _lib.gemm(sv["x"], dqkv, G[pre + "self_attn.in_proj_weight"], 96, 288, R, 1, 96, 288, 1, 1, scn=96, accumulate=2, splitk=-1, compute_bf16=True, splitk_ws=_splitk_ws(dh2.device))
                L.call("step_pt_colsum_bf16", L.ptr(dqkv), R, 288, L.ptr(G[pre + "self_attn.in_proj_bias"]), st)
            if pj is not None:
                L.call("step_pt_rows_linear", L.ptr(dqkv), 1, R, L.ptr(pj[3]), 3, 1, L.ptr(dx), 0, 1, st)               # dx += dqkv @ Wi
            else:
                _lib.gemm(dqkv, wi, dx, R, 96, 288, 288, 1, 96, 1, 96, accumulate=1, compute_bf16=True)                   # dx += dqkv @ Wi
            return dx
        da = _empty(R, 96, like=dh2)
        _linear_bwd(do, sv["a"], P_[pre + "self_attn.out_proj.weight"], G[pre + "self_attn.out_proj.weight"],
                    G[pre + "self_attn.out_proj.bias"], da)
        dqkv = _empty(R, 288, like=dh2)
        L.call("step_pt_attention_bwd", L.ptr(sv["qkv"]), L.ptr(sv["a"]), L.ptr(da), L.ptr(sv["stats"]), S, T, p, seed, site, L.ptr(dqkv), st)
        _linear_bwd(dqkv, sv["x"], P_[pre + "self_attn.in_proj_weight"], G[pre + "self_attn.in_proj_weight"],
                    G[pre + "self_attn.in_proj_bias"], dx, accumulate_dx=True)
        return dx

    @staticmethod
    def backward(ctx, dr):
        global _BF16
        L = _lib
        st = L.stream()
        model, sv = ctx.model, ctx.saved
        _BF16 = model.matmul_precision == "bf16"
        S, P, Pu, Pm = sv["dims"]
        p, seed = sv["p"], sv["seed"]
        fz = sv.get("fz")
        if fz is not None and fz["pool"] is not None and model._pt_pool_tag != (int(seed), float(p)):
            # another training-mode forward has refilled the model's keep-mask pool since this one (fwd-fwd-bwd-bwd, a no_grad pass in
            # between, two views): the feed-forward backward recomputes its hidden-layer masks from the pool, so the pool of THIS
            # forward is rebuilt first -- the fill is a pure function of (seed, p)
            fz["pool"], fz["words"] = model._pt_pool(dr.device, p, seed)
        P_ = {n: prm for n, prm in zip(model._pt_names, model._pt_params())}
        flat, G = model._pt_grad_buffers()
        dev = dr.device
        dr = dr.contiguous().float().view(S * P, 12)
        pos = P_["positional_encoding.position_embedding"]
        # output layer, decoder norm, decoder layer(s)
        dd2 = _empty(S * P, 96, like=dr)
        _linear_bwd(dr, sv["d2"], P_["output_layer.weight"], G["output_layer.weight"], G["output_layer.bias"], dd2)
        dd = _empty(S * P, 96, like=dr)
        L.call("step_pt_layernorm_bwd_dropout", L.ptr(dd2), L.ptr(sv["d_out"]), S * P, L.ptr(P_["decoder_norm.weight"]), L.ptr(sv["st_dec"]), L.ptr(dd),
               None, 0.0, 0, 0, L.ptr(G["decoder_norm.weight"]), L.ptr(G["decoder_norm.bias"]), None, st)
        for lsv in reversed(sv["dec_layers"]):
            dd = _PretrainFunction._layer_bwd(dd, lsv, S, P_, G, p, seed)
        # decoder input: split into d z and the mask-token / positional part
        dz = _empty(S * Pu, 96, like=dr)
        if model.fused_embed:
            # d z and the positional / mask-token gradients of the masked positions in one pass over d d (no [S, Pm, 96] scratch)
            L.call("step_pt_dec_input_bwd_sums", L.ptr(dd), S, P, Pu, p, seed, 101, L.ptr(sv["mk"]), L.ptr(dz),
                   L.ptr(G["positional_encoding.position_embedding"]), L.ptr(G["mask_token"]), st)
        else:
            dm = _empty(S * Pm, 96, like=dr)
            L.call("step_pt_dec_input_bwd", L.ptr(dd), S, P, Pu, p, seed, 101, L.ptr(dz), L.ptr(dm), st)
            L.call("step_pt_sum_over_seq", L.ptr(dm), S, Pm, 0, Pm, L.ptr(sv["mk"]), L.ptr(G["positional_encoding.position_embedding"]), st)
            L.call("step_colsum", L.ptr(dm), S * Pm, 96, 96, L.ptr(G["mask_token"]), st)
        # enc_2_dec_emb, encoder norm, encoder layers
        dy = _empty(S * Pu, 96, like=dr)
        _linear_bwd(dz, sv["y"], P_["enc_2_dec_emb.weight"], G["enc_2_dec_emb.weight"], G["enc_2_dec_emb.bias"], dy)
        dx = _empty(S * Pu, 96, like=dr)
        L.call("step_pt_layernorm_bwd_dropout", L.ptr(dy), L.ptr(sv["x_enc_out"]), S * Pu, L.ptr(P_["encoder_norm.weight"]), L.ptr(sv["st_enc"]), L.ptr(dx),
               None, 0.0, 0, 0, L.ptr(G["encoder_norm.weight"]), L.ptr(G["encoder_norm.bias"]), None, st)
        for lsv in reversed(sv["layers"]):
            dx = _PretrainFunction._layer_bwd(dx, lsv, S, P_, G, p, seed)
        if sv["embed_unmasked"]:
            L.call("step_pt_embed_unmasked_bwd", L.ptr(dx), L.ptr(sv["series"]), L.ptr(sv["um"]), S, P * 12, Pu, p, seed, 100,
                   L.ptr(G["positional_encoding.position_embedding"]), L.ptr(G["patch_embedding.input_embedding.weight"]),
                   L.ptr(G["patch_embedding.input_embedding.bias"]), st)
            ctx.saved = None
            model._backward_count += 1
            return (None, None, None, None) + tuple(G[n] for n in model._pt_names)
        # scatter back to all token positions (zeros at masked ones), dropout, positional and patch embedding
        de = torch.zeros(S * P, 96, device=dev)
        L.call("step_pt_token_scatter", L.ptr(dx), S, P, L.ptr(sv["um"]), Pu, SQRT_D, L.ptr(de), st)
        if p > 0:
            L.call("step_pt_dropout", L.ptr(de), L.ptr(de), de.numel(), p, seed, 100, st)
        L.call("step_pt_sum_over_seq", L.ptr(de), S, P, 0, P, None, L.ptr(G["positional_encoding.position_embedding"]), st)
        patches = sv["series"].view(S * P, 12)
        _linear_bwd(de, patches, None, G["patch_embedding.input_embedding.weight"].view(96, 12), G["patch_embedding.input_embedding.bias"])
        ctx.saved = None
        model._backward_count += 1
        return (None, None, None, None) + tuple(G[n] for n in model._pt_names)


class TSFormer(nn.Module):
    """Drop-in for the reference TSFormer (forecasting mode: fused encoder kernel; pre-train mode: native forward + backward)."""

    def __init__(self, patch_size, in_channel, embed_dim, num_heads, mlp_ratio, dropout, num_token, mask_ratio,
                 encoder_depth, decoder_depth, mode="pre-train"):
        super().__init__()
        assert mode in ["pre-train", "forecasting"], "Error mode."
        if (patch_size, in_channel, embed_dim, num_heads, mlp_ratio) != (12, 1, 96, 4, 4):
            raise NotImplementedError("the HIP encoder is specialised to patch 12, 1 channel, d=96, 4 heads, mlp x4 "
                                      "(every STEP/TSFormer config of the reference)")
        self.patch_size, self.in_channel, self.embed_dim, self.num_heads = patch_size, in_channel, embed_dim, num_heads
        self.num_token, self.mask_ratio, self.encoder_depth, self.mode, self.mlp_ratio = \
            num_token, mask_ratio, encoder_depth, mode, mlp_ratio
        self.dropout_p = float(dropout)
        self.selected_feature = 0
        # pre-training GEMMs: "f32" exact (parity default) or "bf16" operands with f32 accumulation; the fused forecasting encoder
        # always runs 16-bit operands (encoder_operand below)
        self.matmul_precision = "f32"
        self.encoder_norm = nn.LayerNorm(embed_dim)
        self.decoder_norm = nn.LayerNorm(embed_dim)
        self.patch_embedding = PatchEmbedding(patch_size, in_channel, embed_dim)
        self.positional_encoding = PositionalEncoding(embed_dim, dropout=dropout)
        self.mask = MaskGenerator(num_token, mask_ratio)
        self.decoder_depth = decoder_depth
        self.encoder = TransformerLayers(embed_dim, encoder_depth, mlp_ratio, num_heads, dropout)
        self.enc_2_dec_emb = nn.Linear(embed_dim, embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, 1, embed_dim))
        self.decoder = TransformerLayers(embed_dim, decoder_depth, mlp_ratio, num_heads, dropout)
        self.output_layer = nn.Linear(embed_dim, patch_size)
        nn.init.uniform_(self.positional_encoding.position_embedding, -.02, .02)
        nn.init.trunc_normal_(self.mask_token, std=.02)
        self._pt_names = [n for n, _ in self.named_parameters()]
        self._seed_ctr2 = 0
        # bf16 mode of the pre-training step: feed-forward blocks through csrc/pretrain_fused.hip (no stored hidden layer; the backward
        # recomputes it).  STEP_PT_FUSED_FFN=0 keeps the layer-by-layer kernels (A/B measurements)
        self.fused_ffn = os.environ.get("STEP_PT_FUSED_FFN", "1") != "0"
        self._ffn_pack_bufs = None
        self._flat_param = None             # flatten_parameters(): all parameters as views into one buffer (FusedAdamClip)
        self._flat_grad = None
        self._backward_count = 0            # native backwards since zero_grad() (FusedAdamClip consumes exactly one)
        self.fused_embed = os.environ.get("STEP_PT_FUSED_EMBED", "1") != "0"     # patch + positional embedding of the unmasked tokens only (both precisions)
        self.fused_proj = os.environ.get("STEP_PT_FUSED_PROJ", "1") != "0"       # qkv / out-projection and their data gradients as row kernels
        self.fused_ln = os.environ.get("STEP_PT_FUSED_LN", "1") != "0"           # residual add + dropout + LayerNorm as the output stage of the forward row kernels
        self._proj_pack_bufs = None
        self._pt_pool_buf = None
        self._pt_pool_tag = None
        self._packed = None
        self._packed_key = None
        self._plist = None
        # 16-bit operand type of the fused forecasting encoder: "f16" (default) or "bf16".  Same MFMA rate and kernel time; on
        # MI355X float16 fragments bring the hidden-state error vs the fp32 reference from 1.2-1.9e-2 (bf16) down to 2.2e-3 and
        # the prediction error from 7.5e-3 to 8e-4 (profiles/r01_x_encoder_f16_lcg_ab.log, tests/test_gpu_kernels.py).  float16
        # overflows at 65 504: operands on this path stay below ~200 (tools/encoder_precision_study.py); the attention
        # probabilities and V, which need exponent range rather than mantissa, are bfloat16 in both modes.
        self.encoder_operand = "f16"
        # Range guard of the float16 path (the reference computes in fp32 and has no 65 504 limit, transformer_layers.py:13-20):
        #  * pack time: a weight (or LayerNorm / bias value) that float16 cannot hold makes `packed_weights` pack bfloat16 fragments;
        #  * run time: the kernel raises a device flag when a hidden state it writes is not finite (STEP_ENC_RANGE_FLAG, free: it rides on
        #    the squared norms of the epilogue).  The first `range_check_launches` launches after every (re)pack are checked at once (one
        #    stream synchronize each) and RE-RUN on bfloat16 fragments when it is up; later launches are polled without a synchronize every
        #    `range_poll_every` launches: a raised flag then switches all following launches to bfloat16 and warns (the affected batches
        #    already went on as NaN, which the loss shows).  `range_fallbacks` counts the switches, `encoder_operand_in_use` names the type.
        self.range_guard = os.environ.get("STEP_ENC_RANGE_GUARD", "1") != "0"
        self.range_check_launches = 2
        self.range_poll_every = 32
        self.range_fallbacks = 0
        self._range_forced_bf16 = False     # the guard's decision; cleared when the weights change (a new pack is judged afresh)
        self._range_checked = 0
        self._range_status = None           # uint32 [65 + 15] device: 64 slow-path counters (bench / tests read them) + the flag word
        self._range_poll = None             # (pinned host int32 [1], event) of the poll in flight
        self._range_launches = 0
        # 0 / None: one workgroup per sequence (the whole chip, 2456 workgroups at PEMS04).  n > 0: a persistent launch of n workgroups --
        # a workgroup fills a compute unit, so the encoder takes n of the 256 units and leaves the others to whatever runs on the other
        # streams.  That is what makes STEP.prefetch pay: the frozen branch of the next batch on 160 units for 3.1 ms next to this batch's
        # 3.7 ms chain of small kernels, instead of 2.1 ms + 2.1 ms one after the other (profiles/r05_k_persist_prefetch.log).
        self.encoder_workgroups = 0
        self._seed_counter = 0
        # training-mode dropout: pool of Bernoulli(1 - p) keep bits the encoder kernel reads its lane masks from, refilled from
        # the step's seed before every launch (step_dropout_pool_fill).  2^18 words = 2 MB stay resident in every XCD's L2 (with
        # 2^20 words the launch reads 0.94 GB more through L2 misses and takes 6 % longer, profiles/r03_a_*); every (sequence,
        # layer) reads its 10 912 words (PEMS04) at a hashed WORD offset, so two of the 12 280 chunks of a launch coincide with
        # probability 2^-18 per pair (tests/test_encoder_dropout_pool.py counts them)
        self.dropout_pool_words = 1 << 18
        self._dyn = None                    # device StepDynState of a replayed (graph-captured) step: the pool's Philox key moves with it (step_amd/graphed.py)
        self._drop_pool = None
        self._pool_override = None          # tests: int64 cuda tensor of keep-mask words used instead of the Philox fill
        self.encoder_debug_flags = 0        # tests: _lib.ENC_ALWAYS_RESHIFT
        self._events = None          # bench.py: list collecting (start, end) events around the encoder launch
        self.fallback_counter = None # bench.py / tests: int32 cuda tensor [65] whose first 64 words the kernel raises by its slow-path softmax units (word 64: range flag)

    # ------------------------------------------------------------------ packed operand cache
    def _apply(self, fn, recurse=True):
        self._plist = None
        out = super()._apply(fn, recurse)
        flat = self._flat_param
        if flat is not None:
            # .cuda() / .to(device) / .double() re-home p.data: the flat buffer (and an optimizer built on it) would go on updating
            # storage the forward no longer reads.  Unchanged storage (a no-op .to()) keeps the buffer.
            lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
            if not all(lo <= q.data_ptr() < hi and q.dtype == flat.dtype for q in self._pt_params()):
                self._flat_param = None
                self._flat_grad = None
        return out

    def _pack_key(self, P):
        ps = self._plist                      # the Parameter objects of the tree, listed once (the walk costs 0.1 ms per step)
        if ps is None:
            ps = self._plist = list(self.parameters())
        return (P, self.encoder_operand, tuple((p.data_ptr(), p._version) for p in ps))

    @property
    def encoder_operand_in_use(self):
        """the 16-bit operand type the next launch packs / runs: `encoder_operand` unless the range guard took float16 away"""
        return "bf16" if (self.encoder_operand == "f16" and self._range_forced_bf16) else self.encoder_operand

    @staticmethod
    def f16_operands_fit(sd, limit=65504.0):
        """-> (fits, name, max-abs): can float16 hold every encoder-side value of this state_dict (the fragments are rounded to nearest:
        anything beyond 65 504 becomes inf)?  Host-side, at pack time."""
        worst, name = 0.0, None
        for k, v in sd.items():
            if not v.is_floating_point() or k.startswith(("decoder.", "mask_token", "enc_2_dec_emb", "output_layer", "decoder_norm")):
                continue
            m = float(v.abs().max()) if v.numel() else 0.0
            if m != m:                        # NaN: nothing to fit
                return False, k, m
            if m > worst:
                worst, name = m, k
        return bool(worst <= limit), name, worst

    def packed_weights(self, P, device):
        key = self._pack_key(P)
        if self._packed is None or self._packed_key[:1] + self._packed_key[2:] != key[:1] + key[2:] or self._packed.device != device:
            self._range_forced_bf16, self._range_checked = False, 0          # other weights: judged afresh
        key = (key[0], self.encoder_operand_in_use, key[2])
        if self._packed is None or self._packed_key != key or self._packed.device != device:
            sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
            if key[1] == "f16" and self.range_guard:
                fits, name, m = self.f16_operands_fit(sd)
                if not fits:
                    import warnings
                    warnings.warn(f"step_amd.TSFormer: {name} reaches {m:.4g}, beyond float16's 65504: the fused encoder packs bfloat16 "
                                  "operand fragments instead (encoder_operand='f16' kept for weights that fit)")
                    self._range_forced_bf16, self.range_fallbacks = True, self.range_fallbacks + 1
                    key = (key[0], "bf16", key[2])
            self._packed = pack_tsformer(sd, P, depth=self.encoder_depth, operand=key[1]).to(device)
            self._packed_key = key
        return self._packed

    def _range_poll_result(self, wait=False):
        """the flag word of the last poll, once its copy has landed (None: nothing landed yet)"""
        if self._range_poll is None:
            return None
        host, ev = self._range_poll
        if wait:
            ev.synchronize()
        elif not ev.query():
            return None
        self._range_poll = None
        return int(host[0])

    def _range_overflowed(self, where, status=None):
        import warnings
        self._range_forced_bf16, self.range_fallbacks = True, self.range_fallbacks + 1
        for st in (status, self._range_status):
            if st is not None:
                st[64:65].zero_()
        warnings.warn(f"step_amd.TSFormer: float16 operand overflow in the fused encoder ({where}): non-finite hidden states; "
                      "the encoder now runs bfloat16 operand fragments (exponent range of fp32)")

    def dropout_pool(self, device, drop, seed, L):
        """The keep-mask pool for this launch: (int64 tensor viewed as 64-bit words, number of words).  ``_pool_override``
        (tests) supplies the bits instead of the Philox fill, so a test can reproduce every mask on the host."""
        if self._pool_override is not None:
            pool = self._pool_override
            words = pool.numel()
            if words & (words - 1) == 0:            # bare pool: append the wrap-around copy of its first 16 words
                pool = torch.cat([pool, pool[:16]])
            else:
                words -= 16
            return pool, words
        need = 2 * _lib.lib().step_tsformer_dropout_words(int(L), self.encoder_depth)
        words = self.dropout_pool_words
        while words < need:
            words *= 2
        if self._drop_pool is None or self._drop_pool.numel() != words + 16 or self._drop_pool.device != device:
            self._drop_pool = torch.empty(words + 16, dtype=torch.int64, device=device)      # + the wrap-around copy of the first 16
        _lib.call("step_dropout_pool_fill_dyn", _lib.ptr(self._drop_pool), words, float(drop), int(seed), _lib.ptr(self._dyn), _lib.stream())
        return self._drop_pool, words

    # ------------------------------------------------------------------ device entry points
    def encode_series(self, series, want_f32=False, want_bf16=True):
        """series: f32 cuda [S, L] (one row per (sample, node)).  Returns dict with hidden_bf16 [S,P,96],
        optional hidden_f32, last [S,96] and sqnorm [S,16] (per-wave partial squared norms)."""
        if not series.is_cuda:
            raise RuntimeError("step_amd.TSFormer runs only on an AMD GPU (no CPU fallback)")
        S, L = series.shape
        if L % self.patch_size != 0:
            raise AssertionError("long history length must be a multiple of the patch size")   # patch.py:41
        P = L // self.patch_size
        guard = self.range_guard and self.encoder_operand == "f16"
        if guard and not self._range_forced_bf16 and self._range_poll_result() and not torch.cuda.is_current_stream_capturing():
            self._range_overflowed("found by the periodic poll; batches since the previous poll carried NaN")
        pk = self.packed_weights(P, series.device)
        out = {"hidden_bf16": torch.empty(S, P, 96, device=series.device, dtype=torch.bfloat16) if want_bf16 else None,
               "hidden_f32": torch.empty(S, P, 96, device=series.device) if want_f32 else None,
               "last": torch.empty(S, 96, device=series.device),
               "sqnorm": torch.empty(S, 16, device=series.device)}
        drop = self.dropout_p if self.training else 0.0
        self._seed_counter += 1
        seed = (torch.initial_seed() * 1000003 + self._seed_counter) & ((1 << 63) - 1) if drop > 0 else 0
        pool, pool_words = None, 0
        if drop > 0:
            pool, pool_words = self.dropout_pool(series.device, drop, seed, L)
        if self._events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        def launch(pk, operand, counters, extra_flags):
            flags = (_lib.ENC_F16 if operand == "f16" else 0) | self.encoder_debug_flags | extra_flags
            if self.encoder_workgroups:
                flags |= (int(self.encoder_workgroups) & 0xffff) << 8          # STEP_ENC_WORKGROUPS(n): persistent launch on n compute units
            _lib.call("step_tsformer_encode", _lib.ptr(series), S, L, _lib.ptr(pk), pk.numel(), self.encoder_depth,
                      flags, _lib.ptr(out["hidden_bf16"]), _lib.ptr(out["hidden_f32"]), _lib.ptr(out["last"]),
                      _lib.ptr(out["sqnorm"]), float(drop), _lib.ptr(pool), pool_words, int(seed), _lib.ptr(counters),
                      _lib.stream())
        operand = self.encoder_operand_in_use
        guard = guard and operand == "f16" and not torch.cuda.is_current_stream_capturing()
        if not guard:
            launch(pk, operand, self.fallback_counter, 0)
        else:
            st = self._range_status
            if st is None or st.device != series.device:
                st = self._range_status = torch.zeros(80, dtype=torch.int32, device=series.device)
            if self.fallback_counter is not None:
                if self.fallback_counter.numel() < 65:
                    raise ValueError("TSFormer.fallback_counter needs 65 words with the range guard on (64 counters + the range flag)")
                st = self.fallback_counter          # bench / tests count the slow softmax units in the same words
            launch(pk, "f16", st, _lib.ENC_RANGE_FLAG)
            self._range_launches += 1
            if self._range_checked < self.range_check_launches:
                # the first launches on these weights: looked at right away (the frozen weights and the data's scale decide the operand
                # range; one synchronize each), and re-run on bfloat16 fragments when float16 was not enough
                self._range_checked += 1
                if int(st[64].item()) != 0:
                    self._range_overflowed("first launches on these weights; this launch is re-run", st)
                    launch(self.packed_weights(P, series.device), "bf16", self.fallback_counter, 0)
            elif self._range_poll is None and self._range_launches % self.range_poll_every == 0:
                host = torch.empty(1, dtype=torch.int32).pin_memory()
                host.copy_(st[64:65], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record()
                self._range_poll = (host, landed)
        if self._events is not None:
            ev[1].record()
            self._events.append(ev)
        return out

    def forward(self, history_data, future_data=None, batch_seen=None, epoch=None, **kwargs):
        """history_data [B, L*P, N, C] -> hidden [B, N, P, 96] (forecasting mode, tsformer.py:189-191)."""
        if self.mode == "pre-train":
            return self._forward_pretrain(history_data)
        if not history_data.is_cuda:
            raise RuntimeError("step_amd.TSFormer runs only on an AMD GPU (no CPU fallback)")
        B, L, N, Cc = history_data.shape
        x = history_data.contiguous().float()
        series = torch.empty(B * N, L, device=x.device)
        _lib.call("step_pack_long_history", _lib.ptr(x), B, L, N, Cc, self.selected_feature, _lib.ptr(series), _lib.stream())
        out = self.encode_series(series, want_f32=True, want_bf16=False)
        return out["hidden_f32"].view(B, N, L // self.patch_size, 96)

    # ------------------------------------------------------------------ pre-training (tsformer.py:180-188)
    def _pt_params(self):
        d = dict(self.named_parameters())
        return [d[n] for n in self._pt_names]

    def flatten_parameters(self):
        """Re-home every parameter as a view into ONE flat f32 buffer with the layout of the native backward's gradient buffer
        (`_pt_grad_buffers`: parameters in `_pt_names` order, each starting at a multiple of 4 floats), so that clip + Adam are one
        fused pass over (parameters, gradients, moments) -- `step_amd.optim.FusedAdamClip(tsformer, ...)`.  Values and `state_dict` are
        unchanged; call it after `.to(device)` and before building the optimizer."""
        prm = self._pt_params()
        total = sum((p.numel() + 3) & ~3 for p in prm)
        flat = torch.zeros(total, device=prm[0].device, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in prm:
                v = flat[off:off + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                off += (p.numel() + 3) & ~3
        self._flat_param = flat
        self._plist = None
        return flat

    def zero_grad(self, set_to_none=True):
        self._backward_count = 0
        super().zero_grad(set_to_none=set_to_none)

    def _pt_grad_buffers(self):
        prm = self._pt_params()
        total = sum((p.numel() + 3) & ~3 for p in prm)
        flat = torch.zeros(total, device=prm[0].device, dtype=torch.float32)
        G, off = {}, 0
        for n, p in zip(self._pt_names, prm):
            G[n] = flat[off:off + p.numel()].view(p.shape)
            off += (p.numel() + 3) & ~3
        self._flat_grad = flat
        return flat, G

    def _ffn_packs(self, device):
        """per-layer operand-fragment buffers of the fused feed-forward blocks (rewritten by step_pt_ffn_pack every step)"""
        if self._ffn_pack_bufs is None or next(iter(self._ffn_pack_bufs.values())).device != device:
            n = _lib.lib().step_pt_ffn_pack_bytes()
            keys = [f"encoder.transformer_encoder.layers.{l}." for l in range(self.encoder_depth)] + \
                   [f"decoder.transformer_encoder.layers.{l}." for l in range(self.decoder_depth)]
            self._ffn_pack_bufs = {k: torch.empty(n, dtype=torch.uint8, device=device) for k in keys}
        return self._ffn_pack_bufs

    def _proj_packs(self, device):
        """per-layer operand fragments of the attention block's projections: (qkv, out-projection, d a = d o . Wo, d x += d qkv . Wi)"""
        if self._proj_pack_bufs is None or next(iter(self._proj_pack_bufs.values()))[0].device != device:
            nb = _lib.lib().step_pt_rows_linear_pack_bytes
            keys = [f"encoder.transformer_encoder.layers.{l}." for l in range(self.encoder_depth)] + \
                   [f"decoder.transformer_encoder.layers.{l}." for l in range(self.decoder_depth)]
            self._proj_pack_bufs = {k: tuple(torch.empty(nb(kc, og), dtype=torch.uint8, device=device) for kc, og in ((1, 3), (1, 1), (1, 1), (3, 1)))
                                    for k in keys}
        return self._proj_pack_bufs

    def _pt_pool(self, device, p, seed):
        """this step's keep-mask pool of the fused feed-forward blocks: (int64 tensor, words)"""
        if self._pt_pool_buf is None or self._pt_pool_buf.device != device:
            self._pt_pool_buf = torch.empty(PT_POOL_WORDS + 16, dtype=torch.int64, device=device)
        _lib.call("step_dropout_pool_fill", _lib.ptr(self._pt_pool_buf), PT_POOL_WORDS, float(p), int(seed) ^ 0x5EED_F00D, _lib.stream())
        self._pt_pool_tag = (int(seed), float(p))          # whose masks the buffer holds (checked by the backward)
        return self._pt_pool_buf, PT_POOL_WORDS

    def _next_seed(self):
        self._seed_ctr2 += 1
        return (torch.initial_seed() * 6364136223846793005 + self._seed_ctr2 * 1442695040888963407) & ((1 << 63) - 1)

    def _forward_pretrain(self, history_data):
        """history_data [B, L, N, C] -> (reconstruction of the masked tokens [B, Pm*12, N], their labels [B, Pm*12, N])."""
        if not history_data.is_cuda:
            raise RuntimeError("step_amd.TSFormer runs only on an AMD GPU (no CPU fallback)")
        B, L, N, Cc = history_data.shape
        x = history_data.contiguous().float()
        series = torch.empty(B * N, L, device=x.device)
        _lib.call("step_pack_long_history", _lib.ptr(x), B, L, N, Cc, self.selected_feature, _lib.ptr(series), _lib.stream())
        um_list, mk_list = self.mask()
        um = torch.tensor(um_list, dtype=torch.int32, device=x.device)
        mk = torch.tensor(mk_list, dtype=torch.int32, device=x.device)
        P = L // self.patch_size
        r = _PretrainFunction.apply(self, series, um, mk, *self._pt_params())              # [S, P, 12]
        Pu = len(um_list)
        recon = r.view(B, N, P, self.patch_size)[:, :, Pu:, :].reshape(B, N, -1).transpose(1, 2)      # tsformer.py:153-154
        label = series.view(B, N, P, self.patch_size)[:, :, mk.long(), :].reshape(B, N, -1).transpose(1, 2)   # :156-158
        return recon, label
