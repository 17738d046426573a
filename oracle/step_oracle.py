"""CPU oracle for the STEP training-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a functional restatement (plain torch CPU tensors, explicit math, fp32 or
fp64) of the arithmetic the reference performs on the hot path.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` as
the *checker*; nothing under ``step_amd/`` may import it.

Pinning: the reference ships no golden vectors, so this restatement
is pinned against outputs of the reference's own modules run in the build container
(``tools/make_golden.py`` imports ``/root/reference/step/step_arch`` unmodified and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).

Every function cites the reference lines it restates (paths relative to the reference
checkout).  Parameter dictionaries use the reference's ``state_dict`` key names.
Autograd is left enabled on purpose: gradients of these functions are the gradient
oracle for the hand-written backward kernels.
"""
import math

import torch

PATCH = 12          # step/STEP_PEMS04.py:45  patch_size
EMBED = 96          # step/STEP_PEMS04.py:47  embed_dim
HEADS = 4           # step/STEP_PEMS04.py:48  num_heads
HDIM = EMBED // HEADS
LN_EPS = 1e-5       # torch.nn.LayerNorm default, used by tsformer.py:41-42 and nn.TransformerEncoderLayer
BN_EPS = 1e-5       # torch.nn.BatchNorm default
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------- helpers
def layer_norm(x, w, b, eps=LN_EPS):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def batch_norm_train(x, w, b, dims, eps=BN_EPS):
    """Training-mode batch norm over ``dims`` (biased variance for normalisation).
    Returns (y, batch_mean, batch_var_unbiased) so callers can update running stats the
    way torch.nn.BatchNorm does (momentum 0.1, unbiased var)."""
    n = 1
    for d in dims:
        n *= x.shape[d]
    mu = x.mean(dims, keepdim=True)
    var = ((x - mu) ** 2).mean(dims, keepdim=True)
    y = (x - mu) / torch.sqrt(var + eps)
    shape = [1] * x.dim()
    ch = [d for d in range(x.dim()) if d not in dims][0]
    shape[ch] = -1
    y = y * w.view(shape) + b.view(shape)
    return y, mu.flatten(), var.flatten() * (n / max(n - 1, 1))


def batch_norm_eval(x, w, b, rm, rv, ch_dim, eps=BN_EPS):
    shape = [1] * x.dim()
    shape[ch_dim] = -1
    return (x - rm.view(shape)) / torch.sqrt(rv.view(shape) + eps) * w.view(shape) + b.view(shape)


# ----------------------------------------------------------------------------- TSFormer
def _drop(x, mask, keep):
    """torch.nn.functional.dropout with the realisation given: mask is 0/1 (1 = keep), survivors scaled by 1/keep."""
    return x if mask is None else x * mask.to(x.dtype) / keep


def encoder_layer(h, p, pre, drop=None, keep=1.0):
    """One post-norm encoder layer on h[S, P, 96] (S sequences).
    Restates torch.nn.TransformerEncoderLayer(d, 4, 4d, dropout) as instantiated at
    step/step_arch/tsformer/transformer_layers.py:10-11 (norm_first=False, relu).  ``drop`` = None (dropout off) or the
    keep-masks of the layer's four dropout sites: attn[S,H,P,P] on the softmax output (F.multi_head_attention_forward),
    drop1[S,P,96] on the attention block output, ffn[S,P,384] after the activation, drop2[S,P,96] on the FFN output."""
    drop = drop or {}
    S, P, D = h.shape
    qkv = h @ p[pre + "self_attn.in_proj_weight"].T + p[pre + "self_attn.in_proj_bias"]
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(S, P, HEADS, HDIM).transpose(1, 2)
    k = k.reshape(S, P, HEADS, HDIM).transpose(1, 2)
    v = v.reshape(S, P, HEADS, HDIM).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(HDIM), dim=-1)
    att = _drop(att, drop.get("attn"), keep)
    o = (att @ v).transpose(1, 2).reshape(S, P, D)
    o = o @ p[pre + "self_attn.out_proj.weight"].T + p[pre + "self_attn.out_proj.bias"]
    h = layer_norm(h + _drop(o, drop.get("drop1"), keep), p[pre + "norm1.weight"], p[pre + "norm1.bias"])
    f = torch.relu(h @ p[pre + "linear1.weight"].T + p[pre + "linear1.bias"])
    f = _drop(f, drop.get("ffn"), keep) @ p[pre + "linear2.weight"].T + p[pre + "linear2.bias"]
    return layer_norm(h + _drop(f, drop.get("drop2"), keep), p[pre + "norm2.weight"], p[pre + "norm2.bias"])


def transformer_layers(h, p, pre, depth, drop_layers=None, keep=1.0):
    """transformer_layers.py:13-20: scale by sqrt(d) then ``depth`` layers."""
    h = h * math.sqrt(EMBED)
    for i in range(depth):
        h = encoder_layer(h, p, f"{pre}transformer_encoder.layers.{i}.", None if drop_layers is None else drop_layers[i], keep)
    return h


def patch_embed(x, p, pre="tsformer."):
    """patch.py:20-42: Conv2d(1->96, k=(12,1), s=(12,1)) == per-patch linear 12->96.
    x[S, L] -> [S, P, 96]."""
    S, L = x.shape
    w = p[pre + "patch_embedding.input_embedding.weight"][:, 0, :, 0]      # [96, 12]
    b = p[pre + "patch_embedding.input_embedding.bias"]
    return x.reshape(S, L // PATCH, PATCH) @ w.T + b


def tsformer_encode(long_hist, p, pre="tsformer.", depth=4, drop=None, keep=1.0):
    """Forecasting-mode TSFormer (tsformer.py:71-105,179,190; positional_encoding.py:28-32).
    long_hist[B, L, N] (channel 0 only) -> hidden[B, N, P, 96].  ``drop`` = None (dropout off) or the keep-masks (tensors of
    0/1, sequences in (b, n) order) of a training-mode forward, ``keep`` = 1 - p: {"pos": [S,P,96] (positional_encoding.py:32),
    "layers": [per encoder layer the dict encoder_layer takes]}."""
    B, L, N = long_hist.shape
    x = long_hist.permute(0, 2, 1).reshape(B * N, L)
    h = patch_embed(x, p, pre)
    P = h.shape[1]
    h = h + p[pre + "positional_encoding.position_embedding"][:P]
    if drop is not None:
        h = _drop(h, drop["pos"], keep)
    h = transformer_layers(h, p, pre + "encoder.", depth, None if drop is None else drop["layers"], keep)
    h = layer_norm(h, p[pre + "encoder_norm.weight"], p[pre + "encoder_norm.bias"])
    return h.reshape(B, N, P, EMBED)


def tsformer_pretrain(hist, p, unmasked, masked, pre="tsformer.", enc_depth=4, dec_depth=1):
    """Pre-train mode (tsformer.py:71-160,180-188) with the mask index lists given
    (mask.py:15-28 draws them on the host).  hist[B, L, N, C] -> (recon, label), both
    [B, len(masked)*12, N].  Dropout off."""
    B, L, N, _ = hist.shape
    x = hist[..., 0].permute(0, 2, 1).reshape(B * N, L)
    pos = p[pre + "positional_encoding.position_embedding"]
    h = patch_embed(x, p, pre)
    P = h.shape[1]
    h = h + pos[:P]
    um = torch.as_tensor(unmasked, dtype=torch.long)
    mk = torch.as_tensor(masked, dtype=torch.long)
    h = transformer_layers(h[:, um, :], p, pre + "encoder.", enc_depth)
    h = layer_norm(h, p[pre + "encoder_norm.weight"], p[pre + "encoder_norm.bias"])
    z = h @ p[pre + "enc_2_dec_emb.weight"].T + p[pre + "enc_2_dec_emb.bias"]
    m = p[pre + "mask_token"].reshape(1, 1, EMBED) + pos[mk].unsqueeze(0)
    full = torch.cat([z, m.expand(B * N, len(masked), EMBED)], dim=1)
    d = transformer_layers(full, p, pre + "decoder.", dec_depth)
    d = layer_norm(d, p[pre + "decoder_norm.weight"], p[pre + "decoder_norm.bias"])
    r = d @ p[pre + "output_layer.weight"].T + p[pre + "output_layer.bias"]      # [S, P, 12]
    recon = r[:, len(unmasked):, :].reshape(B, N, -1).transpose(1, 2)
    label = x.reshape(B * N, P, PATCH)[:, mk, :].reshape(B, N, -1).transpose(1, 2)
    return recon, label


# ----------------------------------------------------------------------------- DGL
def conv1d_valid(x, w, b):
    """Plain 'valid' cross-correlation, x[N, Ci, T], w[Co, Ci, K] -> [N, Co, T-K+1]
    (what torch.nn.Conv1d(stride=1, padding=0) computes, discrete_graph_learning.py:63-64)."""
    return torch.nn.functional.conv1d(x, w, b)


def dgl_global_feature(node_feats, p, pre="discrete_graph_learning.", training=True, stats=None):
    """discrete_graph_learning.py:131-136.  node_feats[T, N] -> g[N, 100].
    ``stats`` (dict) receives the batch statistics of bn1/bn2/bn3 when training."""
    s = node_feats.transpose(0, 1).unsqueeze(1)                                   # [N,1,T]

    def bn(x, name, dims, ch):
        if training:
            y, mu, var_u = batch_norm_train(x, p[pre + name + ".weight"], p[pre + name + ".bias"], dims)
            if stats is not None:
                stats[name] = (mu.detach(), var_u.detach())
            return y
        return batch_norm_eval(x, p[pre + name + ".weight"], p[pre + name + ".bias"],
                               p[pre + name + ".running_mean"], p[pre + name + ".running_var"], ch)

    c1 = bn(torch.relu(conv1d_valid(s, p[pre + "conv1.weight"], p[pre + "conv1.bias"])), "bn1", (0, 2), 1)
    c2 = bn(torch.relu(conv1d_valid(c1, p[pre + "conv2.weight"], p[pre + "conv2.bias"])), "bn2", (0, 2), 1)
    flat = c2.reshape(c2.shape[0], -1)                                            # channel-major
    g = torch.relu(flat @ p[pre + "fc.weight"].T + p[pre + "fc.bias"])
    return bn(g, "bn3", (0,), 1)


def dgl_edge_logits(g, p, pre="discrete_graph_learning.", row_chunk=None):
    """discrete_graph_learning.py:148-153 with the one-hot matmuls (rel_rec / rel_send,
    :81-89) replaced by their meaning: edge e = i*N + j has receiver i and sender j, and
    the concat order is [sender, receiver].  g[N,100] -> logits[N*N, 2] (batch-invariant).
    ``row_chunk``: evaluate ``row_chunk`` receiver rows at a time and re-compute each block in the
    backward pass (torch.utils.checkpoint), so that the [N, N, 100] hidden tensor -- 6.7 GB at
    N = 4096, several copies of it under autograd -- never exists; same arithmetic per edge."""
    N, E = g.shape
    w = p[pre + "fc_out.weight"]
    snd = g @ w[:, :E].T                    # indexed by j
    rcv = g @ w[:, E:].T                    # indexed by i

    def rows(rcv_rows, snd_all, b_out, w_cat, b_cat):
        hid = torch.relu(rcv_rows.unsqueeze(1) + snd_all.unsqueeze(0) + b_out)        # [i, j, 100]
        return hid @ w_cat.T + b_cat
    args = (p[pre + "fc_out.bias"], p[pre + "fc_cat.weight"], p[pre + "fc_cat.bias"])
    if row_chunk is None or row_chunk >= N:
        return rows(rcv, snd, *args).reshape(N * N, 2)
    from torch.utils.checkpoint import checkpoint
    need_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (rcv, snd) + args)
    out = []
    for i0 in range(0, N, row_chunk):
        blk = rcv[i0:i0 + row_chunk]
        out.append(checkpoint(rows, blk, snd, *args, use_reentrant=False) if need_grad else rows(blk, snd, *args))
    return torch.cat(out, 0).reshape(N * N, 2)


def gumbel_hard_sample(logits, u, temperature=0.5, eps=1e-10):
    """discrete_graph_learning.py:11-45 as called at :157 (eps=1e-10 reaches sample_gumbel).
    logits[B, E, 2], u ~ U[0,1) same shape.  Value is one-hot, gradient is that of the
    soft sample (straight-through)."""
    gmb = -torch.log(-torch.log(u + eps) + eps)
    y = torch.softmax((logits + gmb) / temperature, dim=-1)
    hard = torch.zeros_like(y).scatter_(-1, y.detach().argmax(-1, keepdim=True), 1.0)
    return (hard - y).detach() + y


def cosine_knn_graph(hidden, k_total):
    """similarity.py:6-16 + discrete_graph_learning.py:91-111,164-166.
    hidden[B, N, F] -> {0,1}[B, N, N]: the k_total largest entries of the flattened cosine
    matrix whose value is non-zero, diagonal cleared afterwards."""
    B, N, _ = hidden.shape
    nrm = hidden.norm(dim=2) + 1e-7
    sim = (hidden @ hidden.transpose(1, 2)) / (nrm.unsqueeze(2) * nrm.unsqueeze(1))
    flat = sim.reshape(B, N * N)
    val, idx = torch.topk(flat, k_total, dim=-1)
    res = torch.zeros_like(flat).scatter_(-1, idx, val)
    adj = (res != 0).to(hidden.dtype).reshape(B, N, N)
    adj = adj * (1.0 - torch.eye(N, dtype=hidden.dtype))
    return adj.detach(), sim.detach()


def dgl_forward(long_hist0, node_feats, p, u, k, training=True, stats=None,
                pre="discrete_graph_learning.", hidden=None, edge_row_chunk=None, g=None):
    """DiscreteGraphLearning.forward (discrete_graph_learning.py:113-168).
    long_hist0[B, L, N] is channel 0 of the long history; u is the uniform noise the
    reference draws with torch.rand at :12.  Returns (logits[B,N*N,2], hidden[B,N,P,96],
    adj_knn[B,N,N], sampled_adj[B,N,N])."""
    B, _, N = long_hist0.shape
    if g is None:             # (tests of very large graphs pass the global feature in, e.g. computed without autograd)
        g = dgl_global_feature(node_feats, p, pre, training, stats)
    if hidden is None:        # tests may inject the device encoder's (bf16) hidden states instead
        hidden = tsformer_encode(long_hist0, p).detach()
    logits = dgl_edge_logits(g, p, pre, row_chunk=edge_row_chunk).unsqueeze(0).expand(B, N * N, 2)
    samp = gumbel_hard_sample(logits, u)[..., 0].reshape(B, N, N)
    samp = samp * (1.0 - torch.eye(N, dtype=samp.dtype))
    adj_knn, _ = cosine_knn_graph(hidden.reshape(B, N, -1), k * N)
    return logits, hidden, adj_knn, samp


# ----------------------------------------------------------------------------- GraphWaveNet
def random_walk(adj):
    """graphwavenet/model.py:121-130: D^-1 (A + I), rows with zero degree -> 0."""
    N = adj.shape[-1]
    a = adj + torch.eye(N, dtype=adj.dtype)
    d = a.sum(2)
    dinv = torch.where(d == 0, torch.zeros_like(d), 1.0 / d)
    return dinv.unsqueeze(2) * a


def nconv(x, a):
    """graphwavenet/model.py:10-16.  x[B,C,V,T]; a[B,V,W] or [V,W]; contraction over the
    FIRST index of a."""
    if a.dim() == 3:
        return torch.einsum("ncvl,nvw->ncwl", x, a)
    return torch.einsum("ncvl,vw->ncwl", x, a)


def gwnet_forward(hist, hidden_last, sampled_adj, p, pre="backend.", training=True,
                  drop_masks=None, stats=None, blocks=4, layers=2):
    """GraphWaveNet.forward (graphwavenet/model.py:132-224), STEP's fork.
    hist[B,12,N,3], hidden_last[B,N,96], sampled_adj[B,N,N] -> [B,N,12].
    ``drop_masks[i]`` (already scaled by 1/(1-0.3), shape of layer i's gcn output) restates
    F.dropout at model.py:47; None means dropout off.  BN uses batch stats when training."""
    x = hist.transpose(1, 3)                                     # [B,C,N,T]
    x = torch.nn.functional.pad(x, (1, 0, 0, 0))[:, :2]          # left-pad T by 1, keep 2 channels
    w = p[pre + "start_conv.weight"][:, :, 0, 0]
    x = torch.einsum("oc,bcnt->bont", w, x) + p[pre + "start_conv.bias"].view(1, -1, 1, 1)
    supports = [random_walk(sampled_adj), random_walk(sampled_adj.transpose(-1, -2)),
                torch.softmax(torch.relu(p[pre + "nodevec1"] @ p[pre + "nodevec2"]), dim=1)]
    skip = 0
    nl = blocks * layers
    for i in range(nl):
        dil = 2 ** (i % layers)
        res = x
        T = x.shape[3] - dil

        def tconv(name):
            wt = p[f"{pre}{name}.{i}.weight"][:, :, 0, :]        # [32,32,2]
            return (torch.einsum("oc,bcnt->bont", wt[:, :, 0], res[..., :T]) +
                    torch.einsum("oc,bcnt->bont", wt[:, :, 1], res[..., dil:dil + T]) +
                    p[f"{pre}{name}.{i}.bias"].view(1, -1, 1, 1))
        x = torch.tanh(tconv("filter_convs")) * torch.sigmoid(tconv("gate_convs"))
        s = torch.einsum("oc,bcnt->bont", p[f"{pre}skip_convs.{i}.weight"][:, :, 0, 0], x) + \
            p[f"{pre}skip_convs.{i}.bias"].view(1, -1, 1, 1)
        skip = s + (skip[..., -T:] if i > 0 else 0)
        if i == nl - 1:
            break                                                # model.py:202-213 result is dead for the last layer
        out = [x]
        for a in supports:
            x1 = nconv(x, a)
            x2 = nconv(x1, a)
            out += [x1, x2]
        h = torch.cat(out, dim=1)
        h = torch.einsum("oc,bcnt->bont", p[f"{pre}gconv.{i}.mlp.mlp.weight"][:, :, 0, 0], h) + \
            p[f"{pre}gconv.{i}.mlp.mlp.bias"].view(1, -1, 1, 1)
        if drop_masks is not None:
            h = h * drop_masks[i]
        x = h + res[..., -T:]
        if training:
            x, mu, var_u = batch_norm_train(x, p[f"{pre}bn.{i}.weight"], p[f"{pre}bn.{i}.bias"], (0, 2, 3))
            if stats is not None:
                stats[f"bn.{i}"] = (mu.detach(), var_u.detach())
        else:
            x = batch_norm_eval(x, p[f"{pre}bn.{i}.weight"], p[f"{pre}bn.{i}.bias"],
                                p[f"{pre}bn.{i}.running_mean"], p[f"{pre}bn.{i}.running_var"], 1)
    hs = torch.relu(hidden_last @ p[pre + "fc_his.0.weight"].T + p[pre + "fc_his.0.bias"])
    hs = torch.relu(hs @ p[pre + "fc_his.2.weight"].T + p[pre + "fc_his.2.bias"])       # [B,N,256]
    x = torch.relu(skip + hs.transpose(1, 2).unsqueeze(-1))
    x = torch.relu(torch.einsum("oc,bcnt->bont", p[pre + "end_conv_1.weight"][:, :, 0, 0], x) +
                   p[pre + "end_conv_1.bias"].view(1, -1, 1, 1))
    x = torch.einsum("oc,bcnt->bont", p[pre + "end_conv_2.weight"][:, :, 0, 0], x) + \
        p[pre + "end_conv_2.bias"].view(1, -1, 1, 1)
    return x.squeeze(-1).transpose(1, 2)                          # [B,N,12]


# ----------------------------------------------------------------------------- STEP + loss
def step_forward(hist, long_hist, node_feats, p, u, k, epoch, training=True,
                 drop_masks=None, stats=None, hidden=None, hidden_last=None, aux=None, edge_row_chunk=None, g=None):
    """STEP.forward (step/step_arch/step.py:37-72).  Returns
    (prediction[B,12,N,1], theta[B,N,N], adj_knn[B,N,N], gsl_coefficient)."""
    B, _, N, _ = hist.shape
    logits, hidden, adj_knn, samp = dgl_forward(long_hist[..., 0], node_feats, p, u, k, training, stats, hidden=hidden,
                                                edge_row_chunk=edge_row_chunk, g=g)
    if aux is not None:
        aux["sampled_adj"] = samp.detach()
    last = hidden[:, :, -1, :] if hidden_last is None else hidden_last
    y = gwnet_forward(hist, last, samp, p, training=training,
                      drop_masks=drop_masks, stats=stats).transpose(1, 2)
    coef = 1 / (int(epoch / 6) + 1) if epoch is not None else 0
    theta = torch.softmax(logits, -1)[..., 0].reshape(B, N, N)
    return y.unsqueeze(-1), theta, adj_knn, coef


def masked_mae(pred, label, null_val=0.0):
    """basicts/metrics/mae.py:5-28 with a finite null_val."""
    mask = (~torch.isclose(label, torch.full_like(label, null_val), atol=5e-5, rtol=0.0)).to(pred.dtype)
    mask = mask / mask.mean()
    mask = torch.where(torch.isnan(mask), torch.zeros_like(mask), mask)
    loss = (pred - label).abs() * mask
    return torch.where(torch.isnan(loss), torch.zeros_like(loss), loss).mean()


def step_loss(pred, real, theta, prior, coef, null_val=0.0):
    """step/step_loss/step_loss.py:5-16 (BCELoss mean + masked MAE).  torch's BCELoss clamps
    each log term at -100."""
    B, N, _ = theta.shape
    t = theta.reshape(B, N * N)
    y = prior.reshape(B, N * N)
    bce = -(y * torch.log(t).clamp(min=-100) + (1 - y) * torch.log(1 - t).clamp(min=-100)).mean()
    return masked_mae(pred, real, null_val) + bce * coef


def rescale(x, mean, std):
    """basicts/data/transform.py:49-65 re_standard_transform with scalar mean/std
    (applied by base_tsf_runner.py:240-241 before the loss)."""
    return x * std + mean
