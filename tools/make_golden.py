"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU.

Runs only in the build container (needs /root/reference).  The reference's arithmetic
(`forward` of STEP / DiscreteGraphLearning / TSFormer / GraphWaveNet, `step_loss`,
`masked_mae`) is executed unmodified; the only liberties taken are
  * in-memory stubs for the absent third-party imports (easytorch, timm) -- SURVEY.md 8c;
  * STEP / DiscreteGraphLearning objects are assembled attribute-by-attribute instead of
    through their constructors, because the constructors hard-code per-dataset sizes and
    read files from the cwd (discrete_graph_learning.py:55-57,61,73; step.py:27-35).
    The attributes set here are exactly the ones those constructors set.
  * dropout probabilities are set to 0 so the only random draw left is the Gumbel noise
    (`torch.rand`, discrete_graph_learning.py:12), which is recorded and stored; the one case that keeps dropout on
    (run_dropout_case) records every mask instead.

Usage:  python tools/make_golden.py            (writes tests/golden/step_tiny.npz, ...)
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.reference_loader import reference_root  # noqa: E402

REF = reference_root() or "/root/reference"          # the build container's checkout, or the staged copy (tools/stage_reference.sh)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    sys.modules.setdefault("easytorch", types.ModuleType("easytorch"))
    timm = types.ModuleType("timm")
    timm.models = types.ModuleType("timm.models")
    timm.models.vision_transformer = types.ModuleType("timm.models.vision_transformer")
    timm.models.vision_transformer.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules["timm"] = timm
    sys.modules["timm.models"] = timm.models
    sys.modules["timm.models.vision_transformer"] = timm.models.vision_transformer
    # basicts/__init__ pulls in launcher/runners (easytorch); the hot path needs only
    # basicts.utils.load_pkl and basicts.losses/metrics -> stub the package shell.
    bts = types.ModuleType("basicts")
    bts.__path__ = [os.path.join(REF, "basicts")]
    sys.modules["basicts"] = bts
    sys.path.insert(0, REF)
    step_pkg = types.ModuleType("step")
    step_pkg.__path__ = [os.path.join(REF, "step")]
    sys.modules["step"] = step_pkg
    from step.step_arch.step import STEP
    from step.step_arch.tsformer import TSFormer
    from step.step_arch.graphwavenet import GraphWaveNet
    from step.step_arch.discrete_graph_learning import DiscreteGraphLearning
    import step.step_arch.discrete_graph_learning as dgl_mod
    from step.step_loss.step_loss import step_loss
    from basicts.metrics.mae import masked_mae
    return STEP, TSFormer, GraphWaveNet, DiscreteGraphLearning, dgl_mod, step_loss, masked_mae


def synth_params(state_dict, rng):
    """Deterministic non-trivial values for every tensor of a state_dict."""
    out = {}
    for k, v in state_dict.items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            out[k] = torch.tensor(rng.uniform(0.5, 1.5, shp), dtype=torch.float32)
        elif k.endswith("running_mean"):
            out[k] = torch.tensor(rng.normal(0, 0.1, shp), dtype=torch.float32)
        elif ("norm" in k or ".bn" in k or "bn." in k) and k.endswith("weight"):
            out[k] = torch.tensor(rng.uniform(0.8, 1.2, shp), dtype=torch.float32)
        elif k.endswith("bias"):
            out[k] = torch.tensor(rng.normal(0, 0.05, shp), dtype=torch.float32)
        elif "position_embedding" in k or "mask_token" in k:
            out[k] = torch.tensor(rng.uniform(-0.02, 0.02, shp), dtype=torch.float32)
        elif "nodevec" in k:
            out[k] = torch.tensor(rng.normal(0, 1.0, shp), dtype=torch.float32)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
            out[k] = torch.tensor(rng.normal(0, 1.0 / np.sqrt(fan_in), shp), dtype=torch.float32)
    return out


def synth_series(T, N, rng):
    """[T, N, 3] in the processed-data format."""
    t = np.arange(T)[:, None]
    phase = rng.uniform(0, 2 * np.pi, (1, N))
    amp = rng.uniform(0.5, 1.5, (1, N))
    ch0 = amp * np.sin(2 * np.pi * t / 48.0 + phase) + 0.5 * rng.normal(size=(T, N))
    ch1 = np.broadcast_to((t % 288) / 288.0, (T, N))
    ch2 = np.broadcast_to((t // 288) % 7, (T, N)).astype(np.float64)
    return np.stack([ch0, ch1, ch2], -1).astype(np.float32)


def zero_dropout(mod):
    for m in mod.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
        if hasattr(m, "dropout") and isinstance(getattr(m, "dropout"), float):
            m.dropout = 0.0


def build_step(ref, N, L, T_train, series, k):
    STEP, TSFormer, GraphWaveNet, DGL, dgl_mod, _, _ = ref
    targs = dict(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1,
                 num_token=L / 12, mask_ratio=0.75, encoder_depth=4, decoder_depth=1, mode="forecasting")
    bargs = dict(num_nodes=N, support_len=2, dropout=0.3, gcn_bool=True, addaptadj=True, aptinit=None,
                 in_dim=2, out_dim=12, residual_channels=32, dilation_channels=32, skip_channels=256,
                 end_channels=512, kernel_size=2, blocks=4, layers=2)
    model = STEP.__new__(STEP)
    torch.nn.Module.__init__(model)
    model.dataset_name = "SYNTH"
    model.pre_trained_tsformer_path = None
    model.tsformer = TSFormer(**targs)
    model.backend = GraphWaveNet(**bargs)
    for prm in model.tsformer.parameters():          # step.py:34-35
        prm.requires_grad = False
    d = DGL.__new__(DGL)
    torch.nn.Module.__init__(d)
    d.k = k
    d.num_nodes = N
    d.train_length = T_train
    d.node_feats = torch.from_numpy(series).float()[:T_train, :, 0]
    d.dim_fc = 16 * (T_train - 18)
    d.embedding_dim = 100
    d.conv1 = torch.nn.Conv1d(1, 8, 10, stride=1)
    d.conv2 = torch.nn.Conv1d(8, 16, 10, stride=1)
    d.fc = torch.nn.Linear(d.dim_fc, d.embedding_dim)
    d.bn1 = torch.nn.BatchNorm1d(8)
    d.bn2 = torch.nn.BatchNorm1d(16)
    d.bn3 = torch.nn.BatchNorm1d(d.embedding_dim)
    d.dim_fc_mean = (L // 12) * 96
    d.fc_mean = torch.nn.Linear(d.dim_fc_mean, 100)
    d.fc_cat = torch.nn.Linear(d.embedding_dim, 2)
    d.fc_out = torch.nn.Linear(d.embedding_dim * 2, d.embedding_dim)
    d.dropout = torch.nn.Dropout(0.5)
    eye = np.eye(N, dtype=np.float32)
    rows, cols = np.where(np.ones((N, N)))
    d.rel_rec = torch.FloatTensor(eye[rows])          # one-hot of the receiver index (== encode_one_hot)
    d.rel_send = torch.FloatTensor(eye[cols])
    model.discrete_graph_learning = d
    return model


def run_step_case(ref, name, N, L, T_train, B, k, epoch, seed, training=True):
    _, _, _, _, dgl_mod, step_loss, _ = ref
    rng = np.random.default_rng(seed)
    T_all = T_train + L + 64
    series = synth_series(T_all, N, rng)
    model = build_step(ref, N, L, T_train, series, k)
    params = synth_params(model.state_dict(), rng)
    model.load_state_dict(params)
    zero_dropout(model)
    model.train(training)
    # windows with full long history (t >= L): forecasting_dataset.py:62-71
    ts = rng.integers(L, T_all - 12, size=B)
    data = torch.from_numpy(series)
    hist = torch.stack([data[t - 12:t] for t in ts])
    fut = torch.stack([data[t:t + 12] for t in ts])
    longh = torch.stack([data[t - L:t] for t in ts])
    recorded = {}
    real_rand = torch.rand

    def rec_rand(*a, **kw):
        r = real_rand(*a, **kw)
        recorded["u"] = r.clone()
        return r
    torch.manual_seed(seed)
    dgl_mod.torch.rand = rec_rand
    try:
        pred, theta, knn, coef = model(history_data=hist, long_history_data=longh, future_data=None,
                                       batch_seen=0, epoch=epoch)
    finally:
        dgl_mod.torch.rand = real_rand
    mean, std = 200.0, 150.0
    loss = step_loss(pred[..., [0]] * std + mean, fut[..., [0]] * std + mean, theta, knn, coef, null_val=0.0)
    out = {"in.hist": hist, "in.long_hist0": longh[..., 0], "in.future": fut, "in.node_feats": model.discrete_graph_learning.node_feats,
           "in.u": recorded["u"], "out.pred": pred, "out.theta": theta, "out.knn": knn,
           "out.loss": loss.detach(), "meta": np.array([N, L, T_train, B, k, epoch if epoch is not None else -1, int(training)]),
           "meta.coef": np.float64(coef), "meta.scaler": np.array([mean, std])}
    with torch.no_grad():
        hidden = model.tsformer(longh[..., [0]])
    out["out.hidden"] = hidden
    if training:
        loss.backward()
        for n_, p_ in model.named_parameters():
            if p_.grad is not None:
                out["grad." + n_] = p_.grad
        out["meta.nograd"] = np.array([n_ for n_, p_ in model.named_parameters()
                                       if p_.requires_grad and p_.grad is None])
        for k_, v_ in model.state_dict().items():
            if "running_" in k_ or "num_batches" in k_:
                out["after." + k_] = v_
    for k_, v_ in params.items():
        out["param." + k_] = v_
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k_: (v_.detach().numpy() if torch.is_tensor(v_) else v_) for k_, v_ in out.items()})
    print(name, "loss", float(loss), "pred", tuple(pred.shape), "knn ones", float(knn.sum()),
          "theta mean", float(theta.mean()))


def run_pretrain_case(ref, name, N, L, B, seed):
    _, TSFormer, _, _, _, _, masked_mae = ref
    rng = np.random.default_rng(seed)
    series = synth_series(L + 32, N, rng)
    targs = dict(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1,
                 num_token=L / 12, mask_ratio=0.75, encoder_depth=4, decoder_depth=1, mode="pre-train")
    model = TSFormer(**targs)
    params = synth_params(model.state_dict(), rng)
    model.load_state_dict(params)
    zero_dropout(model)
    model.train()
    data = torch.from_numpy(series)
    ts = rng.integers(L, L + 32, size=B)
    x = torch.stack([data[t - L:t] for t in ts])[..., [0]]
    import random
    random.seed(seed)
    recon, label = model(history_data=x, future_data=None, batch_seen=0, epoch=1)
    um, mk = model.mask.unmasked_tokens, model.mask.masked_tokens
    loss = masked_mae(recon * 150.0 + 200.0, label * 150.0 + 200.0, 0.0)
    loss.backward()
    out = {"in.x": x, "in.unmasked": np.array(um), "in.masked": np.array(mk), "out.recon": recon,
           "out.label": label, "out.loss": loss.detach()}
    for n_, p_ in model.named_parameters():
        out["param." + n_] = params[n_]
        if p_.grad is not None:
            out["grad." + n_] = p_.grad
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k_: (v_.detach().numpy() if torch.is_tensor(v_) else v_) for k_, v_ in out.items()})
    print(name, "loss", float(loss), tuple(recon.shape))


def run_dropout_case(ref, name, N, L, B, seed, p_drop=0.1):
    """Training-mode forward of the reference TSFormer (forecasting mode) with every dropout realisation recorded.
    torch.nn.functional.dropout is wrapped (same arithmetic: x * mask / (1 - p)) so the masks can be stored, and
    torch.nn.functional.scaled_dot_product_attention -- the call F.multi_head_attention_forward makes for
    need_weights=False -- is replaced by its documented definition softmax(q k^T / sqrt(d)) -> dropout -> @ v, the only way to
    observe the attention-probability masks.  Everything else is the reference's unmodified code in train mode."""
    import math
    import torch.nn.functional as F
    _, TSFormer, _, _, _, _, _ = ref
    rng = np.random.default_rng(seed)
    series = synth_series(L + 32, N, rng)
    targs = dict(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=p_drop,
                 num_token=L / 12, mask_ratio=0.75, encoder_depth=4, decoder_depth=1, mode="forecasting")
    model = TSFormer(**targs)
    params = synth_params(model.state_dict(), rng)
    model.load_state_dict(params)
    model.train()
    data = torch.from_numpy(series)
    ts = rng.integers(L, L + 32, size=B)
    x = torch.stack([data[t - L:t] for t in ts])[..., [0]]
    masks = []
    real_dropout, real_sdpa = F.dropout, F.scaled_dot_product_attention

    def rec_dropout(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        m = (torch.rand_like(inp) >= p)
        masks.append(m.clone())
        return inp * m.to(inp.dtype) / (1.0 - p)

    def explicit_sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, **kw):
        assert attn_mask is None and not is_causal
        w = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(q.shape[-1]), dim=-1)
        return rec_dropout(w, dropout_p, True) @ v
    torch.manual_seed(seed)
    F.dropout, F.scaled_dot_product_attention = rec_dropout, explicit_sdpa
    try:
        with torch.no_grad():
            hidden = model(x)
    finally:
        F.dropout, F.scaled_dot_product_attention = real_dropout, real_sdpa
    S, P = B * N, L // 12
    assert len(masks) == 1 + 4 * 4, len(masks)
    out = {"in.x": x, "out.hidden": hidden, "meta": np.array([N, L, B]), "meta.p": np.float64(p_drop)}
    # positional_encoding.py:32 sees [B, N, P, d]; the encoder layers run seq-first [P, S, d] (transformer_layers.py:14-17),
    # attention weights are [S * heads (batch-major), P, P] -> store everything sequence-major
    out["mask.pos"] = masks[0].reshape(S, P, 96)
    for l in range(4):
        a, d1, f, d2 = masks[1 + 4 * l: 5 + 4 * l]
        out[f"mask.{l}.attn"] = a.reshape(S, 4, P, P)
        out[f"mask.{l}.drop1"] = d1.permute(1, 0, 2)
        out[f"mask.{l}.ffn"] = f.permute(1, 0, 2)
        out[f"mask.{l}.drop2"] = d2.permute(1, 0, 2)
    for n_, _ in model.named_parameters():
        out["param." + n_] = params[n_]
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k_: (v_.detach().numpy() if torch.is_tensor(v_) else v_) for k_, v_ in out.items()})
    print(name, "hidden", tuple(hidden.shape), "keep rates", [round(float(m.float().mean()), 3) for m in masks[:5]])


if __name__ == "__main__":
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    run_step_case(ref, "step_tiny", N=20, L=96, T_train=120, B=2, k=3, epoch=1, seed=1)
    run_step_case(ref, "step_small", N=37, L=288, T_train=200, B=3, k=4, epoch=7, seed=2)
    run_step_case(ref, "step_tiny_eval", N=20, L=96, T_train=120, B=2, k=3, epoch=None, seed=3, training=False)
    run_pretrain_case(ref, "tsformer_pretrain_tiny", N=9, L=192, B=2, seed=4)
    run_dropout_case(ref, "tsformer_dropout_tiny", N=5, L=96, B=2, seed=5)
