"""Make the "pre-trained TSFormer_<DS>.pt" that BASELINE.json's config 2 loads (step/step_arch/step.py:27-35): the real blob is
absent (reference .gitignore), so it is produced here by the repository's OWN pre-training path -- K native masked-pre-training
steps (step_amd.TSFormer(mode="pre-train"), the reference's step/TSFormer_PEMS04.py settings: batch 6, Adam lr 1e-3, betas
(0.9, 0.95), clip 5.0, masked MAE on rescaled values) on the synthetic series of SURVEY.md 8d -- and saved in the reference's
checkpoint format ({"model_state_dict": TSFormer(mode="pre-train").state_dict()}).  Deterministic in (seed, steps); ~10 s on an
MI355X for 300 steps, so bench.py regenerates it outside its timed region instead of committing a blob.

    python tools/pretrain_checkpoint.py --out tsformer_ckpt/TSFormer_PEMS04.pt --steps 300
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TSFORMER_ARGS = dict(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, mask_ratio=0.75,
                     encoder_depth=4, decoder_depth=1)


def pretrain(data, L, steps=300, batch=6, device="cuda", matmul="bf16", seed=0, log=None):
    """data: float32 [T, N, C] (channel 0 is the signal).  Returns (state_dict on the CPU, list of losses)."""
    from step_amd import TSFormer
    from step_amd.step_loss import masked_mae
    torch.manual_seed(seed)
    random.seed(seed)                                   # MaskGenerator shuffles with python's generator (mask.py:15-28)
    model = TSFormer(**TSFORMER_ARGS, num_token=L / 12, mode="pre-train").to(device)
    model.train()
    model.matmul_precision = matmul
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=0.001, weight_decay=0, eps=1.0e-8, betas=(0.9, 0.95))       # TSFormer_PEMS04.py:55-61
    dser = torch.as_tensor(data[:, :, :1]).to(device)
    rng = np.random.default_rng(seed + 17)
    losses = []
    for it in range(steps):
        ts = rng.integers(L, dser.shape[0] - 12, size=batch)
        x = torch.stack([dser[t - L:t] for t in ts])
        opt.zero_grad(set_to_none=True)
        recon, label = model(history_data=x, future_data=None, batch_seen=it, epoch=1)
        loss = masked_mae(recon * 150.0 + 200.0, label * 150.0 + 200.0, 0.0)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, max_norm=5.0)                                      # TSFormer_PEMS04.py:74-76
        opt.step()
        if log is not None or it in (0, steps - 1):
            losses.append(float(loss.detach()))
            if log is not None:
                log(it, losses[-1])
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, losses


def attention_sharpness(sd, series, depth=4):
    """Mean over (layer, head, query) of the largest attention probability of the oracle's dropout-free forward on a few
    sequences [S, L] -- 1 / P for uniform attention, 1 for one-hot (how far from an initialisation the weights are)."""
    from oracle import step_oracle as O          # tools/ and bench.py's reporting only, never the product path
    import math
    p = {"tsformer." + k: v for k, v in sd.items()}
    S, L = series.shape
    h = O.patch_embed(series, p) + p["tsformer.positional_encoding.position_embedding"][:L // 12]
    h = h * math.sqrt(96)
    tops = []
    for i in range(depth):
        pre = f"tsformer.encoder.transformer_encoder.layers.{i}."
        qkv = h @ p[pre + "self_attn.in_proj_weight"].T + p[pre + "self_attn.in_proj_bias"]
        q, k, _ = qkv.split(96, dim=-1)
        P = h.shape[1]
        q = q.reshape(S, P, 4, 24).transpose(1, 2)
        k = k.reshape(S, P, 4, 24).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(24), dim=-1)
        tops.append(float(att.amax(-1).mean()))
        h = O.encoder_layer(h, p, pre)
    return tops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="tsformer_ckpt/TSFormer_PEMS04.pt")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--config", default="STEP_PEMS04")
    args = ap.parse_args()
    import bench
    cfg = bench.CONFIGS[args.config]
    data = bench.synth_series(cfg["T_all"], cfg["N"])
    sd, losses = pretrain(data, cfg["L"], args.steps, args.batch, log=lambda it, l: print(f"step {it}: loss {l:.3f}", flush=True) if it % 20 == 0 else None)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    torch.save({"model_state_dict": sd}, args.out)
    print("saved", args.out, "first/last loss", losses[0], losses[-1])


if __name__ == "__main__":
    main()
