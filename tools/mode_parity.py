"""Matmul-mode parity at full size (GPU box): train the same STEP_PEMS04-shaped model from the same initial weights on
the same synthetic windows and noise seeds, once with matmul_precision="f32" and once with "bf16"; report the training-loss
curves (smoothed) and the horizon-12 MAE on held-out windows.  Prints one JSON object."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn


def run(mode, cfg, data, steps, seed):
    from step_amd.step_loss import step_loss_native as step_loss
    from step_amd.optim import FusedAdamClip
    torch.manual_seed(seed)
    dev = torch.device("cuda", 0)
    model = Bn.make_model(cfg, data).to(dev)
    model.train()
    model.matmul_precision = mode
    opt = FusedAdamClip(model, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8, max_norm=3.0)
    dser = torch.from_numpy(data).to(dev)
    rng = np.random.default_rng(99)
    Lh, B = cfg["L"], cfg["B"]
    T_train = cfg["T_train"]                             # same seed + same call order -> same dropout / Gumbel streams in both runs
    mean, std = 200.0, 150.0
    losses = []
    def batch(lo, hi):
        ts = rng.integers(lo, hi, size=B)
        return (torch.stack([dser[t - 12:t] for t in ts]), torch.stack([dser[t - Lh:t] for t in ts]), torch.stack([dser[t:t + 12] for t in ts]))
    for i in range(steps):
        hist, longh, fut = batch(Lh, T_train - 12)
        opt.zero_grad(set_to_none=True)
        pred, theta, knn, coef = model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=i, epoch=1)
        loss = step_loss(pred[..., :1] * std + mean, fut[..., :1] * std + mean, theta, knn, coef, null_val=0.0)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    model.eval()
    rng = np.random.default_rng(7)
    errs = []
    with torch.no_grad():
        for _ in range(8):
            hist, longh, fut = batch(max(T_train, Lh), cfg["T_all"] - 12)
            pred, _, _, _ = model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=0, epoch=1)
            errs.append(float(((pred[:, 11, :, 0] - fut[:, 11, :, 0]).abs() * std).mean()))
    return losses, float(np.mean(errs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--config", default="STEP_PEMS04")
    a = ap.parse_args()
    cfg = dict(Bn.CONFIGS[a.config])
    data = Bn.synth_series(cfg["T_all"], cfg["N"])
    out = {"config": a.config, "steps": a.steps}
    for mode in ("f32", "bf16"):
        losses, h12 = run(mode, cfg, data, a.steps, 0)
        k = max(1, a.steps // 10)
        out[mode] = {"loss_first": float(np.mean(losses[:k])), "loss_last": float(np.mean(losses[-k:])),
                     "loss_curve": [round(float(np.mean(losses[i:i + k])), 3) for i in range(0, a.steps, k)], "h12_mae_heldout": h12}
    out["h12_rel_diff"] = abs(out["bf16"]["h12_mae_heldout"] - out["f32"]["h12_mae_heldout"]) / out["f32"]["h12_mae_heldout"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
