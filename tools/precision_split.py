"""Where do the bf16-mode errors of the graph learner's small gradients come from?  One training step at a full-size config (dropout
off, fixed Gumbel noise), evaluated with exact-f32 contractions everywhere (the reference point: it matches the CPU oracle to ~1e-3,
tests/test_gpu_full_size.py) and with bf16 operands in ONE half of the step at a time -- the graph learner (conv / fc / stored
activations) or the GraphWaveNet (hops, 1x1s, adjacency gradients) -- and in both.  Prints per-tensor rel-L2 against the f32 run."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn          # noqa: E402


def grads(cfg, data, B, override, env=None):
    from oracle import step_oracle as O          # (loss only: a tool, not the product)
    for k, v in (env or {}).items():
        os.environ[k] = v
    N, L = cfg["N"], cfg["L"]
    torch.manual_seed(0)
    model = Bn.make_model(cfg, data).cuda()
    model.train()
    model.matmul_precision = "bf16" if override is None else "f32"
    model._precision_override = override
    model.backend.dropout = 0.0
    model.tsformer.dropout_p = 0.0
    model._noise_override = torch.rand(B, N * N, 2, generator=torch.Generator().manual_seed(5))
    d = torch.from_numpy(data)
    ts = [L + 17 + 301 * i for i in range(B)]
    hist = torch.stack([d[a - 12:a] for a in ts]).cuda(); fut = torch.stack([d[a:a + 12] for a in ts]).cuda(); longh = torch.stack([d[a - L:a] for a in ts]).cuda()
    pred, theta, knn, coef = model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=0, epoch=1)
    loss = O.step_loss(O.rescale(pred[..., [0]], 200.0, 150.0), O.rescale(fut[..., [0]], 200.0, 150.0), theta, knn, coef)
    loss.backward()
    torch.cuda.synchronize()
    for k in (env or {}):
        os.environ.pop(k)
    return {k: v.grad.detach().double().cpu() for k, v in model._trainable() if v.grad is not None}, float(loss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="STEP_PEMS04")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--env", default="", help="KEY=VALUE,... set for the bf16 runs only (A/B knobs of the library)")
    a = ap.parse_args()
    cfg = dict(Bn.CONFIGS[a.config])
    data = Bn.synth_series(cfg["T_all"], cfg["N"])
    env = dict(kv.split("=") for kv in a.env.split(",") if kv)
    ref, l0 = grads(cfg, data, a.batch, {"dgl": 0, "backend": 0})
    out = {"config": a.config, "loss_f32": l0}
    for name, ov in (("dgl_bf16", {"dgl": 1, "backend": 0}), ("backend_bf16", {"dgl": 0, "backend": 1}), ("both_bf16", {"dgl": 1, "backend": 1})):
        g, l = grads(cfg, data, a.batch, ov, env)
        errs = {}
        for k, r in ref.items():
            if float(r.abs().max()) < 1e-4:
                continue
            errs[k] = float(((g[k] - r) ** 2).sum().sqrt() / (r ** 2).sum().sqrt())
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:14]
        out[name] = {"loss": l, "worst": [(k, round(v, 4)) for k, v in worst]}
        print(name, "loss", l, "worst:", [(k, round(v, 4)) for k, v in worst], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
