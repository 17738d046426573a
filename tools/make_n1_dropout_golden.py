"""Oracle side of the horizon-12 MAE parity test WITH DROPOUT ON (tests/test_gpu_training_parity.py::test_h12_mae_parity_dropout_on):
the configuration bench.py times trains with the frozen TSFormer in train mode -- dropout 0.1 at its 17 sites
(positional_encoding.py:32, transformer_layers.py:10) -- and F.dropout(0.3) after every gcn (graphwavenet/model.py:47).
Here the CPU oracle runs the same 200 optimizer steps as tools/make_n1_golden.py, but re-encodes every minibatch with
INDEPENDENT Bernoulli keep-masks (torch.bernoulli, a different generator seed per run), i.e. with the i.i.d. dropout of the
reference, not with the device's keep-mask pool.  The test then compares the MEAN held-out horizon-12 MAE of several native
runs (pool masks, different seeds) with the mean of these runs.  Writes tests/golden/n1_oracle_dropout.npz.  ~2 min per run on 8 cores.

    python tools/make_n1_dropout_golden.py [n_runs=4]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import step_oracle as O              # noqa: E402
from tests import train_problem as TPb          # noqa: E402

CFG = dict(N=64, L=2016, T_train=1200, steps=200, B=4, k=10)
KEEP_ENC, KEEP_GCN = 0.9, 0.7


def encoder_masks(gen, S, P, depth=4):
    b = lambda *shape: torch.bernoulli(torch.full(shape, KEEP_ENC), generator=gen)
    return {"pos": b(S, P, 96), "layers": [{"attn": b(S, 4, P, P), "drop1": b(S, P, 96), "ffn": b(S, P, 384), "drop2": b(S, P, 96)}
                                           for _ in range(depth)]}


def gcn_masks(gen, B, N):
    """model.py:47 for layers 0..6 (the last layer's gcn is dead code): shape of layer i's gcn output, survivors scaled."""
    out, T = [], 13
    for i in range(8):
        T -= 2 ** (i % 2)
        out.append(torch.bernoulli(torch.full((B, 32, N, T), KEEP_GCN), generator=gen) / KEEP_GCN)
    return out


def train_run(prob, sd, schedule, noises, seed, k):
    gen = torch.Generator().manual_seed(seed)
    p = TPb.trainable(sd)
    train = [v for v in p.values() if v.requires_grad]
    opt = torch.optim.Adam(train, lr=TPb.LR0, weight_decay=1e-5, eps=1e-8)
    P = prob.L // 12
    losses = []
    for it, ts in enumerate(schedule):
        for grp in opt.param_groups:
            grp["lr"] = TPb.lr_at(it, True)
        hist, longh, fut = prob.batch(ts)
        B = len(ts)
        with torch.no_grad():
            hid = O.tsformer_encode(longh[..., 0], sd, drop=encoder_masks(gen, B * prob.N, P), keep=KEEP_ENC)
        stats = {}
        opt.zero_grad(set_to_none=True)
        pred, theta, knn, coef = O.step_forward(hist, torch.zeros(B, prob.L, prob.N, 1), prob.data[:prob.T_train, :, 0], p, noises[it], k, 1,
                                                training=True, stats=stats, hidden=hid, hidden_last=hid[:, :, -1, :],
                                                drop_masks=gcn_masks(gen, B, prob.N))
        loss = O.step_loss(O.rescale(pred, prob.mean, prob.std), O.rescale(fut[..., [0]], prob.mean, prob.std), theta, knn, coef)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([q for q in train if q.grad is not None], 3.0)
        opt.step()
        TPb.update_running_stats(p, stats)
        losses.append(float(loss.detach()))
    return losses, p


def main():
    n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(8)
    prob = TPb.Problem(CFG["N"], CFG["L"], CFG["T_train"])
    sd = {k: v.detach().clone() for k, v in TPb.build_native(CFG["N"], CFG["L"], CFG["T_train"], prob.series, k=CFG["k"]).state_dict().items()}
    hidden_eval = prob.oracle_hidden(sd, prob.eval_t)                 # evaluation runs in eval mode: no dropout
    schedule, noises = prob.schedule(CFG["steps"], CFG["B"]), prob.noises(CFG["steps"], CFG["B"])
    u_eval = torch.rand(len(prob.eval_t), CFG["N"] ** 2, 2, generator=torch.Generator().manual_seed(999))
    rows = []
    for r in range(n_runs):
        losses, p = train_run(prob, sd, schedule, noises, 4242 + r, CFG["k"])
        h12, mae = TPb.oracle_eval(prob, p, hidden_eval, u_eval, CFG["k"])
        rows.append((4242 + r, h12, mae, float(np.mean(losses[-20:])), losses[0]))
        print("seed %d: horizon-12 MAE %.4f, all horizons %.4f, loss tail %.4f, first loss %.4f" % rows[-1], flush=True)
    r = np.array(rows)
    np.savez(os.path.join(ROOT, "tests", "golden", "n1_oracle_dropout.npz"), runs=r,
             cfg=np.array([CFG[k] for k in ("N", "L", "T_train", "steps", "B", "k")]), keep=np.array([KEEP_ENC, KEEP_GCN]))
    print("H12 mean %.4f sd %.2f %%; all horizons mean %.4f sd %.2f %%" % (r[:, 1].mean(), 100 * r[:, 1].std() / r[:, 1].mean(),
                                                                          r[:, 2].mean(), 100 * r[:, 2].std() / r[:, 2].mean()))


if __name__ == "__main__":
    main()
