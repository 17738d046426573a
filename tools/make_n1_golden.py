"""Oracle side of the horizon-12 MAE parity test (tests/test_gpu_training_parity.py::test_h12_mae_parity): 200 free-running
optimizer steps of the CPU oracle on the mid-size problem of tests/train_problem.py (N=64 nodes, L=2016 = 168 tokens, batch 4,
Adam 2e-3 with a MultiStepLR-style decay, clip 3.0), repeated with round-off sized perturbations of its inputs, so that the
test can hold the native module to the oracle's MEAN with a band that is a multiple of the oracle's OWN run-to-run spread.
Writes tests/golden/n1_oracle.npz (a few hundred bytes).  ~2.5 minutes on 8 cores.

    python tools/make_n1_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import train_problem as TPb          # noqa: E402

CFG = dict(N=64, L=2016, T_train=1200, steps=200, B=4, k=10)


def main():
    torch.set_num_threads(8)
    prob = TPb.Problem(CFG["N"], CFG["L"], CFG["T_train"])
    sd = {k: v.detach().clone() for k, v in TPb.build_native(CFG["N"], CFG["L"], CFG["T_train"], prob.series, k=CFG["k"]).state_dict().items()}
    hidden = prob.oracle_hidden(sd, prob.train_t + prob.eval_t)
    schedule, noises = prob.schedule(CFG["steps"], CFG["B"]), prob.noises(CFG["steps"], CFG["B"])
    u_eval = torch.rand(len(prob.eval_t), CFG["N"] ** 2, 2, generator=torch.Generator().manual_seed(999))
    rows, first = [], None
    for pert in (0.0, 1e-6, 1e-5, 1e-4, 1e-3, 3e-3):
        losses, p = TPb.oracle_train(prob, sd, hidden, schedule, noises, k=CFG["k"], perturb=pert, lr_decay=True)
        h12, mae = TPb.oracle_eval(prob, p, hidden, u_eval, CFG["k"])
        rows.append((pert, h12, mae, float(np.mean(losses[-20:]))))
        first = losses[0] if first is None else first
        print("perturbation %g: horizon-12 MAE %.4f, all horizons %.4f, loss tail %.4f" % rows[-1], flush=True)
    r = np.array(rows)
    np.savez(os.path.join(ROOT, "tests", "golden", "n1_oracle.npz"), runs=r, first_loss=np.float64(first),
             cfg=np.array([CFG[k] for k in ("N", "L", "T_train", "steps", "B", "k")]))
    print("H12 mean %.4f sd %.2f %%; all horizons mean %.4f sd %.2f %%" % (r[:, 1].mean(), 100 * r[:, 1].std() / r[:, 1].mean(),
                                                                          r[:, 2].mean(), 100 * r[:, 2].std() / r[:, 2].mean()))


if __name__ == "__main__":
    main()
