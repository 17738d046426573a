"""CPU timing of the UNMODIFIED reference (kind "reference") in the build container: the reference's own STEP module
(/root/reference/step/step_arch, imported through the stubs of tools/make_golden.py), its step_loss on re-scaled outputs,
backward, clip_grad_norm_(3.0) and Adam(lr 2e-3, wd 1e-5) -- the full training step of step/STEP_PEMS04.py -- on synthetic
PEMS04-shaped data (N=307, L=4032, T_train=13599), batch 1, 2 warm-up + 5 timed steps, with 2 threads (what step/run.py:10
sets) and with all cores.  Writes profiles/r02_cpu_reference_baseline.json.  The GPU box has no /root/reference, so this
record is the "reference" companion of bench.py's in-run "port" baseline (the oracle timed on the GPU box's host).

    python tools/cpu_reference_baseline.py [--batch 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import make_golden as MG
    import bench
    ref = MG.import_reference()
    step_loss = ref[5]
    cfg = bench.CONFIGS["STEP_PEMS04"]
    N, L, Ttr = cfg["N"], cfg["L"], cfg["T_train"]
    data = bench.synth_series(cfg["T_all"], N)
    torch.manual_seed(0)
    model = MG.build_step(ref, N, L, Ttr, data, cfg["k"])       # the reference's classes; dropout stays on (train mode), as in training
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8)
    d = torch.from_numpy(data)
    rng = np.random.default_rng(0)

    def one_step():
        ts = rng.integers(L, cfg["T_all"] - 12, size=args.batch)
        hist = torch.stack([d[t - 12:t] for t in ts]); fut = torch.stack([d[t:t + 12] for t in ts]); longh = torch.stack([d[t - L:t] for t in ts])
        t0 = time.perf_counter()
        opt.zero_grad()
        pred, theta, knn, coef = model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=0, epoch=1)
        loss = step_loss(pred[..., [0]] * 150.0 + 200.0, fut[..., [0]] * 150.0 + 200.0, theta, knn, coef, null_val=0.0)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=3.0)
        opt.step()
        return time.perf_counter() - t0
    out = {"kind": "reference", "what": "unmodified /root/reference step_arch.STEP + step_loss + backward + clip_grad_norm_ + Adam, torch "
           + torch.__version__ + " CPU fp32, synthetic STEP_PEMS04 shapes (N=307, L=4032, T_train=13599)", "batch": args.batch,
           "host": f"build container, {os.cpu_count()} cores", "runs": []}
    for threads in (2, os.cpu_count()):
        torch.set_num_threads(threads)
        for _ in range(args.warmup):
            one_step()
        ts = [one_step() for _ in range(args.steps)]
        out["runs"].append({"cores": threads, "value": args.batch / float(np.median(ts)), "unit": "windows/s", "step_s": [round(t, 2) for t in ts],
                            "sample": f"{args.warmup} warm-up + {args.steps} timed training steps of {args.batch} window(s)"})
        print(out["runs"][-1], flush=True)
    with open(os.path.join(ROOT, "profiles", "r02_cpu_reference_baseline.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
