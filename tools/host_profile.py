"""Where the HOST time of a training step goes (cProfile over the enqueue of N steps, no device wait inside).
usage: python tools/host_profile.py [config] [steps]   -> top functions by own time and by cumulative time"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "STEP_METR-LA"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    args = argparse.Namespace(matmul="bf16", eval_dropout_off=False, no_shard=False, torch_optim=False, prefetch=False, forward_only=False,
                              no_prefetch="--no-prefetch" in sys.argv, resident_batches=False, encoder_workgroups=None, collectives="auto")
    dev = torch.device("cuda:0")
    sb = bench.StepBench(name, bench.CONFIGS[name], args, 1, 0, dev, None)
    for i in range(20):
        sb.train_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        sb.train_step(20 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step, with device wait {1e3 * (t2 - t0) / steps:.3f} ms/step")
    # phases of a step, host time only
    ph = {"zero_grad": 0.0, "forward": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0}
    for i in range(steps):
        hist, longh, fut = sb.batch(1000 + i)
        a = time.perf_counter()
        sb.opt.zero_grad(set_to_none=True)
        b = time.perf_counter()
        pred, theta, knn, coef = sb.model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=i, epoch=1)
        c = time.perf_counter()
        loss = sb.step_loss(pred[..., :1] * sb.std + sb.mean, fut[..., :1] * sb.std + sb.mean, theta, knn, coef, null_val=0.0)
        d = time.perf_counter()
        loss.backward()
        e = time.perf_counter()
        sb.opt.step()
        f = time.perf_counter()
        for k, v in zip(ph, (b - a, c - b, d - c, e - d, f - e)):
            ph[k] += v
    torch.cuda.synchronize()
    print("host ms per step by phase:", {k: round(1e3 * v / steps, 3) for k, v in ph.items()})
    pr = cProfile.Profile()
    pr.enable()
    for i in range(steps):
        sb.train_step(200 + i)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(40)
        print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
