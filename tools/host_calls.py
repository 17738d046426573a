"""Host time of every libstep_hip entry point during training steps (wraps step_amd._lib.call; no device wait inside the timed calls).
usage: python tools/host_calls.py [config] [steps] [extra bench args...]"""
import argparse
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1] if len(sys.argv) > 1 else "STEP_PEMS04"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
import bench  # noqa: E402
on, wgs = bench._prefetch_policy(["bench.py", "--config", name] + sys.argv[3:])
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if on else "2")
import torch  # noqa: E402
from step_amd import _lib  # noqa: E402


def main():
    args = argparse.Namespace(matmul="bf16", eval_dropout_off=False, no_shard=False, torch_optim=False, prefetch=False, no_prefetch="--no-prefetch" in sys.argv,
                              forward_only=False, resident_batches=False, encoder_workgroups=None, collectives="auto")
    dev = torch.device("cuda:0")
    sb = bench.StepBench(name, bench.CONFIGS[name], args, 1, 0, dev, None)
    for i in range(20):
        sb.train_step(i)
    torch.cuda.synchronize()
    acc = collections.defaultdict(lambda: [0, 0.0])
    orig = _lib.call

    def timed(fn, *a):
        t0 = time.perf_counter()
        orig(fn, *a)
        d = time.perf_counter() - t0
        acc[fn][0] += 1
        acc[fn][1] += d
    _lib.call = timed
    import step_amd.step_arch.step as S
    import step_amd.step_arch.tsformer as T
    import step_amd.optim as OP
    import step_amd.step_loss as SL
    for mod in (S, T, OP, SL):
        if hasattr(mod, "_lib"):
            mod._lib.call = timed
    t0 = time.perf_counter()
    for i in range(steps):
        sb.train_step(20 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tot = sum(v[1] for v in acc.values())
    print(f"{name}: host enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step (with device wait {1e3 * (t2 - t0) / steps:.3f}); inside libstep_hip calls {1e3 * tot / steps:.3f} ms/step, "
          f"python / torch around them {1e3 * ((t1 - t0) - tot) / steps:.3f} ms/step")
    for fn, (n, d) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {fn:36s} {n / steps:5.1f} calls/step  {1e6 * d / n:8.1f} us/call  {1e3 * d / steps:7.3f} ms/step")


if __name__ == "__main__":
    main()
