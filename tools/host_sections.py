"""Host time of a training step by section: model.forward (of which library calls), loss, run_backward (of which the native backward's
python body, of which library calls), optimizer -- all without device waits.  usage: python tools/host_sections.py [config] [steps]"""
import argparse, collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1] if len(sys.argv) > 1 else "STEP_PEMS04"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
import bench
on, wgs = bench._prefetch_policy(["bench.py", "--config", name] + sys.argv[3:])
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if on else "2")
import torch
from step_amd import _lib
import step_amd.step_arch.step as S

args = argparse.Namespace(matmul="bf16", eval_dropout_off=False, no_shard=False, torch_optim=False, prefetch=False, no_prefetch="--no-prefetch" in sys.argv,
                          forward_only=False, resident_batches=False, encoder_workgroups=None, collectives="auto")
sb = bench.StepBench(name, bench.CONFIGS[name], args, 1, 0, torch.device("cuda:0"), None)
for i in range(20):
    sb.train_step(i)
torch.cuda.synchronize()
acc = collections.defaultdict(float)
lib_t = [0.0]
orig_call = _lib.call
def timed_call(fn, *a):
    t0 = time.perf_counter(); orig_call(fn, *a); lib_t[0] += time.perf_counter() - t0
_lib.call = timed_call
orig_bwd = S._StepFunction.backward
def bwd(ctx, *g):
    t0 = time.perf_counter(); l0 = lib_t[0]
    out = orig_bwd(ctx, *g)
    acc["backward body (python + library)"] += time.perf_counter() - t0
    acc["  library calls inside it"] += lib_t[0] - l0
    return out
S._StepFunction.backward = staticmethod(bwd)
for i in range(steps):
    a = time.perf_counter(); l0 = lib_t[0]
    hist, longh, fut = sb.batch(100 + i)
    sb.opt.zero_grad(set_to_none=True)
    b = time.perf_counter()
    pred, theta, knn, coef = sb.model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=i, epoch=1)
    c = time.perf_counter(); l1 = lib_t[0]
    nxt = sb.stage(101 + i)
    if sb.prefetch:
        sb.model.prefetch(nxt[1])
    d = time.perf_counter(); l2 = lib_t[0]
    loss = sb.step_loss(pred[..., :1], fut[..., :1], theta, knn, coef, null_val=0.0, rescale=(sb.mean, sb.std))
    e = time.perf_counter(); l3 = lib_t[0]
    loss.backward()
    f = time.perf_counter()
    sb.opt.step()
    g = time.perf_counter()
    acc["batch + zero_grad"] += b - a
    acc["model.forward"] += c - b; acc["  library calls inside forward"] += l1 - l0
    acc["loader stage + prefetch"] += d - c; acc["  library calls inside stage + prefetch"] += l2 - l1
    acc["loss"] += e - d; acc["  library calls inside loss"] += l3 - l2
    acc["loss.backward() (engine + body)"] += f - e
    acc["optimizer.step"] += g - f
    acc["step total"] += g - a
torch.cuda.synchronize()
print(name, "host ms per step (no device wait):")
for k, v in acc.items():
    print(f"  {k:44s} {1e3 * v / steps:7.3f}")
