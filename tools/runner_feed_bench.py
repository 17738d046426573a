"""The training loop of the REFERENCE's runner (easytorch Runner.train -> BaseTimeSeriesForecastingRunner.train_iters -> STEPRunner.forward,
unmodified sources from /root/reference or oracle/_ref/reference.tar.gz; tests/_shims stands in for easytorch) around step_amd.STEP at
config C2 (STEP_PEMS04 shape, batch 8, bf16 mode, dropout on), timed.  One JSON line.  What bench.py quotes as `other_input_feed.runner*`.

    python tools/runner_feed_bench.py --runner native --dataset device [--loss native] [--iters 60] [--profile]

--runner native     CFG.RUNNER = step_amd.runner.native_runner(STEPRunner)        (reference: the config's STEPRunner as it is)
--dataset device    CFG.DATASET_CLS = step_amd.runner.DeviceForecastingDataset     (host: the reference's ForecastingDataset)
--loss native       CFG.TRAIN.LOSS = step_amd.step_loss.step_loss_native            (reference: the config's step_loss)
"""
import argparse
import cProfile
import json
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")          # like bench.py when the frozen branch is prefetched (four streams carry work)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runner", default="native", choices=["native", "reference"])
    ap.add_argument("--dataset", default="device", choices=["device", "host"])
    ap.add_argument("--loss", default="reference", choices=["native", "reference"])
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    import torch
    from tests import dropin_common as DC
    from oracle.reference_loader import reference_root
    if reference_root() is None:
        print(json.dumps({"error": "reference sources not staged (tools/stage_reference.sh)"}))
        return
    from step_amd.runner import DeviceForecastingDataset, native_runner
    with tempfile.TemporaryDirectory() as root, DC.Workspace(root, "PEMS04", n_train=a.batch * (a.iters + a.warmup + 2), full_history_only=False) as ws:
        cfg = ws.config(batch=a.batch, dropout=True)
        cfg.TRAIN.DATA.SHUFFLE = True
        if a.runner == "native":
            cfg.RUNNER = native_runner(cfg.RUNNER)
        if a.dataset == "device":
            cfg.DATASET_CLS = DeviceForecastingDataset
        if a.loss == "native":
            from step_amd.step_loss import step_loss_native
            cfg.TRAIN.LOSS = step_loss_native
        torch.manual_seed(0)
        runner = cfg.RUNNER(cfg)
        runner.model.matmul_precision = "bf16"
        t = {}

        def clock(module, args, kwargs):
            if kwargs["batch_seen"] == a.warmup:
                torch.cuda.synchronize()
                t["t0"] = time.perf_counter()
                if a.profile:
                    t["prof"] = cProfile.Profile()
                    t["prof"].enable()
        runner.model.register_forward_pre_hook(clock, with_kwargs=True)
        losses = runner.train(cfg, max_iters=a.warmup + a.iters)
        if a.profile:
            t["prof"].disable()
        host = time.perf_counter() - t["t0"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t["t0"]
        out = {"runner": a.runner, "dataset": a.dataset, "loss": a.loss, "ms_per_step": 1e3 * dt / a.iters, "windows_per_s": a.batch * a.iters / dt,
               "host_enqueue_ms_per_step": 1e3 * host / a.iters, "iters": a.iters, "batch": a.batch, "final_loss": losses[-1],
               "optimizer": type(runner.optim).__name__, "encoder_workgroups": int(runner.model.tsformer.encoder_workgroups)}
        print(json.dumps(out))
        if a.profile:
            pstats.Stats(t["prof"], stream=sys.stderr).sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
