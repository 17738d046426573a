"""bench.py -- STEP training windows/s on MI355X (BASELINE.json metric, config C2 = STEP_PEMS04).

One "step" = one full training step of the native STEP model on one synthetic minibatch that is already
resident in HBM: TSFormer encoder forward (frozen) + kNN prior + DiscreteGraphLearning forward/backward +
GraphWaveNet forward/backward + step_loss + gradient all-reduce (N>1) + clip_grad_norm_(3.0) + Adam.
Prints ONE JSON line (rank 0).  Launch:  python bench.py [--gpus N --steps K --warmup W]
(for N>1:  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (nodes, long-history length L, train_length, total series length, batch per GPU)
    "STEP_PEMS04": dict(N=307, L=288 * 7 * 2, T_train=13599, T_all=16992, B=8, k=10),
    "STEP_PEMS07": dict(N=883, L=288 * 7, T_train=16513, T_all=28224, B=4, k=10),
    "STEP_METR-LA": dict(N=207, L=288 * 7, T_train=23990, T_all=34272, B=2, k=10, train_ratio=0.7),
    "SYNTH_4096": dict(N=4096, L=288 * 7, T_train=16513, T_all=28224, B=1, k=10),      # BASELINE config 5 (N-scaling stress)
    # BASELINE config 3: masked pre-training of TSFormer (reference step/TSFormer_PEMS-BAY.py: B=16... batch per GPU)
    "TSFormer_PEMS-BAY": dict(N=325, L=288 * 7, T_train=36482, T_all=52116, B=16, k=0, pretrain=True),
}


def synth_series(T, N, seed=0):
    """ch0 z-scored signal sin(2 pi t/288 + phi_n) + 0.5 N(0,1); ch1 time of day; ch2 day of week."""
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float32)[:, None]
    phase = rng.uniform(0, 2 * np.pi, (1, N)).astype(np.float32)
    ch0 = np.sin(2 * np.pi * t / 288.0 + phase) + 0.5 * rng.standard_normal((T, N), dtype=np.float32)
    ch1 = np.broadcast_to((t % 288) / 288.0, (T, N))
    ch2 = np.broadcast_to((t // 288) % 7, (T, N))
    return np.stack([ch0, ch1, ch2], -1).astype(np.float32)


def pmc_traffic(config, B):
    """HBM bytes per encoder launch from the committed rocprofv3 PMC passes (profiles/encoder_pmc.json, produced by
    tools/pmc_enc_ab.sh: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md "HBM").  A counter run cannot be nested inside this process, so the number is NOT a measurement of
    this run: it is looked up for the same kernel / config / batch and labelled "static": true; None when no matching
    measurement is committed."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "encoder_pmc.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        ent = rec.get(f"{config}:B{B}")
        return None if ent is None else {"hbm_bytes_per_launch": ent["read_bytes"] + ent["write_bytes"], "read_bytes": ent["read_bytes"],
                                          "write_bytes": ent["write_bytes"], "static": True, "source": ent["source"]}
    except (OSError, ValueError, KeyError):
        return None


def make_model(cfg, data):
    from step_amd import STEP
    N, L = cfg["N"], cfg["L"]
    targs = dict(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, num_token=L / 12,
                 mask_ratio=0.75, encoder_depth=4, decoder_depth=1, mode="forecasting")
    bargs = dict(num_nodes=N, support_len=2, dropout=0.3, gcn_bool=True, addaptadj=True, aptinit=None, in_dim=2, out_dim=12,
                 residual_channels=32, dilation_channels=32, skip_channels=256, end_channels=512, kernel_size=2, blocks=4, layers=2)
    dargs = dict(dataset_name="SYNTH", k=cfg["k"], input_seq_len=12, output_seq_len=12, data=data, train_length=cfg["T_train"],
                 tsformer_tokens=L // 12)
    torch.manual_seed(0)
    return STEP("SYNTH", None, targs, bargs, dargs)


def encoder_flops(cfg, B):
    """Algorithmic FLOPs of one encoder launch (SURVEY.md 8d, row T4 + T1): per window
    N*P*(4*(221184 + 384*P) + 2304)."""
    P = cfg["L"] // 12
    return B * cfg["N"] * P * (4 * (221184 + 384 * P) + 2304)


def cpu_baseline(cfg, data, seed=0):
    """The CPU oracle (restatement of the reference algorithm, kind "port") timed on this host's cores on a bounded sample of
    the same workload: full training steps (forward + step_loss + backward + clip_grad_norm_ + Adam) of ONE window each --
    two warm-up and five timed steps on all cores (median reported), then one step with two threads, the thread count the
    reference ships with (step/run.py:10 torch.set_num_threads(2)).  The reference itself cannot run on the GPU box
    (no /root/reference there); its own CPU timing, taken in the build container, is profiles/r02_cpu_reference_baseline.json."""
    from oracle import step_oracle as O
    torch.manual_seed(seed)
    N, L, Ttr = cfg["N"], cfg["L"], cfg["T_train"]
    model = make_model(cfg, data)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and not k.startswith("tsformer.") and "running_" not in k:
            v.requires_grad_(True)
    train = [v for v in p.values() if v.requires_grad]
    opt = torch.optim.Adam(train, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8)
    d = torch.from_numpy(data)

    def one_step(i):
        t = L + 17 + 301 * i
        hist, fut, longh = d[t - 12:t][None], d[t:t + 12][None], d[t - L:t][None]
        u = torch.rand(1, N * N, 2)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        pred, theta, knn, coef = O.step_forward(hist, longh[..., [0]], d[:Ttr, :, 0], p, u, cfg["k"], 1, training=True)
        loss = O.step_loss(O.rescale(pred, 200.0, 150.0), O.rescale(fut[..., [0]], 200.0, 150.0), theta, knn, coef)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([q for q in train if q.grad is not None], 3.0)
        opt.step()
        return time.perf_counter() - t0
    cores = min(os.cpu_count() or 1, 32)          # torch CPU ops stop scaling (and oversubscribe) beyond a few tens of threads
    torch.set_num_threads(cores)
    one_step(0); one_step(1)
    ts = sorted(one_step(2 + i) for i in range(5))
    torch.set_num_threads(2)
    t2 = one_step(7)
    torch.set_num_threads(cores)
    med = ts[len(ts) // 2]
    return {"value": 1.0 / med, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"full training steps (fwd+loss+bwd+clip+Adam) of 1 window of the same workload, torch CPU fp32 oracle: 2 warm-up + 5 timed "
                      f"on {cores} threads ({ts[0]:.1f} .. {med:.1f} .. {ts[-1]:.1f} s, median reported), 1 step on 2 threads ({t2:.1f} s)",
            "two_threads": {"value": 1.0 / t2, "unit": "windows/s", "cores": 2}}


def average_grads(params, world):
    """Data-parallel exchange step of the pre-training bench: one all-reduce (mean) over the flattened gradients
    (2.67 MB at C3), the job DistributedDataParallel does for easytorch in the reference."""
    if world <= 1:
        return
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat.mul_(1.0 / world)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def pretrain_main(args, cfg, world, rank, dev):
    """Config C3: one step = TSFormer masked pre-training forward + masked_mae + backward + clip(5.0) + Adam
    (reference step/TSFormer_PEMS-BAY.py:52-72), windows/s = sequences of L steps x N nodes per second."""
    import random
    from step_amd import TSFormer
    from step_amd.step_loss import masked_mae
    N, Lh, B = cfg["N"], cfg["L"], cfg["B"]
    data = synth_series(cfg["T_all"], N)
    torch.manual_seed(0)
    random.seed(0)
    model = TSFormer(12, 1, 96, 4, 4, 0.1, Lh / 12, 0.75, 4, 1, mode="pre-train").to(dev)
    model.train()
    model.matmul_precision = args.matmul
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=0.001, weight_decay=0, eps=1.0e-8, betas=(0.9, 0.95))
    dser = torch.from_numpy(data[:, :, :1]).to(dev)
    rng = np.random.default_rng(99 + rank)
    batches = []
    for _ in range(4):
        ts = rng.integers(Lh, cfg["T_all"] - 12, size=B)
        batches.append(torch.stack([dser[t - Lh:t] for t in ts]))

    def step(i):
        opt.zero_grad(set_to_none=True)
        recon, label = model(history_data=batches[i % len(batches)], future_data=None, batch_seen=i, epoch=1)
        loss = masked_mae(recon * 150.0 + 200.0, label * 150.0 + 200.0, 0.0)
        loss.backward()
        average_grads(params, world)
        torch.nn.utils.clip_grad_norm_(params, max_norm=5.0)
        opt.step()
        return loss

    for i in range(args.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tdt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tdt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tdt)
    if rank == 0:
        print(json.dumps({"metric": "TSFormer masked pre-training windows/s (config C3)", "value": B * world * args.steps / dt,
                          "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": args.matmul, "data": "synthetic",
                          "config": {"workload": f"TSFormer_PEMS-BAY pre-train: N={N}, L={Lh} (P={Lh // 12}, 42 unmasked), batch {B}/GPU, "
                                                 "fwd+bwd" + ("+grad all-reduce" if world > 1 else "") + "+clip+Adam, unfused path, "
                                                 + ("bf16 operands / f32 accumulate GEMMs" if args.matmul == "bf16" else "exact-f32 GEMMs"),
                                     "global_batch": B * world, "parallelism": f"dp{world}",
                                     "final_loss": float(loss.detach())}}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)          # >= 20 warm-up, >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="STEP_PEMS04", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the reference config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-loader-figure", action="store_true", help="skip the second timed loop through the device-resident window loader")
    ap.add_argument("--eval-dropout-off", action="store_true", help="disable dropout (parity runs)")
    ap.add_argument("--matmul", default="bf16", choices=["bf16", "f32"], help="operand precision of the GraphWaveNet / DGL contractions")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--forward-only", action="store_true",
                    help="validation / test path (SURVEY 8f-4): eval-mode forward + metric under no_grad, no backward / optimizer")
    ap.add_argument("--no-shard", action="store_true", help="--gpus > 1: keep the whole graph learner (and fc.weight) on every rank")
    ap.add_argument("--torch-optim", action="store_true", help="torch clip_grad_norm_ + torch.optim.Adam instead of the fused kernel")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local = local % max(torch.cuda.device_count(), 1)          # (test rigs with fewer devices than ranks share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)         # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)                  # e.g. gloo: exercises the data-parallel path on one device
    from step_amd.step_loss import step_loss_native as step_loss
    import step_amd._lib as L
    L.lib()

    N, Lh, B = cfg["N"], cfg["L"], cfg["B"]
    if cfg.get("pretrain"):
        return pretrain_main(args, cfg, world, rank, dev)
    data = synth_series(cfg["T_all"], N)
    model = make_model(cfg, data).to(dev)
    model.train()
    model.matmul_precision = args.matmul
    if args.eval_dropout_off:
        model.backend.dropout = 0.0
        model.tsformer.dropout_p = 0.0
    if world > 1:
        # each rank keeps one time slice of the graph learner's global branch and of fc.weight (bf16 mode)
        model.enable_native_data_parallel(shard_graph_learner=args.matmul == "bf16" and not args.no_shard and not args.torch_optim)
    sharded = model.discrete_graph_learning._shard is not None
    params = [p for p in model.parameters() if p.requires_grad]
    if args.torch_optim:
        opt = torch.optim.Adam(params, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8)    # step/STEP_PEMS04.py:90-96
    else:
        from step_amd.optim import FusedAdamClip
        opt = FusedAdamClip(model, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8, max_norm=3.0)   # same rule, one fused pass
    dser = torch.from_numpy(data).to(dev)
    rng = np.random.default_rng(1234 + rank)
    nb = args.steps + args.warmup
    batches = []
    for _ in range(min(nb, 8)):                       # resident input batches, cycled (119 MB each at C2)
        ts = rng.integers(Lh, cfg["T_all"] - 12, size=B)
        hist = torch.stack([dser[t - 12:t] for t in ts])
        fut = torch.stack([dser[t:t + 12] for t in ts])
        longh = torch.stack([dser[t - Lh:t] for t in ts])
        batches.append((hist, longh, fut))
    mean, std = 200.0, 150.0

    def eval_step(i):
        hist, longh, fut = batches[i % len(batches)]
        with torch.no_grad():
            pred, theta, knn, coef = model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=i, epoch=None)
            from step_amd.step_loss import masked_mae
            return masked_mae(pred[..., :1] * std + mean, fut[..., :1] * std + mean, 0.0)       # base_tsf_runner.py:257-318

    def step(i, epoch=1):
        if args.forward_only:
            return eval_step(i)
        hist, longh, fut = batches[i % len(batches)]
        opt.zero_grad(set_to_none=True)
        pred, theta, knn, coef = model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=i, epoch=epoch)
        # target-feature selection + inverse scaling, as the runner does (step_runner.py:86-92); slices, not index kernels
        loss = step_loss(pred[..., :1] * std + mean, fut[..., :1] * std + mean, theta, knn, coef, null_val=0.0)
        loss.backward()
        if args.torch_optim:
            torch.nn.utils.clip_grad_norm_(params, max_norm=3.0)                    # STEP_PEMS04.py:103-105
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.forward_only:
        model.eval()
    for i in range(args.warmup):
        step(i)
    model.tsformer._events = []
    if world > 1:
        model._reduce_wait_ms = []
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step(args.warmup + i)
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        model.collect_reduce_waits()
    per_step = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    if world > 1:
        tdt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
        dt = float(tdt)
    ev = model.tsformer._events
    enc_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")
    model.tsformer._events = None
    # In the timed steps the graph learner and the WaveNet layers run on a second stream next to the encoder (step.py), so the
    # encoder's launch-to-completion time above includes the compute units it lends them.  A few extra (untimed) steps with the
    # overlap off give the kernel's own duration.
    enc_alone_ms = None
    if getattr(model, "overlap_streams", False) and not args.forward_only:
        model.overlap_streams = False
        model.tsformer._events = []
        for i in range(6):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        ev2 = model.tsformer._events[1:]
        enc_alone_ms = float(np.mean([a.elapsed_time(b) for a, b in ev2])) if ev2 else None
        model.tsformer._events = None
        model.overlap_streams = True
    # ---- second figure: the same step fed by the index-only loader over the device-resident series, forecast
    # origins drawn over the WHOLE training split -- including the windows that start before a full long history exists
    # (all-zero history, forecasting_dataset.py:66-67: 39 % of PEMS04's training windows) -- gather launches inside the timed region
    loader_fig = None
    if not args.forward_only and not args.no_loader_figure:
        from step_amd.step_arch.step import DeviceWindowLoader
        loader = DeviceWindowLoader(dser, Lh)
        n_train = int((cfg["T_all"] - 23) * cfg.get("train_ratio", 0.6))
        lrng = np.random.default_rng(4321 + rank)
        nl = max(args.steps // 2, 10)
        origins = [torch.from_numpy(lrng.integers(12, 12 + n_train, size=B)).to(dev) for _ in range(nl + 3)]
        zero_frac = float(np.mean([float((o < Lh).float().mean()) for o in origins[3:]]))

        def loader_step(i):
            hist, ref, fut = loader.batch(origins[i])
            opt.zero_grad(set_to_none=True)
            pred, theta, knn, coef = model(history_data=hist, long_history_data=ref, future_data=None, batch_seen=i, epoch=1)
            loss = step_loss(pred[..., :1] * std + mean, fut[..., :1] * std + mean, theta, knn, coef, null_val=0.0)
            loss.backward()
            if args.torch_optim:
                torch.nn.utils.clip_grad_norm_(params, max_norm=3.0)
            opt.step()
        for i in range(3):
            loader_step(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(nl):
            loader_step(3 + i)
        barrier()
        dtl = time.perf_counter() - t1
        if world > 1:
            tdl = torch.tensor([dtl], device=dev, dtype=torch.float64)
            dist.all_reduce(tdl, op=dist.ReduceOp.MAX)
            dtl = float(tdl)
        loader_fig = {"value": B * world * nl / dtl, "unit": "windows/s", "ms_per_step": dtl / nl * 1e3, "steps": nl,
                      "zero_history_fraction": zero_frac,
                      "what": "same training step, windows gathered on the device from the resident series by forecast origin "
                              "(step_gather_windows, LongHistoryRef), origins uniform over the training split"}
    # ---- data-parallel exchange: the flat-gradient all-reduce alone (isolated) and the part of it the step does not hide
    comm = None
    if world > 1 and not args.forward_only:
        lay = model._grad_layout()
        fo, fn, _ = lay["items"]["dgl.fc_w"]
        buf = torch.zeros(lay["total"], device=dev)
        reps = 5
        for _ in range(2):
            dist.all_reduce(buf[:fo])          # (with time slices the fc chunk has a different length on every rank and is never reduced)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            if not sharded:
                dist.all_reduce(buf[fo:fo + fn])
            dist.all_reduce(buf[:fo])
        e1.record()
        torch.cuda.synchronize()
        iso = e0.elapsed_time(e1) / reps
        exposed = float(np.mean(model._reduce_wait_ms)) if model._reduce_wait_ms else float("nan")
        t_ex = torch.tensor([iso, exposed], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t_ex) for _ in range(world)]
        dist.all_gather(gathered, t_ex)
        comm = {"allreduce_bytes": int((fo if sharded else lay["total"]) * 4), "chunks": [int(fo * 4)] if sharded else [int(fn * 4), int(fo * 4)],
                "graph_learner_time_slices": bool(sharded),
                "per_rank_allreduce_ms_isolated": [float(g[0]) for g in gathered],
                "per_rank_exposed_wait_ms": [float(g[1]) for g in gathered],
                "overlap_fraction": [float(1.0 - g[1] / g[0]) if float(g[0]) > 0 else None for g in gathered],
                "what": "isolated = the two chunked all-reduces of the flat gradient alone; exposed = time the compute stream waits "
                        "for them at the end of backward (events around the waits), averaged over the timed steps"}
    if rank == 0:
        flops = encoder_flops(cfg, B)
        ach = flops / (enc_ms * 1e-3) / 1e12
        out = {
            "metric": ("validation windows/sec (eval-mode forward + masked MAE)" if args.forward_only else
                       "training windows/sec on PEMS04, horizon-12 MAE parity, 1/2/4/8 MI355X"),
            "value": B * world * args.steps / dt, "unit": "windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.matmul == "bf16" else "bf16 (TSFormer) + f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: N={N} nodes, long history L={Lh} (P={Lh // 12} patches), 12->12, "
                                   f"train series T={cfg['T_train']}, batch {B}/GPU, random-init weights, "
                                   + ("eval forward" if args.forward_only else "full train step (fwd+bwd+clip+Adam)"),
                       "global_batch": B * world,
                       "parallelism": f"dp{world}" + (" + graph-learner time slices (fc.weight sharded)" if sharded else ""),
                       "final_loss": float(loss.detach())},
            "step_ms": {"p10": float(np.percentile(per_step, 10)), "p50": float(np.percentile(per_step, 50)),
                        "p90": float(np.percentile(per_step, 90))},
            "roofline": {"kernel": "tsformer_encoder_kernel", "bound": "mfma", "achieved": ach, "peak": 2500.0,
                         "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": pmc_traffic(args.config, B), "ms_per_launch": enc_ms,
                         "algorithmic_flop_per_launch": flops,
                         "ms_per_launch_alone": enc_alone_ms,
                         "achieved_alone": (flops / (enc_alone_ms * 1e-3) / 1e12) if enc_alone_ms else None,
                         "note": "achieved / ms_per_launch: events around the kernel inside the timed steps, where it shares the GPU with the "
                                 "second stream's kernels (graph learner + WaveNet layers); *_alone: the same kernel in 5 extra steps with "
                                 "the overlap off"},
        }
        if loader_fig is not None:
            out["device_loader"] = loader_fig
        if comm is not None:
            out["data_parallel"] = comm
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, data)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
