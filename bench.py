"""bench.py -- STEP training windows/s on MI355X (BASELINE.json metric, config C2 = STEP_PEMS04).

One "step" = one full training step of the native STEP model on one synthetic minibatch whose data is already resident in HBM (the
processed series; the windows are gathered from it on the device, inside the timed region, by forecast origin): TSFormer encoder
forward (frozen, pre-trained weights) + kNN prior + DiscreteGraphLearning forward/backward + GraphWaveNet forward/backward +
step_loss + gradient all-reduce (N>1) + clip_grad_norm_(3.0) + Adam.
Prints ONE JSON line (rank 0).  Launch:  python bench.py [--gpus N --steps K --warmup W]
(for N>1:  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...).

"pre-trained TSFormer_PEMS04.pt loaded" (BASELINE config 2, step/step_arch/step.py:27-35): the checkpoint is produced before the
timed region by tools/pretrain_checkpoint.py -- `--pretrain-steps` native masked-pre-training steps (the repository's own config-C3
path) on the same synthetic series -- saved in the reference's format and loaded through `pre_trained_tsformer_path`.  The encoder's
softmax schedule is data dependent (units whose scores outrun the fixed shift are redone by the re-shifting loop); the line reports
how many did (`roofline.fallback_units_per_launch`) and a second figure with random-init weights (`random_init`).

Schedule of the training loop (`schedule` in the line; round 5): the frozen branch (TSFormer + kNN prior) of batch i+1 is queued on its
own stream before the backward pass of batch i (`STEP.prefetch`, step_amd/step_arch/step.py: it reads nothing the optimizer updates
and its outputs are bit-identical, tests/test_gpu_step.py::test_prefetched_frozen_branch_is_bit_identical), and the encoder runs as a
PERSISTENT launch on part of the compute units (`TSFormer.encoder_workgroups`, ENC_SPLIT below) next to the rest of the step on the
others: 4.26 -> 3.73 ms per step at PEMS04 (profiles/r05_k_*, r05_l_*).  Every timed step still holds exactly one encoder launch, one
forward, one backward and one optimizer step.  `other_schedule` = the round-4 schedule (frozen branch inside forward(), encoder over
the whole chip), `other_input_feed` = eight resident batches cycled (the round 1-4 headline loop); `--no-prefetch` /
`--resident-batches` make them the headline.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  This step uses three streams
# (main; "side": graph learner + WaveNet layers next to the encoder; "aux": the leaves of the backward) of which at most two carry work
# at a time, and runs 1.6 % faster at PEMS04 / 0.8 % at PEMS07 with TWO queues than with three or four (one queue: no overlap at all,
# 5.27 ms; profiles/r03_ar_hw_queues_ab.log, r03_as_*).  It has to be in the environment before the runtime initialises, i.e. before
# torch is imported; an explicit setting wins, and multi-process runs keep the default (the collective library brings its own stream).
# Only when bench.py IS the program: a process that imports it as a module (the tests do, for the configuration table) keeps its own
# runtime settings -- a captured-graph replay (step_amd.GraphedTrainStep) crashes inside hipGraphLaunch with two hardware queues.
# Round 5: (a) the one-rank process-group runs (--force-process-group) take the same setting -- their extra cost in round 4 was the step's
# second stream landing on the main stream's hardware queue (RCCL / torch.distributed create streams of their own), not the queue count; the
# streams are now chosen with a concurrency probe (step_arch/step.py _concurrent_stream) and two queues win there too: 4.36 vs 4.44 ms
# (profiles/r05_j_dp_one_rank.log).  (b) With the next batch's frozen branch prefetched (the default where ENC_SPLIT has an entry) FOUR
# streams carry work -- main, side / aux, and the prefetch stream with the persistent encoder -- and need a queue each: 3.73 ms with four
# queues against 5.4 with two and 5.6 with three (profiles/r05_k_persist_prefetch.log, r05_l_*).  Real multi-rank runs keep the default.
ENC_SPLIT = {"STEP_PEMS04": 160, "STEP_PEMS07": 256}      # config -> one-sequence workgroups of the persistent encoder launch when the frozen branch is prefetched
# (PEMS07, round 6: at 168 tokens the encoder now puts TWO sequences into a twelve-wave workgroup -- 1.7 -> 1.1 ms alone -- so 256 units = 128
#  workgroups = 128 compute units; 416 before.  profiles/r06_za_C4_split_sweep.log)
# ... and WHEN the next batch's frozen branch is queued: at the start of the step (next to the whole step: 3.73 -> 3.48 ms at PEMS04, where the
# encoder is then done before the bandwidth-bound backward of the graph learner starts) or behind the forward (next to the backward only: better
# at PEMS07, 5.57 vs 5.97 ms: there the graph learner's forward wants the whole chip too) -- profiles/r05_w_prefetch_early_sweep.log
PREFETCH_EARLY = {"STEP_PEMS04": True, "STEP_PEMS07": False}


def _prefetch_policy(argv):
    """(prefetch on?, encoder workgroups) from the command line, before torch is imported"""
    name = argv[argv.index("--config") + 1] if "--config" in argv and argv.index("--config") + 1 < len(argv) else "STEP_PEMS04"
    if "--no-prefetch" in argv or "--forward-only" in argv or "--graph-child" in argv:
        return False, 0
    if "--prefetch" in argv or name in ENC_SPLIT:
        return True, ENC_SPLIT.get(name, 0)
    return False, 0


if __name__ == "__main__" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "--graph-child" not in sys.argv:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if _prefetch_policy(sys.argv)[0] else "2")

import numpy as np  # noqa: E402
import torch  # noqa: E402

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
PEAK_TFLOPS = 2500.0            # dense bf16 / f16 MFMA peak, MI355X_MICROARCH.md


def synth_series(T, N, seed=0):
    """ch0 z-scored signal sin(2 pi t/288 + phi_n) + 0.5 N(0,1); ch1 time of day; ch2 day of week."""
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float32)[:, None]
    phase = rng.uniform(0, 2 * np.pi, (1, N)).astype(np.float32)
    ch0 = np.sin(2 * np.pi * t / 288.0 + phase) + 0.5 * rng.standard_normal((T, N), dtype=np.float32)
    ch1 = np.broadcast_to((t % 288) / 288.0, (T, N))
    ch2 = np.broadcast_to((t // 288) % 7, (T, N))
    return np.stack([ch0, ch1, ch2], -1).astype(np.float32)


def tsformer_args(L, mode):
    return dict(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, num_token=L / 12,
                mask_ratio=0.75, encoder_depth=4, decoder_depth=1, mode=mode)


def make_model(cfg, data, ckpt=None):
    from step_amd import STEP
    N, L = cfg["N"], cfg["L"]
    bargs = dict(num_nodes=N, support_len=2, dropout=0.3, gcn_bool=True, addaptadj=True, aptinit=None, in_dim=2, out_dim=12,
                 residual_channels=32, dilation_channels=32, skip_channels=256, end_channels=512, kernel_size=2, blocks=4, layers=2)
    dargs = dict(dataset_name="SYNTH", k=cfg["k"], input_seq_len=12, output_seq_len=12, data=data, train_length=cfg["T_train"],
                 tsformer_tokens=L // 12)
    torch.manual_seed(0)
    return STEP("SYNTH", ckpt, tsformer_args(L, "forecasting"), bargs, dargs)


def native_checkpoint(name, cfg, data, steps, dev, workdir):
    """tsformer_ckpt/TSFormer_<name>.pt made by `steps` native pre-training steps (tools/pretrain_checkpoint.py); returns
    (path, info).  Outside every timed region."""
    from tools.pretrain_checkpoint import pretrain
    t0 = time.perf_counter()
    sd, losses = pretrain(data, cfg["L"], steps=steps, batch=6, device=dev, matmul="bf16", seed=0)
    path = os.path.join(workdir, "tsformer_ckpt", f"TSFormer_{name}.pt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"model_state_dict": sd}, path)
    torch.cuda.synchronize()
    return path, {"steps": steps, "batch": 6, "first_loss": losses[0], "last_loss": losses[-1], "seconds": time.perf_counter() - t0, "sd": sd}


def encoder_flops(cfg, B):
    """Algorithmic FLOPs of one encoder launch (SURVEY.md 8d, row T4 + T1): per window
    N*P*(4*(221184 + 384*P) + 2304)."""
    P = cfg["L"] // 12
    return B * cfg["N"] * P * (4 * (221184 + 384 * P) + 2304)


def step_flops(cfg, B):
    """Algorithmic FLOPs of one whole training step: B * W_window + W_step."""
    N, P, T = cfg["N"], cfg["L"] // 12, cfg["T_train"]
    w_window = N * P * (4 * (221184 + 384 * P) + 2304) + 2 * N * N * P * 96 + 3 * (19968 * N * N + 2.66e6 * N)
    w_step = 3 * (2 * N * (80 * (T - 9) + 1280 * (T - 18) + 1600 * (T - 18)) + N * N * 500 + 40000 * N)
    return B * w_window + w_step


def static_pmc_traffic(config, B):
    """HBM bytes per encoder launch from the committed rocprofv3 PMC passes (profiles/encoder_pmc.json, tools/pmc_enc_ab.sh):
    the fallback when no counter run can be made from inside this process (no rocprofv3 on the box)."""
    path = os.path.join(ROOT, "profiles", "encoder_pmc.json")
    try:
        with open(path) as f:
            ent = json.load(f).get(f"{config}:B{B}")
        return None if ent is None else {"hbm_bytes_per_launch": ent["read_bytes"] + ent["write_bytes"], "read_bytes": ent["read_bytes"],
                                          "write_bytes": ent["write_bytes"], "static": True, "source": ent["source"]}
    except (OSError, ValueError, KeyError):
        return None


def live_pmc_traffic(config, B, ckpt, timeout=150, enc_wgs=0):
    """HBM bytes of the encoder launch measured IN THIS RUN: two rocprofv3 counter passes (FETCH_SIZE, then WRITE_SIZE -- they do
    not fit one pass; --kernel-trace only, no other trace domain) over a child process that loads the same checkpoint and launches
    the training-mode encoder at this config's size (bench.py --pmc-child).  Units and the gfx950 correction as
    MI355X_MICROARCH.md "HBM" prescribes: both counters in KiB, FETCH_SIZE doubled.  None if rocprofv3 is unavailable or fails."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    import glob
    import sqlite3
    vals = {}
    work = tempfile.mkdtemp(prefix="step_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = os.path.join(work, ctr)
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", ckpt or "-",
               "--config", config, "--batch", str(B), "--encoder-workgroups", str(int(enc_wgs))]
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            for db in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
                cur = sqlite3.connect(db).cursor()
                rows = cur.execute("""select p.name, sum(e.value), count(distinct d.id) from rocpd_pmc_event e
                    join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id
                    join rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like '%tsformer_encoder_kernel%'
                    group by p.name""").fetchall()
                for name, total, n in rows:
                    vals[name] = total / max(n, 1)
        except Exception:          # noqa: BLE001 -- any failure of the optional counter run falls back to the committed measurement
            pass
    shutil.rmtree(work, ignore_errors=True)
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    rd, wr = int(vals["FETCH_SIZE"] * 1024 * 2), int(vals["WRITE_SIZE"] * 1024)
    return {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "static": False,
            "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) around `bench.py --pmc-child` in this run: "
                      "same checkpoint, same launch size, dropout on; KiB counters, FETCH_SIZE x2 (gfx950)"}


def graph_replay_figure(config, B, ckpt, args, steps=30, timeout=240):
    """The step replayed from one captured hipGraph (step_amd.GraphedTrainStep), measured in a CHILD process: this process pins
    GPU_MAX_HW_QUEUES=2 for its eager three-stream schedule, and a replay of a graph with three parallel branches needs the runtime's
    default queue count (with two queues hipGraphLaunch crashes the process, profiles/r04_h_*).  The child runs the eager loop and the
    replayed loop back to back under the default and prints both."""
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--graph-child", "--config", config, "--batch", str(B), "--steps", str(steps), "--warmup", "8",
           "--no-extras", "--no-cpu-baseline", "--no-pmc", "--matmul", args.matmul, "--pretrain-steps", "0" if ckpt is None else str(args.pretrain_steps)]
    try:
        out = subprocess.run(cmd, env=env, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
        for line in reversed(out.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": "no output from the graph child"}
    except Exception as ex:          # noqa: BLE001 -- an optional figure must not take the headline line with it
        return {"error": f"{type(ex).__name__}: {ex}"[:300]}


def runner_feed_figure(timeout=300):
    """The same config driven by the REFERENCE's own training loop (tools/runner_feed_bench.py: the reference's unmodified config file, STEPRunner,
    train_iters and scaler registry from the staged sources, tests/_shims for easytorch) with `CFG.RUNNER = step_amd.runner.native_runner(STEPRunner)`,
    `CFG.DATASET_CLS = step_amd.runner.DeviceForecastingDataset`, in child processes: what a maintainer gets from the three config lines of
    INTEGRATION.md, and -- 24 iterations -- the reference's runner and host dataset as they are around the same module."""
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "runner_feed_bench.py")
    out = {}
    for name, extra in (("native_runner_device_dataset", ["--runner", "native", "--dataset", "device", "--loss", "native", "--iters", "60"]),
                        ("native_runner_device_dataset_reference_loss", ["--runner", "native", "--dataset", "device", "--iters", "60"]),
                        ("reference_runner_host_dataset", ["--runner", "reference", "--dataset", "host", "--iters", "24"])):
        try:
            r = subprocess.run([sys.executable, tool] + extra, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
            out[name] = next((json.loads(l) for l in reversed(r.strip().splitlines()) if l.startswith("{")), {"error": "no output"})
        except Exception as ex:          # noqa: BLE001 -- an optional figure must not take the headline line with it
            out[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    out["what"] = ("runner.train() of the reference around step_amd.STEP at this config (batch 8, bf16 mode, dropout on, shuffled windows over the whole "
                   "training split): windows/s and ms per iteration, host side included")
    return out


def pmc_child(args):
    """Child of live_pmc_traffic: three training-mode encoder launches at the config's size, nothing else."""
    from step_amd import TSFormer
    cfg = dict(CONFIGS[args.config])
    B = args.batch or cfg["B"]
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = TSFormer(**tsformer_args(cfg["L"], "forecasting")).to(dev)
    if args.pmc_child != "-":
        m.load_state_dict(torch.load(args.pmc_child, map_location="cpu")["model_state_dict"])
    m.train()
    m.encoder_workgroups = int(args.encoder_workgroups or 0)          # the launch form the timed loop uses (persistent, or one workgroup per sequence)
    S = B * cfg["N"]
    rng = np.random.default_rng(0)
    t = np.arange(cfg["L"], dtype=np.float32)[None]
    x = np.sin(2 * np.pi * t / 288.0 + rng.uniform(0, 6.28, (S, 1)).astype(np.float32)) + 0.5 * rng.standard_normal((S, cfg["L"]), dtype=np.float32)
    series = torch.from_numpy(x.astype(np.float32)).to(dev)
    for _ in range(3):
        m.encode_series(series)
    torch.cuda.synchronize()


def cpu_baseline(cfg, data, seed=0, warm=2, timed=5, two_threads=True):
    """The CPU oracle (restatement of the reference algorithm, kind "port") timed on this host's cores on a bounded sample of
    the same workload: full training steps (forward + step_loss + backward + clip_grad_norm_ + Adam) of ONE window each --
    `warm` warm-up and `timed` timed steps on all cores (median reported), then (headline config) one step with two threads, the thread
    count the reference ships with (step/run.py:10 torch.set_num_threads(2)).  The reference itself cannot run on the GPU box
    (no /root/reference there); its own CPU timing, taken in the build container, is profiles/r02_cpu_reference_baseline.json.
    On graphs of >= 2048 nodes the oracle evaluates the edge MLP in receiver-row blocks (the reference's formulation does not exist
    there: 2 x 275 GB of one-hot matrices at N = 4096, discrete_graph_learning.py:88-89) -- same arithmetic, restated path."""
    from oracle import step_oracle as O
    torch.manual_seed(seed)
    N, L, Ttr = cfg["N"], cfg["L"], cfg["T_train"]
    model = make_model(cfg, data)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    for k, v in p.items():
        if v.is_floating_point() and not k.startswith("tsformer.") and "running_" not in k:
            v.requires_grad_(True)
    train = [v for v in p.values() if v.requires_grad]
    opt = torch.optim.Adam(train, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8)
    d = torch.from_numpy(data)
    chunk = 128 if N >= 2048 else None

    def one_step(i):
        t = L + 17 + 301 * i
        hist, fut, longh = d[t - 12:t][None], d[t:t + 12][None], d[t - L:t][None]
        u = torch.rand(1, N * N, 2)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        pred, theta, knn, coef = O.step_forward(hist, longh[..., [0]], d[:Ttr, :, 0], p, u, cfg["k"], 1, training=True, edge_row_chunk=chunk)
        loss = O.step_loss(O.rescale(pred, 200.0, 150.0), O.rescale(fut[..., [0]], 200.0, 150.0), theta, knn, coef)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([q for q in train if q.grad is not None], 3.0)
        opt.step()
        return time.perf_counter() - t0
    cores = min(os.cpu_count() or 1, 32)          # torch CPU ops stop scaling (and oversubscribe) beyond a few tens of threads
    torch.set_num_threads(cores)
    for i in range(warm):
        one_step(i)
    ts = sorted(one_step(warm + i) for i in range(timed))
    med = ts[len(ts) // 2]
    out = {"value": 1.0 / med, "unit": "windows/s", "cores": cores, "kind": "port",
           "sample": f"full training steps (fwd+loss+bwd+clip+Adam) of 1 window of the same workload, torch CPU fp32 oracle"
                     + (" (edge MLP in receiver-row blocks: the restated path, the reference cannot build this graph size)" if chunk else "")
                     + f": {warm} warm-up + {timed} timed on {cores} threads ({ts[0]:.1f} .. {med:.1f} .. {ts[-1]:.1f} s, median reported)"}
    if two_threads:
        torch.set_num_threads(2)
        t2 = one_step(warm + timed)
        torch.set_num_threads(cores)
        out["sample"] += f", 1 step on 2 threads ({t2:.1f} s)"
        out["two_threads"] = {"value": 1.0 / t2, "unit": "windows/s", "cores": 2}
    return out


def cpu_baseline_reference(cfg, data, warm=1, timed=3):
    """The reference's OWN modules (kind "reference") timed on this host's cores, when its sources are here -- /root/reference in the build
    container, the archive oracle/_ref/reference.tar.gz (unpacked into the temporary directory) on a GPU box after tools/stage_reference.sh: step_arch.STEP.forward + step_loss on re-scaled outputs +
    backward + clip_grad_norm_(3.0) + Adam (the training step of step/STEP_PEMS04.py with easytorch's Runner.backward), fp32, train mode,
    ONE window per step of the same synthetic workload; `warm` + `timed` steps on all cores (median), then one step on two threads (what
    step/run.py:10 ships).  Returns None where the reference cannot run: no sources, or a graph it cannot build (the [N^2, N] one-hot
    matrices, discrete_graph_learning.py:81-89, need 2 x N^3 x 4 bytes: 275 GB each at N = 4096)."""
    from oracle.reference_loader import reference_root
    N, L, Ttr = cfg["N"], cfg["L"], cfg["T_train"]
    if reference_root() is None or N > 1024:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden as MG
    ref = MG.import_reference()
    step_loss = ref[5]
    torch.manual_seed(0)
    model = MG.build_step(ref, N, L, Ttr, data, cfg["k"])      # the reference's classes, attribute by attribute (its constructor reads files by dataset name)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8)
    d = torch.from_numpy(data)

    def one_step(i):
        t = L + 17 + (301 * i) % (data.shape[0] - L - 40)
        hist, fut, longh = d[t - 12:t][None], d[t:t + 12][None], d[t - L:t][None]
        t0 = time.perf_counter()
        opt.zero_grad()
        pred, theta, knn, coef = model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=i, epoch=1)
        loss = step_loss(pred[..., [0]] * 150.0 + 200.0, fut[..., [0]] * 150.0 + 200.0, theta, knn, coef, null_val=0.0)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=3.0)
        opt.step()
        return time.perf_counter() - t0
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    for i in range(warm):
        one_step(i)
    ts = sorted(one_step(warm + i) for i in range(timed))
    med = ts[len(ts) // 2]
    torch.set_num_threads(2)
    t2 = one_step(warm + timed)
    torch.set_num_threads(cores)
    return {"value": 1.0 / med, "unit": "windows/s", "cores": cores, "kind": "reference",
            "sample": f"the reference's own step_arch.STEP + step_loss + backward + clip_grad_norm_ + Adam (sources: {reference_root()}), torch "
                      f"{torch.__version__} CPU fp32, full training steps of 1 window of the same workload: {warm} warm-up + {timed} timed on "
                      f"{cores} threads ({ts[0]:.1f} .. {med:.1f} .. {ts[-1]:.1f} s, median reported), 1 step on 2 threads ({t2:.1f} s)",
            "two_threads": {"value": 1.0 / t2, "unit": "windows/s", "cores": 2}}


def cpu_baseline_pretrain(cfg, data, seed=0, warm=1, timed=3):
    """Config C3's CPU baseline: the oracle's masked pre-training step (tsformer_pretrain + masked MAE on rescaled values + backward +
    clip 5.0 + Adam, reference step/TSFormer_PEMS-BAY.py:52-76) of ONE window (N sequences of L steps) on this host's cores."""
    import random
    from oracle import step_oracle as O
    from step_amd import TSFormer
    torch.manual_seed(seed)
    random.seed(seed)
    N, L = cfg["N"], cfg["L"]
    m = TSFormer(**tsformer_args(L, "pre-train"))
    p = {"tsformer." + k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    train = [v for v in p.values() if v.requires_grad]
    opt = torch.optim.Adam(train, lr=0.001, weight_decay=0, eps=1.0e-8, betas=(0.9, 0.95))
    d = torch.from_numpy(data[:, :, 0])
    P = L // 12

    def one_step(i):
        t = L + 31 + 211 * i
        hist = d[t - L:t][None, :, :, None]                # [1, L, N, 1]
        idx = list(range(P))
        random.shuffle(idx)
        nm = int(P * 0.75)
        masked, unmasked = sorted(idx[:nm]), sorted(idx[nm:])
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        recon, label = O.tsformer_pretrain(hist, p, unmasked, masked)
        loss = O.masked_mae(recon * 150.0 + 200.0, label * 150.0 + 200.0, 0.0)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([q for q in train if q.grad is not None], 5.0)
        opt.step()
        return time.perf_counter() - t0
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    for i in range(warm):
        one_step(i)
    ts = sorted(one_step(warm + i) for i in range(timed))
    med = ts[len(ts) // 2]
    return {"value": 1.0 / med, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"masked pre-training steps (fwd+loss+bwd+clip+Adam) of 1 window ({N} sequences x {L} steps), torch CPU fp32 oracle: "
                      f"{warm} warm-up + {timed} timed on {cores} threads ({ts[0]:.2f} .. {med:.2f} .. {ts[-1]:.2f} s, median reported)"}


def static_dominant_kernels(name, top=4):
    """The config's heaviest kernels by time per step with their roofline fractions, from the committed rocprofv3 table of this code
    (profiles/kernel_roofline.json, written by tools/roofline_table.py --json from a `--kernel-trace --stats` summary): static evidence,
    not measured in this run -- the line's own live figures are `roofline` (events around the encoder) and the step time."""
    path = os.path.join(ROOT, "profiles", "kernel_roofline.json")
    try:
        with open(path) as f:
            ent = json.load(f).get(name)
        if ent is None:
            return None
        ks = [{k: r[k] for k in ("what", "bound", "launches_per_step", "avg_us", "ms_per_step", "GBps", "frac_of_hbm_peak", "TFLOPps", "frac_of_mfma_peak")}
              for r in ent["kernels"][:top]]
        return {"static": True, "source": ent["source"], "kernel_ms_per_step_no_overlap": ent["kernel_ms_per_step"],
                "dispatches_per_step": ent["dispatches_per_step"], "kernels": ks}
    except (OSError, ValueError, KeyError):
        return None


def static_roofline(name):
    """`roofline` object of a config's heaviest kernel from the committed per-kernel table (static: durations are the rocprofv3 averages of
    the run named in `source`, bytes / FLOPs analytic); None without a table."""
    dom = static_dominant_kernels(name, top=1)
    if not dom or not dom["kernels"]:
        return None
    k = dom["kernels"][0]
    mf = k.get("TFLOPps") is not None and k["bound"] not in ("hbm", "hbm/L2")
    return {"kernel": k["what"], "bound": "mfma" if mf else "hbm", "kernel_class": k["bound"],
            "achieved": k["TFLOPps"] if mf else k["GBps"], "peak": PEAK_TFLOPS if mf else 8000.0, "unit": "TFLOP/s" if mf else "GB/s",
            "frac": k["frac_of_mfma_peak"] if mf else k["frac_of_hbm_peak"], "ms_per_launch": k["avg_us"] / 1e3,
            "launches_per_step": k["launches_per_step"], "traffic": None, "static": True, "source": dom["source"]}


def average_grads(model, params, world):
    """Data-parallel exchange step of the pre-training bench (the job DistributedDataParallel does for easytorch in the reference):
    the native backward leaves every gradient in ONE flat buffer (TSFormer._flat_grad) that autograd adopts as the parameters'
    .grad, so the exchange is one all-reduce (mean) of 2.67 MB with no flatten / scatter-back copies.  If autograd cloned instead
    of adopting (checked by address), fall back to the flatten + scatter form."""
    if world <= 1:
        return
    import torch.distributed as dist
    flat = model._flat_grad
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    grads = [p.grad for p in params if p.grad is not None]
    dist.all_reduce(flat)               # (the flat buffer is also what step_amd.optim.FusedAdamClip consumes)
    flat.mul_(1.0 / world)
    if all(lo <= g.data_ptr() < hi for g in grads):
        return
    buf = torch.cat([g.reshape(-1) for g in grads])          # autograd cloned: average the clones too (what torch.optim reads)
    dist.all_reduce(buf)
    buf.mul_(1.0 / world)
    off = 0
    for g in grads:
        g.copy_(buf[off:off + g.numel()].view_as(g))
        off += g.numel()


HOST = {"enqueue_s": None}


def timed_loop(step, warmup, steps, barrier, start=0):
    """`warmup` untimed steps, then exactly `steps` timed ones between two barriers (+ device synchronize): seconds, per-step ms."""
    for i in range(warmup):
        step(start + i)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    out = None
    for i in range(steps):
        out = step(start + warmup + i)
        marks[i + 1].record()
    HOST["enqueue_s"] = time.perf_counter() - t0          # host time to ENQUEUE the timed steps (no device wait in it unless a step syncs)
    barrier()
    dt = time.perf_counter() - t0
    per_step = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(steps)])
    return dt, per_step, out


DIST = {"on": False}        # a process group exists (world > 1, or --force-process-group on one rank)


def _c_stdio_flush():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:          # noqa: BLE001
        pass
    sys.stdout.flush()


def stdout_is_for_the_json_line(rank):
    """The driver reads ONE JSON line from this job's stdout.  RCCL prints a version banner through C stdio when a communicator is created;
    on a pipe that buffer is flushed at process exit, i.e. AFTER the line (seen with --force-process-group).  Ranks other than 0 therefore
    send their stdout to stderr from the start; rank 0 flushes C stdio before the line and sends whatever comes after it to stderr."""
    if rank != 0:
        _c_stdio_flush()
        os.dup2(2, 1)


def print_json_line(obj):
    sys.stdout.flush()
    saved = os.dup(1)                   # what C stdio has buffered so far (the banner) leaves through stderr: the line is the ONLY one on stdout
    os.dup2(2, 1)
    _c_stdio_flush()
    os.dup2(saved, 1)
    os.close(saved)
    print(json.dumps(obj), flush=True)
    _c_stdio_flush()
    os.dup2(2, 1)


def max_over_ranks(dt, world, dev):
    if DIST["on"]:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t)
    return dt


def pretrain_run(args, cfg, world, rank, dev, warmup, steps):
    """Config C3: one step = TSFormer masked pre-training forward + masked_mae + backward + clip(5.0) + Adam
    (reference step/TSFormer_PEMS-BAY.py:52-72), windows/s = sequences of L steps x N nodes per second."""
    import random
    from step_amd import TSFormer
    from step_amd.step_loss import masked_mae
    N, Lh, B = cfg["N"], cfg["L"], cfg["B"]
    data = synth_series(cfg["T_all"], N)
    torch.manual_seed(0)
    random.seed(0)
    model = TSFormer(**tsformer_args(Lh, "pre-train")).to(dev)
    model.train()
    model.matmul_precision = args.matmul
    params = [p for p in model.parameters() if p.requires_grad]
    if args.torch_optim:
        opt = torch.optim.Adam(params, lr=0.001, weight_decay=0, eps=1.0e-8, betas=(0.9, 0.95))      # step/TSFormer_PEMS-BAY.py:60-66
    else:
        # same update rule and the runner's clip (5.0) as one fused pass over the flat parameter / gradient buffers; the loss below is the
        # same masked MAE on the rescaled values, value and gradient from two launches (the reference's runner makes ~60 element-wise ones)
        from step_amd.optim import FusedAdamClip
        from step_amd.step_loss import masked_mae_native
        model.flatten_parameters()
        opt = FusedAdamClip(model, lr=0.001, weight_decay=0.0, eps=1.0e-8, betas=(0.9, 0.95), max_norm=5.0)
    dser = torch.from_numpy(data[:, :, :1]).to(dev)
    rng = np.random.default_rng(99 + rank)
    batches = []
    for _ in range(4):
        ts = rng.integers(Lh, cfg["T_all"] - 12, size=B)
        batches.append(torch.stack([dser[t - Lh:t] for t in ts]))

    def step(i):
        opt.zero_grad(set_to_none=True)
        recon, label = model(history_data=batches[i % len(batches)], future_data=None, batch_seen=i, epoch=1)
        if args.torch_optim:
            loss = masked_mae(recon * 150.0 + 200.0, label * 150.0 + 200.0, 0.0)
        else:
            loss = masked_mae_native(recon.transpose(1, 2), label.transpose(1, 2), 0.0, rescale=(200.0, 150.0))
        loss.backward()
        average_grads(model, params, world)
        if args.torch_optim:
            torch.nn.utils.clip_grad_norm_(params, max_norm=5.0)
        opt.step()
        return loss

    def barrier():
        if DIST["on"]:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    dt, per_step, loss = timed_loop(step, warmup, steps, barrier)
    dt = max_over_ranks(dt, world, dev)
    flops = B * 87.2e9
    return {"value": B * world * steps / dt, "unit": "windows/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "whole_step_tflops": flops / (dt / steps) / 1e12, "whole_step_frac_of_mfma_peak": flops / (dt / steps) / 1e12 / PEAK_TFLOPS,
            "workload": f"TSFormer_PEMS-BAY pre-train: N={N}, L={Lh} (P={Lh // 12}, 42 unmasked), batch {B}/GPU, fwd+bwd"
                        + ("+grad all-reduce" if world > 1 else "") + "+clip+Adam, "
                        + ("bf16 operands / f32 accumulate GEMMs" if args.matmul == "bf16" else "exact-f32 GEMMs"),
            "final_loss": float(loss.detach())}


class StepBench:
    """One STEP config on this rank: model, resident batches, optimizer, step functions."""

    def __init__(self, name, cfg, args, world, rank, dev, ckpt):
        from step_amd.step_loss import step_loss_native
        self.name, self.cfg, self.args, self.world, self.rank, self.dev = name, cfg, args, world, rank, dev
        self.step_loss = step_loss_native
        N, Lh, B = cfg["N"], cfg["L"], cfg["B"]
        self.data = synth_series(cfg["T_all"], N)
        self.model = make_model(cfg, self.data, ckpt).to(dev)
        self.model.train()
        self.model.matmul_precision = args.matmul
        self.prefetch = (args.prefetch or (name in ENC_SPLIT)) and not args.no_prefetch and not args.forward_only
        self.enc_wgs = 0
        self.prefetch_early = (PREFETCH_EARLY.get(name, False) or getattr(args, "prefetch_early", False)) and not getattr(args, "prefetch_late", False)
        if self.prefetch:
            self.enc_wgs = int(args.encoder_workgroups) if getattr(args, "encoder_workgroups", None) is not None else ENC_SPLIT.get(name, 0)
        elif getattr(args, "encoder_workgroups", None):
            self.enc_wgs = int(args.encoder_workgroups)
        self.model.tsformer.encoder_workgroups = self.enc_wgs
        self.prefetch_ahead = max(1, int(getattr(args, "prefetch_ahead", 1) or 1))
        self.model.prefetch_fifo = 1 + self.prefetch_ahead
        if os.environ.get("STEP_PREFETCH_KNN_STREAM") is None:          # the announced batch's kNN prior on its own stream: with the early announcement only
            self.model.prefetch_knn_stream = bool(self.prefetch and self.prefetch_early)
        if args.eval_dropout_off:
            self.model.backend.dropout = 0.0
            self.model.tsformer.dropout_p = 0.0
        if DIST["on"]:
            # each rank keeps one time slice of the graph learner's global branch and of fc.weight (bf16 mode)
            self.model.enable_native_data_parallel(shard_graph_learner=args.matmul == "bf16" and not args.no_shard and not args.torch_optim,
                                                   single_rank_collectives=world == 1, collectives=args.collectives)
        self.sharded = self.model.discrete_graph_learning._shard is not None
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        if args.torch_optim:
            self.opt = torch.optim.Adam(self.params, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8)    # step/STEP_PEMS04.py:90-96
        else:
            from step_amd.optim import FusedAdamClip
            # same rule, one fused pass over the flat buffers; the gradients stay in the native backward's flat buffer (no per-parameter .grad)
            self.opt = FusedAdamClip(self.model, lr=0.002, weight_decay=1.0e-5, eps=1.0e-8, max_norm=3.0, param_grads=False)
        self.dser = torch.from_numpy(self.data).to(dev)
        rng = np.random.default_rng(1234 + rank)
        self.batches = []
        for _ in range(8):                               # resident input batches, cycled (119 MB each at C2)
            ts = rng.integers(Lh, cfg["T_all"] - 12, size=B)
            hist = torch.stack([self.dser[t - 12:t] for t in ts])
            fut = torch.stack([self.dser[t:t + 12] for t in ts])
            longh = torch.stack([self.dser[t - Lh:t] for t in ts])
            self.batches.append((hist, longh, fut))
        self.mean, self.std = 200.0, 150.0
        # the training loop's input: the index-only loader over the device-resident series ("H2D + gather inside the timed
        # region"), one batch ahead like a DataLoader's prefetch; forecast origins uniform over the training split, INCLUDING the windows
        # that start before a full long history exists (all-zero history, forecasting_dataset.py:66-67: 36-39 % at PEMS04)
        from step_amd.step_arch.step import DeviceWindowLoader
        self.use_loader = not args.resident_batches and not args.forward_only
        self.loader = DeviceWindowLoader(self.dser, Lh)
        self.n_train = int((cfg["T_all"] - 23) * cfg.get("train_ratio", 0.6))
        self._staged = None
        self._announced = None
        self._pin = None

    def barrier(self):
        if DIST["on"]:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def eval_step(self, i):
        from step_amd.step_loss import masked_mae
        hist, longh, fut = self.batches[i % len(self.batches)]
        with torch.no_grad():
            pred, theta, knn, coef = self.model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=i, epoch=None)
            return masked_mae(pred[..., :1] * self.std + self.mean, fut[..., :1] * self.std + self.mean, 0.0)       # base_tsf_runner.py:257-318

    def origins(self, i):
        """forecast origins of step i (reproducible per step and rank)"""
        r = np.random.default_rng([4321, self.rank, i])
        idx = torch.from_numpy(r.integers(12, 12 + self.n_train, size=self.cfg["B"]))
        if getattr(self.args, "pageable_origins", False):
            return idx.to(self.dev, non_blocking=True)          # (A/B) from pageable memory the copy blocks the host until the stream has drained
        # through a ring of PINNED slots, like the reference's DataLoader (pin_memory=True, STEP_PEMS04.py:122): the copy is asynchronous and
        # the host keeps enqueueing ahead of the device (64 slots: a slot is rewritten 64 steps later)
        if self._pin is None:
            self._pin = torch.empty((64, self.cfg["B"]), dtype=torch.int64).pin_memory()
        slot = self._pin[i % 64]
        slot.copy_(idx)
        return slot.to(self.dev, non_blocking=True)

    def batch(self, i):
        if not self.use_loader:
            return self.batches[i % len(self.batches)]
        if isinstance(self._staged, dict) and i in self._staged:
            return self._staged.pop(i)
        return self.loader.batch(self.origins(i))

    def stage(self, i):
        """the loader runs ahead of the step: gather batch i now (one or two steps early); kept until step i takes it"""
        if not self.use_loader:
            return self.batches[i % len(self.batches)]
        if not isinstance(self._staged, dict):
            self._staged = {}
        if i not in self._staged:
            for old in [k for k in self._staged if k < i - 3]:
                del self._staged[old]
            self._staged[i] = self.loader.batch(self.origins(i))
        return self._staged[i]

    def train_step(self, i, epoch=1):
        hist, longh, fut = self.batch(i)
        early = self.prefetch and self.prefetch_early
        if early:                         # the next batch's frozen branch (TSFormer + kNN prior) runs next to the whole of this step
            ahead = self.prefetch_ahead
            if ahead > 1 and not getattr(self, "_announced", None) == i:          # first step of a loop: batch i + 1 was not announced a step ago
                self.model.prefetch(self.stage(i + 1)[1])
            self.model.prefetch(self.stage(i + ahead)[1])
            self._announced = i + 1
        self.opt.zero_grad(set_to_none=True)
        pred, theta, knn, coef = self.model(history_data=hist, long_history_data=longh, future_data=None, batch_seen=i, epoch=epoch)
        if not early:
            nxt = self.stage(i + 1)
            if self.prefetch:             # ... or only next to this batch's backward + Adam (round-3 placement)
                self.model.prefetch(nxt[1])
        # target-feature selection + inverse scaling, as the runner does (step_runner.py:86-92); slices, not index kernels
        loss = self.step_loss(pred[..., :1], fut[..., :1], theta, knn, coef, null_val=0.0, rescale=(self.mean, self.std))
        loss.backward()
        if self.args.torch_optim:
            torch.nn.utils.clip_grad_norm_(self.params, max_norm=3.0)                    # STEP_PEMS04.py:103-105
        self.opt.step()
        return loss

    def step(self, i):
        return self.eval_step(i) if self.args.forward_only else self.train_step(i)

    def run(self, warmup, steps, start=0):
        """-> dict(value, ms_per_step, p10/p50/p90, encoder ms in-step, fallback units per launch, final loss)"""
        m = self.model
        if self.args.forward_only:
            m.eval()
        for i in range(warmup):
            self.step(start + i)
        m.tsformer._events = []
        m.tsformer.fallback_counter = torch.zeros(65, dtype=torch.int32, device=self.dev)
        if DIST["on"]:
            m._reduce_wait_ms = []
        allocs0 = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0)
        dt, per_step, loss = timed_loop(self.step, 0, steps, self.barrier, start + warmup)
        allocs = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0) - allocs0
        if DIST["on"]:
            m.collect_reduce_waits()
            self.reduce_waits, self.small = list(m._reduce_wait_ms), m.collect_small_collectives()
            m._reduce_wait_ms = None
        dt = max_over_ranks(dt, self.world, self.dev)
        ev = m.tsformer._events
        enc_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")
        launches = max(len(ev), 1)
        slow = int(m.tsformer.fallback_counter[:64].sum().item())
        m.tsformer._events, m.tsformer.fallback_counter = None, None
        B = self.cfg["B"]
        return {"value": B * self.world * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
                "p10": float(np.percentile(per_step, 10)), "p50": float(np.percentile(per_step, 50)), "p90": float(np.percentile(per_step, 90)),
                "host_enqueue_ms_per_step": HOST["enqueue_s"] / steps * 1e3, "device_allocs": int(allocs), "enc_ms": enc_ms, "enc_launches": len(ev), "fallback_units_per_launch": slow / launches, "final_loss": float(loss.detach())}

    def softmax_units(self):
        P = self.cfg["L"] // 12
        return self.cfg["B"] * self.cfg["N"] * 4 * 4 * ((P + 31) // 32)

    def encoder_alone_ms(self, start):
        """the kernel's own duration: a few untimed steps with neither the second stream nor the prefetch next to it"""
        m = self.model
        keep = (m.overlap_streams, self.prefetch, m.tsformer.encoder_workgroups)
        m.overlap_streams, self.prefetch, m.tsformer.encoder_workgroups = False, False, 0       # one workgroup per sequence: the whole chip
        m.cancel_prefetch()
        self._staged = None
        self._announced = None
        m.tsformer._events = []
        for i in range(6):
            self.step(start + i)
        torch.cuda.synchronize()
        ev = m.tsformer._events[1:]
        m.tsformer._events = None
        m.overlap_streams, self.prefetch, m.tsformer.encoder_workgroups = keep
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None

    def alternative_figure(self, steps, start, prefetch, enc_wgs, loader, what):
        """the same training step under another schedule / input feed, in this process (same runtime settings)"""
        m = self.model
        keep = (self.prefetch, m.tsformer.encoder_workgroups, self.use_loader)
        m.cancel_prefetch()
        self._staged = None
        self._announced = None
        self.prefetch, m.tsformer.encoder_workgroups, self.use_loader = prefetch, enc_wgs, loader
        r = self.run(3, steps, start)
        m.cancel_prefetch()
        self._staged = None
        self._announced = None
        self.prefetch, m.tsformer.encoder_workgroups, self.use_loader = keep
        torch.cuda.synchronize()
        return {"value": r["value"], "unit": "windows/s", "ms_per_step": r["ms_per_step"], "steps": steps, "encoder_ms_per_launch": r["enc_ms"],
                "what": what}

    def graph_figure(self, steps):
        """the same training step replayed from ONE captured hipGraph (step_amd.GraphedTrainStep): host time per step = one graph launch
        + the copies of the batch into the graph's static input buffers (inside the timed region)"""
        from step_amd import GraphedTrainStep
        m = self.model
        m.tsformer._events, m.tsformer.fallback_counter = None, None
        m.cancel_prefetch()
        gs = GraphedTrainStep(m, self.opt, self.batches[0], scaler=(self.mean, self.std), epoch=1, warmup=3)
        try:
            def gstep(i):
                hist, longh, fut = self.batches[i % len(self.batches)]
                return gs(hist, longh, fut)
            dt, per_step, loss = timed_loop(gstep, 5, steps, self.barrier)
            host = HOST["enqueue_s"] / steps * 1e3
            same = timed_loop(lambda i: gs(), 3, max(steps // 2, 5), self.barrier)[0] / max(steps // 2, 5) * 1e3
        finally:
            gs.close()
        B = self.cfg["B"]
        return {"value": B * steps / dt, "unit": "windows/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
                "p10": float(np.percentile(per_step, 10)), "p50": float(np.percentile(per_step, 50)), "p90": float(np.percentile(per_step, 90)),
                "host_enqueue_ms_per_step": host, "ms_per_step_without_input_copies": same, "final_loss": float(loss.detach()),
                "what": "the same full training step (zero_grad, forward, step_loss, backward, clip + Adam, dropout on) captured once into a "
                        "hipGraph and replayed: one launch per step; seeds / Adam step count / learning rate / loss coefficient are read from "
                        "a device-resident state the graph's first node advances; each replay is preceded by the copies of the batch into the "
                        "graph's static inputs (`ms_per_step_without_input_copies`: replays of the resident batch)"}

    def comm_figure(self):
        """data-parallel exchange: the flat-gradient all-reduce alone (isolated) and the part of it the step does not hide"""
        import torch.distributed as dist
        model, dev, world = self.model, self.dev, self.world
        lay = model._grad_layout()
        fo, fn, _ = lay["items"]["dgl.fc_w"]
        buf = torch.zeros(lay["total"], device=dev)
        reps = 5
        nc = model._comm                        # the step's own communicator (RCCL C API), or None: torch.distributed
        red = (lambda t: nc.allreduce_(t, average=True)) if nc is not None else (lambda t: dist.all_reduce(t))
        for _ in range(2):
            red(buf[:fo])          # (with time slices the fc chunk has a different length on every rank and is never reduced)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            if not self.sharded:
                red(buf[fo:fo + fn])
            red(buf[:fo])
        e1.record()
        torch.cuda.synchronize()
        iso = e0.elapsed_time(e1) / reps
        exposed = float(np.mean(self.reduce_waits)) if self.reduce_waits else float("nan")
        small = self.small
        # proof of the group the step's collectives ran in: a one summed over the ranks through the SAME transport (RCCL's C API when the step
        # uses it), and what every rank's communicator says about itself
        probe = torch.ones(1, device=dev)
        if nc is not None:
            nc.allreduce_(probe, average=False)
        else:
            dist.all_reduce(probe)
        seen = float(probe.item())
        step_bytes = float((fo if self.sharded else lay["total"]) * 4)          # the flat gradient (the time slices' small sums are listed in small_collectives)
        t_ex = torch.tensor([iso, exposed, seen, float(nc.world if nc is not None else dist.get_world_size()),
                             float(nc.rank if nc is not None else dist.get_rank()), step_bytes], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t_ex) for _ in range(world)]
        dist.all_gather(gathered, t_ex)
        return {"allreduce_bytes": int((fo if self.sharded else lay["total"]) * 4),
                "per_rank": [{"rank": int(g[4]), "communicator_world_size": int(g[3]), "ranks_summed_by_a_probe_allreduce": int(round(float(g[2]))),
                              "bytes_allreduced_per_step": int(g[5])} for g in gathered],
                "chunks": [int(fo * 4)] if self.sharded else [int(fn * 4), int(fo * 4)],
                "graph_learner_time_slices": bool(self.sharded),
                "per_rank_allreduce_ms_isolated": [float(g[0]) for g in gathered],
                "per_rank_exposed_wait_ms": [float(g[1]) for g in gathered],
                "overlap_fraction": [float(1.0 - g[1] / g[0]) if float(g[0]) > 0 else None for g in gathered],
                "small_collectives": small,
                "collectives": ("rccl C API through libstep_hip (step_grad_allreduce_begin / _join, step_comm_allreduce), RCCL "
                                + str(nc.version)) if nc is not None else "torch.distributed (" + str(dist.get_backend()) + ")",
                "what": "isolated = the chunked all-reduces of the flat gradient alone; exposed = time the compute stream waits "
                        "for them at the end of backward (events around the waits), averaged over the timed steps; small_collectives = "
                        "count and exposed ms per step of the time-sliced graph learner's blocking sums"}

    def close(self):
        self.model.cancel_prefetch()
        del self.model, self.opt, self.batches, self.dser, self.params
        import gc
        gc.collect()
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)          # >= 20 warm-up, >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="STEP_PEMS04", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the reference config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-loader-figure", action="store_true", help="skip the second timed loop through the device-resident window loader")
    ap.add_argument("--no-runner-figure", action="store_true", help="skip the child runs that drive the config through the reference's runner loop")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no random-init / no-prefetch / other-config figures, no counter run")
    ap.add_argument("--other-configs", default=None, help="comma list of further configs measured after the headline (default: the three "
                    "north-star configs at N=1, none at N>1; '-' = none)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 counter passes (roofline.traffic falls back to the committed record)")
    ap.add_argument("--pretrain-steps", type=int, default=300, help="native TSFormer pre-training steps behind the loaded checkpoint (0: random init)")
    ap.add_argument("--prefetch", action="store_true", help="queue the frozen branch (encoder + kNN prior) of the next batch on its own stream "
                    "before this batch's backward (STEP.prefetch).  DEFAULT for the configs of ENC_SPLIT, where the encoder then runs as a "
                    "persistent launch on part of the compute units (--encoder-workgroups) next to the rest of the step")
    ap.add_argument("--no-prefetch", action="store_true", help="frozen branch inside forward(), encoder over the whole chip (the round-4 schedule)")
    ap.add_argument("--prefetch-early", action="store_true", help="queue the next batch's frozen branch at the start of the step (default at PEMS04)")
    ap.add_argument("--prefetch-late", action="store_true", help="queue the next batch's frozen branch behind this batch's forward (before its backward) "
                    "instead of at the start of the step")
    ap.add_argument("--prefetch-ahead", type=int, default=1, help="announce the frozen branch of batch i + k at the start of step i (2: the encoders of successive batches run back to back)")
    ap.add_argument("--pageable-origins", action="store_true", help="(A/B) copy the forecast origins from pageable host memory: the copy then blocks the host every step")
    ap.add_argument("--resident-batches", action="store_true", help="cycle eight resident input batches instead of the index-only device loader")
    ap.add_argument("--eval-dropout-off", action="store_true", help="disable dropout (parity runs)")
    ap.add_argument("--matmul", default="bf16", choices=["bf16", "f32"], help="operand precision of the GraphWaveNet / DGL contractions")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--forward-only", action="store_true",
                    help="validation / test path (SURVEY 8f-4): eval-mode forward + metric under no_grad, no backward / optimizer")
    ap.add_argument("--no-shard", action="store_true", help="--gpus > 1: keep the whole graph learner (and fc.weight) on every rank")
    ap.add_argument("--torch-optim", action="store_true", help="torch clip_grad_norm_ + torch.optim.Adam instead of the fused kernel")
    ap.add_argument("--graph", action="store_true", help="also measure the step replayed from one captured hipGraph (step_amd.GraphedTrainStep) in a "
                    "child process with the runtime's default hardware queues (measured slower than the eager three-stream schedule on this "
                    "stack: 4.72 vs 4.42 ms at PEMS04, profiles/r04_h_graph_replay_*.json)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="--gpus 1 only: create a one-rank process group and issue every collective of the data-parallel path (parameter "
                         "broadcast, chunked async all-reduce, the time-sliced graph learner's small sums) through it, so that RCCL, its "
                         "stream and the event ordering against the step's streams run on a one-GPU box; leaves GPU_MAX_HW_QUEUES at the "
                         "runtime default unless set explicitly")
    ap.add_argument("--encoder-workgroups", type=int, default=None,
                    help="persistent TSFormer encoder launch of at most this many workgroups (TSFormer.encoder_workgroups; a workgroup fills a "
                         "compute unit at 336 tokens, two share one at 168); 0 = one workgroup per sequence")
    ap.add_argument("--collectives", default="auto", choices=["auto", "rccl", "torch"],
                    help="data-parallel collectives of the step: RCCL C-API calls issued in stream order by libstep_hip (rccl; auto picks it on "
                         "the nccl backend) or torch.distributed.all_reduce (torch)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--graph-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child is not None:
        return pmc_child(args)
    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    stdout_is_for_the_json_line(rank)
    local = local % max(torch.cuda.device_count(), 1)          # (test rigs with fewer devices than ranks share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    DIST["on"] = world > 1 or args.force_process_group
    if DIST["on"]:
        import torch.distributed as dist
        if world == 1:          # --force-process-group: a group of one rank, rendezvous on the loopback address
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29581")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)         # RCCL over xGMI
        else:
            dist.init_process_group(args.backend)                  # e.g. gloo: exercises the data-parallel path on one device
    import step_amd._lib as L
    L.lib()
    workdir = tempfile.mkdtemp(prefix=f"step_bench_r{rank}_")
    N, Lh, B = cfg["N"], cfg["L"], cfg["B"]

    if cfg.get("pretrain"):
        r = pretrain_run(args, cfg, world, rank, dev, args.warmup, args.steps)
        if rank == 0:
            print_json_line({"metric": "TSFormer masked pre-training windows/s (config C3)", "value": r["value"], "unit": "windows/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "bf16 operands, f32 accumulate" if args.matmul == "bf16" else "f32", "data": "synthetic",
                              "config": {"workload": r["workload"], "global_batch": B * world, "parallelism": f"dp{world}",
                                         "final_loss": r["final_loss"]},
                              "whole_step": {"tflops": r["whole_step_tflops"], "frac_of_mfma_peak": r["whole_step_frac_of_mfma_peak"]},
                              "roofline": static_roofline(args.config), "dominant_kernels": static_dominant_kernels(args.config)})
        if DIST["on"]:
            torch.distributed.destroy_process_group()
        return

    # ---- the checkpoint the model loads: native pre-training, before any timed region
    ckpt, ckpt_info = None, None
    if args.pretrain_steps > 0:
        ckpt, ckpt_info = native_checkpoint(args.config.replace("STEP_", ""), cfg, synth_series(cfg["T_all"], N), args.pretrain_steps, dev, workdir)
    bench = StepBench(args.config, cfg, args, world, rank, dev, ckpt)
    res = bench.run(args.warmup, args.steps)
    nxt = args.warmup + args.steps
    extras = not args.no_extras
    enc_alone_ms = bench.encoder_alone_ms(nxt) if (not args.forward_only and extras) else None
    loader_fig = None
    if not args.forward_only and not args.no_loader_figure and extras:
        loader_fig = bench.alternative_figure(max(args.steps // 2, 10), nxt + 8, bench.prefetch, bench.enc_wgs, not bench.use_loader,
                                              ("same step fed by the index-only loader over the device-resident series (step_gather_windows inside the timed region)"
                                               if not bench.use_loader else
                                               "same step cycling eight RESIDENT input batches (the round 1-4 headline): no gather launches, no zero-history windows"))
    comm = bench.comm_figure() if (DIST["on"] and not args.forward_only) else None
    if args.graph_child:
        # child of graph_replay_figure(): the eager loop above and the replayed loop in ONE process with the runtime's default hardware queues
        gf = bench.graph_figure(max(args.steps, 10))
        gf["eager_same_process"] = {"value": res["value"], "ms_per_step": res["ms_per_step"], "host_enqueue_ms_per_step": res["host_enqueue_ms_per_step"]}
        print(json.dumps(gf), flush=True)
        return
    graph_fig = None
    if not args.forward_only and not DIST["on"] and not args.torch_optim and args.graph:
        graph_fig = graph_replay_figure(args.config, cfg["B"], ckpt, args)
    # ---- secondary figures of the same config: frozen branch inside forward(); random-init TSFormer
    no_prefetch, random_init = None, None
    short = max(min(args.steps // 3, 40), 5)
    if extras and not args.forward_only:
        if bench.prefetch:
            no_prefetch = bench.alternative_figure(short, nxt + 16, False, 0, bench.use_loader,
                                                   "same step with the frozen branch INSIDE forward() and the encoder over the whole chip (one workgroup per "
                                                   "sequence): the round-4 schedule, in this process (GPU_MAX_HW_QUEUES as in runtime_env; two queues suit it "
                                                   "1.6 % better)")
        else:
            no_prefetch = bench.alternative_figure(short, nxt + 16, True, ENC_SPLIT.get(args.config, 160), bench.use_loader,
                                                   "same step with the frozen branch of the NEXT batch prefetched next to this batch's backward and the "
                                                   "encoder as a persistent launch on part of the compute units (needs four hardware queues to pay)")
        if ckpt is not None:
            from step_amd import TSFormer
            torch.manual_seed(0)
            fresh = TSFormer(**tsformer_args(Lh, "forecasting")).state_dict()
            bench.model.tsformer.load_state_dict(fresh)
            r3 = bench.run(3, short, nxt + 64)
            random_init = {"value": r3["value"], "unit": "windows/s", "ms_per_step": r3["ms_per_step"], "steps": short,
                           "encoder_ms_per_launch": r3["enc_ms"], "fallback_units_per_launch": r3["fallback_units_per_launch"],
                           "what": "same step with the TSFormer at its random initialisation (no checkpoint): the encoder's softmax schedule is data dependent"}
            bench.model.tsformer.load_state_dict(ckpt_info["sd"])
    sharded = bench.sharded
    prefetch_on, enc_wgs, loader_on, early_on = bench.prefetch, bench.enc_wgs, bench.use_loader, bench.prefetch_early
    units = bench.softmax_units()
    data = bench.data
    bench.close()
    # ---- the other north-star configs (short loops, after and outside the headline's timed region)
    others = None
    names = args.other_configs
    if names is None:
        names = "STEP_PEMS07,SYNTH_4096,TSFormer_PEMS-BAY" if (world == 1 and extras and not args.forward_only and args.config == "STEP_PEMS04") else "-"
    if names != "-":
        others = {}
        for name in [n for n in names.split(",") if n]:
            try:
                # a fresh allocator cache per config: with the blocks the previous config left cached, the first steps of a larger config kept
                # going back to hipMalloc / hipFree inside the timed region (PEMS07 6.83 ms here against 6.07 ms in its own process,
                # profiles/r03_ag_*)
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                c2 = dict(CONFIGS[name])
                if c2.get("pretrain"):
                    r = pretrain_run(args, c2, world, rank, dev, 5, 15)
                    others[name] = {"value": r["value"], "unit": "windows/s", "ms_per_step": r["ms_per_step"], "steps": 15,
                                    "whole_step_frac_of_mfma_peak": r["whole_step_frac_of_mfma_peak"], "workload": r["workload"]}
                    others[name]["roofline"] = static_roofline(name)
                    dom = static_dominant_kernels(name)
                    if dom is not None:
                        others[name]["dominant_kernels"] = dom
                    if not args.no_cpu_baseline:
                        others[name]["cpu_baseline"] = cpu_baseline_pretrain(c2, synth_series(c2["T_all"], c2["N"]))
                else:
                    ck2 = None
                    if args.pretrain_steps > 0:
                        ck2, _ = native_checkpoint(name.replace("STEP_", ""), c2, synth_series(c2["T_all"], min(c2["N"], 512)), min(args.pretrain_steps, 100), dev, workdir)
                    b2 = StepBench(name, c2, args, world, rank, dev, ck2)
                    r = b2.run(8, 20)
                    fl = step_flops(c2, c2["B"])
                    others[name] = {"value": r["value"], "unit": "windows/s", "ms_per_step": r["ms_per_step"], "steps": 20,
                                    "encoder_ms_per_launch": r["enc_ms"], "fallback_units_per_launch": r["fallback_units_per_launch"],
                                    "host_enqueue_ms_per_step": r["host_enqueue_ms_per_step"], "device_allocs_in_timed_region": r["device_allocs"],
                                    "whole_step_tflops": fl / (r["ms_per_step"] * 1e-3) / 1e12,
                                    "whole_step_frac_of_mfma_peak": fl / (r["ms_per_step"] * 1e-3) / 1e12 / PEAK_TFLOPS,
                                    "workload": f"{name}: N={c2['N']}, L={c2['L']}, batch {c2['B']}/GPU, full train step, natively pre-trained TSFormer"}
                    efl = encoder_flops(c2, c2["B"])
                    others[name]["roofline"] = {"kernel": "tsformer_encoder_kernel", "bound": "mfma", "achieved": efl / (r["enc_ms"] * 1e-3) / 1e12,
                                                "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": efl / (r["enc_ms"] * 1e-3) / 1e12 / PEAK_TFLOPS,
                                                "ms_per_launch": r["enc_ms"], "algorithmic_flop_per_launch": efl,
                                                "traffic": (static_pmc_traffic(name, c2["B"]) or {}).get("hbm_bytes_per_launch"),
                                                "traffic_detail": static_pmc_traffic(name, c2["B"]),
                                                "note": "events around the encoder on its launch stream inside the timed steps (it shares the GPU with "
                                                        "the second stream's kernels there)"}
                    dom = static_dominant_kernels(name)
                    if dom is not None:
                        others[name]["dominant_kernels"] = dom
                    data2 = b2.data
                    b2.close()
                    if not args.no_cpu_baseline:
                        # one timed oracle step (plus a warm-up below 2048 nodes): ~10-40 s on 32 threads
                        big = c2["N"] >= 2048
                        others[name]["cpu_baseline"] = cpu_baseline(c2, data2, warm=0 if big else 1, timed=1, two_threads=False)
                    del data2
            except Exception as ex:          # noqa: BLE001 -- a failing extra must not take the headline line with it
                others[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    if rank == 0:
        flops = encoder_flops(cfg, B)
        enc_ms = res["enc_ms"]
        ach = flops / (enc_ms * 1e-3) / 1e12
        traffic = None
        if extras and not args.no_pmc and world == 1:
            traffic = live_pmc_traffic(args.config, B, ckpt, enc_wgs=enc_wgs)
        if traffic is None:
            traffic = static_pmc_traffic(args.config, B)
        wl = (f"{args.config}: N={N} nodes, long history L={Lh} (P={Lh // 12} patches), 12->12, train series T={cfg['T_train']}, batch {B}/GPU, "
              + (f"pre-trained TSFormer (native C3 path, {ckpt_info['steps']} steps, masked MAE {ckpt_info['first_loss']:.1f} -> {ckpt_info['last_loss']:.1f}) "
                 f"loaded from tsformer_ckpt/, " if ckpt_info else "random-init weights, ")
              + ("eval forward" if args.forward_only else "full train step (fwd+bwd+clip+Adam)")
              + (f", frozen branch of the next batch prefetched on its own stream (encoder: persistent launch on {enc_wgs} compute units)" if prefetch_on else "")
              + (", windows gathered on the device by forecast origin" if loader_on else ""))
        sfl = step_flops(cfg, B)
        out = {
            "metric": ("validation windows/sec (eval-mode forward + masked MAE)" if args.forward_only else
                       "training windows/sec on PEMS04, horizon-12 MAE parity, 1/2/4/8 MI355X"),
            "value": res["value"], "unit": "windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f16 (TSFormer Q/K/weights) + bf16 (P, V, GraphWaveNet / graph-learner GEMMs) operands, f32 accumulate" if args.matmul == "bf16"
                      else "f16/bf16 operands (TSFormer) + f32 (everything else), f32 accumulate"),
            "data": "synthetic",
            "config": {"workload": wl, "global_batch": B * world,
                       "parallelism": f"dp{world}" + (" + graph-learner time slices (fc.weight sharded)" if sharded else ""),
                       "final_loss": res["final_loss"]},
            "step_ms": {"p10": res["p10"], "p50": res["p50"], "p90": res["p90"]},
            "host_enqueue_ms_per_step": res["host_enqueue_ms_per_step"],
            "runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            "whole_step": {"algorithmic_flop": sfl, "tflops": sfl / (res["ms_per_step"] * 1e-3) / 1e12,
                           "frac_of_mfma_peak": sfl / (res["ms_per_step"] * 1e-3) / 1e12 / PEAK_TFLOPS},
            "roofline": {"kernel": "tsformer_encoder_kernel", "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS,
                         "traffic": (traffic or {}).get("hbm_bytes_per_launch"), "traffic_unit": "HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE)",
                         "traffic_detail": traffic, "ms_per_launch": enc_ms,
                         "launches_in_timed_region": res["enc_launches"],
                         "algorithmic_flop_per_launch": flops,
                         "ms_per_launch_alone": enc_alone_ms,
                         "achieved_alone": (flops / (enc_alone_ms * 1e-3) / 1e12) if enc_alone_ms else None,
                         "frac_alone": (flops / (enc_alone_ms * 1e-3) / 1e12 / PEAK_TFLOPS) if enc_alone_ms else None,
                         "fallback_units_per_launch": res["fallback_units_per_launch"], "softmax_units_per_launch": units,
                         "compute_units": (enc_wgs if enc_wgs else 256), "compute_units_of_chip": 256,
                         "frac_of_occupied_units": ach / PEAK_TFLOPS * 256.0 / (enc_wgs if enc_wgs else 256),
                         "note": "achieved / ms_per_launch: events around the kernel on its launch stream inside the timed steps; frac is against "
                                 "the WHOLE chip's peak.  With the frozen branch prefetched (the default here) the kernel is a persistent launch of "
                                 "`compute_units` workgroups -- one per compute unit -- running next to the rest of the step on the others, so its "
                                 "launch lasts longer on purpose (frac_of_occupied_units = frac x 256 / compute_units); *_alone: "
                                 "the same kernel over the whole chip (one workgroup per sequence) in 5 extra steps with nothing next to it; "
                                 "fallback_units: (32-token tile, head, layer, sequence) units whose softmax left the fixed-shift schedule"},
        }
        out["schedule"] = {"frozen_branch": ("prefetched: the encoder + kNN prior of batch i + 1 queued on their own stream "
                                             + ("at the start of step i" if early_on else "before the backward of batch i")
                                             + " (STEP.prefetch; bit-identical outputs)") if prefetch_on else "inside forward()",
                           "encoder_workgroups": enc_wgs, "input": "index-only loader over the device-resident series, one batch ahead (gather inside "
                           "the timed region)" if loader_on else "eight resident batches, cycled"}
        if no_prefetch is not None:
            out["other_schedule"] = no_prefetch
            try:        # the figure earlier rounds quoted as roofline.frac: the same kernel over the WHOLE chip inside forward(), next to the side stream
                ms4 = float(no_prefetch.get("encoder_ms_per_launch") or 0.0)
                if ms4 > 0.0 and ms4 == ms4:
                    out["roofline"]["whole_chip_in_step"] = {
                        "ms_per_launch": ms4, "achieved": flops / (ms4 * 1e-3) / 1e12, "frac": flops / (ms4 * 1e-3) / 1e12 / PEAK_TFLOPS,
                        "what": "the same kernel launched over the whole chip inside forward() (`other_schedule`, this process): how rounds 1-4 "
                                "measured roofline.frac; with the persistent launch on `compute_units` units `frac` is lower by construction"}
            except (TypeError, ValueError, KeyError):
                pass
        if random_init is not None:
            out["random_init"] = random_init
        if loader_fig is not None:
            out["other_input_feed"] = loader_fig
        if world == 1 and extras and not args.forward_only and args.config == "STEP_PEMS04" and not args.no_runner_figure:
            out.setdefault("other_input_feed", {})["reference_runner_loop"] = runner_feed_figure()
        if graph_fig is not None:
            out["graph_replay"] = graph_fig
        if comm is not None:
            out["data_parallel"] = comm
        if others is not None:
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            # the reference's own modules when their sources are on this box (tools/stage_reference.sh), with the oracle's figure next to it;
            # otherwise the oracle ("port")
            refb = cpu_baseline_reference(cfg, data)
            port = cpu_baseline(cfg, data, warm=1 if refb else 2, timed=3 if refb else 5)
            out["cpu_baseline"] = dict(refb, port=port) if refb else port
        print_json_line(out)
    shutil.rmtree(workdir, ignore_errors=True)
    if DIST["on"]:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
