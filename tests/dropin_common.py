"""Shared pieces of the drop-in tests (TEST INFRASTRUCTURE): a scratch working directory holding exactly the files the
reference reads with cwd-relative paths --

    datasets/<DS>/data_in12_out12.pkl   {"processed_data": float32 [T, N, 3]}          discrete_graph_learning.py:57, dataset
    datasets/<DS>/index_in12_out12.pkl  {"train" | "valid" | "test": [(t-12, t, t+12)]}  forecasting_dataset.py:29
    datasets/<DS>/scaler_in12_out12.pkl {"func": "re_standard_transform", "args": {"mean", "std"}}   base_tsf_runner.py:40
    tsformer_ckpt/TSFormer_<DS>.pt      {"model_state_dict": TSFormer(mode="pre-train").state_dict()}   step.py:31-32

-- and the model keyword arguments of the reference's config files (step/STEP_<DS>.py), restated so that the GPU box, which
has no /root/reference, can build the same module.  Everything is synthetic and seeded."""
import os
import pickle
import sys

import numpy as np
import torch

DATASETS = {   # name: (nodes, series length, long-history length) -- scripts/data_preparation/*/generate_training_data.py
    "METR-LA": (207, 34272, 288 * 7),
    "PEMS04": (307, 16992, 288 * 7 * 2),
}
MEAN, STD = 200.0, 150.0


def synth_series(T, N, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(T, dtype=np.float32)[:, None]
    phase = rng.uniform(0, 2 * np.pi, (1, N)).astype(np.float32)
    ch0 = np.sin(2 * np.pi * t / 288.0 + phase) + 0.5 * rng.standard_normal((T, N), dtype=np.float32)
    ch1 = np.broadcast_to((t % 288) / 288.0, (T, N))
    ch2 = np.broadcast_to((t // 288) % 7, (T, N))
    return np.stack([ch0, ch1, ch2], -1).astype(np.float32)


def model_param(ds):
    """CFG.MODEL.PARAM of step/STEP_<DS>.py (STEP_METR-LA.py:41-82, STEP_PEMS04.py:41-81)."""
    N, _, L = DATASETS[ds]
    return {
        "dataset_name": ds,
        "pre_trained_tsformer_path": f"tsformer_ckpt/TSFormer_{ds}.pt",
        "tsformer_args": {"patch_size": 12, "in_channel": 1, "embed_dim": 96, "num_heads": 4, "mlp_ratio": 4, "dropout": 0.1,
                          "num_token": L / 12, "mask_ratio": 0.75, "encoder_depth": 4, "decoder_depth": 1, "mode": "forecasting"},
        "backend_args": {"num_nodes": N, "support_len": 2, "dropout": 0.3, "gcn_bool": True, "addaptadj": True, "aptinit": None,
                         "in_dim": 2, "out_dim": 12, "residual_channels": 32, "dilation_channels": 32, "skip_channels": 256,
                         "end_channels": 512, "kernel_size": 2, "blocks": 4, "layers": 2},
        "dgl_args": {"dataset_name": ds, "k": 10, "input_seq_len": 12, "output_seq_len": 12},
    }


def train_origins(ds, n=6, seed=0):
    """a few forecast origins with a full long history (t >= L)"""
    N, T, L = DATASETS[ds]
    rng = np.random.default_rng(seed + 17)
    return sorted(int(t) for t in rng.choice(np.arange(L, T - 12), n, replace=False))


def make_workspace(root, ds="METR-LA", seed=0, n_train=6, full_history_only=True):
    """Write the four files under `root`; returns the series [T, N, 3].  n_train forecast origins (6: the recorded runs); with
    full_history_only=False they are drawn from the whole training split, windows without a full long history included (the reference's
    dataset hands those an all-zero history, forecasting_dataset.py:66-67)."""
    from step_amd.step_arch.tsformer import TSFormer
    N, T, L = DATASETS[ds]
    series = synth_series(T, N, seed)
    d = os.path.join(root, "datasets", ds)
    os.makedirs(d, exist_ok=True)
    os.makedirs(os.path.join(root, "tsformer_ckpt"), exist_ok=True)
    with open(os.path.join(d, "data_in12_out12.pkl"), "wb") as f:
        pickle.dump({"processed_data": series}, f)
    if full_history_only:
        idx = [(t - 12, t, t + 12) for t in train_origins(ds, n=n_train, seed=seed)]
    else:
        rng = np.random.default_rng(seed + 23)
        idx = [(int(t) - 12, int(t), int(t) + 12) for t in rng.choice(np.arange(12, int(0.6 * T)), n_train, replace=False)]
    with open(os.path.join(d, "index_in12_out12.pkl"), "wb") as f:
        pickle.dump({"train": idx, "valid": idx[:2], "test": idx[:2]}, f)
    with open(os.path.join(d, "scaler_in12_out12.pkl"), "wb") as f:
        pickle.dump({"func": "re_standard_transform", "args": {"mean": MEAN, "std": STD}}, f)
    torch.manual_seed(seed)
    a = dict(model_param(ds)["tsformer_args"], mode="pre-train")
    torch.save({"model_state_dict": TSFormer(**a).state_dict()}, os.path.join(root, "tsformer_ckpt", f"TSFormer_{ds}.pt"))
    return series


PKGS = ("step", "basicts", "easytorch", "easydict", "timm", "setproctitle")


class Workspace:
    """a scratch cwd holding the files the reference reads + the reference and the easytorch stand-ins on sys.path, for the duration of a
    `with` block; `config()` imports the reference's own config file and points CFG.MODEL.ARCH at step_amd.STEP"""
    def __init__(self, root, ds, **kw):
        self.root, self.ds = root, ds
        self.series = make_workspace(root, ds, **kw)

    def __enter__(self):
        self.old = os.getcwd()
        os.chdir(self.root)
        from oracle.reference_loader import reference_root
        self.added = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shims"), reference_root()]
        for p in self.added:
            sys.path.insert(0, p)
        self.saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in PKGS}
        for k in self.saved:
            del sys.modules[k]
        return self

    def __exit__(self, *exc):
        os.chdir(self.old)
        for p in self.added:
            sys.path.remove(p)
        for k in [k for k in sys.modules if k.split(".")[0] in PKGS]:
            del sys.modules[k]
        sys.modules.update(self.saved)

    def config(self, batch, dropout=False):
        import importlib
        cfg = importlib.import_module("step.STEP_" + self.ds).CFG          # the reference's config file
        from step_amd import STEP
        cfg.MODEL.ARCH = STEP
        if not dropout:
            cfg.MODEL.PARAM["tsformer_args"]["dropout"] = 0.0
            cfg.MODEL.PARAM["backend_args"]["dropout"] = 0.0
        cfg.TRAIN.DATA.BATCH_SIZE = batch
        cfg.TRAIN.DATA.SHUFFLE = False
        cfg["_DEVICE"] = "cuda"
        return cfg


