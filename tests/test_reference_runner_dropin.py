"""Drop-in proof on the caller side (BASELINE.json config 1 "STEP_METR-LA ... batch=2 on CPU via
BasicTS runner (plumbing, no GPU)"): the REFERENCE's own config file, STEPRunner, BaseTimeSeriesForecastingRunner.train_iters,
ForecastingDataset and scaler registry run unmodified (tests/_shims supplies the absent easytorch / easydict / setproctitle /
timm names) around ``step_amd.STEP`` wired in the way INTEGRATION.md describes -- ``CFG.MODEL.ARCH = step_amd.STEP`` and nothing
else.  There is no GPU in the build container and libstep_hip has no CPU path, so the device arithmetic is stood in for by
the oracle THROUGH THE NATIVE MODULE'S OWN PARAMETERS (the module surface, its state_dict, the keyword call, the returned tuple
and the loss hook are the real ones); the numbers the reference runner produces are stored in
tests/golden/runner_metr_la.json, and tests/test_gpu_runner_golden.py replays the same two iterations on the GPU against them.

Needs the reference sources: /root/reference, or the staged archive (oracle/reference_loader.py)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import step_oracle as O
from tests import dropin_common as DC
from tests.train_problem import update_running_stats

from oracle.reference_loader import reference_root

REF = reference_root()              # the build container's checkout, or the archive tools/stage_reference.sh packs, unpacked
pytestmark = pytest.mark.skipif(REF is None, reason="needs the reference sources")
DS = "METR-LA"
GUMBEL_SEED = 1234
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "runner_metr_la.json")


class OracleBackedSTEP(torch.nn.Module):
    """Stand-in for the device arithmetic only: same parameters (the native module's own nn.Parameters, so autograd fills their
    .grad), same keyword signature and return tuple as STEP.forward; the math is the oracle's."""

    def __init__(self, native):
        super().__init__()
        self.native = native
        self.calls = []

    def forward(self, history_data, long_history_data, future_data, batch_seen, epoch, **kwargs):
        assert not kwargs and future_data is None
        B, L12, N, C = history_data.shape
        self.calls.append(dict(hist=tuple(history_data.shape), long=tuple(long_history_data.shape), batch_seen=batch_seen, epoch=epoch,
                               training=self.training))
        p = dict(self.native.named_parameters())
        p.update(dict(self.native.named_buffers()))
        torch.manual_seed(GUMBEL_SEED + batch_seen)                   # (the DataLoader iterator draws from the same generator)
        u = torch.rand(B, N * N, 2)                                   # discrete_graph_learning.py:12 (CPU generator)
        stats = {}
        dgl = self.native.discrete_graph_learning
        pred, theta, knn, coef = O.step_forward(history_data, long_history_data[..., [0]], dgl.node_feats, p, u, dgl.k, epoch,
                                                training=self.training, stats=stats)
        if self.training:
            update_running_stats(p, stats)
        return pred, theta, knn, coef


@pytest.fixture(scope="module")
def workspace(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("dropin"))
    series = DC.make_workspace(root, DS)
    old = os.getcwd()
    os.chdir(root)                       # the reference reads datasets/... and tsformer_ckpt/... relative to the cwd
    added = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shims"), REF]
    for p in added:
        sys.path.insert(0, p)
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("step", "basicts", "easytorch", "easydict", "timm", "setproctitle")}
    for k in saved:
        del sys.modules[k]
    try:
        yield root, series
    finally:
        os.chdir(old)
        for p in added:
            sys.path.remove(p)
        for k in [k for k in sys.modules if k.split(".")[0] in ("step", "basicts", "easytorch", "easydict", "timm", "setproctitle")]:
            del sys.modules[k]
        sys.modules.update(saved)


def _cfg():
    cfg = importlib.import_module("step.STEP_METR-LA").CFG
    from step_amd import STEP
    cfg.MODEL.ARCH = STEP                                  # <- the whole integration
    # determinism of the recorded numbers: dropout off (it cannot be bit-matched), two windows per step, fixed order, CPU
    cfg.MODEL.PARAM["tsformer_args"]["dropout"] = 0.0
    cfg.MODEL.PARAM["backend_args"]["dropout"] = 0.0
    cfg.TRAIN.DATA.BATCH_SIZE = 2
    cfg.TRAIN.DATA.SHUFFLE = False
    cfg["_DEVICE"] = "cpu"
    return cfg


def test_reference_config_builds_native_module_with_reference_state_dict(workspace):
    """D1 / S2: cfg.MODEL.ARCH(**cfg.MODEL.PARAM) (base_runner.py:49-51) with the reference's exact keyword arguments -- the
    four-key dgl_args, the cwd-relative data file, the pre-trained checkpoint -- and the resulting state_dict against the
    REFERENCE STEP built from the same config in the same directory."""
    cfg = _cfg()
    from step.step_arch import STEP as RefSTEP
    assert cfg.MODEL.PARAM == DC.model_param(DS) | {"tsformer_args": dict(DC.model_param(DS)["tsformer_args"], dropout=0.0),
                                                    "backend_args": dict(DC.model_param(DS)["backend_args"], dropout=0.0)}
    torch.manual_seed(0)
    native = cfg.MODEL.ARCH(**cfg.MODEL.PARAM)
    ref = RefSTEP(**cfg.MODEL.PARAM)
    rs, ns = ref.state_dict(), native.state_dict()
    assert list(rs.keys()) == list(ns.keys())
    for k in rs:
        assert rs[k].shape == ns[k].shape and rs[k].dtype == ns[k].dtype, k
    assert ns["discrete_graph_learning.fc.weight"].shape == (100, 383552)          # discrete_graph_learning.py:61
    assert torch.equal(native.discrete_graph_learning.node_feats, ref.discrete_graph_learning.node_feats)     # :57
    # the pre-trained TSFormer was loaded and frozen in both (step.py:27-35)
    ck = torch.load("tsformer_ckpt/TSFormer_METR-LA.pt")["model_state_dict"]
    for k, v in ck.items():
        assert torch.equal(ns["tsformer." + k], v) and torch.equal(rs["tsformer." + k], v), k
    assert not any(p.requires_grad for p in native.tsformer.parameters())
    assert {n for n, p in native.named_parameters() if p.requires_grad} == {n for n, p in ref.named_parameters() if p.requires_grad}
    native.load_state_dict(rs, strict=True)               # reference checkpoints load unchanged, and back
    ref.load_state_dict(native.state_dict(), strict=True)


def test_reference_runner_trains_native_module(workspace):
    """R1 / R2: two iterations of the reference's training loop body -- ForecastingDataset.__getitem__, STEPRunner.forward,
    BaseTimeSeriesForecastingRunner.train_iters (re-scaling, curriculum slice, step_loss through metric_forward, the three
    metric .item() calls) -- followed by zero_grad / backward / clip_grad_norm_ / Adam.step with the config's settings."""
    root, series = workspace
    cfg = _cfg()
    torch.manual_seed(0)
    runner = cfg.RUNNER(cfg)                               # STEPRunner -> ... -> Runner.__init__ -> define_model
    from step_amd import STEP
    native = runner.model
    assert isinstance(native, STEP)
    with pytest.raises(RuntimeError, match="no CPU fallback"):          # the product has no CPU path: it says so
        native(history_data=torch.zeros(1, 12, 207, 3), long_history_data=torch.zeros(1, 2016, 207, 3), future_data=None, batch_seen=0, epoch=1)
    runner.model = OracleBackedSTEP(native)
    before = {n: p.detach().clone() for n, p in native.named_parameters() if p.requires_grad}
    losses = runner.train(cfg, max_iters=2)
    calls = runner.model.calls
    assert [c["hist"] for c in calls] == [(2, 12, 207, 3)] * 2 and [c["long"] for c in calls] == [(2, 2016, 207, 3)] * 2
    assert [c["batch_seen"] for c in calls] == [0, 1] and [c["epoch"] for c in calls] == [1, 1] and all(c["training"] for c in calls)
    assert runner.iter_per_epoch == 3 and all(np.isfinite(losses))
    # the runner's numbers against a direct evaluation: first iteration, curriculum length 1 at epoch 1 (base_tsf_runner.py:170-190)
    origins = DC.train_origins(DS)
    d = torch.from_numpy(series)
    hist = torch.stack([d[t - 12:t] for t in origins[:2]])
    fut = torch.stack([d[t:t + 12] for t in origins[:2]])
    longh = torch.stack([d[t - 2016:t] for t in origins[:2]])
    torch.manual_seed(0)
    fresh = cfg.MODEL.ARCH(**cfg.MODEL.PARAM)
    p = dict(fresh.named_parameters()) | dict(fresh.named_buffers())
    torch.manual_seed(GUMBEL_SEED)
    u = torch.rand(2, 207 * 207, 2)
    pred, theta, knn, coef = O.step_forward(hist, longh[..., [0]], fresh.discrete_graph_learning.node_feats, p, u, 10, 1, training=True)
    want = O.step_loss(O.rescale(pred, DC.MEAN, DC.STD)[:, :1], O.rescale(fut[..., [0]], DC.MEAN, DC.STD)[:, :1], theta, knn, coef)
    assert losses[0] == pytest.approx(float(want), rel=1e-5)
    mae0 = float(O.masked_mae(O.rescale(pred, DC.MEAN, DC.STD)[:, :1], O.rescale(fut[..., [0]], DC.MEAN, DC.STD)[:, :1], 0.0))
    assert runner.meters["train_MAE"].n == 2 and runner.meters["train_RMSE"].n == 2 and runner.meters["train_MAPE"].n == 2
    # the optimizer of the config moved every parameter that has a gradient; the frozen TSFormer and the dead tensors stayed
    moved = {n for n, q in native.named_parameters() if q.requires_grad and not torch.equal(q.detach(), before[n])}
    dead = {n for n, q in native.named_parameters() if q.requires_grad and q.grad is None}
    assert dead == {n for n in before if n.startswith("backend.residual_convs") or n.startswith("backend.bn.7") or
                    n.startswith("backend.gconv.7") or n.startswith("discrete_graph_learning.fc_mean")}
    assert moved == set(before) - dead
    rec = {"dataset": DS, "origins": origins[:4], "losses": losses, "train_MAE_first": mae0, "cl_length": 1, "gumbel_seed": GUMBEL_SEED,
           "init_seed": 0, "optimizer": dict(cfg.TRAIN.OPTIM.PARAM, max_norm=cfg.TRAIN.CLIP_GRAD_PARAM["max_norm"])}
    assert cfg.TRAIN.OPTIM.TYPE == "Adam" and rec["optimizer"] == {"lr": 0.005, "weight_decay": 1.0e-5, "eps": 1.0e-8, "max_norm": 3.0}
    if os.environ.get("STEP_WRITE_GOLDEN") == "1":
        with open(GOLDEN, "w") as f:
            json.dump(rec, f, indent=1)
    with open(GOLDEN) as f:
        gold = json.load(f)
    assert gold["origins"] == rec["origins"]
    assert gold["losses"] == pytest.approx(losses, rel=1e-4)          # the committed record is what this code produces
