"""R1 / R2 in ONE process (missing #4): the REFERENCE's own config file, STEPRunner.forward,
BaseTimeSeriesForecastingRunner.train_iters, ForecastingDataset and scaler registry -- unmodified sources, found under /root/reference in
the build container or unpacked from oracle/_ref/reference.tar.gz on the GPU box (tools/stage_reference.sh; one git-ignored archive, shipped by gpurun) -- drive
``step_amd.STEP`` ON THE GPU through libstep_hip: ``CFG.MODEL.ARCH = step_amd.STEP`` and nothing else.  The two
training losses the runner returns must match the record the same runner produced around the fp32 oracle
(tests/golden/runner_metr_la.json, written by tests/test_reference_runner_dropin.py).

tests/_shims supplies the absent easytorch / easydict / setproctitle / timm names (test-only stand-ins, see its README).  Skipped where the
reference sources are not present."""
import importlib
import json
import os
import sys

import pytest
import torch

from oracle.reference_loader import reference_root
from tests import dropin_common as DC

REF = reference_root()
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(REF is None, reason="needs the reference sources (tools/stage_reference.sh stages them for the GPU box)")]
DS = "METR-LA"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "runner_metr_la.json")
PKGS = ("step", "basicts", "easytorch", "easydict", "timm", "setproctitle")


@pytest.fixture()
def workspace(tmp_path):
    root = str(tmp_path)
    series = DC.make_workspace(root, DS)
    old = os.getcwd()
    os.chdir(root)                       # the reference reads datasets/... and tsformer_ckpt/... relative to the cwd
    added = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shims"), REF]
    for p in added:
        sys.path.insert(0, p)
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in PKGS}
    for k in saved:
        del sys.modules[k]
    try:
        yield root, series
    finally:
        os.chdir(old)
        for p in added:
            sys.path.remove(p)
        for k in [k for k in sys.modules if k.split(".")[0] in PKGS]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_reference_runner_trains_the_hip_module_in_one_process(workspace):
    root, series = workspace
    with open(GOLDEN) as f:
        gold = json.load(f)
    cfg = importlib.import_module("step.STEP_METR-LA").CFG          # the reference's config file
    from step_amd import STEP
    cfg.MODEL.ARCH = STEP                                  # <- the whole integration
    cfg.MODEL.PARAM["tsformer_args"]["dropout"] = 0.0      # as in the recorded run: dropout cannot be bit-matched, two windows per step, fixed order
    cfg.MODEL.PARAM["backend_args"]["dropout"] = 0.0
    cfg.TRAIN.DATA.BATCH_SIZE = 2
    cfg.TRAIN.DATA.SHUFFLE = False
    cfg["_DEVICE"] = "cuda"
    torch.manual_seed(gold["init_seed"])
    runner = cfg.RUNNER(cfg)                               # STEPRunner -> BaseTimeSeriesForecastingRunner -> Runner.__init__ -> define_model
    native = runner.model
    assert isinstance(native, STEP) and next(native.parameters()).is_cuda
    N = DC.DATASETS[DS][0]
    seen = []

    def gumbel_like_the_record(module, args, kwargs):
        # the recorded run drew the Gumbel uniforms with torch.rand under seed gumbel_seed + batch_seen (discrete_graph_learning.py:12 draws
        # them on the host); nothing else of the call is touched
        seen.append((tuple(kwargs["history_data"].shape), tuple(kwargs["long_history_data"].shape), kwargs["batch_seen"], kwargs["epoch"],
                     kwargs["history_data"].device.type))
        torch.manual_seed(gold["gumbel_seed"] + kwargs["batch_seen"])
        module._noise_override = torch.rand(kwargs["history_data"].shape[0], N * N, 2)
        return None

    native.register_forward_pre_hook(gumbel_like_the_record, with_kwargs=True)
    losses = runner.train(cfg, max_iters=2)                # train_iters x 2 + easytorch's backward (zero_grad, backward, clip_grad_norm_, Adam.step)
    torch.cuda.synchronize()
    assert seen == [((2, 12, N, 3), (2, 2016, N, 3), 0, 1, "cuda"), ((2, 12, N, 3), (2, 2016, N, 3), 1, 1, "cuda")]
    print("reference runner around the HIP module: losses", losses, "record (same runner around the fp32 oracle)", gold["losses"],
          "train_MAE meter", runner.meters["train_MAE"].avg)
    assert losses[0] == pytest.approx(gold["losses"][0], rel=3e-3)
    assert losses[1] == pytest.approx(gold["losses"][1], rel=2e-2)
    assert runner.meters["train_MAE"].n == 2 and runner.meters["train_RMSE"].n == 2 and runner.meters["train_MAPE"].n == 2
    # the config's optimizer moved the native module's parameters (views of nothing: plain nn.Parameters here), the frozen TSFormer stayed
    assert all(p.grad is not None for n, p in native.named_parameters() if p.requires_grad and n.startswith("backend.start_conv"))
    assert not any(p.requires_grad for p in native.tsformer.parameters())
