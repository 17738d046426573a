"""GPU half of the drop-in proof.  tests/test_reference_runner_dropin.py ran the REFERENCE's STEPRunner / train_iters / dataset
(unmodified, in the build container) around step_amd.STEP built by the reference's STEP_METR-LA config and recorded the two
training losses in tests/golden/runner_metr_la.json.  The GPU box has no /root/reference, so here the same two iterations are
replayed on the device with the caller's arithmetic restated from the reference lines cited below: same synthetic files in a
scratch cwd, same constructor keywords, same seeds, same windows, same optimizer settings -- the losses must match the
reference runner's record (whose device arithmetic was stood in for by the fp32 oracle)."""
import json
import os

import pytest
import torch

from oracle import step_oracle as O
from tests import dropin_common as DC

pytestmark = pytest.mark.gpu
DS = "METR-LA"


def test_two_runner_iterations_match_the_reference_runner_record(tmp_path):
    from step_amd import STEP
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "runner_metr_la.json")) as f:
        gold = json.load(f)
    old = os.getcwd()
    series = DC.make_workspace(str(tmp_path), DS)
    os.chdir(str(tmp_path))
    try:
        param = DC.model_param(DS)
        param["tsformer_args"]["dropout"] = 0.0
        param["backend_args"]["dropout"] = 0.0
        torch.manual_seed(gold["init_seed"])
        model = STEP(**param).cuda()                       # reads datasets/METR-LA/... and tsformer_ckpt/... from the cwd
    finally:
        os.chdir(old)
    model.train()
    o = gold["optimizer"]
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=o["lr"], weight_decay=o["weight_decay"], eps=o["eps"])
    d = torch.from_numpy(series)
    N, _, L = DC.DATASETS[DS]
    losses = []
    for it in range(2):
        ts = gold["origins"][2 * it:2 * it + 2]
        # ForecastingDataset.__getitem__ (forecasting_dataset.py:62-71) + default collate + to_running_device (step_runner.py:57-63)
        hist = torch.stack([d[t - 12:t] for t in ts]).cuda()
        fut = torch.stack([d[t:t + 12] for t in ts]).cuda()
        longh = torch.stack([d[t - L:t] for t in ts]).cuda()
        torch.manual_seed(gold["gumbel_seed"] + it)
        model._noise_override = torch.rand(2, N * N, 2)     # discrete_graph_learning.py:12
        # STEPRunner.forward (step_runner.py:60-75): FORWARD_FEATURES [0,1,2], keyword call, TARGET_FEATURES [0]
        pred, theta, knn, coef = model(history_data=hist[..., [0, 1, 2]], long_history_data=longh[..., [0, 1, 2]], future_data=None,
                                       batch_seen=it, epoch=1)
        assert list(pred.shape)[:3] == [2, 12, N]
        pred, real = pred[..., [0]], fut[..., [0]]
        # train_iters (base_tsf_runner.py:237-250): re_standard_transform, curriculum slice, loss(*forward_return, null_val)
        pr, rl = pred * DC.STD + DC.MEAN, real * DC.STD + DC.MEAN
        cl = gold["cl_length"]
        loss = O.step_loss(pr[:, :cl], rl[:, :cl], theta, knn, coef, null_val=0.0)
        if it == 0:
            mae = float(O.masked_mae(pr[:, :cl], rl[:, :cl], 0.0))
        # easytorch Runner.backward: zero_grad, backward, clip_grad_norm_(model.parameters(), max_norm), step
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=o["max_norm"])
        opt.step()
        losses.append(float(loss.detach()))
    print("runner replay: losses", losses, "reference runner record", gold["losses"], "train_MAE", mae, gold["train_MAE_first"])
    assert losses[0] == pytest.approx(gold["losses"][0], rel=3e-3)
    assert mae == pytest.approx(gold["train_MAE_first"], rel=3e-3)
    assert losses[1] == pytest.approx(gold["losses"][1], rel=2e-2)
