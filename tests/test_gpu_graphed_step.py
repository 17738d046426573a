"""step_amd.GraphedTrainStep: one training iteration (zero_grad, forward, step_loss, backward, clip + Adam) captured into a hipGraph
and replayed.  Checks, on the reference-golden problem `step_small` (37 nodes, so the bf16 / f32 paths and all three streams are
on the path):
* a replay computes the eager step's loss and gradient at the same parameters (dropout off, explicit Gumbel noise): a twin model
  is set to the graphed model's parameters before every replay and stepped eagerly;
* what must change from replay to replay does -- Adam's step count, the keep-mask pool, the Gumbel sample, the gcn dropout masks --
  although the graph's launch arguments are frozen (the device-resident StepDynState, include/step_hip.h);
* learning rate and graph-term coefficient written by the host between replays take effect."""
import numpy as np
import pytest
import torch

from tests.helpers import load_golden, rel_l2
from tests.test_gpu_step import build_native, inputs_of

pytestmark = pytest.mark.gpu


def _setup(mode, dropout):
    from step_amd.optim import FusedAdamClip
    g = load_golden("step_small")
    torch.manual_seed(0)
    m = build_native(g)
    m.train()
    m.matmul_precision = mode
    if not dropout:
        m.backend.dropout = 0.0
        m.tsformer.dropout_p = 0.0
        m._noise_override = g["in.u"].cuda().float().contiguous()
    opt = FusedAdamClip(m, lr=2e-3, weight_decay=1e-5, eps=1e-8, max_norm=3.0)
    return g, m, opt


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_replay_matches_eager_step(mode):
    from step_amd import GraphedTrainStep
    from step_amd.step_loss import step_loss_native
    g, model, opt = _setup(mode, dropout=False)
    _, twin, topt = _setup(mode, dropout=False)
    mean, std = [float(x) for x in g["meta.scaler"]]
    hist, long_hist, fut = inputs_of(g)
    step = GraphedTrainStep(model, opt, (hist, long_hist, fut), scaler=(mean, std), epoch=1, warmup=2)
    assert step.state()["adam_step"] == 2
    rng = np.random.default_rng(0)
    for k in range(3):
        # a different batch every time: the static input buffers are refilled, not re-captured
        hk = hist + torch.from_numpy(rng.normal(size=tuple(hist.shape)).astype(np.float32)).cuda() * 0.05
        lk = long_hist.clone()
        lk[..., 0] += torch.from_numpy(rng.normal(size=tuple(long_hist.shape[:-1])).astype(np.float32)).cuda() * 0.05
        with torch.no_grad():
            twin._flat_param.copy_(model._flat_param)                       # the parameters this replay starts from
        loss = step(hk, lk, fut)
        g_graph = model._flat_grad.clone()
        topt.zero_grad(set_to_none=True)
        pred, theta, knn, coef = twin(history_data=hk, long_history_data=lk, future_data=None, batch_seen=k, epoch=1)
        l2 = step_loss_native(pred[..., :1], fut[..., :1], theta, knn, coef, null_val=0.0, rescale=(mean, std))
        l2.backward()
        torch.cuda.synchronize()
        e_loss = abs(float(loss) - float(l2)) / abs(float(l2))
        e_grad = rel_l2(g_graph.cpu(), twin._flat_grad.cpu())
        print(f"[{mode}] replay {k}: loss {float(loss):.6f} vs eager {float(l2):.6f} (rel {e_loss:.1e}), flat gradient rel-L2 {e_grad:.2e}")
        assert e_loss < 1e-5 and e_grad < (2e-4 if mode == "f32" else 2e-3)
        assert not torch.equal(twin._flat_param, model._flat_param)       # the replay's Adam moved the parameters
    st = step.state()
    assert st["adam_step"] == 5 and step.replays == 3
    # learning rate and graph-term coefficient: host writes between replays
    before = model._flat_param.clone()
    step(hist, long_hist, fut)
    d_full = float((model._flat_param - before).norm())
    step.set_lr(2e-5)
    before = model._flat_param.clone()
    step(hist, long_hist, fut)
    d_small = float((model._flat_param - before).norm())
    print(f"[{mode}] |delta params| at lr 2e-3: {d_full:.4e}, at lr 2e-5: {d_small:.4e}")
    assert 30.0 < d_full / d_small < 300.0
    l_e1 = float(step(hist, long_hist, fut))
    step.set_epoch(13)                                                       # coefficient 1 / (13 // 6 + 1) = 1/3
    l_e13 = float(step(hist, long_hist, fut))
    assert abs(step.state()["gsl_coef"] - 1.0 / 3.0) < 1e-6 and l_e13 < l_e1
    step.close()
    assert opt.dyn is None and model._dyn is None and opt.step_count == step.state()["adam_step"]
    # and the model steps eagerly again
    opt.zero_grad(set_to_none=True)
    pred, theta, knn, coef = model(history_data=hist, long_history_data=long_hist, future_data=None, batch_seen=0, epoch=1)
    step_loss_native(pred[..., :1], fut[..., :1], theta, knn, coef, null_val=0.0, rescale=(mean, std)).backward()
    opt.step()
    assert opt.step_count == step.state()["adam_step"] + 1


def test_replays_draw_fresh_randomness():
    """dropout and the device Gumbel stream on: the graph is frozen, the random streams are not"""
    from step_amd import GraphedTrainStep
    import step_amd._lib as L
    g, model, opt = _setup("bf16", dropout=True)
    mean, std = [float(x) for x in g["meta.scaler"]]
    hist, long_hist, fut = inputs_of(g)
    B, _, N, _ = hist.shape
    step = GraphedTrainStep(model, opt, (hist, long_hist, fut), scaler=(mean, std), epoch=1, warmup=2)
    off = L.lib().step_gwnet_saved_offset(B, N, 1, 0, 0)                      # dropout mask of layer 0 inside the saved buffer
    seen = []
    for k in range(3):
        step(hist, long_hist, fut)
        torch.cuda.synchronize()
        pool = model.tsformer._drop_pool.clone()
        adj = model._last["sampled_adj"].clone()
        mask = model._last["saved_gwnet"][off:off + B * N * 12 * 32].clone()
        assert set(mask.unique().tolist()) <= {0.0, 1.0 / 0.7} or torch.allclose(mask[mask > 0], torch.tensor(1.0 / 0.7, device=mask.device))
        seen.append((pool, adj, mask, step.state()["seed_xor"]))
    for a in range(3):
        for b in range(a + 1, 3):
            assert not torch.equal(seen[a][0], seen[b][0]), "the keep-mask pool did not change between replays"
            assert not torch.equal(seen[a][1], seen[b][1]), "the Gumbel sample did not change between replays"
            assert not torch.equal(seen[a][2], seen[b][2]), "the gcn dropout mask did not change between replays"
            assert seen[a][3] != seen[b][3]
    keep = float((seen[0][2] > 0).float().mean())
    assert 0.6 < keep < 0.8
    step.close()
