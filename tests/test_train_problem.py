"""CPU check of the multi-step parity infrastructure (tests/train_problem.py): the oracle trainer learns the synthetic
problem, its BatchNorm bookkeeping matches torch.nn.BatchNorm's, and two runs with identical inputs are bit-identical."""
import torch

from tests import train_problem as TPb


def test_oracle_trainer_learns_and_is_deterministic():
    prob = TPb.Problem(N=10, L=96, T_train=150, n_train=12, n_eval=6)
    model = TPb.build_native(10, 96, 150, prob.series, k=3)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    hidden = prob.oracle_hidden(sd, prob.train_t + prob.eval_t)
    sched, noises = prob.schedule(25, 3), prob.noises(25, 3)
    l1, p1 = TPb.oracle_train(prob, sd, hidden, sched, noises, k=3)
    l2, p2 = TPb.oracle_train(prob, sd, hidden, sched, noises, k=3)
    assert l1 == l2
    assert sum(l1[-5:]) < 0.8 * sum(l1[:5])
    u = torch.rand(len(prob.eval_t), 100, 2, generator=torch.Generator().manual_seed(1))
    h12, mae = TPb.oracle_eval(prob, p1, hidden, u, 3)
    assert 0 < mae < 2 * prob.std and 0 < h12 < 2 * prob.std
    # running statistics moved away from their initial values (0 / 1) and stayed finite
    rm = p1["backend.bn.0.running_mean"]
    assert torch.isfinite(rm).all() and float(rm.abs().max()) > 0


def test_running_stat_update_matches_torch_batchnorm():
    from oracle import step_oracle as O
    x = torch.randn(5, 4, 7, 3)
    bn = torch.nn.BatchNorm2d(4)
    bn.train()
    y = bn(x)
    w, b = bn.weight.detach(), bn.bias.detach()
    yo, mu, var_u = O.batch_norm_train(x, w, b, (0, 2, 3))
    p = {"backend.bn.0.running_mean": torch.zeros(4), "backend.bn.0.running_var": torch.ones(4)}
    TPb.update_running_stats(p, {"bn.0": (mu, var_u)})
    assert torch.allclose(yo, y, atol=1e-5)
    assert torch.allclose(p["backend.bn.0.running_mean"], bn.running_mean, atol=1e-6)
    assert torch.allclose(p["backend.bn.0.running_var"], bn.running_var, atol=1e-6)
