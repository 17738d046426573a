"""Pin the oracle (oracle/step_oracle.py) against outputs of the reference's own modules
(tests/golden/*.npz, produced by tools/make_golden.py inside the build container)."""
import pytest
import torch

from oracle import step_oracle as O
from tests.helpers import load_golden, params_of, rel_l2, max_abs


def _run_step(g, dtype):
    N, L, T_train, B, k, epoch, training = [int(x) for x in g["meta"]]
    p = params_of(g, dtype=dtype)
    mean, std = [float(x) for x in g["meta.scaler"]]
    stats = {}
    pred, theta, knn, coef = O.step_forward(g["in.hist"].to(dtype), g["in.long_hist0"].to(dtype).unsqueeze(-1),
                                            g["in.node_feats"].to(dtype), p, g["in.u"].to(dtype), k,
                                            epoch if epoch >= 0 else None, training=bool(training), stats=stats)
    loss = O.step_loss(O.rescale(pred, mean, std), O.rescale(g["in.future"].to(dtype)[..., [0]], mean, std),
                       theta, knn, coef)
    return p, pred, theta, knn, coef, loss, stats


@pytest.mark.parametrize("name", ["step_tiny", "step_small", "step_tiny_eval"])
def test_step_forward_matches_reference(name):
    g = load_golden(name)
    p, pred, theta, knn, coef, loss, _ = _run_step(g, torch.float32)
    assert coef == pytest.approx(float(g["meta.coef"]))
    # The kNN restatement itself is exact on the reference's hidden states ...
    N, B, k = int(g["meta"][0]), int(g["meta"][3]), int(g["meta"][4])
    knn_ref_h, sim = O.cosine_knn_graph(g["out.hidden"].reshape(B, N, -1), k * N)
    assert torch.equal(knn_ref_h, g["out.knn"]), "kNN prior graph differs from the reference"
    # ... and end-to-end it may differ only where the cut falls on near-ties (hard parts)
    diff = (knn != g["out.knn"]).nonzero()
    assert diff.shape[0] <= 2 * B
    kth = torch.topk(sim.reshape(B, -1), k * N, -1).values[:, -1]
    for b, i, j in diff.tolist():
        assert abs(float(sim[b, i, j] - kth[b])) < 1e-4
    assert max_abs(theta, g["out.theta"]) < 2e-5
    assert rel_l2(pred, g["out.pred"]) < 2e-4
    assert float(loss) == pytest.approx(float(g["out.loss"]), rel=2e-4)


@pytest.mark.parametrize("name", ["step_tiny", "step_small"])
def test_hidden_matches_reference(name):
    g = load_golden(name)
    p = params_of(g, requires_grad=False)
    h = O.tsformer_encode(g["in.long_hist0"], p)
    assert max_abs(h, g["out.hidden"]) < 5e-5


@pytest.mark.parametrize("name", ["step_tiny", "step_small"])
def test_step_grads_match_reference(name):
    g = load_golden(name)
    p, pred, theta, knn, coef, loss, stats = _run_step(g, torch.float32)
    loss.backward()
    nograd = set(str(s) for s in g["meta.nograd"])
    checked = 0
    for k, v in g.items():
        if not k.startswith("grad."):
            continue
        n = k[len("grad."):]
        assert p[n].grad is not None, n
        # biases feeding a train-mode BatchNorm have an analytically zero gradient: both sides
        # are round-off there, so allow an absolute floor
        if float(v.abs().max()) < 1e-4:
            assert max_abs(p[n].grad, v) < 1e-4, n
        else:
            err = rel_l2(p[n].grad, v)
            assert err < 5e-3, (n, err)
        checked += 1
    assert checked > 40
    for n in nograd:
        assert p[n].grad is None or float(p[n].grad.abs().max()) == 0.0, n
    # running statistics after one training forward (momentum 0.1, unbiased variance)
    for k, v in g.items():
        if k.startswith("after.") and k.endswith("running_mean"):
            mod = k[len("after."):-len(".running_mean")]
            key = mod.split(".", 1)[1] if mod.startswith("backend.") else mod.split(".")[-1]
            if key not in stats:
                continue           # bn.7: dead layer, see DESIGN.md
            mu, var_u = stats[key]
            want_m = 0.9 * g["param." + mod + ".running_mean"] + 0.1 * mu
            want_v = 0.9 * g["param." + mod + ".running_var"] + 0.1 * var_u
            assert max_abs(want_m, v) < 1e-4, k
            assert rel_l2(want_v, g["after." + mod + ".running_var"]) < 1e-4, k


def test_pretrain_matches_reference():
    g = load_golden("tsformer_pretrain_tiny")
    p = {k[len("param."):]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("param.")}
    pp = {"tsformer." + k: v for k, v in p.items()}
    recon, label = O.tsformer_pretrain(g["in.x"], pp, g["in.unmasked"].tolist(), g["in.masked"].tolist())
    assert max_abs(label, g["out.label"]) == 0.0
    assert max_abs(recon, g["out.recon"]) < 5e-5
    loss = O.masked_mae(recon * 150.0 + 200.0, label * 150.0 + 200.0, 0.0)
    assert float(loss) == pytest.approx(float(g["out.loss"]), rel=1e-4)
    loss.backward()
    n = 0
    for k, v in g.items():
        if k.startswith("grad."):
            err = rel_l2(p[k[len("grad."):]].grad, v)
            assert err < 5e-3, (k, err)
            n += 1
    assert n > 50


def _dropout_masks_of(g):
    t = lambda a: torch.from_numpy(a.astype("float32"))
    return {"pos": t(g["mask.pos"]),
            "layers": [{k: t(g[f"mask.{l}.{k}"]) for k in ("attn", "drop1", "ffn", "drop2")} for l in range(4)]}


def test_training_mode_dropout_sites_match_reference():
    """The reference TSFormer in train mode with every dropout realisation recorded (tools/make_golden.py run_dropout_case):
    the oracle replaying those masks must reproduce its hidden states, i.e. the five kinds of dropout sites
    (positional_encoding.py:32; attention probabilities, dropout1, FFN, dropout2 of each encoder layer) sit where the
    reference's do and scale survivors the same way."""
    g = load_golden("tsformer_dropout_tiny")
    N, L, B = [int(x) for x in g["meta"]]
    p = {"tsformer." + k[len("param."):]: v for k, v in g.items() if k.startswith("param.")}
    keep = 1.0 - float(g["meta.p"])
    x = g["in.x"][..., 0]
    h = O.tsformer_encode(x, p, drop=_dropout_masks_of(g), keep=keep)
    assert max_abs(h, g["out.hidden"]) < 1e-4
    # the masks matter: without them the states differ visibly
    assert rel_l2(O.tsformer_encode(x, p), g["out.hidden"]) > 0.05
