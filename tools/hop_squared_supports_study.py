"""CPU study (no GPU, test infrastructure): the two diffusion orders of a GraphWaveNet layer from ONE input read.

Today a layer launches x1 = A x and then x2 = A x1 (model.py:10-16, 35-48: order 2, three supports), two dependent GEMM
launches per layer and two more in the backward.  With the squared supports A2 = A A built once per step, [x1; x2] = [A; A2] x
is one launch, dx = A^T g1 + A2^T g2 is one launch (two K segments), and the adjacency gradient gains the chain-rule term of A2:

    dA_total = dA + dA2 A^T + A^T dA2          (dA = g1 x^T and dA2 = g2 x^T summed over layers, samples' own matrices)

This script checks those identities against autograd on the PEMS04 shapes (N = 307, 32 channels, the 8 layers' time lengths)
and prices the operand rounding of the bf16 mode: rounding A2 once versus rounding the intermediate x1 (what the current
kernels do).  Output: profiles/r02_hop_squared_supports_study.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def hop(A, x):
    """(A x)[b, c, w, t] = sum_v x[b, c, v, t] A[b, v, w]   -- nconv of model.py:10-16 with a per-sample support"""
    return torch.einsum("bcvt,bvw->bcwt", x, A)


def main():
    torch.manual_seed(0)
    B, N, C = 2, 307, 32
    T_layers = [12, 10, 9, 7, 6, 4, 3, 1]
    dt = torch.float64
    A = torch.softmax(torch.randn(B, N, N, dtype=dt), dim=-1).requires_grad_(True)       # a row-stochastic support
    xs = [torch.randn(B, C, N, T, dtype=dt, requires_grad=True) for T in T_layers]
    g1 = [torch.randn(B, C, N, T, dtype=dt) for T in T_layers]
    g2 = [torch.randn(B, C, N, T, dtype=dt) for T in T_layers]
    # reference formulation: two dependent hops per layer, autograd
    loss = 0
    for x, a, b in zip(xs, g1, g2):
        x1 = hop(A, x)
        x2 = hop(A, x1)
        loss = loss + (x1 * a).sum() + (x2 * b).sum()
    loss.backward()
    dA_ref = A.grad.clone()
    dx_ref = [x.grad.clone() for x in xs]
    # squared-support formulation, explicit backward
    Ad = A.detach()
    A2 = Ad @ Ad
    err_fwd, err_dx = 0.0, 0.0
    dA = torch.zeros_like(Ad)
    dA2 = torch.zeros_like(Ad)
    for x, a, b, dxr in zip(xs, g1, g2, dx_ref):
        xd = x.detach()
        x1, x2 = hop(Ad, xd), hop(A2, xd)
        err_fwd = max(err_fwd, float((x2 - hop(Ad, x1)).norm() / x2.norm()))
        dx = torch.einsum("bcwt,bvw->bcvt", a, Ad) + torch.einsum("bcwt,bvw->bcvt", b, A2)     # one launch, two K segments
        err_dx = max(err_dx, float((dx - dxr).norm() / dxr.norm()))
        dA += torch.einsum("bcvt,bcwt->bvw", xd, a)                                              # the existing adjacency-gradient contraction
        dA2 += torch.einsum("bcvt,bcwt->bvw", xd, b)
    dA_total = dA + dA2 @ Ad.transpose(1, 2) + Ad.transpose(1, 2) @ dA2
    err_dA = float((dA_total - dA_ref).norm() / dA_ref.norm())
    # bf16 operand rounding (f32 accumulation): today A and x1 are rounded; with A2, A2 is rounded instead of x1
    r = lambda t: t.to(torch.bfloat16).to(dt)
    x = xs[0].detach()
    exact = hop(Ad, hop(Ad, x))
    today = hop(r(Ad), r(hop(r(Ad), r(x))))
    squared = hop(r(A2), r(x))
    rec = {"shapes": {"B": B, "N": N, "C": C, "T": T_layers},
           "forward_x2_vs_two_hops_f64": err_fwd, "dx_vs_autograd_f64": err_dx, "dA_vs_autograd_f64": err_dA,
           "bf16_operands_x2_rel_l2": {"two_hops_round_x1": float((today - exact).norm() / exact.norm()),
                                       "squared_round_A2": float((squared - exact).norm() / exact.norm())},
           "flops_extra_per_step": {"A2 = A A (24 matrices)": 24 * 2 * N ** 3, "dA chain-rule terms": 2 * 24 * 2 * N ** 3,
                                    "one hop launch at T=12 (3 supports x 8 samples)": 24 * 2 * N * N * C * 12}}
    out = os.path.join(ROOT, "profiles", "r02_hop_squared_supports_study.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))
    assert err_fwd < 1e-12 and err_dx < 1e-12 and err_dA < 1e-12


if __name__ == "__main__":
    sys.exit(main())
