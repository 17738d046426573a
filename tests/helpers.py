"""Shared helpers for the test-suite (loading golden fixtures, error metrics)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name, dtype=torch.float32):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    g = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in "fc":
            g[k] = torch.from_numpy(v.astype(np.float64)).to(dtype)
        elif v.dtype.kind in "iu":
            g[k] = torch.from_numpy(v.astype(np.int64))
        else:
            g[k] = v
    return g


def params_of(g, requires_grad=True, dtype=None):
    p = {}
    for k, v in g.items():
        if k.startswith("param."):
            t = v.clone() if dtype is None or not v.is_floating_point() else v.to(dtype)
            if requires_grad and t.is_floating_point() and not k.startswith("param.tsformer.") \
                    and "running_" not in k:
                t.requires_grad_(True)
            p[k[len("param."):]] = t
    return p


def rel_l2(a, b):
    a = a.detach().double().flatten()
    b = b.detach().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())
