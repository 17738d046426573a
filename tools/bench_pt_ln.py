"""Micro-benchmark of the fused LayerNorm kernels of the pre-training step at config C3's decoder / encoder row counts.
usage: [STEP_LN_BWD_BLOCKS=n] python tools/bench_pt_ln.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib as L  # noqa: E402
from tools.bench_pt_ffn import timed  # noqa: E402


def main():
    tag = "blocks=" + os.environ.get("STEP_LN_BWD_BLOCKS", "default")
    st = L.stream()
    for R in (5200 * 168, 5200 * 42):
        a, b, dy = [torch.randn(R, 96, device="cuda") for _ in range(3)]
        g, beta = torch.ones(96, device="cuda"), torch.zeros(96, device="cuda")
        pre, y, stats, dx, dxd = [torch.empty(R, 96, device="cuda") for _ in range(2)] + [torch.empty(R, 2, device="cuda")] + [torch.empty(R, 96, device="cuda") for _ in range(2)]
        dg, db, col = [torch.zeros(96, device="cuda") for _ in range(3)]
        t_f = timed(lambda: L.call("step_pt_add_layernorm_fwd", L.ptr(a), L.ptr(b), R, 0.1, 7, 3, L.ptr(g), L.ptr(beta), L.ptr(pre), L.ptr(y), L.ptr(stats), st))
        t_b = timed(lambda: L.call("step_pt_layernorm_bwd_dropout", L.ptr(dy), L.ptr(pre), R, L.ptr(g), L.ptr(stats), L.ptr(dx), L.ptr(dxd), 0.1, 7, 4,
                                   L.ptr(dg), L.ptr(db), L.ptr(col), st))
        gb = R * 96 * 4 * 4 / 1e3
        print(f"{tag} R={R}: add + dropout + LayerNorm forward {t_f:.0f} us ({gb / t_f:.0f} GB/s), backward + dropout {t_b:.0f} us ({gb / t_b:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
