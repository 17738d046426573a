"""Micro-benchmark of the fused TSFormer encoder kernel alone (C2 shapes by default).
usage: python tools/bench_encoder.py [S P iters]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd.step_arch.tsformer import TSFormer  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2456
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 336
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    torch.manual_seed(0)
    m = TSFormer(12, 1, 96, 4, 4, 0.1, P, 0.75, 4, 1, mode="forecasting").cuda()
    x = torch.randn(S, P * 12, device="cuda")
    flops = S * P * (4 * (221184 + 384 * P) + 2304)
    for drop in (False, True):
        m.train(drop)
        for _ in range(3):
            m.encode_series(x)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(iters):
            m.encode_series(x)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / iters
        print(f"S={S} P={P} dropout={drop}: {ms:.3f} ms/launch  {flops / ms / 1e9:.1f} TFLOP/s ({flops / ms / 1e9 / 25:.2f}% of 2.5 PF)", flush=True)


if __name__ == "__main__":
    main()
