"""Micro-benchmark of step_gemm on the GraphWaveNet diffusion-hop shape and friends."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib as L

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

def main():
    B, N, T, CAT, C = 8, 307, 12, 224, 32
    P = torch.rand(B, N, N, device="cuda")
    cat = torch.randn(B, N, T, CAT, device="cuda")
    # nconv forward: Out[b][w][n] = sum_v P[b][v][w] X[b][v][n]  with slot remaps
    f = lambda: L.gemm(P, cat, cat, N, T * C, N, 1, N, T * CAT, 1, T * CAT, batch=B, sab=N * N, sbb=N * T * CAT, scb=N * T * CAT,
                       b_off=0, c_off=32, b_n=(32, CAT), c_n=(32, CAT))
    us = timeit(f); fl = 2.0 * B * N * N * T * C
    print(f"nconv fwd (TN, remap)  {us:8.1f} us  {fl / us / 1e6:7.2f} TF/s")
    X = torch.randn(B, N, T * C, device="cuda"); O = torch.empty(B, N, T * C, device="cuda")
    f = lambda: L.gemm(P, X, O, N, T * C, N, 1, N, T * C, 1, T * C, batch=B, sab=N * N, sbb=N * T * C, scb=N * T * C)
    us = timeit(f); print(f"nconv fwd (TN, dense)  {us:8.1f} us  {fl / us / 1e6:7.2f} TF/s")
    f = lambda: L.gemm(P, X, O, N, T * C, N, N, 1, T * C, 1, T * C, batch=B, sab=N * N, sbb=N * T * C, scb=N * T * C)
    us = timeit(f); print(f"nconv bwd data (NN)    {us:8.1f} us  {fl / us / 1e6:7.2f} TF/s")
    dP = torch.empty(B, N, N, device="cuda")
    f = lambda: L.gemm(X, X, dP, N, N, T * C, T * C, 1, 1, T * C, N, batch=B, sab=N * T * C, sbb=N * T * C, scb=N * N)
    us = timeit(f); fl2 = 2.0 * B * N * N * T * C; print(f"nconv bwd adj (NT)     {us:8.1f} us  {fl2 / us / 1e6:7.2f} TF/s")
    A = torch.randn(2456 * 12, 224, device="cuda"); W = torch.randn(32, 224, device="cuda"); H = torch.empty(2456 * 12, 32, device="cuda")
    f = lambda: L.gemm(A, W, H, 2456 * 12, 32, 224, 224, 1, 1, 224, 32)
    us = timeit(f); fl3 = 2.0 * 2456 * 12 * 32 * 224; print(f"mix 224->32 (NT)       {us:8.1f} us  {fl3 / us / 1e6:7.2f} TF/s")
    f = lambda: None
    print(f"empty python loop      {timeit(f):8.1f} us")

if __name__ == "__main__":
    main()
