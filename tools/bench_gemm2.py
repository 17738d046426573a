import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib as L
from tools.bench_gemm import timeit
for (M, N, K, B) in [(307, 384, 64, 8), (307, 384, 307, 8), (307, 384, 1024, 8), (32, 32, 1024, 1), (32, 32, 64, 1), (307, 384, 307, 1), (2456, 512, 256, 1), (4096, 4096, 1024, 1)]:
    A = torch.randn(B, M, K, device="cuda"); Bm = torch.randn(B, K, N, device="cuda"); C = torch.empty(B, M, N, device="cuda")
    f = lambda: L.gemm(A, Bm, C, M, N, K, K, 1, N, 1, N, batch=B, sab=M * K, sbb=K * N, scb=M * N)
    us = timeit(f, 30)
    print(f"NN M={M} N={N} K={K} B={B}: {us:8.1f} us  {2.0 * B * M * N * K / us / 1e6:7.2f} TF/s", flush=True)
