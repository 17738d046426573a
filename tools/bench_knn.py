"""Micro-benchmark of the kNN prior graph (Gram + cosine + top-k) at PEMS04 size (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib as L


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    B, N, F, k = 8, 307, 336 * 96, 10
    H = (torch.randn(B, 1, F, device="cuda") + 0.7 * torch.randn(B, N, F, device="cuda")).to(torch.bfloat16)
    sim = torch.empty(B, N, N, device="cuda"); adj = torch.empty(B, N, N, device="cuda")
    work = torch.empty(L.lib().step_knn_workspace_bytes(B, N, F), dtype=torch.uint8, device="cuda")
    f = lambda: L.call("step_knn_graph", L.ptr(H), None, B, N, F, k * N, L.ptr(sim), L.ptr(adj), L.ptr(work), work.numel(), L.stream())
    print(f"knn_graph total {timeit(f):.1f} us")
    g = lambda: L.call("step_topk_mask", L.ptr(sim), B, N, k * N, L.ptr(adj), L.ptr(work), work.numel(), L.stream())
    print(f"topk_mask {timeit(g):.1f} us")


if __name__ == "__main__":
    main()
