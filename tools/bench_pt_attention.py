"""Micro-benchmark of the pre-training attention kernels (matrix-core path) at config C3's decoder / encoder sizes.
usage: [STEP_HIP_LIB=...] python tools/bench_pt_attention.py [S]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib as L  # noqa: E402


POOL = os.environ.get("ATTN_POOL", "1") != "0"       # keep words of the forward from the Bernoulli pool (what the pre-training step does)


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 5200
    tag = os.environ.get("STEP_HIP_LIB", "default")
    for T in (168, 42):
        gen = torch.Generator().manual_seed(T)
        qkv = (torch.randn(S, T, 288, generator=gen) * 1.0).bfloat16().cuda()
        dout = torch.randn(S, T, 96, generator=gen).bfloat16().cuda()
        out = torch.empty(S, T, 96, device="cuda", dtype=torch.bfloat16)
        stats = torch.empty(S * 4 * T, 2, device="cuda")
        dqkv = torch.zeros(S, T, 288, device="cuda", dtype=torch.bfloat16)
        kb = torch.zeros(S * 4 * T * ((T + 31) // 32), dtype=torch.int32, device="cuda")
        st = L.stream()
        pool = torch.zeros((1 << 18) + 16, dtype=torch.int64, device="cuda")
        L.call("step_dropout_pool_fill", L.ptr(pool), 1 << 18, 0.1, 77, st)
        for p in (0.1, 0.0):
            def fwd():
                L.call("step_pt_attention_fwd_bf16", L.ptr(qkv), S, T, p, 1234, 7, L.ptr(out), L.ptr(stats), L.ptr(kb), L.ptr(pool) if POOL else None, 1 << 18, st)

            def bwd():
                L.call("step_pt_attention_bwd_bf16", L.ptr(qkv), L.ptr(out), L.ptr(dout), L.ptr(stats), S, T, p, 1234, 7, L.ptr(dqkv), L.ptr(kb), st)
            res = []
            for f in (fwd, bwd):
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    f()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 10)
            print(f"{tag} S={S} T={T} p={p}: forward {res[0] * 1e3:.0f} us, backward {res[1] * 1e3:.0f} us", flush=True)


if __name__ == "__main__":
    main()
