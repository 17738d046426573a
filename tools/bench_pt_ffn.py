"""Micro-benchmark of the fused feed-forward kernels of the pre-training step (csrc/pretrain_fused.hip) at config C3's decoder / encoder
row counts, next to the kernels they replace.  usage: [STEP_HIP_LIB=...] python tools/bench_pt_ffn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from step_amd import _lib as L  # noqa: E402


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    tag = os.path.basename(os.environ.get("STEP_HIP_LIB", "default"))
    gen = torch.Generator().manual_seed(1)
    w1 = (torch.randn(384, 96, generator=gen) * 0.15).cuda()
    b1 = (torch.randn(384, generator=gen) * 0.1).cuda()
    w2 = (torch.randn(96, 384, generator=gen) * 0.1).cuda()
    b2 = (torch.randn(96, generator=gen) * 0.1).cuda()
    st = L.stream()
    words = 1 << 18
    pool = torch.zeros(words + 16, dtype=torch.int64, device="cuda")
    L.call("step_dropout_pool_fill", L.ptr(pool), words, 0.1, 99, st)
    pack = torch.empty(L.lib().step_pt_ffn_pack_bytes(), dtype=torch.uint8, device="cuda")
    L.call("step_pt_ffn_pack", L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(pack), st)
    for R in (5200 * 168, 5200 * 42):
        x = torch.randn(R, 96, device="cuda")
        df2 = torch.randn(R, 96, device="cuda")
        out = torch.empty(R, 96, device="cuda")
        ws = torch.empty(L.lib().step_pt_ffn_wgrad_ws_floats(R), device="cuda")
        dw1, db1, dw2 = torch.zeros(384, 96, device="cuda"), torch.zeros(384, device="cuda"), torch.zeros(96, 384, device="cuda")
        for p in (0.1, 0.0):
            t_f = timed(lambda: L.call("step_pt_ffn_fused_fwd", L.ptr(x), R, L.ptr(pack), p, L.ptr(pool), words, 5, 2, L.ptr(out), st))
            t_d = timed(lambda: L.call("step_pt_ffn_fused_bwd_data", L.ptr(df2), L.ptr(x), R, L.ptr(pack), p, L.ptr(pool), words, 5, 2, L.ptr(out), st))
            t_w = timed(lambda: L.call("step_pt_ffn_fused_bwd_weights", L.ptr(df2), L.ptr(x), R, L.ptr(pack), L.ptr(b1), p, L.ptr(pool), words, 5, 2,
                                       L.ptr(ws), L.ptr(dw1), L.ptr(db1), L.ptr(dw2), st))
            gb_f, gb_d = R * 96 * 4 * 2 / 1e3, R * 96 * 4 * 4 / 1e3
            print(f"{tag} R={R} p={p}: forward {t_f:.0f} us ({gb_f / t_f:.0f} GB/s), backward-data {t_d:.0f} us ({gb_d / t_d:.0f} GB/s), "
                  f"backward-weights (W2 + W1 + 2 reductions) {t_w:.0f} us", flush=True)
        nb = L.lib().step_pt_rows_linear_pack_bytes
        wi = (torch.randn(288, 96, generator=gen) * 0.15).cuda()
        packs = [torch.empty(nb(kc, og), dtype=torch.uint8, device="cuda") for kc, og in ((1, 3), (1, 1), (3, 1))]
        L.call("step_pt_rows_linear_pack", L.ptr(wi), 96, 1, 1, 3, None, L.ptr(packs[0]), st)
        L.call("step_pt_rows_linear_pack", L.ptr(w1), 96, 1, 1, 1, None, L.ptr(packs[1]), st)
        L.call("step_pt_rows_linear_pack", L.ptr(wi), 1, 96, 3, 1, None, L.ptr(packs[2]), st)
        qkv = torch.empty(R, 288, device="cuda", dtype=torch.bfloat16)
        ab = torch.randn(R, 96, device="cuda").bfloat16()
        t_q = timed(lambda: L.call("step_pt_rows_linear", L.ptr(x), 0, R, L.ptr(packs[0]), 1, 3, L.ptr(qkv), 1, 0, st))
        t_q0 = timed(lambda: L.call("step_pt_linear_bf16out", L.ptr(x), L.ptr(wi), 1, 96, None, R, 288, 96, L.ptr(qkv), st))
        t_o = timed(lambda: L.call("step_pt_rows_linear", L.ptr(ab), 1, R, L.ptr(packs[1]), 1, 1, L.ptr(out), 0, 0, st))
        t_a = timed(lambda: L.call("step_pt_rows_linear", L.ptr(x), 0, R, L.ptr(packs[1]), 1, 1, L.ptr(ab), 1, 0, st))
        t_x = timed(lambda: L.call("step_pt_rows_linear", L.ptr(qkv), 1, R, L.ptr(packs[2]), 3, 1, L.ptr(out), 0, 1, st))
        wsp = torch.empty(L.lib().step_pt_proj_wgrad_ws_floats(R), device="cuda")
        gwi, gbi, gwo = torch.zeros(288, 96, device="cuda"), torch.zeros(288, device="cuda"), torch.zeros(96, 96, device="cuda")
        t_g = timed(lambda: L.call("step_pt_proj_wgrad", L.ptr(x), L.ptr(qkv), L.ptr(df2), L.ptr(ab), R, L.ptr(wsp), L.ptr(gwi), L.ptr(gbi), L.ptr(gwo), st))
        print(f"{tag} R={R} projection weight gradients (dWi, dbi, dWo + 2 reductions): {t_g:.0f} us ({R * (576 + 384 + 384 + 192) / 1e3 / t_g:.0f} GB/s)", flush=True)
        print(f"{tag} R={R} projections: qkv {t_q:.0f} us ({R * (384 + 576) / 1e3 / t_q:.0f} GB/s; step_pt_linear_bf16out {t_q0:.0f} us), out-projection {t_o:.0f} us, "
              f"da {t_a:.0f} us, dx += {t_x:.0f} us ({R * (576 + 768) / 1e3 / t_x:.0f} GB/s)", flush=True)
        t_p = timed(lambda: L.call("step_pt_ffn_pack", L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(pack), st))
        print(f"{tag} pack {t_p:.1f} us", flush=True)


if __name__ == "__main__":
    main()
