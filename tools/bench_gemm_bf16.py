"""Micro-benchmark of step_gemm on the production shapes (GPU box): bf16 fc / hop GEMMs, and with the argument `f32` the exact-f32
GraphWaveNet GEMMs.  Use STEP_HIP_LIB=<other build> for A/B comparisons."""
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


N, EMB, T2 = 307, 100, 13581
K = 16 * T2
a2 = torch.randn(N, K, device="cuda"); fcw = torch.randn(EMB, K, device="cuda") * 0.01
gpre = torch.zeros(N, EMB, device="cuda"); dg = torch.randn(N, EMB, device="cuda")
wraw = torch.empty(EMB, K, device="cuda"); da2 = torch.empty(N, K, device="cuda")
sc, sh = torch.rand(16, device="cuda") + 0.5, torch.randn(16, device="cuda")
cases = {
    "fc_fwd_kscale": (lambda: L.gemm(a2, fcw, gpre, N, EMB, K, K, 1, 1, K, EMB, accumulate=2, splitk=-1, a_kscale=sc, a_kshift=sh, a_kperiod=T2, compute_bf16=True), (N + EMB) * K * 4),
    "fc_fwd_plain": (lambda: L.gemm(a2, fcw, gpre, N, EMB, K, K, 1, 1, K, EMB, accumulate=2, splitk=-1, compute_bf16=True), (N + EMB) * K * 4),
    "fc_dW": (lambda: L.gemm(dg, a2, wraw, EMB, K, N, 1, EMB, K, 1, K, compute_bf16=True), (N + EMB) * K * 4),
    "fc_da2": (lambda: L.gemm(dg, fcw, da2, N, K, EMB, EMB, 1, K, 1, K, compute_bf16=True), (N + EMB) * K * 4),
}
Bn, S = 8, 7
N8 = (N + 7) // 8 * 8
for T in (12, 6, 1):
    P16 = torch.rand(3, Bn, N, N8, device="cuda").to(torch.bfloat16)
    cat = torch.randn(Bn, N, T, S * 32, device="cuda")
    dP = torch.zeros(3, Bn, N, N, device="cuda")
    cases[f"hop_fwd_T{T}"] = ((lambda P16=P16, cat=cat, T=T: L.gemm(P16, cat, cat, N, T * 32, N, N8, 1, T * S * 32, 1, T * S * 32, batch=3 * Bn, batch0=Bn,
                               sab=N * N8, sab1=Bn * N * N8, sbb=N * T * S * 32, sbb1=64, scb=N * T * S * 32, scb1=64, b_off=32, c_off=64,
                               b_n=(32, S * 32), c_n=(32, S * 32), compute_bf16=True)), 3 * Bn * (N * N8 * 2 + 2 * N * T * 32 * 4))
    cases[f"hop_adj_T{T}"] = ((lambda cat=cat, dP=dP, T=T: L.gemm(cat, cat, dP, N, N, T * 32, T * S * 32, 1, 1, T * S * 32, N, batch=3 * Bn, batch0=Bn,
                               sab=N * T * S * 32, sab1=64, sbb=N * T * S * 32, sbb1=64, scb=N * N, scb1=Bn * N * N, a_off=32, b_off=64,
                               a_k=(32, S * 32), b_k=(32, S * 32), accumulate=1, compute_bf16=True)), 3 * Bn * (2 * N * N * 4 + 2 * N * T * 32 * 4))
sel = [a for a in sys.argv[1:] if a != 'f32'] or ([] if sys.argv[1:] else list(cases))
for name in sel:
    f, nbytes = cases[name]
    us = timeit(f)
    print(f"{name:16s} {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s (algorithmic bytes)", flush=True)

# exact-f32 GEMMs of the GraphWaveNet layers
if not sys.argv[1:] or "f32" in sys.argv[1:]:
    npos = 8 * 307 * 12
    x64 = torch.randn(npos, 64, device="cuda"); w64 = torch.randn(64, 64, device="cuda"); o64 = torch.empty(npos, 64, device="cuda")
    cat = torch.randn(npos, 224, device="cuda"); wg = torch.randn(32, 224, device="cuda"); h = torch.empty(npos, 32, device="cuda")
    dcat = torch.empty(npos, 224, device="cuda"); dwg = torch.zeros(32, 224, device="cuda"); dw64 = torch.zeros(64, 64, device="cuda")
    BNn = 2456
    xh = torch.randn(BNn, 256, device="cuda"); e1w = torch.randn(512, 256, device="cuda"); e1 = torch.empty(BNn, 512, device="cuda")
    bias = torch.randn(512, device="cuda")
    f32cases = {
        "gate_fwd  npos x64x64": lambda: L.gemm(x64, w64, o64, npos, 64, 64, 64, 1, 1, 64, 64, bias=bias),
        "gconv_fwd npos x32x224": lambda: L.gemm(cat, wg, h, npos, 32, 224, 224, 1, 1, 224, 32, bias=bias),
        "gconv_dgrad npos x224x32": lambda: L.gemm(h, wg, dcat, npos, 224, 32, 32, 1, 224, 1, 224),
        "gconv_wgrad 32x224xnpos": lambda: L.gemm(h, cat, dwg, 32, 224, npos, 1, 32, 224, 1, 224, accumulate=2, splitk=-1),
        "gate_wgrad 64x64xnpos": lambda: L.gemm(o64, x64, dw64, 64, 64, npos, 1, 64, 64, 1, 64, accumulate=2, splitk=-1),
        "gate_dgrad npos x64x64": lambda: L.gemm(o64, w64, x64, npos, 64, 64, 64, 1, 64, 1, 64),
        "end1_fwd 2456x512x256": lambda: L.gemm(xh, e1w, e1, BNn, 512, 256, 256, 1, 1, 256, 512, bias=bias, relu=True),
    }
    for name, f in f32cases.items():
        us = timeit(f)
        print(f"{name:26s} {us:8.1f} us", flush=True)
