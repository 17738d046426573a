"""Per-kernel roofline table of one STEP config from a tools/prof_summary.py markdown summary (rocprofv3 --kernel-trace --stats of
`bench.py --config <name>` with STEP_NO_OVERLAP=1, so that durations are the kernels' own).

Algorithmic bytes / FLOPs per launch are analytic (shapes of the config, bench.CONFIGS); durations are the rocprofv3 averages.
usage: python tools/roofline_table.py <summary.md> [--config STEP_PEMS04] [--json profiles/kernel_roofline.json] > table.md
`--json` also records the rows (and the dominant kernels by time per step) under the config's name in that file, which is what
bench.py quotes as `other_configs[*].dominant_kernels` (marked static, with the summary's path as source)."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MB = 1e6
HBM, MFMA_BF16 = 8000.0, 2500.0          # GB/s, TFLOP/s (MI355X_MICROARCH.md)
EMB = 100


def rows_of(N, T, B, P, L):
    """kernel-name substring -> (label, bound, algorithmic bytes per launch or None, algorithmic FLOP per launch or None)"""
    T1, T2 = T - 9, T - 18
    K = 16 * T2
    a1h, a2h = N * 8 * T1 * 2, N * 16 * T2 * 2           # bf16 mode: channels-last bf16 rows
    fcw = EMB * K * 4
    wp = EMB * K * 2                                      # the per-step bf16 fc weight copy
    npar = 486252 + 20 * N + EMB * K + 21934 + 300        # parameters the optimizer touches
    S = B * N
    Np = (N + 7) // 8 * 8
    big = N >= 2048                                       # (the hop GEMM takes 128 x 64 tiles on the large graphs, 64 x 64 otherwise)
    hop_key = ("gemm_fast_kernelILi128ELi64ELi1ELi2ELb0", "gemm_fast_kernelILi64ELi64ELi1ELi2ELb0")
    return [
        ("tsformer_encoder_kernel", f"fused TSFormer encoder ({S} sequences x {P} tokens)", "mfma", S * L * 4 + S * P * 96 * 2, S * P * (4 * (221184 + 384 * P) + 2304)),
        # (graphs of up to 320 nodes: the cosine Gram is gram_sym_kernel + gram_finish_kernel since round 6, and this instantiation is the fc forward alone)
        (("gemm_fast_kernelILi128ELi128ELi1ELi1ELb0",) if N > 320 else ("gemm_fast_kernelILi128ELi128ELi1ELi1ELb0",),
         (f"cosine Gram ({B} x {N}^2 x {P * 96}) and DGL fc forward (bf16 rows x bf16 weight copy), averaged" if N > 320 else
          "DGL fc forward (bf16 rows x bf16 weight copy, split-K through a workspace)"), "hbm/L2",
         (S * P * 96 * 2 + B * N * N * 4 + a2h + wp) / 2 if N > 320 else a2h + wp, (2.0 * B * N * N * P * 96 + 2.0 * N * EMB * K) / 2 if N > 320 else 2.0 * N * EMB * K),
        ("gram_sym_kernel", f"symmetric cosine Gram ({B} x {N}^2 x {P * 96}): the whole output per workgroup, H read once, blocks on / above the diagonal", "mfma/lds",
         S * P * 96 * 2 + B * N * N * 4, 2.0 * B * N * N * P * 96),
        ("gram_finish_kernel", "sum of the Gram product's feature slices + cosine normalisation, both triangles", "hbm", 16 * B * ((N + 63) // 64) * ((N + 63) // 64 + 1) // 2 * 16384 + B * N * N * 4, None),
        ("gemm_fast_kernelILi128ELi128ELi2ELi3ELb0", "DGL fc weight gradient G = dgpre^T a2 (n-contiguous bf16 loader)", "hbm", a2h + fcw, 2.0 * N * EMB * K),
        ("fc_unpermute_kernel", "G -> fc.weight layout, BN2 affine, stored into the gradient, BN2 backward sums", "hbm", 3 * fcw, None),
        ("fc_prep_kernel", "per-step bf16 (t,c)-ordered copy of fc.weight with BN2's scale", "hbm", fcw + wp, None),
        ("gemm_fast_kernelILi128ELi128ELi0ELi3ELb0ELb0ELb1", "DGL fc input gradient + fused BN2 backward / ReLU mask (reads a2h, writes dz2h)", "hbm", wp + 2 * a2h, 2.0 * N * EMB * K),
        ("conv1_fwd_cl_kernel", "DGL conv1 forward (series -> a1h)", "hbm", N * T * 4 + a1h, 2.0 * N * T1 * 8 * 10),
        ("conv2_fwd_cl_kernel", "DGL conv2 forward (a1h -> a2h)", "hbm", a1h + a2h, 2.0 * N * T2 * 16 * 80),
        ("conv2_dgrad_cl_kernel", "DGL conv2 input gradient + fused BN1 backward (dz2h, a1h -> dz1h)", "hbm", a2h + 2 * a1h, 2.0 * N * T2 * 16 * 80),
        ("conv2_wgrad_cl_kernel", "DGL conv2 weight-gradient sums (reads dz2h and a1h)", "hbm", a2h + a1h, 2.0 * N * T2 * 16 * 80),
        ("conv1_wgrad_cl_kernel", "DGL conv1 weight gradient (reads dz1h and the series)", "hbm", a1h + N * T * 4, 2.0 * N * T1 * 8 * 10),
        ("adam_clip_kernel", f"fused clip + Adam ({npar / 1e6:.1f} M parameters, 28 B each)", "hbm", npar * 28, None),
        ("pack_long_history_kernel", "long history [B,L,N,3] -> [B*N, L] (reads all three channels' lines)", "hbm", B * L * N * 3 * 4 + B * L * N * 4, None),
        (hop_key, f"diffusion hop, 3 supports x {B} samples per launch (bf16 adjacency stack x 32-channel slots; T = 12 here, it shrinks per layer)", "hbm/L2" if big else "latency", 3 * B * (N * Np * 2 + 2 * N * 12 * 32 * 4), 2.0 * 3 * B * N * N * 12 * 32),
        ("ELb0ELb1ELb0", f"adjacency gradients of all layers, one segmented contraction (K = 3264, {3 * B} x {N}^2 outputs)", "mfma/L2", 3 * B * (2 * N * 3264 * 4 + N * N * 4), 2.0 * 3 * B * N * N * 3264),
        ("edge_logit_kernel", f"edge MLP forward ({N}^2 edges x 100 hidden units, recomputed in the backward)", "valu", N * N * 8 + 2 * N * 100 * 4, 300.0 * N * N),
        ("edge_bwd_row_kernel", "edge MLP backward, receiver pass", "valu", N * N * 8 + 3 * N * 100 * 4, 600.0 * N * N),
    ]


def rows_of_pretrain(N, B, P, Pu=None):
    """Config C3 (TSFormer pre-training, bf16 mode): per-launch AVERAGES over the step's launches of a kernel (4 encoder layers of
    S * Pu rows + 1 decoder layer of S * P rows; the LayerNorm kernels also run once for each final norm)."""
    S = B * N
    Pu = Pu or P // 4
    Re, Rd = S * Pu, S * P
    lay = (4 * Re + Rd) / 5.0                                   # rows of an average layer launch
    ln = (2 * (4 * Re + Rd) + Re + Rd) / 12.0                   # rows of an average LayerNorm launch
    att = lambda k: (4 * S * 4 * Pu * Pu + S * 4 * P * P) / 5.0 * 2.0 * 24 * k      # k products of T x T x 24 per (sequence, head)
    f32, b16 = 96 * 4, 96 * 2
    return [
        (("attn2_bwd_kernel", "attn_mfma_bwd_kernel"), "attention backward on the matrix cores (dQ, dK, dV; scores recomputed in both orientations; round 6: pretrain_attn2.hip)", "valu issue / hbm",
         lay * (2 * 3 * b16 + 2 * b16 + 4 * 8 + 4 * 4 * ((P + 31) // 32)), att(5)),
        ("ln_bwd_drop_kernel", "LayerNorm backward + dropout of the continuing gradient + parameter / bias gradient sums", "hbm", ln * 4 * f32 * (10.0 / 12) + ln * 3 * f32 * (2.0 / 12), None),
        ("add_ln_fwd_kernel", "LayerNorm forward of the two final norms (the layers' residual + dropout + LayerNorm steps are output stages of row kernels)", "hbm", (Re + Rd) / 2.0 * 2 * f32, None),
        ("ffn_rows_kernelILi8ELb1", "fused feed-forward, input gradient (hidden layer recomputed; reads h1, d f2, read-modify-writes d h1)", "hbm", lay * 4 * f32, lay * 3 * 2.0 * 96 * 384),
        (("attn2_fwd_kernel", "attn_mfma_fwd_kernel"), "attention forward on the matrix cores (round 6: pretrain_attn2.hip)", "valu issue / hbm", lay * (3 * b16 + b16 + 4 * 8 + 4 * 4 * ((P + 31) // 32)), att(2)),
        ("ffn_wgrad_kernelILb1", "fused feed-forward, d W1 and d b1 (hidden layer and its gradient recomputed)", "lds", lay * 2 * f32, lay * 3 * 2.0 * 96 * 384),
        ("ffn_rows_kernelILi8ELb0", "fused feed-forward, forward (hidden layer in registers) + residual + dropout + LayerNorm 2 as its output stage", "hbm", lay * 3 * f32, lay * 2 * 2.0 * 96 * 384),
        ("rows_linear_kernelILi3ELi1ELb1ELb0ELb1", "d x += d qkv . Wi (row kernel, LDS-resident weights)", "hbm", lay * (3 * b16 + 2 * f32), lay * 2.0 * 96 * 288),
        ("ffn_wgrad_kernelILb0", "fused feed-forward, d W2 (hidden layer recomputed)", "lds", lay * 2 * f32, lay * 2 * 2.0 * 96 * 384),
        ("rows_linear_kernelILi1ELi3ELb0ELb1ELb0", "qkv = x . Wi^T + bi (row kernel, LDS-resident weights)", "hbm", lay * (f32 + 3 * b16), lay * 2.0 * 96 * 288),
        ("rows_linear_kernelILi1ELi1ELb1ELb0ELb0", "o = a . Wo^T + bo (row kernel) + residual + dropout + LayerNorm 1 as its output stage", "hbm", lay * (b16 + 3 * f32), lay * 2.0 * 96 * 96),
        ("proj_wgrad_kernel", "d Wi, d bi, d Wo in one pass over x, d qkv, d o, a", "lds", lay * (f32 + 3 * b16 + f32 + b16), lay * 2.0 * 96 * (288 + 96)),
        ("rows_linear_kernelILi1ELi1ELb0ELb1ELb0", "d a = d o . Wo (row kernel)", "hbm", lay * (f32 + b16), lay * 2.0 * 96 * 96),
    ]


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--config", default="STEP_PEMS04", choices=list(bench.CONFIGS))
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    N, T, B, L = cfg["N"], cfg["T_train"], cfg["B"], cfg["L"]
    P = L // 12
    pre = bool(cfg.get("pretrain"))
    rows, total_ms, ndisp = {}, None, None
    for line in open(args.summary):
        m = re.match(r"\| (\S+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            rows[m.group(1)] = (int(m.group(2)), float(m.group(4)), float(m.group(3)))
        m = re.match(r"kernel dispatches: (\d+)\s+sum of kernel time: ([\d.]+) ms", line)
        if m:
            ndisp, total_ms = int(m.group(1)), float(m.group(2))
    print(f"# Per-kernel roofline, config {args.config} (N={N}, P={P}, T_train={T}, B={B}), 1x MI355X -- durations: " + args.summary)
    print()
    print("Algorithmic bytes = the tensors a launch has to read and write once (analytic); peak HBM 8 TB/s")
    print("(6.3 TB/s is what a streaming copy reaches on this part), dense bf16 matrix peak 2.5 PFLOP/s.")
    print()
    print("| kernel | launches/step | avg us | ms/step | algorithmic MB | GB/s | % of 8 TB/s | TFLOP/s | % of 2.5 PF | bound |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    steps = None
    out = []
    for key, label, bound, nbytes, flop in (rows_of_pretrain(N, B, P) if pre else rows_of(N, T, B, P, L)):
        hit = []
        for kk in (key if isinstance(key, tuple) else (key,)):
            hit = [(k, v) for k, v in rows.items() if kk in k]
            if hit:
                break
        if not hit:
            continue
        calls, us, tot = hit[0][1]
        if steps is None:
            steps = calls // 5 if pre else calls           # the encoder is launched once per step (pre-training: the attention backward five times)
        gbs = nbytes / us / 1e3 if nbytes else None
        tf = flop / us / 1e6 if flop else None
        print(f"| {label} | {calls / steps:.0f} | {us:.1f} | {tot / steps:.3f} | {nbytes / MB:.0f} | {gbs:.0f} | {100 * gbs / HBM:.0f} % | "
              f"{'' if tf is None else f'{tf:.1f}'} | {'' if tf is None else f'{100 * tf / MFMA_BF16:.1f} %'} | {bound} |")
        out.append({"kernel": hit[0][0][:80], "what": label, "bound": bound, "launches_per_step": calls / steps, "avg_us": us, "ms_per_step": tot / steps,
                    "algorithmic_bytes_per_launch": nbytes, "algorithmic_flop_per_launch": flop, "GBps": gbs, "frac_of_hbm_peak": gbs / HBM if gbs else None,
                    "TFLOPps": tf, "frac_of_mfma_peak": tf / MFMA_BF16 if tf else None})
    if total_ms is not None and steps:
        print()
        print(f"Sum of kernel time per step (streams not overlapping): {total_ms / steps:.3f} ms in {ndisp / steps:.0f} dispatches.")
    # measured HBM traffic of the dominant kernel (PMC passes of tools/pmc_enc_ab.sh), next to its algorithmic bytes
    pmc = os.path.join(ROOT, "profiles", "encoder_pmc.json")
    if os.path.exists(pmc):
        ent = json.load(open(pmc)).get(f"{args.config}:B{B}")
        if ent:
            print()
            print(f"Encoder, measured HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, `profiles/encoder_pmc.json`): "
                  f"{ent['read_bytes'] / MB:.0f} MB read + {ent['write_bytes'] / MB:.0f} MB written, against {B * N * L * 4 / MB:.0f} MB of input and "
                  f"{B * N * P * 96 * 2 / MB:.0f} MB of bf16 hidden states that have to move.")
    if args.json:
        db = json.load(open(args.json)) if os.path.exists(args.json) else {}
        out.sort(key=lambda r: -r["ms_per_step"])
        db[args.config] = {"source": os.path.relpath(os.path.abspath(args.summary), ROOT), "kernel_ms_per_step": total_ms / steps if total_ms and steps else None,
                           "dispatches_per_step": ndisp / steps if ndisp and steps else None, "kernels": out}
        with open(args.json, "w") as f:
            json.dump(db, f, indent=1)


if __name__ == "__main__":
    main()
