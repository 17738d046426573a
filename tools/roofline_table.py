"""Per-kernel roofline table for config C2 (STEP_PEMS04, B=8) from a tools/prof_summary.py markdown summary.

Algorithmic bytes / FLOPs per launch are analytic (shapes of the config); durations are the rocprofv3 averages of the summary.
usage: python tools/roofline_table.py profiles/r02_w_C2_train_step_no_overlap.md > profiles/r02_w_roofline_table.md
(rows describe the round-2 kernels; the round-1 table profiles/r01_u_roofline_table.md was made by the round-1 version of this file)"""
import re
import sys

N, T, B, P, EMB = 307, 13599, 8, 336, 100
T1, T2 = T - 9, T - 18
K = 16 * T2
MB = 1e6
a1, a2 = N * 8 * T1 * 4, N * 16 * T2 * 4
fcw = EMB * K * 4
npar = 25.3e6
HBM, MFMA_BF16 = 8000.0, 2500.0          # GB/s, TFLOP/s (MI355X_MICROARCH.md)

# kernel-name substring -> (label, bound, algorithmic bytes per launch or None, algorithmic FLOP per launch or None)
a1h, a2h = a1 // 2, a2 // 2               # round 2, bf16 mode: channels-last bf16 rows
wp = EMB * K * 2                          # the per-step bf16 fc weight copy

# kernel-name substring -> (label, bound, algorithmic bytes per launch or None, algorithmic FLOP per launch or None)
ROWS = [
    ("tsformer_encoder_kernel", "fused TSFormer encoder (2456 sequences x 336 tokens)", "mfma", 39.6e6 + 158.4e6, B * N * P * (4 * (221184 + 384 * P) + 2304)),
    ("gemm_fast_kernelILi128ELi128ELi1ELi1ELb0", "cosine Gram (8 x 307^2 x 32256) and DGL fc forward (bf16 rows x bf16 weight copy), averaged", "hbm/L2", (B * N * P * 96 * 2 + B * N * N * 4 + a2h + wp) / 2, (2.0 * B * N * N * P * 96 + 2.0 * N * EMB * K) / 2),
    ("gemm_fast_kernelILi128ELi128ELi2ELi3ELb0", "DGL fc weight gradient G = dgpre^T a2 (n-contiguous bf16 loader, writes 87 MB)", "hbm", a2h + fcw, 2.0 * N * EMB * K),
    ("fc_unpermute_kernel", "G -> fc.weight layout, BN2 affine, += into the gradient, BN2 backward sums", "hbm", 4 * fcw, None),
    ("fc_prep_kernel", "per-step bf16 (t,c)-ordered copy of fc.weight with BN2's scale", "hbm", fcw + wp, None),
    ("gemm_fast_kernelILi128ELi128ELi0ELi3ELb0ELb0ELb1", "DGL fc input gradient + fused BN2 backward / ReLU mask (reads a2h, writes dz2h)", "hbm", wp + 2 * a2h, 2.0 * N * EMB * K),
    ("conv1_fwd_cl_kernel", "DGL conv1 forward (series -> a1h)", "hbm", N * T * 4 + a1h, 2.0 * N * T1 * 8 * 10),
    ("conv2_fwd_cl_kernel", "DGL conv2 forward (a1h -> a2h)", "hbm", a1h + a2h, 2.0 * N * T2 * 16 * 80),
    ("conv2_dgrad_cl_kernel", "DGL conv2 input gradient + fused BN1 backward (dz2h, a1h -> dz1h)", "hbm", a2h + 2 * a1h, 2.0 * N * T2 * 16 * 80),
    ("conv2_wgrad_cl_kernel", "DGL conv2 weight-gradient sums (reads dz2h and a1h)", "hbm", a2h + a1h, 2.0 * N * T2 * 16 * 80),
    ("conv1_wgrad_cl_kernel", "DGL conv1 weight gradient (reads dz1h and the series)", "hbm", a1h + N * T * 4, 2.0 * N * T1 * 8 * 10),
    ("adam_clip_kernel", "fused clip + Adam (25.3 M parameters, 28 B each)", "hbm", npar * 28, None),
    ("pack_long_history_kernel", "long history [B,L,N,3] -> [B*N, L] (reads all three channels' lines)", "hbm", B * 4032 * N * 3 * 4 + B * 4032 * N * 4, None),
    ("gemm_fast_kernelILi64ELi64ELi1ELi2ELb0", "diffusion hop, 3 supports x 8 samples per launch (bf16 stack x f32 slots)", "latency", 24 * (N * 312 * 2 + 2 * N * 12 * 32 * 4), 2.0 * 24 * N * N * 12 * 32),
    ("ELb0ELb1ELb0", "adjacency gradients of all layers, one segmented contraction (K = 3264, 24 x 307^2 outputs)", "mfma/L2", 24 * (2 * N * 3264 * 4 + N * N * 4), 2.0 * 24 * N * N * 3264),
]


def main():
    rows = {}
    for line in open(sys.argv[1]):
        m = re.match(r"\| (\S+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
        if m:
            rows[m.group(1)] = (int(m.group(2)), float(m.group(4)))
    print("# Per-kernel roofline, config C2 (STEP_PEMS04, B=8), 1x MI355X -- durations: " + sys.argv[1])
    print()
    print("Algorithmic bytes = the tensors a launch has to read and write once (analytic, f32 unless noted); peak HBM 8 TB/s")
    print("(6.3 TB/s is what a streaming copy reaches on this part), dense bf16 matrix peak 2.5 PFLOP/s.")
    print()
    print("| kernel | launches/step | avg us | algorithmic MB | GB/s | % of 8 TB/s | TFLOP/s | bound |")
    print("|---|---|---|---|---|---|---|---|")
    steps = None
    for key, label, bound, nbytes, flop in ROWS:
        hit = [(k, v) for k, v in rows.items() if key in k]
        if not hit:
            continue
        calls, us = hit[0][1]
        if steps is None:
            steps = calls                                  # the encoder is launched once per step
        gbs = nbytes / us / 1e3 if nbytes else None
        tf = flop / us / 1e6 if flop else None
        print(f"| {label} | {calls / steps:.0f} | {us:.1f} | {nbytes / MB:.0f} | {gbs:.0f} | {100 * gbs / HBM:.0f} % | "
              f"{'' if tf is None else f'{tf:.1f}'} | {bound} |")
    # measured HBM traffic of the dominant kernel (PMC passes of tools/pmc_enc_ab.sh), next to its algorithmic bytes
    import json
    import os
    pmc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "encoder_pmc.json")
    if os.path.exists(pmc):
        ent = json.load(open(pmc)).get("STEP_PEMS04:B8")
        if ent:
            print()
            print(f"Encoder, measured HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, `profiles/encoder_pmc.json`): "
                  f"{ent['read_bytes'] / MB:.0f} MB read + {ent['write_bytes'] / MB:.0f} MB written, against 40 MB of input and 158 MB of "
                  f"bf16 hidden states that have to move.")


if __name__ == "__main__":
    main()
