"""Per-kernel roofline table for config C2 (STEP_PEMS04, B=8) from a tools/prof_summary.py markdown summary.

Algorithmic bytes / FLOPs per launch are analytic (shapes of the config); durations are the rocprofv3 averages of the summary.
usage: python tools/roofline_table.py profiles/r01_u_final_bf16mode_train_step.md > profiles/r01_u_roofline_table.md"""
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
ROWS = [
    ("tsformer_encoder_kernel", "fused TSFormer encoder (2456 sequences x 336 tokens)", "mfma", 39.6e6 + 158.4e6, B * N * P * (4 * (221184 + 384 * P) + 2304)),
    ("gemm_fast_kernelILi128ELi128ELi1ELi1ELb0", "cosine Gram, bf16 x bf16 (8 x 307^2 x 32256)", "hbm/L2", B * N * P * 96 * 2 + B * N * N * 4, 2.0 * B * N * N * P * 96),
    ("gemm_fast_kernelILi128ELi128ELi0ELi0ELb0", "DGL fc forward (307 x 100 x 217296, split-K)", "hbm", a2 + fcw, 2.0 * N * EMB * K),
    ("gemm_fast_kernelILi128ELi128ELi2ELi2ELb0", "DGL fc weight gradient + BN2 affine (+= into 87 MB)", "hbm", a2 + 2 * fcw, 2.0 * N * EMB * K),
    ("gemm_fast_kernelILi128ELi128ELi0ELi2ELb0", "DGL fc input gradient d_a2 (307 x 217296 x 100)", "hbm", fcw + a2, 2.0 * N * EMB * K),
    ("conv2_fwd_mfma_kernel", "DGL conv2 forward (reads a1, writes a2)", "hbm", a1 + a2, 2.0 * N * T2 * 16 * 80),
    ("conv2_dgrad_mfma_kernel", "DGL conv2 input gradient", "hbm", a2 + a1, 2.0 * N * T2 * 16 * 80),
    ("conv2_wgrad_mfma_kernel", "DGL conv2 weight gradient (reads dz2 and a1)", "hbm", a2 + a1, 2.0 * N * T2 * 16 * 80),
    ("conv1_wgrad_mfma_kernel", "DGL conv1 weight gradient (reads dz1 and the series)", "hbm", a1 + N * T * 4, 2.0 * N * T1 * 8 * 10),
    ("conv_relu_fwd_kernelILi1ELi8", "DGL conv1 forward (f32 VALU)", "hbm", N * T * 4 + a1, 2.0 * N * T1 * 8 * 10),
    ("bn_bwd_reduce_kernelILi16", "BN2 backward reduce (reads d_a2, a2)", "hbm", 2 * a2, None),
    ("bn_bwd_apply_kernelILi16", "BN2 backward apply (reads d_a2, a2; writes dz2)", "hbm", 3 * a2, None),
    ("bn_bwd_reduce_kernelILi8", "BN1 backward reduce", "hbm", 2 * a1, None),
    ("bn_bwd_apply_kernelILi8", "BN1 backward apply", "hbm", 3 * a1, None),
    ("adam_clip_kernel", "fused clip + Adam (25.3 M parameters, 28 B each)", "hbm", npar * 28, None),
    ("pack_long_history_kernel", "long history [B,L,N,3] -> [B*N, L] (reads all three channels' lines)", "hbm", B * 4032 * N * 3 * 4 + B * 4032 * N * 4, None),
    ("gemm_fast_kernelILi64ELi64ELi1ELi2ELb0", "diffusion hop, 3 supports x 8 samples per launch (bf16 stack x f32 slots)", "latency", 24 * (N * 312 * 2 + 2 * N * 12 * 32 * 4), 2.0 * 24 * N * N * 12 * 32),
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


if __name__ == "__main__":
    main()
