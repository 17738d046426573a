"""CPU study of cheaper keep-mask generators for the encoder's dropout (no GPU involved).

The kernel draws four Bernoulli bytes from one xorshift32 step (6 full-rate VALU ops per word = 1.5 ops per draw,
csrc/tsformer_device.h `Dropper`); the dropout sites are ~35 % of the kernel's VALU instructions.  Candidates are judged with
the statistics of tests/test_gpu_kernels.py::test_dropout_generator_statistics (drop rate at threshold 26/256, serial
correlation inside a stream at several lags, correlation between neighbouring streams, spread of the per-stream rate), on
streams seeded exactly like `Dropper::seed` (mix32 of base + id * golden ratio).

  xorshift32       the shipped generator: 4 draws per 6 ops
  lcg24_b2         x <- (x * A + C) mod 2^24, A = 0x43FD45 (= 1 mod 4: full period), one v_mad_u32_u24; draw = bits 16..23: 1 draw per op
  lcg24_b12        same state, draws = bits 16..23 then bits 8..15: 2 draws per op
  lcg32hi_b23      x <- x * A + C mod 2^32 (v_mad_u64_u32 / v_mul_lo_u32 class: quarter rate, listed for reference), bytes 3, 2

    python tools/dropout_generator_study.py            # prints one JSON object
"""
import json

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def mix32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & M32
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & M32
    x ^= x >> np.uint64(16)
    return x


def seeds(streams, base=0x1234567):
    ids = np.arange(streams, dtype=np.uint64)
    return mix32((np.uint64(base) + ids * np.uint64(0x9E3779B1)) & M32) | np.uint64(1)


def gen_xorshift32(streams, draws):
    st = seeds(streams)
    out = np.empty((streams, draws), np.uint8)
    for w in range(draws // 4):
        st ^= (st << np.uint64(13)) & M32; st ^= st >> np.uint64(17); st ^= (st << np.uint64(5)) & M32
        for b in range(4):
            out[:, 4 * w + b] = (st >> np.uint64(8 * b)) & np.uint64(0xFF)
    return out


def gen_lcg24(streams, draws, bytes_used, A=0x43FD45, C=0xC39EC3):
    st = seeds(streams) & np.uint64(0xFFFFFF)
    out = np.empty((streams, draws), np.uint8)
    n = len(bytes_used)
    for w in range(draws // n):
        st = (st * np.uint64(A) + np.uint64(C)) & np.uint64(0xFFFFFF)
        for i, b in enumerate(bytes_used):
            out[:, n * w + i] = (st >> np.uint64(8 * b)) & np.uint64(0xFF)
    return out


def gen_lcg32(streams, draws, bytes_used, A=1664525, C=1013904223):
    st = seeds(streams)
    out = np.empty((streams, draws), np.uint8)
    n = len(bytes_used)
    for w in range(draws // n):
        st = (st * np.uint64(A) + np.uint64(C)) & M32
        for i, b in enumerate(bytes_used):
            out[:, n * w + i] = (st >> np.uint64(8 * b)) & np.uint64(0xFF)
    return out


def judge(by, thresh=26):
    drop = (by < thresh).astype(np.float64)
    p = thresh / 256
    n = drop.size
    var = p * (1 - p)
    z = drop - p
    lags = {k: float((z[:, :-k] * z[:, k:]).mean() / var) for k in (1, 2, 3, 4, 8, 32)}
    cross = float((z[:-1] * z[1:]).mean() / var)
    tol = 6 / n ** 0.5
    sd = (var / n) ** 0.5
    res = {"drop_rate": float(drop.mean()), "rate_z": float((drop.mean() - p) / sd), "serial_corr": lags, "neighbour_stream_corr": cross,
           "per_stream_sd_over_binomial": float(drop.mean(1).std() / (var / drop.shape[1]) ** 0.5), "corr_tolerance": tol}
    res["passes"] = bool(abs(res["rate_z"]) < 6 and all(abs(v) < tol for v in lags.values()) and abs(cross) < tol
                         and res["per_stream_sd_over_binomial"] < 1.3)
    return res


def main():
    streams = 4096
    out = {}
    for draws in (176, 1024):          # 176 = draws per lane per (head, layer) at P = 336; 1024 = the GPU self-test's stream length
        out[f"draws_{draws}"] = {
            "xorshift32 (1.5 ops/draw)": judge(gen_xorshift32(streams, draws)),
            "lcg24_b2 (1 op/draw)": judge(gen_lcg24(streams, draws, (2,))),
            "lcg24_b12 (0.5 op/draw)": judge(gen_lcg24(streams, draws, (2, 1))),
            "lcg32_b32 (quarter-rate multiply)": judge(gen_lcg32(streams, draws, (3, 2))),
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
