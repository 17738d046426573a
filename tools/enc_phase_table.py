"""Per-phase time table of the fused TSFormer encoder from the s_memtime stamps of a -DTSF_TIMING=1 build.

    python tools/enc_phase_table.py gpurun_out/enc_timing_<tag>_drop_p336.bin [more.bin ...] > profiles/<name>.md

tools/enc_ab.cpp writes the files (header: groups, 16, 4, stamps; then uint64 [group][wave][layer][stamp], zero = not written).
A stamp is taken by every wave of the sampled workgroups (sequence index 7 mod 64) at the phase boundaries of
csrc/tsformer_encoder.hip (TSF_STAMP): 0 layer start; per head hd: 1+5hd after the stage boundary (barrier + weight DMA wait),
2+5hd Q/K/V done and K/V fragments written, 3+5hd behind the K/V hand-over barrier, 4+5hd key-tile loop done, 5+5hd out-projection
issued; 21 residual + LayerNorm 1 done; 22 behind the first feed-forward stage boundary; 21+2j / 22+2j around the later ones;
34 feed-forward done; 35 residual + LayerNorm 2 done.  The instrumentation itself costs time (a scalar memory-time read drains the
wave's counters), so the table is for PROPORTIONS, not for absolute cycles.
"""
import sys

import numpy as np


def load(fn):
    raw = open(fn, "rb").read()
    g, w, l, n = np.frombuffer(raw[:16], dtype=np.int32)
    t = np.frombuffer(raw[16:], dtype=np.uint64).reshape(g, w, l, n).astype(np.int64)
    return t


def table(fn):
    t = load(fn)
    G, W, L, N = t.shape
    waves = [w for w in range(W) if (t[:, w, :, 0] > 0).any()]
    groups = [g for g in range(G) if (t[g, waves[0], :, 0] > 0).all()]
    t = t[groups][:, waves]                         # [g, wave, layer, stamp]
    nw = len(waves)
    rows = []                                       # (name, kind, per-wave durations [g, wave, layer])

    def d(a, b):
        return (t[..., b] - t[..., a]).astype(np.float64)

    prev = 0
    for hd in range(4):
        b = 1 + 5 * hd
        rows.append((f"head {hd}: stage boundary (weight DMA wait + barrier)", "wait", d(prev, b)))
        rows.append((f"head {hd}: Q / K / V projections, K/V fragments to LDS", "mfma", d(b, b + 1)))
        rows.append((f"head {hd}: K/V hand-over barrier", "wait", d(b + 1, b + 2)))
        rows.append((f"head {hd}: key-tile loop (scores, exp2, keep masks, P V)", "attn", d(b + 2, b + 3)))
        rows.append((f"head {hd}: normalise + out-projection", "mfma", d(b + 3, b + 4)))
        prev = b + 4
    rows.append(("dropout 1 + residual + LayerNorm 1", "valu", d(20, 21)))
    rows.append(("feed-forward: first stage boundary", "wait", d(21, 22)))
    last = 22
    for j in range(1, 6):
        a, b = 21 + 2 * j, 22 + 2 * j
        if (t[..., a] > 0).all():
            rows.append((f"feed-forward chunks up to block {j}", "mfma", d(last, a)))
            rows.append((f"feed-forward: stage boundary in front of block {j}", "wait", d(a, b)))
            last = b
    rows.append(("feed-forward chunks (rest)", "mfma", d(last, 34)))
    rows.append(("dropout 2 + residual + LayerNorm 2", "valu", d(34, 35)))
    layer = d(0, 35)
    out = []
    out.append(f"### {fn}\n")
    out.append(f"{len(groups)} sampled workgroups x {nw} waves x {L} layers; cycles are s_memtime ticks of an instrumented build.\n")
    out.append(f"layer time (stamp 0 -> 35): mean {layer.mean():.0f} cycles, fastest wave {layer.min():.0f}, slowest {layer.max():.0f}\n")
    out.append("| phase | kind | mean cycles per wave | % of layer | fastest wave (mean over workgroups) | slowest wave | spread across the waves of a workgroup |")
    out.append("|---|---|---|---|---|---|---|")
    tot = {}
    for name, kind, v in rows:
        per_wave = v.mean(axis=(0, 2))             # mean over workgroups and layers, per wave
        spread = (v.max(axis=1) - v.min(axis=1)).mean()
        out.append(f"| {name} | {kind} | {v.mean():.0f} | {100 * v.mean() / layer.mean():.1f} | {per_wave.min():.0f} | {per_wave.max():.0f} | {spread:.0f} |")
        tot[kind] = tot.get(kind, 0.0) + v.mean()
    out.append("")
    out.append("| kind | cycles per layer and wave | % |")
    out.append("|---|---|---|")
    for k, v in tot.items():
        out.append(f"| {k} | {v:.0f} | {100 * v / layer.mean():.1f} |")
    # who waits: per wave index, the share of its layer time spent in the 'wait' rows
    waitsum = sum(v for _, kind, v in rows if kind == "wait")
    pw = (waitsum.mean(axis=(0, 2)) / layer.mean(axis=(0, 2)))
    out.append("")
    out.append("share of a wave's layer time spent at stage / hand-over barriers, by wave index: " + ", ".join(f"{100 * x:.0f} %" for x in pw))
    # time between the first and the last wave passing the hand-over barrier's arrival (how long the earliest wave waits for the last)
    arr = t[..., [2 + 5 * hd for hd in range(4)]]
    out.append(f"K/V hand-over: first arrival -> last arrival {((arr.max(axis=1) - arr.min(axis=1)).mean()):.0f} cycles on average")
    return "\n".join(out)


if __name__ == "__main__":
    for fn in sys.argv[1:]:
        print(table(fn))
        print()
