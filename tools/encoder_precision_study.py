"""What the 16-bit operand type of the fused encoder costs in accuracy (CPU only, no GPU involved).

The lane-level emulator of the kernel (tests/emu_encoder.py) is run at the full PEMS04 sequence length (P = 336 tokens,
11 waves) on a freshly initialised TSFormer, with the operand fragments -- weights in the packed buffer and the activation
fragments the kernel rounds when an accumulator tile becomes the next MFMA operand -- held as

  bf16/bf16   what csrc/tsformer_encoder.hip does today (v_mfma_f32_32x32x16_bf16)
  f16/f16     the same data flow on v_mfma_f32_32x32x16_f16 (same rate on gfx950), weights packed as float16
  f16/bf16    float16 weights, bfloat16 activations (isolates the weight rounding)
  bf16/exact  bfloat16 weights, activations never rounded (the weight-rounding floor of today's kernel)
  f16/exact   float16 weights, activations never rounded

against the CPU oracle (oracle/step_oracle.py, fp32).  Also reports the largest activation magnitude that becomes an operand
(float16 overflows at 65504).  Output: one JSON object; committed as profiles/r01_v_encoder_precision_study.json.

    python tools/encoder_precision_study.py [--sequences 3] [--tokens 336] [--weight-scale 1|3]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import step_oracle as O                   # noqa: E402
from step_amd import tsformer_pack as TP              # noqa: E402
from tests import emu_encoder as E                    # noqa: E402
from tests.helpers import rel_l2                      # noqa: E402


def packer_for(dtype_name):
    """The product packer with its 16-bit conversion retargeted (same fragment layout: both types are 2 bytes)."""
    op = {"bfloat16": "bf16", "float16": "f16"}[dtype_name]
    return lambda sd, P: TP.pack_tsformer(sd, P, operand=op)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=3)
    ap.add_argument("--tokens", type=int, default=336)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weight-scale", type=float, default=1.0,
                    help="multiply every weight matrix of the fresh initialisation (1 = as initialised; 3 = sharper attention, "
                         "closer to a trained checkpoint's weight norms)")
    args = ap.parse_args()
    P = args.tokens
    L = 12 * P
    torch.manual_seed(args.seed)
    from step_amd.step_arch.tsformer import TSFormer
    model = TSFormer(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, num_token=P, mask_ratio=0.75,
                     encoder_depth=4, decoder_depth=1, mode="forecasting")
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # a trained checkpoint has larger projection weights than the initialisation: --weight-scale makes attention non-uniform;
    # biases get a perturbation so that no term is exactly zero
    g = torch.Generator().manual_seed(args.seed + 1)
    for k, v in sd.items():
        if v.ndim >= 2 and "position" not in k and "mask_token" not in k:
            sd[k] = v * args.weight_scale
        elif k.endswith("bias"):
            sd[k] = v + 0.1 * torch.randn(v.shape, generator=g)
    p = {"tsformer." + k: v for k, v in sd.items()}
    rng = np.random.default_rng(args.seed)
    t = np.arange(L)
    series = np.stack([np.sin(2 * np.pi * t / 288 + rng.uniform(0, 6)) * rng.uniform(0.5, 1.5) + 0.3 * np.sin(2 * np.pi * t / 2016)
                       + 0.25 * rng.standard_normal(L) for _ in range(args.sequences)]).astype(np.float32)
    want = O.tsformer_encode(torch.from_numpy(series.T.copy())[None], p).reshape(args.sequences, P, 96)
    variants = [("bf16/bf16", "bfloat16", torch.bfloat16, True), ("f16/f16", "float16", torch.float16, True),
                ("f16/bf16", "float16", None, True), ("bf16/exact", "bfloat16", torch.bfloat16, False), ("f16/exact", "float16", torch.float16, False)]
    out = {"tokens": P, "sequences": args.sequences, "weight_scale": args.weight_scale, "variants": {}}
    for name, wtype, atype, rnd in variants:
        packed = packer_for(wtype)(sd, P)
        errs = []
        E.PEAK["abs"] = 0.0
        for s in range(args.sequences):
            if name == "f16/bf16":
                # weights decoded as float16, activations rounded to bfloat16
                E.OPERAND = torch.float16
                frag = E.Buf.frag
                E.bf16_round = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float64).numpy()
                got = E.encode_sequence(series[s].astype(np.float64), packed, P, 4, round_bf16=True)
                E.bf16_round = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(E.OPERAND).to(torch.float64).numpy()
                del frag
            else:
                E.OPERAND = atype
                got = E.encode_sequence(series[s].astype(np.float64), packed, P, 4, round_bf16=rnd)
            errs.append(float(rel_l2(torch.from_numpy(got), want[s])))
        E.OPERAND = torch.bfloat16
        out["variants"][name] = {"hidden_rel_l2": errs, "mean": float(np.mean(errs)), "peak_operand_abs": E.PEAK["abs"]}
        print(name, out["variants"][name], file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
