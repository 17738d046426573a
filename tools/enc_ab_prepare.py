"""Stage the torch-free encoder A/B harness (tools/enc_ab.cpp) for a GPU box: packed weights (bf16 and f16 fragments), 12
full-length series, the CPU oracle's hidden states, the harness binary and the library builds to compare, all under
scratch_ab/ (git-ignored, shipped by gpurun).  A gpurun call of the harness costs ~20-45 s of GPU budget including the box:

    python tools/enc_ab_prepare.py                     # default build  -> scratch_ab/libstep_default.so
    python tools/enc_ab_prepare.py lcg -DTSF_DROPOUT_LCG=1    # + a variant build -> scratch_ab/libstep_lcg.so
    gpurun --timeout 120 -- 'mkdir -p gpurun_out && cd scratch_ab && ./enc_ab ./libstep_default.so default 0; ./enc_ab ./libstep_lcg.so lcg 2'

Third harness argument: generator whose selftest stream is dumped to gpurun_out/dropout_stream_gen<k>.bin (-1 / absent: none).
"""
import os
import shutil
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "scratch_ab")


def main():
    from oracle import step_oracle as O
    from step_amd import build as B, tsformer_pack as TP
    from step_amd.step_arch.tsformer import TSFormer
    os.makedirs(OUT, exist_ok=True)
    P, S0 = 336, 12
    L = 12 * P
    torch.manual_seed(0)
    m = TSFormer(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, num_token=P, mask_ratio=0.75,
                 encoder_depth=4, decoder_depth=1, mode="forecasting")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    for k, v in sd.items():                       # sharper attention than the 0.02-std initialisation, no exactly-zero biases
        if v.ndim >= 2 and "position" not in k and "mask_token" not in k:
            sd[k] = v * 3.0
        elif k.endswith("bias"):
            sd[k] = v + 0.1 * torch.randn(v.shape, generator=g)
    p = {"tsformer." + k: v for k, v in sd.items()}
    rng = np.random.default_rng(0)
    t = np.arange(L)
    series = np.stack([np.sin(2 * np.pi * t / 288 + rng.uniform(0, 6)) * rng.uniform(0.5, 1.5) + 0.3 * np.sin(2 * np.pi * t / 2016)
                       + 0.25 * rng.standard_normal(L) for _ in range(S0)]).astype(np.float32)
    want = O.tsformer_encode(torch.from_numpy(series.T.copy())[None], p).reshape(S0, P, 96).numpy().astype(np.float32)
    series.tofile(os.path.join(OUT, "series_small.bin"))
    want.tofile(os.path.join(OUT, "want_hidden.bin"))
    for op in ("bf16", "f16"):
        TP.pack_tsformer(sd, P, operand=op).numpy().tofile(os.path.join(OUT, f"pack_{op}.bin"))
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-o", os.path.join(OUT, "enc_ab"), os.path.join(ROOT, "tools", "enc_ab.cpp"), "-ldl"])
    shutil.copy(B.build(verbose=False), os.path.join(OUT, "libstep_default.so"))
    if len(sys.argv) > 2:
        name = sys.argv[1]
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), name, *sys.argv[2:]])
        shutil.copy(os.path.join(B.HERE, f"libstep_hip_{name}.so"), os.path.join(OUT, f"libstep_{name}.so"))
    print(sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
