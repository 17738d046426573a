"""Stage the torch-free encoder A/B harness (tools/enc_ab.cpp) for a GPU box under scratch_ab/ (git-ignored, shipped by gpurun):
packed weights (bf16 and f16 fragments), 12 full-length series, the CPU oracle's hidden states without dropout and with the
dropout masks of a keep-mask pool (drop_pool.bin / drop_seed.bin, masks rebuilt on the host by tests/enc_dropout_host.py), the
harness binary, and one small library per variant (csrc/tsformer_encoder.hip + errors.cpp only, so a variant builds in ~15 s):

    python tools/enc_ab_prepare.py default pipe0:-DTSF_PIPE=0 pipe1:-DTSF_PIPE=1 prio1:-DTSF_SETPRIO=1 old@HEAD~3
    gpurun --timeout 300 -- 'cd scratch_ab && ./enc_ab default=./libenc_default.so pipe0=./libenc_pipe0.so ... > ../gpurun_out/enc_ab.log'

`name:-Dflag[,-Dflag]` = extra defines; `name@rev` = the sources of git revision `rev` (an ABI 3 kernel is called through its old
signature, which lets the previous round's kernel be timed in the same process).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "scratch_ab")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build_variant(name, defines, rev=None):
    lib = os.path.join(OUT, f"libenc_{name}.so")
    with tempfile.TemporaryDirectory() as tmp:
        if rev is None:
            csrc = os.path.join(ROOT, "step_amd", "csrc")
        else:                           # check the revision's csrc/ and include/ out into a scratch tree
            for d in ("step_amd/csrc", "include"):
                os.makedirs(os.path.join(tmp, d), exist_ok=True)
                files = subprocess.check_output(["git", "-C", ROOT, "ls-tree", "--name-only", rev, d + "/"]).decode().split()
                for f in files:
                    with open(os.path.join(tmp, f), "wb") as fh:
                        fh.write(subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:{f}"]))
            csrc = os.path.join(tmp, "step_amd", "csrc")
        objs = []
        for src in ("tsformer_encoder.hip", "errors.cpp"):
            o = os.path.join(tmp, src + ".o")
            subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", *defines, "-c",
                                   os.path.join(csrc, src), "-o", o])
            objs.append(o)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def main():
    from oracle import step_oracle as O
    from step_amd import tsformer_pack as TP
    from step_amd.step_arch.tsformer import TSFormer
    from tests import enc_dropout_host as DH
    os.makedirs(OUT, exist_ok=True)
    for P in (336, 168):
        fixtures(O, TP, TSFormer, DH, P)
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "-o", os.path.join(OUT, "enc_ab"), os.path.join(ROOT, "tools", "enc_ab.cpp"), "-ldl"])
    from concurrent.futures import ThreadPoolExecutor

    def one(spec):
        if "@" in spec:
            name, rev = spec.split("@", 1)
            return build_variant(name, [], rev)
        name, _, d = spec.partition(":")
        return build_variant(name, [x for x in d.split(",") if x])
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:          # one hipcc per variant, in parallel
        for lib in ex.map(one, sys.argv[1:] or ["default"]):
            print(lib)
    print(sorted(os.listdir(OUT)))


def fixtures(O, TP, TSFormer, DH, P):
    """weights, series, oracle outputs for P tokens per sequence; files of P != 336 carry the suffix _p<P> (ENC_AB_P of the harness)"""
    S0 = 12
    sfx = "" if P == 336 else f"_p{P}"
    L = 12 * P
    torch.manual_seed(0)
    m = TSFormer(patch_size=12, in_channel=1, embed_dim=96, num_heads=4, mlp_ratio=4, dropout=0.1, num_token=P, mask_ratio=0.75,
                 encoder_depth=4, decoder_depth=1, mode="forecasting")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # the untouched initialisation (what bench.py's random-init model holds): timing only, see enc_ab.cpp
    TP.pack_tsformer(sd, P, operand="f16").numpy().tofile(os.path.join(OUT, f"pack_f16_plain{sfx}.bin"))
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
    x = torch.from_numpy(series.T.copy())[None]
    want = O.tsformer_encode(x, p).reshape(S0, P, 96).numpy().astype(np.float32)
    # dropout: a host-made pool, the dense masks the kernel will take from it, the oracle replaying them
    keep, seed, words = 0.9, 0x5EED_0123_4567_89AB, 1 << 16
    bits = (rng.random((words, 64)) < keep).astype(np.uint64)
    pool = (bits << np.arange(64, dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)
    masks = DH.encoder_masks(pool, seed, S0, P)
    tt = torch.from_numpy
    md = {"pos": tt(masks["pos"]), "layers": [{k: tt(v) for k, v in Lr.items()} for Lr in masks["layers"]]}
    wantd = O.tsformer_encode(x, p, drop=md, keep=keep).reshape(S0, P, 96).numpy().astype(np.float32)
    series.tofile(os.path.join(OUT, f"series_small{sfx}.bin"))
    want.tofile(os.path.join(OUT, f"want_hidden{sfx}.bin"))
    wantd.tofile(os.path.join(OUT, f"want_hidden_drop{sfx}.bin"))
    pool.tofile(os.path.join(OUT, "drop_pool.bin"))
    np.array([seed], dtype=np.uint64).tofile(os.path.join(OUT, "drop_seed.bin"))
    for op in ("bf16", "f16"):
        TP.pack_tsformer(sd, P, operand=op).numpy().tofile(os.path.join(OUT, f"pack_{op}{sfx}.bin"))


if __name__ == "__main__":
    main()
