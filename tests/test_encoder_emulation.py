"""CPU check of the fused-encoder data flow: the lane-level emulation of the HIP kernel, fed by
the real packed weight buffer, must reproduce the oracle's TSFormer hidden states."""
import numpy as np
import pytest
import torch

from oracle import step_oracle as O
from step_amd import tsformer_pack as TP
from tests import emu_encoder as E
from tests.emu_encoder import encode_sequence
from tests.helpers import load_golden, params_of, rel_l2


@pytest.fixture(params=["bf16", "f16"])
def operand(request):
    E.OPERAND = TP.OPERAND_DTYPES[request.param]
    yield request.param
    E.OPERAND = torch.bfloat16


# hidden-state rel-L2 of the emulated kernel vs the oracle: (weights-only rounding, weights + activations)
TOL = {"bf16": (1.5e-2, 2.5e-2), "f16": (2e-3, 4e-3)}


@pytest.mark.parametrize("name,seqs", [("step_tiny", [0, 7]), ("step_small", [3])])
@pytest.mark.parametrize("rnd", [False, True])
def test_emulated_kernel_matches_oracle(name, seqs, rnd, operand):
    g = load_golden(name)
    p = params_of(g, requires_grad=False)
    long0 = g["in.long_hist0"]                       # [B, L, N]
    B, L, N = long0.shape
    P = L // 12
    sd = {k[len("tsformer."):]: v for k, v in p.items() if k.startswith("tsformer.")}
    packed = TP.pack_tsformer(sd, P, operand=operand)
    assert packed.numel() == TP.total_bytes(4, P)
    assert int(packed[12:16].view(torch.int32)) == int(operand == "f16")       # header word 3
    want = O.tsformer_encode(long0, p).reshape(B * N, P, 96)
    series = long0.permute(0, 2, 1).reshape(B * N, L).double().numpy()
    for s in seqs:
        got = encode_sequence(series[s], packed, P, 4, round_bf16=rnd)
        err = rel_l2(torch.from_numpy(got), want[s])
        assert err < TOL[operand][int(rnd)], (s, operand, err)


def test_emulated_kernel_multi_wave(operand):
    """P = 40 tokens -> two waves (second one partially filled): exercises the LDS fragment
    exchange and the key mask."""
    rng = np.random.default_rng(5)
    g = load_golden("step_tiny")
    p = params_of(g, requires_grad=False)
    L = 480
    x = torch.tensor(rng.normal(size=(1, L, 2)), dtype=torch.float32)
    sd = {k[len("tsformer."):]: v for k, v in p.items() if k.startswith("tsformer.")}
    packed = TP.pack_tsformer(sd, L // 12, operand=operand)
    want = O.tsformer_encode(x, p).reshape(2, L // 12, 96)
    got = encode_sequence(x[0, :, 1].double().numpy(), packed, L // 12, 4, round_bf16=True)
    assert rel_l2(torch.from_numpy(got), want[1]) < TOL[operand][1]


def _random_pool(words, keep, seed):
    rng = np.random.default_rng(seed)
    bits = (rng.random((words, 64)) < keep).astype(np.uint64)
    return (bits << np.arange(64, dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)


def _masks_to_torch(m):
    t = torch.from_numpy
    return {"pos": t(m["pos"]), "layers": [{k: t(v) for k, v in L.items()} for L in m["layers"]]}


@pytest.mark.parametrize("P", [8, 40])
def test_emulated_dropout_matches_oracle_with_host_masks(P):
    """Training-mode dropout: the emulated kernel consumes the keep-mask words exactly like csrc/tsformer_encoder.hip (word
    offsets, lane / register maps, survivor scales folded into sqrt(d), 1/denominator, b2 and the residual fma); the oracle
    replays the dense masks tests/enc_dropout_host.py builds from the same pool through an independent index map.  With
    exact (unrounded) operands the two must agree to round-off -- except for the residual stream, which the kernel re-reads
    from its 16-bit operand copy in training mode; rnd=False keeps that copy exact too."""
    from tests import enc_dropout_host as DH
    g = load_golden("step_tiny")
    p = params_of(g, requires_grad=False)
    rng = np.random.default_rng(P)
    S, L = 3, P * 12
    x = torch.tensor(rng.normal(size=(1, L, S)), dtype=torch.float32)
    sd = {k[len("tsformer."):]: v for k, v in p.items() if k.startswith("tsformer.")}
    packed = TP.pack_tsformer(sd, P, operand="f16")
    keep, seed = 0.9, 0x1234_5678_9ABC_DEF1
    nkt = (P + 31) // 32
    pool = _random_pool(1 << 12 if nkt == 1 else 1 << 13, keep, 7)
    assert pool.shape[0] >= 2 * DH.DropLayout(nkt).words
    masks = DH.encoder_masks(pool, seed, S, P)
    for k in ("pos",):
        assert abs(masks[k].mean() - keep) < 0.02
    assert abs(masks["layers"][1]["attn"].mean() - keep) < 0.02
    pd = {k: v.double() for k, v in p.items()}
    want = O.tsformer_encode(x.double(), pd, drop=_masks_to_torch(masks), keep=keep).reshape(S, P, 96)
    nodrop = O.tsformer_encode(x.double(), pd).reshape(S, P, 96)
    E.OPERAND = torch.float16
    try:
        for s in range(S):
            got = encode_sequence(x[0, :, s].double().numpy(), packed, P, 4, round_bf16=False,
                                  drop=dict(pool=pool, seed=seed, seq=s, keep=keep))
            # weights are float16-rounded in the packed buffer, activations exact: same band as the dropout-free check above
            err = rel_l2(torch.from_numpy(got), want[s])
            assert err < TOL["f16"][0], (s, err)
            assert rel_l2(torch.from_numpy(got), nodrop[s]) > 10 * err
    finally:
        E.OPERAND = torch.bfloat16


def test_emulated_online_softmax_reshift_paths():
    """The single-pass softmax with a biased running shift: weights scaled up so that scores reach the thousands and the
    running maximum jumps by more than the head room between key tiles (the re-shift path then runs on later tiles, not only
    on the first), plus the test mode that re-shifts on every new maximum.  In exact arithmetic the result does not depend on
    when the shift moves: the two schedules must agree to round-off, and both must match the exact softmax of the oracle up
    to the rounding of the packed weights (amplified by the x16 scores)."""
    g = load_golden("step_tiny")
    p = params_of(g, requires_grad=False)
    rng = np.random.default_rng(11)
    P = 72
    L = P * 12
    x = torch.tensor(rng.normal(size=(1, L, 1)) * np.linspace(0.2, 3.0, L)[None, :, None], dtype=torch.float32)
    p = dict(p)
    for l in range(4):
        k = f"tsformer.encoder.transformer_encoder.layers.{l}.self_attn.in_proj_weight"
        p[k] = p[k] * 4.0                                   # scores x16
    sd = {k[len("tsformer."):]: v for k, v in p.items() if k.startswith("tsformer.")}
    packed = TP.pack_tsformer(sd, P, operand="f16")
    want = O.tsformer_encode(x.double(), {k: v.double() for k, v in p.items()}).reshape(P, 96)
    E.OPERAND = torch.float16
    got, counts, redone = {}, {}, {}
    try:
        for always in (False, True):
            E.STATS.update(reshifts=0, tiles=0, redone=0)
            got[always] = encode_sequence(x[0, :, 0].double().numpy(), packed, P, 4, round_bf16=False, always_reshift=always)
            counts[always] = E.STATS["reshifts"]
            redone[always] = E.STATS["redone"]
    finally:
        E.OPERAND = torch.bfloat16
    assert counts[False] > 0, "the input does not exercise the re-shift path"
    # the fixed-shift schedule overflowed on some heads (and only those were redone with the re-shifting loop)
    assert 0 < redone[False] < 4 * 4 * ((P + 31) // 32) and redone[True] == 0, redone
    assert counts[True] > counts[False]
    assert rel_l2(torch.from_numpy(got[False]), torch.from_numpy(got[True])) < 1e-9
    assert rel_l2(torch.from_numpy(got[False]), want) < 16 * TOL["f16"][0]


def test_packed_patch_embedding_section_is_the_f32_mfma_operand_layout():
    """csrc/tsformer_layout.h TSF_G_WPE: value (t, s, lane) = W_pe[32 t + lane % 32][2 s + lane / 32] -- what lane `lane` hands
    v_mfma_f32_32x32x2_f32 as its A operand in k-step s of feature block t; and the positional table carries b_pe."""
    g = load_golden("step_tiny")
    p = params_of(g, requires_grad=False)
    sd = {k[len("tsformer."):]: v for k, v in p.items() if k.startswith("tsformer.")}
    P = 40
    raw = TP.pack_tsformer(sd, P, operand="f16").numpy().tobytes()
    sec = np.frombuffer(raw, dtype=np.float32, count=3 * 6 * 64, offset=TP.HDR).reshape(3, 6, 64)
    w = sd["patch_embedding.input_embedding.weight"][:, 0, :, 0].numpy()          # [96, 12]
    for t in range(3):
        for s in range(6):
            for lane in (0, 1, 31, 32, 45, 63):
                assert sec[t, s, lane] == w[32 * t + lane % 32, 2 * s + lane // 32]
    pos = np.frombuffer(raw, dtype=np.float32, count=P * 96, offset=TP.LAYER0 + 4 * TP.layer_bytes()).reshape(P, 2, 48)
    want = (sd["positional_encoding.position_embedding"][:P] + sd["patch_embedding.input_embedding.bias"][None, :]).numpy()
    rows = E.ROW                                                                    # [2, 16]: accumulator register -> row of a 32-row tile
    for tok in (0, 7, P - 1):
        for h in (0, 1):
            feat = np.concatenate([32 * t + rows[h] for t in range(3)])
            assert np.array_equal(pos[tok, h], want[tok, feat])
