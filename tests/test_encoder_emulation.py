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
