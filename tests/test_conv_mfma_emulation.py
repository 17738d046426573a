"""CPU check of the index arithmetic of csrc/dgl_conv_mfma.hip through its lane-level emulation (tests/emu_conv_mfma.py):
the emulated kernels must reproduce torch's conv1d forward / input gradient / weight gradient on bf16-rounded operands."""
import numpy as np
import torch
import torch.nn.functional as F

from tests import emu_conv_mfma as E


def _case(T1, seed):
    g = torch.Generator().manual_seed(seed)
    a1 = torch.randn(8, T1, generator=g, dtype=torch.float64)
    w = torch.randn(16, 8, 10, generator=g, dtype=torch.float64) * 0.2
    b = torch.randn(16, generator=g, dtype=torch.float64) * 0.1
    sc = torch.rand(8, generator=g, dtype=torch.float64) + 0.5
    sh = torch.randn(8, generator=g, dtype=torch.float64) * 0.1
    return a1, w, b, sc, sh


def _r(t):
    return torch.from_numpy(E.bf16(t.numpy()))


def test_conv2_forward_emulation():
    for T1 in (9 + 40, 9 + 64, 9 + 101):
        a1, w, b, sc, sh = _case(T1, T1)
        xbn = _r(a1 * sc[:, None] + sh[:, None])
        want = torch.relu(F.conv1d(xbn[None], _r(w), b)[0])
        got = E.conv2_fwd(a1.numpy(), w.numpy(), b.numpy(), sc.numpy(), sh.numpy())
        assert np.abs(got - want.numpy()).max() < 1e-9


def test_conv2_dgrad_emulation():
    for T1 in (9 + 40, 9 + 90):
        a1, w, b, sc, sh = _case(T1, 100 + T1)
        g = torch.Generator().manual_seed(T1)
        dz = torch.randn(16, T1 - 9, generator=g, dtype=torch.float64)
        x = torch.zeros(1, 8, T1, dtype=torch.float64, requires_grad=True)
        F.conv1d(x, _r(w)).backward(_r(dz)[None])
        got = E.conv2_dgrad(dz.numpy(), w.numpy(), T1)
        assert np.abs(got - x.grad[0].numpy()).max() < 1e-9


def test_conv2_wgrad_emulation():
    for T1 in (9 + 64, 9 + 150):
        a1, w, b, sc, sh = _case(T1, 200 + T1)
        g = torch.Generator().manual_seed(T1)
        dz = torch.randn(16, T1 - 9, generator=g, dtype=torch.float64)
        wt = torch.zeros(16, 8, 10, dtype=torch.float64, requires_grad=True)
        xbn = _r(a1 * sc[:, None] + sh[:, None])
        F.conv1d(xbn[None], wt).backward(_r(dz)[None])
        dw, db = E.conv2_wgrad(dz.numpy(), a1.numpy(), sc.numpy(), sh.numpy())
        assert np.abs(dw - wt.grad.numpy()).max() < 1e-8
        assert np.abs(db - dz.sum(1).numpy()).max() < 1e-9


def test_conv1_wgrad_emulation():
    for T in (9 + 64, 9 + 130):
        g = torch.Generator().manual_seed(T)
        x = torch.randn(T, generator=g, dtype=torch.float64)
        dz = torch.randn(8, T - 9, generator=g, dtype=torch.float64)
        wt = torch.zeros(8, 1, 10, dtype=torch.float64, requires_grad=True)
        F.conv1d(_r(x)[None, None], wt).backward(_r(dz)[None])
        dw, db = E.conv1_wgrad(dz.numpy(), x.numpy())
        assert np.abs(dw - wt.grad[:, 0].numpy()).max() < 1e-8
        assert np.abs(db - dz.sum(1).numpy()).max() < 1e-9
