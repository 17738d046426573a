"""The operand-fragment index maps of the fused feed-forward kernels (csrc/pretrain_fused.hip), executed on the host with a model of the
matrix instruction (tests/ffn_fused_host.py): forward, input gradient and the three parameter gradients of
f2 = W2 . keep(relu(W1 . h1 + b1)) / (1 - p) + b2 against plain matrix algebra, with and without a keep-mask pool, R not a multiple of 32.
What this pins: the chain k-slot map both operands of a product have to share, the transposed roles in the weight-gradient kernels
(hidden unit as the lane), and that the lane-mask form of the pool words (forward, backward-data) and the word-per-unit form (weight
gradients) address the same bit."""
import numpy as np
import pytest

from tests import ffn_fused_host as F


@pytest.mark.parametrize("drop", [False, True])
def test_fragment_maps_compute_the_feed_forward_block(drop):
    rng = np.random.default_rng(3)
    R = 32 * 2 + 13
    x = rng.standard_normal((R, 96))
    w1 = rng.standard_normal((384, 96)) * 0.15
    b1 = rng.standard_normal(384) * 0.1
    w2 = rng.standard_normal((96, 384)) * 0.1
    b2 = rng.standard_normal(96) * 0.1
    df2 = rng.standard_normal((R, 96))
    dh1_in = rng.standard_normal((R, 96))
    pool, seed, site, p = None, 0x1234_5678_9ABC, 34, 0.0
    keep = np.ones((R, 384), dtype=bool)
    if drop:
        p = 0.25
        pool = rng.integers(0, 1 << 63, size=1024, dtype=np.uint64) | (rng.integers(0, 2, size=1024, dtype=np.uint64) << np.uint64(63))
        keep = F.keep_matrix(pool, seed, site, R)
        assert 0.4 < keep.mean() < 0.6
    ik = 1.0 / (1.0 - p)
    pre = x @ w1.T + b1
    hid = np.maximum(pre, 0) * keep * ik
    want_f2 = hid @ w2.T + b2
    dhid = (df2 @ w2) * (pre > 0) * keep * ik
    want_dh1 = dh1_in + dhid @ w1
    want_dw1, want_db1, want_dw2 = dhid.T @ x, dhid.sum(0), df2.T @ hid
    got_f2 = F.forward(x, w1, b1, w2, b2, pool, seed, site, ik)
    got_dh1 = F.backward_data(df2, x, w1, b1, w2, dh1_in, pool, seed, site, ik)
    got_dw1, got_db1, got_dw2 = F.backward_weights(df2, x, w1, b1, w2, pool, seed, site, ik)
    for name, a, b in (("f2", got_f2, want_f2), ("dh1", got_dh1, want_dh1), ("dw1", got_dw1, want_dw1), ("db1", got_db1, want_db1),
                       ("dw2", got_dw2, want_dw2)):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-9), name


def test_tile_staging_maps_are_bijective_and_conflict_free():
    """The wave-private LDS staging of the row kernels (csrc/pretrain_fused.hip tile_put / tile_put_bf16 / tile_frags / tile_out /
    tile_out_bf16), executed on the host address by address: every element of a 32 x 96 tile lands once, the fragments read back are the
    transposed-layout fragments of the tile (chunk c of row r at position c ^ ((r >> 1) & 7), 8-byte chunks), the 16 lanes of one
    ds_read_b64 pass hit 16 different 8-byte bank pairs, and the output staging returns every row in row-major order."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((32, 96))
    want = F.pack_T(F.rows_T(x, 0))

    def frags_from(lds):                                   # tile_frags; lds in units of one 2-byte element
        out = []
        for f in range(6):
            got = np.zeros((64, 8))
            for lane in range(64):
                r, h = lane % 32, lane // 32
                sw = (r >> 1) & 7
                a, b = (r * 192 + 8 * ((4 * f + h) ^ sw)) // 2, (r * 192 + 8 * ((4 * f + h + 2) ^ sw)) // 2
                got[lane, :4], got[lane, 4:] = lds[a:a + 4], lds[b:b + 4]
            out.append(got)
        return out
    # f32 source: 12 coalesced float4 loads per lane, one 8-byte LDS write each (tile_put)
    lds = np.full(32 * 96, np.nan)
    for k in range(12):
        for lane in range(64):
            flat = k * 256 + lane * 4
            r, c = flat // 96, (flat % 96) >> 2
            pos = (r * 192 + 8 * (c ^ ((r >> 1) & 7))) // 2
            assert np.isnan(lds[pos:pos + 4]).all()
            lds[pos:pos + 4] = x.reshape(-1)[flat:flat + 4]
    assert not np.isnan(lds).any()
    assert all(np.array_equal(g, w) for g, w in zip(frags_from(lds), want))
    # bf16 source: 6 loads of 8 elements per lane, one 16-byte write with the halves swapped where the swizzle is odd (tile_put_bf16)
    lds = np.full(32 * 96, np.nan)
    for k in range(6):
        for lane in range(64):
            idx = k * 64 + lane
            r, c16 = idx // 12, idx % 12
            sw = (r >> 1) & 7
            v = x[r, c16 * 8:c16 * 8 + 8]
            a = (r * 192 + 8 * (((2 * c16) ^ sw) & ~1)) // 2
            lds[a:a + 8] = np.concatenate([v[4:], v[:4]]) if sw & 1 else v
    assert all(np.array_equal(g, w) for g, w in zip(frags_from(lds), want))
    # the 16 lanes of one ds_read_b64 pass (rows r0 .. r0 + 15 of one lane half, one chunk) use 16 different 8-byte slots of the 128-byte line
    for f in range(6):
        for h in range(2):
            for r0 in (0, 16):
                for add in (0, 2):
                    slots = {((r * 192 + 8 * ((4 * f + h + add) ^ ((r >> 1) & 7))) // 8) % 16 for r in range(r0, r0 + 16)}
                    assert len(slots) == 16
    # f32 output staging, one 32-feature block at a time (tile_out): written from the transposed layout, read back as rows
    acc = F.rows_T(x, 0)
    y = np.zeros((32, 96))
    for t in range(3):
        ost = np.full(32 * 32, np.nan)
        for lane in range(64):
            r, h = lane % 32, lane // 32
            for q in range(4):
                a = (r * 128 + 16 * ((2 * q + h) ^ (r & 7))) // 4
                ost[a:a + 4] = acc[t][lane, 4 * q:4 * q + 4]
        assert not np.isnan(ost).any()
        for n in range(4):
            for lane in range(64):
                rr, cq = (n * 64 + lane) >> 3, lane & 7
                a = (rr * 128 + 16 * (cq ^ (rr & 7))) // 4
                y[rr, 32 * t + 4 * cq:32 * t + 4 * cq + 4] = ost[a:a + 4]
    assert np.array_equal(y, x)
    # bf16 output staging over the input region (tile_out_bf16)
    lds = np.full(32 * 96, np.nan)
    for lane in range(64):
        r, h = lane % 32, lane // 32
        sw = (r >> 1) & 7
        for t in range(3):
            for q in range(4):
                a = (r * 192 + 8 * ((8 * t + 2 * q + h) ^ sw)) // 2
                lds[a:a + 4] = acc[t][lane, 4 * q:4 * q + 4]
    y = np.zeros((32, 96))
    for k in range(6):
        for lane in range(64):
            idx = k * 64 + lane
            rr, c16 = idx // 12, idx % 12
            s2 = (rr >> 1) & 7
            a = (rr * 192 + 8 * (((2 * c16) ^ s2) & ~1)) // 2
            v = lds[a:a + 8]
            y[rr, c16 * 8:c16 * 8 + 8] = np.concatenate([v[4:], v[:4]]) if s2 & 1 else v
    assert np.array_equal(y, x)
