"""Host mirrors of the index maps of the pre-training attention kernels (csrc/pretrain.hip) -- CPU checks of properties the kernels
rely on, next to the GPU tests that check the values (tests/test_gpu_pretrain.py).

* `attn_keep_index` / `ma_key`: the 16 keys one lane of a 32 x 32 score tile holds must be 16 CONSECUTIVE 16-bit fields of the dropout
  stream (two Philox calls per lane and tile), every key of a row must own exactly one field, and the f32 kernels (which walk keys in
  order) must address the same fields.
* `ma_unit`: workgroups are dealt round-robin to 8 XCDs; the map must be a bijection onto the (sequence, head) units and hand every
  XCD a contiguous range, so that the four heads of a sequence meet in one L2.
* keep-bit words: bit `ma_key(e, h)` of the word of (query, key tile).
"""
import numpy as np
import pytest


def ma_key(e, h):                       # accumulator register e of lane half h -> row (key) of the 32 x 32 tile
    return (e & 3) + 8 * (e >> 2) + 4 * h


def attn_keep_index(row, T32, k):       # field number of (row, key k) in the attention-dropout stream
    k32 = k & 31
    return row * T32 + (k & ~31) + 16 * ((k32 >> 2) & 1) + 4 * (k32 >> 3) + (k32 & 3)


def ma_unit(b, n):
    per, rem, x = n >> 3, n & 7, b & 7
    return x * per + min(x, rem) + (b >> 3)


@pytest.mark.parametrize("T", [40, 42, 77, 168, 336])
def test_lane_keys_are_consecutive_fields_and_every_key_has_one(T):
    T32 = (T + 31) & ~31
    for row in (0, 5, 4 * T - 1):
        for kt in range(T32 // 32):
            for h in (0, 1):
                f = [attn_keep_index(row, T32, kt * 32 + ma_key(e, h)) for e in range(16)]
                base = row * T32 + kt * 32 + 16 * h
                assert f == list(range(base, base + 16))          # what ma_keep16 draws: blocks base >> 3 and (base >> 3) + 1, fields 0..7 each
                assert base % 8 == 0
        fields = [attn_keep_index(row, T32, k) for k in range(T32)]
        assert sorted(fields) == list(range(row * T32, (row + 1) * T32))          # a bijection onto the row's fields
    # rows do not overlap
    assert attn_keep_index(1, T32, 0) == T32 and max(attn_keep_index(0, T32, k) for k in range(T32)) == T32 - 1


def test_keep_bit_word_layout():
    keys = sorted(ma_key(e, h) for h in (0, 1) for e in range(16))
    assert keys == list(range(32))                               # the two lane halves cover the 32 keys of a tile exactly once
    # bit ma_key(e, h) of the word: the backward's phase A reads (word >> (8 g + 4 h + j)) & 1 for m[4 g + j]
    for h in (0, 1):
        for g in range(4):
            for j in range(4):
                assert ma_key(4 * g + j, h) == 8 * g + 4 * h + j


@pytest.mark.parametrize("n", [8, 20800, 20803, 5, 4 * 5200 + 7])
def test_xcd_contiguous_unit_map_is_a_bijection(n):
    units = [ma_unit(b, n) for b in range(n)]
    assert sorted(units) == list(range(n))
    for x in range(8):                                            # XCD x (blocks b = x mod 8) gets one contiguous, ascending range
        mine = [ma_unit(b, n) for b in range(x, n, 8)]
        assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True
    if n >= 64:
        # the four heads of a sequence (units 4 s .. 4 s + 3) run on one XCD, except where a sequence straddles two ranges
        xcd_of = {ma_unit(b, n): b & 7 for b in range(n)}
        split = sum(len({xcd_of[4 * s + hd] for hd in range(4)}) > 1 for s in range(n // 4))
        assert split <= 7


def test_split_k_reduce_index_math():
    """splitk_reduce_kernel: element e of [batch][M][N] -> (i0, i1, m, n) with batch = i1 * batch0 + i0, written at
    i0 * scb + i1 * scb1 + m * ldc + n * scn; the staged kernel wrote split z of the same element at z * total + e."""
    M, N, batch0, nb1 = 5, 7, 3, 2
    batch, mn = batch0 * nb1, M * N
    scb, scb1, ldc, scn = 1000, 5000, 1, M            # a transposed C, like the weight gradients written as C^T
    seen = set()
    for e in range(mn * batch):
        b, r = divmod(e, mn)
        m, n = divmod(r, N)
        i0, i1 = b % batch0, b // batch0
        assert (i1 * batch0 + i0) * mn + m * N + n == e          # the workspace offset the GEMM used (scb = mn, scb1 = batch0 * mn)
        seen.add(i0 * scb + i1 * scb1 + m * ldc + n * scn)
    assert len(seen) == mn * batch
