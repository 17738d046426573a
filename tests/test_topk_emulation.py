"""CPU check of the multi-workgroup radix top-k (csrc/knn.hip) through tests/emu_topk.py: digit split, state recomputation,
per-slice tie counts and the ordered tie fill against an independent sort-based statement of the selection rule."""
import numpy as np
import pytest

from tests import emu_topk as T

rng = np.random.default_rng(3)


def test_order_key_is_monotone():
    x = np.array([-np.inf, -3.5, -1e-30, 0.0, 1e-30, 0.25, 1.0, 7.0, np.inf], np.float32)
    k = T.order_key(x).astype(np.int64)
    assert np.all(np.diff(k) > 0)


@pytest.mark.parametrize("N,k", [(67, 670), (91, 10 * 91), (130, 1), (65, 65 * 65), (64, 4095)])
def test_distinct_values(N, k):
    sim = rng.uniform(-1, 1, (N, N)).astype(np.float32)
    got, _, need_eq = T.topk_mask(sim, k)
    assert need_eq >= 1
    assert np.array_equal(got, T.reference_mask(sim, k))


def test_ties_at_the_cut_are_taken_in_flat_index_order_across_slices():
    # 3 slices (N*N = 9409 > 2*4096); a quantised matrix puts hundreds of entries on the threshold value, spread over all slices
    N, k = 97, 970
    sim = (rng.integers(-8, 9, (N, N)) / 8.0).astype(np.float32)
    sim[sim == 0] = 0.0625                                    # keep -0.0 / exact zeros out of this case
    got, thr, need_eq = T.topk_mask(sim, k)
    n_thr = int((T.order_key(sim) == thr).sum())
    assert 1 <= need_eq < n_thr                              # the cut really falls inside a run of equal values
    assert np.array_equal(got, T.reference_mask(sim, k))


def test_zero_entries_and_diagonal_are_dropped():
    # the reference keeps `scattered != 0` and clears the diagonal: a selected exact zero does not become an edge
    N, k = 70, 3000
    sim = rng.uniform(-1, 1, (N, N)).astype(np.float32)
    sim[rng.random((N, N)) < 0.3] = 0.0
    np.fill_diagonal(sim, 1.0)
    got, _, _ = T.topk_mask(sim, k)
    assert np.array_equal(got, T.reference_mask(sim, k))
    assert got.diagonal().sum() == 0 and got[sim == 0].sum() == 0


def test_all_equal_matrix():
    N, k = 66, 500
    sim = np.full((N, N), 0.5, np.float32)
    got, _, need_eq = T.topk_mask(sim, k)
    assert need_eq == k
    assert np.array_equal(got, T.reference_mask(sim, k))
