"""Numpy restatement of the multi-workgroup radix top-k of csrc/knn.hip (tk_hist / tk_ties / tk_mask), slice by slice and
thread by thread, so that the digit split (11/11/10 bits), the state recomputation from the finished histograms, the
per-slice tie counts and the ordered tie fill are checked on the CPU.  Test infrastructure only.

The selection rule being implemented (discrete_graph_learning.py:100-108 of the reference: `topk` of the flattened
similarity, scatter, `!= 0`, diagonal cleared) is restated independently in `reference_mask` below.
"""
import numpy as np

TK_BINS, TK_THREADS, TK_EPT = 2048, 256, 16
TK_SLICE = TK_THREADS * TK_EPT
SHIFT = (21, 10, 0)
MASK = (0x7FF, 0x7FF, 0x3FF)


def order_key(x):
    """f32_order_key: monotone map float32 -> uint32 (negative floats bit-flipped, others get the top bit)."""
    u = np.asarray(x, np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def _select(hist, rem):
    """tk_select: thread t owns bins 2047-8t-i (i<8); block exclusive scan of the per-thread totals, then the bin where the
    running count reaches rem."""
    per = TK_BINS // TK_THREADS
    c = np.array([[hist[TK_BINS - 1 - (t * per + i)] for i in range(per)] for t in range(TK_THREADS)], np.uint64)
    run0 = np.concatenate([[0], np.cumsum(c.sum(1))[:-1]])
    found = None
    for t in range(TK_THREADS):
        run = int(run0[t])
        for i in range(per):
            if run < rem and run + int(c[t, i]) >= rem:
                assert found is None, "exactly one thread may publish the bin"
                found = (TK_BINS - 1 - (t * per + i), rem - run)
            run += int(c[t, i])
    assert found is not None
    return found


def _state(hists, upto, k_total):
    prefix, rem = 0, k_total
    for d in range(upto):
        b, rem = _select(hists[d], rem)
        prefix |= b << SHIFT[d]
    return prefix, rem


def topk_mask(sim, k_total):
    """sim [N,N] float32 -> adj [N,N] float32, following the five launches."""
    N = sim.shape[0]
    E = N * N
    v = np.ascontiguousarray(sim, np.float32).reshape(-1)
    key = order_key(v)
    slices = (E + TK_SLICE - 1) // TK_SLICE
    k = min(k_total, E)
    hists = np.zeros((3, TK_BINS), np.uint64)
    for d in range(3):                                            # tk_hist_kernel, one launch per digit
        prefix, _ = _state(hists, d, k)
        himask = 0 if d == 0 else (0xFFFFFFFF << SHIFT[d - 1]) & 0xFFFFFFFF
        for s in range(slices):
            ks = key[s * TK_SLICE:min(E, (s + 1) * TK_SLICE)]
            ok = (ks & np.uint32(himask)) == np.uint32(prefix)
            local = np.bincount(((ks[ok] >> np.uint32(SHIFT[d])) & np.uint32(MASK[d])).astype(np.int64), minlength=TK_BINS)
            hists[d] += local.astype(np.uint64)
    thr, need_eq = _state(hists, 3, k)                            # tk_ties_kernel
    ties = np.array([(key[s * TK_SLICE:min(E, (s + 1) * TK_SLICE)] == np.uint32(thr)).sum() for s in range(slices)], np.int64)
    adj = np.zeros(E, np.float32)                                 # tk_mask_kernel
    for s in range(slices):
        before = int(ties[:s].sum())
        lo = s * TK_SLICE
        mine = np.zeros(TK_THREADS, np.int64)
        for t in range(TK_THREADS):
            a, b = lo + t * TK_EPT, min(E, lo + (t + 1) * TK_EPT)
            if a < E:
                mine[t] = (key[a:b] == np.uint32(thr)).sum()
        excl = np.concatenate([[0], np.cumsum(mine)[:-1]])
        for t in range(TK_THREADS):
            rank = before + int(excl[t])
            for j in range(TK_EPT):
                e = lo + t * TK_EPT + j
                if e >= E:
                    break
                sel = key[e] > thr
                if key[e] == thr:
                    sel = rank < need_eq
                    rank += 1
                i, jj = divmod(e, N)
                adj[e] = 1.0 if (sel and v[e] != 0.0 and i != jj) else 0.0
    return adj.reshape(N, N), thr, need_eq


def reference_mask(sim, k_total):
    """Independent statement: the k largest entries of the flattened matrix, ties at the cut broken by ascending flat index;
    entries that are exactly zero and the diagonal are dropped."""
    N = sim.shape[0]
    v = np.asarray(sim, np.float32).reshape(-1)
    k = min(k_total, v.size)
    order = np.lexsort((np.arange(v.size), -v.astype(np.float64)))       # value descending, index ascending
    adj = np.zeros(v.size, np.float32)
    adj[order[:k]] = 1.0
    adj[v == 0.0] = 0.0
    adj = adj.reshape(N, N)
    np.fill_diagonal(adj, 0.0)
    return adj
