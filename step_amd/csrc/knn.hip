// kNN prior graph on device: cosine similarities of the TSFormer hidden states and an exact
// global top-k over the flattened N*N matrix per sample (radix select, 3 digits of 11/11/10 bits, many workgroups
// per sample).  The Gram matrix runs on the staged bf16 GEMM of gemm_bf16.hip.
//
// Tie rule (documented in DESIGN.md): entries strictly above the k-th largest value are always
// selected; entries equal to it are selected in ascending flat-index order until k are chosen.
// torch.topk cuts ties arbitrarily, so parity is "identical except at ties on the threshold".
#include "common.h"
#include "step_internal.h"

namespace {

__device__ __forceinline__ uint32_t f32_order_key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// raw Gram -> cosine: sim[i][j] = raw / ((|F_i| + 1e-7)(|F_j| + 1e-7))        (similarity.py:8-14)
__global__ __launch_bounds__(256) void cosine_finalize_kernel(float* __restrict__ sim, const float* __restrict__ sqn_part,
                                                              int N) {
    const int b = blockIdx.z;
    const int i = blockIdx.y;
    float* row = sim + ((long)b * N + i) * N;
    auto norm_of = [&](int n) -> float {
        float s = 0.f;
        if (sqn_part) {
            const float* p = sqn_part + ((long)b * N + n) * 16;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += p[w];
        } else {
            s = sim[((long)b * N + n) * N + n];
        }
        return sqrtf(fmaxf(s, 0.f)) + 1e-7f;
    };
    // the diagonal is read by other rows when sqn_part == NULL: those reads race with the
    // in-place write below only for column i of row i, which is written last by one thread.
    const float ni = norm_of(i);
    for (int j = blockIdx.x * 256 + threadIdx.x; j < N; j += gridDim.x * 256) {
        if (j == i && !sqn_part) continue;
        row[j] = row[j] / (ni * norm_of(j));
    }
}
__global__ void cosine_diag_kernel(float* sim, int N) {   // only for the sqn_part == NULL path
    const int b = blockIdx.y;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N) {
        float* d = sim + ((long)b * N + i) * N + i;
        float n = sqrtf(fmaxf(*d, 0.f)) + 1e-7f;
        *d = *d / (n * n);
    }
}

// ------------------------------------------------------------------------------------------
// Exact global top-k over the flattened N*N similarities of a sample: radix select of the k-th largest key in three
// 11-bit digits (most significant first), many workgroups per sample.  Every workgroup owns a contiguous slice of
// TK_SLICE elements (16 consecutive elements per thread: 16 independent loads in flight), histograms its slice in LDS
// and merges the non-empty bins into the sample's global histogram.  The selection state (prefix, how many still to
// take) is recomputed by each workgroup from the finished histograms of the earlier digits -- 2048 bins, a block scan.
// Launches: digit 0, digit 1, digit 2, threshold ties per slice, mask.  (The previous version ran one 1024-thread
// workgroup per sample and spent 240 us in 460 dependent load round trips.)
constexpr int TK_BITS = 11, TK_BINS = 1 << TK_BITS, TK_THREADS = 256, TK_EPT = 16, TK_SLICE = TK_THREADS * TK_EPT;
__host__ __device__ inline int tk_shift(int digit) { return digit == 0 ? 21 : (digit == 1 ? 10 : 0); }       // 11 + 11 + 10 bits
__host__ __device__ inline uint32_t tk_mask(int digit) { return digit == 2 ? 0x3ffu : 0x7ffu; }

struct TkWork {            // per sample, all zeroed before the first launch
    uint32_t* hist;        // [B][3][TK_BINS]
    uint32_t* ties;        // [B][slices]
    uint32_t* sel;         // [B][2]: threshold key, number of threshold-valued entries to keep
};

// exclusive prefix sum of one value per thread over the 256-thread block (wave scans by shuffles + 4 wave totals)
__device__ uint32_t tk_block_scan(uint32_t v, uint32_t* sm /*[4]*/, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sm[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w2 = 0; w2 < wave; ++w2) off += sm[w2];
    total = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    return off + inc - v;
}

// scan one finished histogram from the top bin down: the bin where the running count reaches `rem`, and what is left to
// take inside that bin.  All threads return the same values.
__device__ void tk_select(const uint32_t* __restrict__ hist, uint32_t rem, uint32_t* scratch /*[TK_THREADS + 2]*/, uint32_t& bin_out,
                          uint32_t& rem_out) {
    const int tid = threadIdx.x;
    constexpr int PER = TK_BINS / TK_THREADS;                 // 8 consecutive bins per thread, thread 0 holds the top ones
    uint32_t c[PER], tot = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { c[i] = hist[TK_BINS - 1 - (tid * PER + i)]; tot += c[i]; }
    uint32_t all;
    uint32_t run = tk_block_scan(tot, scratch, all);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (run < rem && run + c[i] >= rem) { scratch[TK_THREADS] = (uint32_t)(TK_BINS - 1 - (tid * PER + i)); scratch[TK_THREADS + 1] = rem - run; }
        run += c[i];
    }
    __syncthreads();
    bin_out = scratch[TK_THREADS];
    rem_out = scratch[TK_THREADS + 1];
    __syncthreads();
}

// selection state before digit `upto`: prefix (the digits already fixed, in place) and the remaining count
__device__ void tk_state(const TkWork& w, int b, int upto, uint32_t k_total, uint32_t* scratch, uint32_t& prefix, uint32_t& rem) {
    prefix = 0u; rem = k_total;
    for (int d = 0; d < upto; ++d) {
        uint32_t bin, r;
        tk_select(w.hist + ((long)b * 3 + d) * TK_BINS, rem, scratch, bin, r);
        prefix |= bin << tk_shift(d);
        rem = r;
    }
}

__global__ __launch_bounds__(TK_THREADS) void tk_hist_kernel(const float* __restrict__ sim, long E, uint32_t k_total, int digit, TkWork w) {
    __shared__ uint32_t hist[TK_BINS];
    __shared__ uint32_t scratch[TK_THREADS + 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < TK_BINS; i += TK_THREADS) hist[i] = 0u;
    uint32_t prefix, rem;
    tk_state(w, b, digit, k_total, scratch, prefix, rem);          // contains barriers: also orders the zeroing above
    const uint32_t himask = digit == 0 ? 0u : (0xFFFFFFFFu << tk_shift(digit - 1));
    const float* v = sim + (long)b * E;
    const long e0 = (long)blockIdx.x * TK_SLICE + (long)tid * TK_EPT;
    float x[TK_EPT];
#pragma unroll
    for (int j = 0; j < TK_EPT; ++j) x[j] = e0 + j < E ? v[e0 + j] : 0.f;
#pragma unroll
    for (int j = 0; j < TK_EPT; ++j) {
        const uint32_t key = f32_order_key(x[j]);
        if (e0 + j < E && (key & himask) == prefix) atomicAdd(&hist[(key >> tk_shift(digit)) & tk_mask(digit)], 1u);
    }
    __syncthreads();
    uint32_t* gh = w.hist + ((long)b * 3 + digit) * TK_BINS;
    for (int i = tid; i < TK_BINS; i += TK_THREADS)
        if (hist[i]) atomicAdd(&gh[i], hist[i]);
}

// number of entries equal to the threshold key in each slice; slice 0 also publishes (threshold, need_eq)
__global__ __launch_bounds__(TK_THREADS) void tk_ties_kernel(const float* __restrict__ sim, long E, uint32_t k_total, TkWork w) {
    __shared__ uint32_t scratch[TK_THREADS + 2];
    __shared__ uint32_t cnt;
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid == 0) cnt = 0u;
    uint32_t thr, need_eq;
    tk_state(w, b, 3, k_total, scratch, thr, need_eq);
    const float* v = sim + (long)b * E;
    const long e0 = (long)blockIdx.x * TK_SLICE + (long)tid * TK_EPT;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < TK_EPT; ++j)
        if (e0 + j < E && f32_order_key(v[e0 + j]) == thr) ++c;
    if (c) atomicAdd(&cnt, c);
    __syncthreads();
    if (tid == 0) {
        w.ties[(long)b * gridDim.x + blockIdx.x] = cnt;
        if (blockIdx.x == 0) { w.sel[b * 2] = thr; w.sel[b * 2 + 1] = need_eq; }
    }
}

// adj = 1 for keys above the threshold and for the first need_eq threshold-valued entries in flat-index order
// (discrete_graph_learning.py:108 keeps scattered values != 0; :165-166 clears the diagonal)
__global__ __launch_bounds__(TK_THREADS) void tk_mask_kernel(const float* __restrict__ sim, long E, int N, TkWork w, float* __restrict__ adj) {
    __shared__ uint32_t scan[TK_THREADS];
    const int b = blockIdx.y, tid = threadIdx.x;
    const uint32_t thr = w.sel[b * 2], need_eq = w.sel[b * 2 + 1];
    uint32_t part = 0;                                         // threshold-valued entries in earlier slices
    for (int g = tid; g < (int)blockIdx.x; g += TK_THREADS) part += w.ties[(long)b * gridDim.x + g];
    uint32_t before;
    tk_block_scan(part, scan, before);
    const float* v = sim + (long)b * E;
    float* out = adj + (long)b * E;
    const long e0 = (long)blockIdx.x * TK_SLICE + (long)tid * TK_EPT;
    float x[TK_EPT];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < TK_EPT; ++j) {
        x[j] = e0 + j < E ? v[e0 + j] : 0.f;
        if (e0 + j < E && f32_order_key(x[j]) == thr) ++mine;
    }
    uint32_t slice_ties;
    uint32_t rank = before + tk_block_scan(mine, scan, slice_ties);
#pragma unroll
    for (int j = 0; j < TK_EPT; ++j) {
        const long e = e0 + j;
        if (e >= E) break;
        const uint32_t key = f32_order_key(x[j]);
        bool sel = key > thr;
        if (key == thr) { sel = rank < need_eq; ++rank; }
        const int i = (int)(e / N), jj = (int)(e % N);
        out[e] = (sel && x[j] != 0.f && i != jj) ? 1.f : 0.f;
    }
}

}  // namespace

static long tk_slices(int N) { return ((long)N * N + TK_SLICE - 1) / TK_SLICE; }

static long tk_bytes(int B, int N) { return (long)B * (3 * TK_BINS + tk_slices(N) + 2) * (long)sizeof(uint32_t); }
// ---------------------------------------------------------------------------------------------------------------------------
// Symmetric Gram product for graphs of up to 320 nodes (METR-LA, PEMS04, PEMS08 of the reference's datasets): raw[b] = H[b] H[b]^T, H bf16 [N][F].
// The staged GEMM computes the full square from 128 x 128 tiles, each of which streams its two 128-row operand panels over the whole
// feature axis: 6 x the 158 MB of H through L2 at PEMS04, 181 us alone and 250 us next to the encoder -- for 48.6 GFLOP (19 us of matrix
// time).  Here ONE workgroup of 16 waves owns the WHOLE (padded) output for a slice of the feature axis: a 64-feature slab of all N rows
// is staged once in LDS (double buffered) and every wave multiplies the row panels of its 64 x 64 block (2 x 2 MFMA tiles: four fragment
// reads for four products) -- only the <= 15 blocks on or above the diagonal exist, one per wave (64 accumulator registers).  H is read exactly once; the partial blocks of the feature
// slices go to a workspace in accumulator order (coalesced), and gram_finish_kernel adds them, applies the cosine normalisation
// (similarity.py:8-14) and writes both triangles.
constexpr int GS_KS = 64, GS_PITCH = GS_KS * 2 + 8;          // slab of 64 features; LDS row = 128 B + 8 (conflict-free 16-byte fragment reads)
__host__ __device__ constexpr int gs_blocks(int nb) { return nb * (nb + 1) / 2; }
// block p (0 .. NB(NB+1)/2 - 1) -> (bi <= bj), row-major over the upper triangle
__device__ __forceinline__ void gs_block_of(int p, int NB, int& bi, int& bj) {
    bi = 0;
    while (p >= NB - bi) { p -= NB - bi; ++bi; }
    bj = bi + p;
}
__global__ __launch_bounds__(1024) void gram_sym_kernel(const uint16_t* __restrict__ H, int N, int F, int NB, int slabs_per_split,
                                                        float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
    const int rows = NB * 64;                                   // padded row count (zero rows beyond N)
    const int buf_bytes = rows * GS_PITCH;
    const uint16_t* Hb = H + (long)b * N * F;
    const int k_begin = s * slabs_per_split * GS_KS;
    const int k_end = min(F, (s + 1) * slabs_per_split * GS_KS);
    // zero both buffers once: rows >= N (and the pad bytes) are never written again
    for (int i = tid; i < 2 * buf_bytes / 16; i += 1024) ((uint4*)gs_lds)[i] = make_uint4(0u, 0u, 0u, 0u);
    const int nblk = gs_blocks(NB);
    constexpr int Q = 1;                                        // blocks per wave (NB <= 5: 15 blocks for 16 waves)
    int bi[Q], bj[Q];
    bool own[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int p = wave + 16 * q;
        own[q] = p < nblk;
        gs_block_of(own[q] ? p : 0, NB, bi[q], bj[q]);
    }
    f32x16 acc[Q][4];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][t][e] = 0.f;
    // staging: chunk c = (row, 16-byte piece) of the slab; up to 3 chunks per thread (384 rows x 8 pieces / 1024 threads)
    const int nchunk = N * (GS_KS / 8);
    uint4 st[3];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = tid + u * 1024;
            st[u] = make_uint4(0u, 0u, 0u, 0u);
            if (c < nchunk) {
                const int row = c >> 3, k = k0 + (c & 7) * 8;
                if (k < k_end) st[u] = *(const uint4*)(Hb + (long)row * F + k);          // (F % 8 == 0 and slices end on slab boundaries or at F: whole pieces)
            }
        }
    };
    auto commit = [&](char* buf) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = tid + u * 1024;
            if (c < nchunk) {
                char* d = buf + (c >> 3) * GS_PITCH + (c & 7) * 16;
                *(uint2*)d = make_uint2(st[u].x, st[u].y);
                *(uint2*)(d + 8) = make_uint2(st[u].z, st[u].w);
            }
        }
    };
    __syncthreads();
    if (k_begin < k_end) {
        fetch(k_begin);
        commit(gs_lds);
        __syncthreads();
        int cur = 0;
        const int r = lane & 31, h = lane >> 5;
        for (int k0 = k_begin; k0 < k_end; k0 += GS_KS) {
            const bool more = k0 + GS_KS < k_end;
            if (more) fetch(k0 + GS_KS);
            const char* buf = gs_lds + cur * buf_bytes;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (!own[q]) continue;
                const char* pa = buf + (bi[q] * 64 + r) * GS_PITCH + h * 16;
                const char* pb = buf + (bj[q] * 64 + r) * GS_PITCH + h * 16;
                const bool diag = bi[q] == bj[q];
#pragma unroll
                for (int ks = 0; ks < GS_KS / 16; ++ks) {
                    auto frag = [&](const char* p) -> bf16x8 {
                        const uint2 lo = *(const uint2*)(p + ks * 32), hi = *(const uint2*)(p + ks * 32 + 8);
                        return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    };
                    const bf16x8 a0 = frag(pa), a1 = frag(pa + 32 * GS_PITCH);
                    const bf16x8 b0 = diag ? a0 : frag(pb), b1 = diag ? a1 : frag(pb + 32 * GS_PITCH);
                    acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[q][0], 0, 0, 0);
                    acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[q][1], 0, 0, 0);
                    if (!diag) acc[q][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[q][2], 0, 0, 0);
                    acc[q][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[q][3], 0, 0, 0);
                }
            }
            if (more) commit(gs_lds + (cur ^ 1) * buf_bytes);
            __syncthreads();
            cur ^= 1;
        }
    }
    // partial blocks -> workspace [split][sample][block][tile 0..3][accumulator register e][lane]: 256-byte runs per store instruction
    float* out = ws + ((long)(s * gridDim.y + b) * nblk) * 4096;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (!own[q]) continue;
        float* o = out + (long)(wave + 16 * q) * 4096 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t == 2 && bi[q] == bj[q]) continue;               // the lower tile of a diagonal block is the transpose of tile 1
#pragma unroll
            for (int e = 0; e < 16; ++e) o[t * 1024 + e * 64] = acc[q][t][e];
        }
    }
    (void)S;
}
// sum of the feature slices, cosine normalisation, both triangles.  grid (blocks of the upper triangle, B), 256 threads; every thread owns
// accumulator elements (t, e, lane) with lane = tid & 63, e = 4 (tid >> 6) .. + 3 of all four tiles.
__global__ __launch_bounds__(256) void gram_finish_kernel(const float* __restrict__ ws, int S, int N, int NB, const float* __restrict__ sqn_part,
                                                          float* __restrict__ sim) {
    __shared__ float tile[64][65];
    __shared__ float nrm[128];
    const int b = blockIdx.y, Bn = gridDim.y, nblk = gs_blocks(NB);
    int bi, bj;
    gs_block_of(blockIdx.x, NB, bi, bj);
    const int tid = threadIdx.x, lane = tid & 63, eg = tid >> 6;
    if (tid < 128) {           // norms of the block's 64 rows and 64 columns (similarity.py:8-14: |F| + 1e-7)
        const int n = (tid < 64 ? bi * 64 : bj * 64 - 64) + tid;
        float sum = 0.f;
        if (n < N) {
            const float* p = sqn_part + ((long)b * N + n) * 16;
#pragma unroll
            for (int w = 0; w < 16; ++w) sum += p[w];
        }
        nrm[tid] = sqrtf(fmaxf(sum, 0.f)) + 1e-7f;
    }
    const bool diag = bi == bj;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t == 2 && diag) continue;
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = eg * 4 + ee;
            float v = 0.f;
            for (int sp = 0; sp < S; ++sp) v += ws[((long)(sp * Bn + b) * nblk + blockIdx.x) * 4096 + t * 1024 + e * 64 + lane];
            // accumulator element (e, lane) of a 32 x 32 product: row 8 (e >> 2) + (e & 3) + 4 (lane >> 5), column lane & 31
            const int rr = (t >> 1) * 32 + 8 * (e >> 2) + (e & 3) + 4 * (lane >> 5), cc = (t & 1) * 32 + (lane & 31);
            tile[rr][cc] = v;
            if (diag && t == 1) tile[cc][rr] = v;                   // (the skipped lower tile)
        }
    }
    __syncthreads();
    // rows of the block (and, off the diagonal, of its transpose) in 256-byte runs
    for (int i = tid; i < 64 * 64; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        const int gi = bi * 64 + rr, gj = bj * 64 + cc;
        if (gi < N && gj < N) sim[((long)b * N + gi) * N + gj] = tile[rr][cc] / (nrm[rr] * nrm[64 + cc]);
    }
    if (!diag) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int cc = i >> 6, rr = i & 63;                      // element (gj, gi) of the output = tile[rr][cc]
            const int gi = bi * 64 + rr, gj = bj * 64 + cc;
            if (gi < N && gj < N) sim[((long)b * N + gj) * N + gi] = tile[rr][cc] / (nrm[rr] * nrm[64 + cc]);
        }
    }
}
// feature slices per sample: enough workgroups for about half the chip (the product runs next to other kernels), at least 2 slabs each
static int gram_sym_splits(int B, int F) {
    static const int forced = []() { const char* e = getenv("STEP_GRAM_SPLITS"); return e ? atoi(e) : 0; }();
    const int slabs = (F + GS_KS - 1) / GS_KS;
    int S = forced > 0 ? forced : (128 + B - 1) / B;
    if (S > slabs / 2) S = slabs / 2;
    return S < 1 ? 1 : S;
}
static bool gram_sym_ok(int N) {
    static const bool off = []() { const char* e = getenv("STEP_GRAM_SYM"); return e && e[0] == '0'; }();
    return !off && N <= 320;
}
static long gram_sym_ws_floats(int B, int N, int F) {
    const int NB = (N + 63) / 64;
    return (long)gram_sym_splits(B, F) * B * gs_blocks(NB) * 4096;
}

// F > 0 (step_knn_graph): the selection state, then -- 256-byte aligned -- the partial tiles of the split-K Gram product
// (StepGemm.splitk_ws: 33 MB at PEMS04, where 11 splits x 8 x 307^2 f32 atomics were most of the launch)
static long gram_ws_floats(int B, int N, int F) {
    const int splits = step_gemm_auto_splitk(N, N, F, B);
    const long staged = splits > 1 ? (long)splits * B * N * N : 0;
    const long sym = gram_sym_ok(N) ? gram_sym_ws_floats(B, N, F) : 0;
    return staged > sym ? staged : sym;
}
extern "C" long step_knn_workspace_bytes(int B, int N, int F) {
    const long base = tk_bytes(B, N);
    return F > 0 ? ((base + 255) & ~255L) + gram_ws_floats(B, N, F) * (long)sizeof(float) : base;
}

extern "C" int step_topk_mask(const float* sim, int B, int N, int k_total, float* adj, void* work, long work_bytes,
                              void* stream) {
    STEP_REQUIRE(sim && adj && work && B > 0 && N > 0 && k_total > 0, "topk_mask: bad arguments");
    STEP_REQUIRE(work_bytes >= step_knn_workspace_bytes(B, N, 0), "topk_mask: workspace of %ld bytes, need %ld", work_bytes,
                 step_knn_workspace_bytes(B, N, 0));
    hipStream_t st = (hipStream_t)stream;
    const long E = (long)N * N;
    const int slices = (int)tk_slices(N);
    TkWork w;
    w.hist = (uint32_t*)work;
    w.ties = w.hist + (long)B * 3 * TK_BINS;
    w.sel = w.ties + (long)B * slices;
    if (hipMemsetAsync(work, 0, (size_t)tk_bytes(B, N), st) != hipSuccess) {
        step_set_error("topk_mask: memset failed");
        return STEP_ERR_HIP;
    }
    const uint32_t k = (uint32_t)((long)k_total < E ? k_total : E);
    dim3 grid(slices, B);
    for (int d = 0; d < 3; ++d) {
        tk_hist_kernel<<<grid, TK_THREADS, 0, st>>>(sim, E, k, d, w);
        STEP_LAUNCH_CHECK("topk digit histogram");
    }
    tk_ties_kernel<<<grid, TK_THREADS, 0, st>>>(sim, E, k, w);
    STEP_LAUNCH_CHECK("topk ties");
    tk_mask_kernel<<<grid, TK_THREADS, 0, st>>>(sim, E, N, w, adj);
    STEP_LAUNCH_CHECK("topk mask");
    return STEP_OK;
}

extern "C" int step_knn_graph(const uint16_t* hidden, const float* sqnorm_part, int B, int N, int F, int k_total,
                              float* sim, float* adj, void* work, long work_bytes, void* stream) {
    STEP_REQUIRE(hidden && sim && adj && B > 0 && N > 0 && F > 0 && k_total > 0, "knn_graph: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    STEP_REQUIRE(F % 8 == 0, "knn_graph: feature length %d must be a multiple of 8", F);
    if (gram_sym_ok(N) && sqnorm_part && work && work_bytes >= step_knn_workspace_bytes(B, N, F) && (((uintptr_t)hidden) & 15) == 0) {
        // small graphs: the whole output per workgroup, H read once (gram_sym_kernel), normalisation in the reduction of the slices
        const int NB = (N + 63) / 64, S = gram_sym_splits(B, F), slabs = (F + GS_KS - 1) / GS_KS;
        float* ws = (float*)((char*)work + ((tk_bytes(B, N) + 255) & ~255L));
        const int lds = 2 * NB * 64 * GS_PITCH;
        STEP_TRY(step_raise_lds_once((const void*)gram_sym_kernel, 160 * 1024, "knn_graph"));
        gram_sym_kernel<<<dim3(S, B), 1024, lds, st>>>(hidden, N, F, NB, (slabs + S - 1) / S, ws);
        STEP_LAUNCH_CHECK("gram_sym");
        gram_finish_kernel<<<dim3(gs_blocks(NB), B), 256, 0, st>>>(ws, S, N, NB, sqnorm_part, sim);
        STEP_LAUNCH_CHECK("gram_finish");
        return step_topk_mask(sim, B, N, k_total, adj, work, work_bytes, stream);
    }
    if (hipMemsetAsync(sim, 0, (size_t)B * N * N * sizeof(float), st) != hipSuccess) {
        step_set_error("knn_graph: memset failed");
        return STEP_ERR_HIP;
    }
    {
        // raw[b] = H[b] H[b]^T on the bf16 matrix cores: the staged GEMM (k-contiguous bf16 rows on both sides, 128x128
        // tiles, split-K with f32 atomics into the zeroed output).  The full square is computed: the symmetric half would
        // save MFMA work the kernel does not wait for (it streams H, 158 MB at PEMS04, and is latency / L2 bound).
        StepGemm g;
        memset(&g, 0, sizeof(g));
        g.M = N; g.N = N; g.K = F; g.batch = B;
        g.A = hidden; g.sam = F; g.sak = 1; g.sab = (long)N * F; g.a_bf16 = 1;
        g.B = hidden; g.sbk = 1; g.sbn = F; g.sbb = (long)N * F; g.b_bf16 = 1;
        g.C = sim; g.ldc = N; g.scn = 1; g.scb = (long)N * N;
        g.alpha = 1.f; g.accumulate = 2; g.splitk = -1; g.compute_bf16 = 1;
        if (work && work_bytes >= step_knn_workspace_bytes(B, N, F) && gram_ws_floats(B, N, F) > 0) {
            g.splitk_ws = (float*)((char*)work + ((tk_bytes(B, N) + 255) & ~255L));
            g.splitk_ws_floats = gram_ws_floats(B, N, F);
        }
        STEP_TRY(step_gemm_launch(g, st));
    }
    dim3 grid(cdiv(N, 256) > 4 ? 4 : cdiv(N, 256), N, B);
    cosine_finalize_kernel<<<grid, 256, 0, st>>>(sim, sqnorm_part, N);
    STEP_LAUNCH_CHECK("cosine_finalize");
    if (!sqnorm_part) {
        cosine_diag_kernel<<<dim3(cdiv(N, 256), B), 256, 0, st>>>(sim, N);
        STEP_LAUNCH_CHECK("cosine_diag");
    }
    return step_topk_mask(sim, B, N, k_total, adj, work, work_bytes, stream);
}
