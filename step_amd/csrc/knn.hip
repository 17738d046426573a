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
// F > 0 (step_knn_graph): the selection state, then -- 256-byte aligned -- the partial tiles of the split-K Gram product
// (StepGemm.splitk_ws: 33 MB at PEMS04, where 11 splits x 8 x 307^2 f32 atomics were most of the launch)
static long gram_ws_floats(int B, int N, int F) {
    const int splits = step_gemm_auto_splitk(N, N, F, B);
    return splits > 1 ? (long)splits * B * N * N : 0;
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
    if (hipMemsetAsync(sim, 0, (size_t)B * N * N * sizeof(float), st) != hipSuccess) {
        step_set_error("knn_graph: memset failed");
        return STEP_ERR_HIP;
    }
    STEP_REQUIRE(F % 8 == 0, "knn_graph: feature length %d must be a multiple of 8", F);
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
