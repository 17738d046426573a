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
constexpr int GS_KS = 64, GS_ROWB = GS_KS * 2;               // slab of 64 features = 128 bytes per row, 8 pieces of 16 bytes
__host__ __device__ constexpr int gs_blocks(int nb) { return nb * (nb + 1) / 2; }
// block p (0 .. NB(NB+1)/2 - 1) -> (bi <= bj), row-major over the upper triangle
__device__ __forceinline__ void gs_block_of(int p, int NB, int& bi, int& bj) {
    bi = 0;
    while (p >= NB - bi) { p -= NB - bi; ++bi; }
    bj = bi + p;
}
// lane i's 16 bytes land at lds_dst + 16 i (lds_dst wave-uniform): global_load_lds_dwordx4, no registers in between
__device__ __forceinline__ void gs_dma_1k(const void* gsrc_lane, uint32_t lds_dst) {
    uint32_t keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_dst)
                 : "memory");
}
// Staging: a slab of all 64 NB (padded) rows goes from global memory straight into one of three LDS buffers by DMA -- NB instructions of 1 KB
// (8 rows) from each of waves 0..7, two slabs ahead of the matrix work.  A row's 8 pieces are stored XOR-swizzled (piece c of row r in slot
// c ^ (r & 7): the DMA fixes where a lane's data lands, not what it loads), so the 16-byte fragment reads of 8 consecutive rows hit all 32
// banks.  Rows beyond N repeat row N - 1: they only feed outputs that are never stored.  LDS moves 1 KB per matrix product here (four
// fragments for the four products of a 64 x 64 block): 2048 cycles per slab on the LDS port against 2048 on the matrix pipe -- with two-way
// bank conflicts (the first version: padded rows, 8-byte reads) the kernel ran at the LDS port's pace, 67 us instead of ~30.
template <int NB>
__global__ __launch_bounds__(1024) void gram_sym_kernel(const uint16_t* __restrict__ H, int N, int F, int slabs_per_split, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char gs_lds[];
    constexpr int ROWS = NB * 64, BUF = ROWS * GS_ROWB, NBLK = gs_blocks(NB);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = blockIdx.x, b = blockIdx.y;
    const uint16_t* Hb = H + (long)b * N * F;
    const int nslab_all = F / GS_KS;
    const int slab0 = s * slabs_per_split;
    const int nslab = min(nslab_all - slab0, slabs_per_split);
    int bi, bj;
    const bool own = wave < NBLK;
    gs_block_of(own ? wave : 0, NB, bi, bj);
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)gs_lds;
    auto issue = [&](int slab, int buf) {              // waves 0..7: NB groups of 8 rows each
        if (wave < 8) {
            const long k0 = (long)(slab0 + slab) * GS_KS;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int g = wave * NB + j;
                int row = g * 8 + (lane >> 3);
                const int c = (lane & 7) ^ (row & 7);
                row = row < N ? row : N - 1;
                gs_dma_1k(Hb + (long)row * F + k0 + c * 8, lds0 + (uint32_t)(buf * BUF + g * 1024));
            }
        }
    };
    if (nslab > 0) {
        const int r = lane & 31, h = lane >> 5;
        issue(0, 0);
        if (nslab > 1) issue(1, 1);
        for (int i = 0; i < nslab; ++i) {
            // slab i has landed: this wave's DMAs of slab i + 1 (NB of them, if requested) may still be in flight
            if (i + 1 < nslab) {
                if constexpr (NB == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if constexpr (NB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if constexpr (NB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                                   // everybody's pieces of slab i are there; everybody is done with slab i - 1
            if (i + 2 < nslab) issue(i + 2, (i + 2) % 3);      // ... whose buffer is refilled
            if (own) {
                const char* buf = gs_lds + (i % 3) * BUF;
                const int ra = bi * 64 + r, rb = bj * 64 + r;
                const char* pa0 = buf + ra * GS_ROWB;
                const char* pa1 = pa0 + 32 * GS_ROWB;
                const char* pb0 = buf + rb * GS_ROWB;
                const char* pb1 = pb0 + 32 * GS_ROWB;
                const bool diag = bi == bj;
#pragma unroll
                for (int ks = 0; ks < GS_KS / 16; ++ks) {
                    const int sl = (((ks * 2 + h) ^ (r & 7)) << 4);        // rows ra, ra + 32, rb, rb + 32 share r & 7
                    const bf16x8 a0 = *(const bf16x8*)(pa0 + sl), a1 = *(const bf16x8*)(pa1 + sl);
                    const bf16x8 b0 = diag ? a0 : *(const bf16x8*)(pb0 + sl), b1 = diag ? a1 : *(const bf16x8*)(pb1 + sl);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
                    if (!diag) acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
                }
            }
        }
    }
    // partial blocks -> workspace [split][sample][block][tile 0..3][accumulator register e][lane]: 256-byte runs per store instruction
    if (own) {
        float* o = ws + ((long)(s * gridDim.y + b) * NBLK + wave) * 4096 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t == 2 && bi == bj) continue;                  // the lower tile of a diagonal block is the transpose of tile 1
#pragma unroll
            for (int e = 0; e < 16; ++e) o[t * 1024 + e * 64] = acc[t][e];
        }
    }
}
// sum of the feature slices, cosine normalisation, both triangles.  grid (4 x blocks of the upper triangle, B): one 32 x 32 MFMA tile per
// workgroup of 256 threads (thread: accumulator elements e = 4 (tid >> 6) .. + 3 of lane tid & 63).
__global__ __launch_bounds__(256) void gram_finish_kernel(const float* __restrict__ ws, int S, int N, int NB, const float* __restrict__ sqn_part,
                                                          float* __restrict__ sim) {
    __shared__ float tile[32][33];
    __shared__ float nrm[64];
    const int b = blockIdx.y, Bn = gridDim.y, nblk = gs_blocks(NB), blk = blockIdx.x >> 2, t = blockIdx.x & 3;
    int bi, bj;
    gs_block_of(blk, NB, bi, bj);
    const bool diag = bi == bj;
    if (diag && t == 2) return;                                    // (never stored: the transpose of tile 1, written by that tile's workgroup)
    const int row0 = bi * 64 + (t >> 1) * 32, col0 = bj * 64 + (t & 1) * 32;
    const int tid = threadIdx.x, lane = tid & 63, eg = tid >> 6;
    if (tid < 64) {           // norms of the tile's 32 rows and 32 columns (similarity.py:8-14: |F| + 1e-7)
        const int n = tid < 32 ? row0 + tid : col0 + tid - 32;
        float sum = 0.f;
        if (n < N) {
            const float* p = sqn_part + ((long)b * N + n) * 16;
#pragma unroll
            for (int w = 0; w < 16; ++w) sum += p[w];
        }
        nrm[tid] = sqrtf(fmaxf(sum, 0.f)) + 1e-7f;
    }
    const long sstr = (long)Bn * nblk * 4096;
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) {
        const int e = eg * 4 + ee;
        // (four independent chains: the loads of four slices are in flight together)
        const float* src = ws + ((long)b * nblk + blk) * 4096 + t * 1024 + e * 64 + lane;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int sp = 0;
        for (; sp + 4 <= S; sp += 4) {
            v0 += src[(long)sp * sstr]; v1 += src[(long)(sp + 1) * sstr]; v2 += src[(long)(sp + 2) * sstr]; v3 += src[(long)(sp + 3) * sstr];
        }
        for (; sp < S; ++sp) v0 += src[(long)sp * sstr];
        // accumulator element (e, lane) of a 32 x 32 product: row 8 (e >> 2) + (e & 3) + 4 (lane >> 5), column lane & 31
        tile[8 * (e >> 2) + (e & 3) + 4 * (lane >> 5)][lane & 31] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    for (int i = tid; i < 32 * 32; i += 256) {                        // rows of the tile in 128-byte runs
        const int rr = i >> 5, cc = i & 31, gi = row0 + rr, gj = col0 + cc;
        if (gi < N && gj < N) sim[((long)b * N + gi) * N + gj] = tile[rr][cc] / (nrm[rr] * nrm[32 + cc]);
    }
    if (!(diag && (t == 0 || t == 3))) {                              // ... and of its transpose (tiles on the diagonal hold both triangles already)
        for (int i = tid; i < 32 * 32; i += 256) {
            const int cc = i >> 5, rr = i & 31, gi = row0 + rr, gj = col0 + cc;          // element (gj, gi) of the output = tile[rr][cc]
            if (gi < N && gj < N) sim[((long)b * N + gj) * N + gi] = tile[rr][cc] / (nrm[rr] * nrm[32 + cc]);
        }
    }
}
// feature slices per sample: ~128 workgroups, at least 2 slabs each (inside the step, next to the encoder's persistent launch, 4 / 8 / 16
// slices per sample give the same step time at PEMS04: 3.411 / 3.404 / 3.399 ms, profiles/r06_h_gram_sweep.log; alone, more slices are faster)
static int gram_sym_splits(int B, int F) {
    static const int forced = []() { const char* e = getenv("STEP_GRAM_SPLITS"); return e ? atoi(e) : 0; }();
    const int slabs = (F + GS_KS - 1) / GS_KS;
    int S = forced > 0 ? forced : (128 + B - 1) / B;
    if (forced <= 0 && S > 16) S = 16;          // (few samples: more slices only lengthen the reduction -- 64 slices at METR-LA's B = 2 made it 89 us)
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
    if (gram_sym_ok(N) && F % GS_KS == 0 && sqnorm_part && work && work_bytes >= step_knn_workspace_bytes(B, N, F) && (((uintptr_t)hidden) & 15) == 0) {
        // small graphs: the whole output per workgroup, H read once (gram_sym_kernel), normalisation in the reduction of the slices
        const int NB = (N + 63) / 64, S = gram_sym_splits(B, F), slabs = F / GS_KS, per = (slabs + S - 1) / S;
        float* ws = (float*)((char*)work + ((tk_bytes(B, N) + 255) & ~255L));
        const int lds = 3 * NB * 64 * GS_ROWB;
        const dim3 grid((slabs + per - 1) / per, B);
#define GS_LAUNCH(nb)                                                                                            \
        case nb:                                                                                                 \
            STEP_TRY(step_raise_lds_once((const void*)gram_sym_kernel<nb>, 160 * 1024, "knn_graph"));            \
            gram_sym_kernel<nb><<<grid, 1024, lds, st>>>(hidden, N, F, per, ws);                                 \
            break;
        switch (NB) {
            GS_LAUNCH(1) GS_LAUNCH(2) GS_LAUNCH(3) GS_LAUNCH(4) GS_LAUNCH(5)
            default: STEP_REQUIRE(false, "knn_graph: %d nodes in the symmetric Gram path", N);
        }
#undef GS_LAUNCH
        STEP_LAUNCH_CHECK("gram_sym");
        const int S_used = (int)grid.x;
        gram_finish_kernel<<<dim3(4 * gs_blocks(NB), B), 256, 0, st>>>(ws, S_used, N, NB, sqnorm_part, sim);
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
