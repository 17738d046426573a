// kNN prior graph on device: cosine similarities of the TSFormer hidden states and an exact
// global top-k over the flattened N*N matrix per sample (radix select, 4 x 8-bit passes).
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
// Symmetric Gram matrix of the bf16 hidden states on the bf16 matrix cores:
//   raw[b][i][j] = sum_k H[b][i][k] H[b][j][k],  K = P*96 (32 256 at PEMS04) -- 6.1 GFLOP per window.
// 128x128 output tile per workgroup (4 waves x 64x64 = 2x2 MFMA 32x32x16 tiles), only tiles with
// tm <= tn are computed and mirrored, split-K over grid.z with f32 atomics into the zeroed output.
// Operands are staged through LDS in 64-deep k-chunks with 16-byte global loads (rows are k-contiguous)
// and read back as MFMA fragments (lane = row, 8 consecutive k) with ds_read_b128; the 144-byte row
// pitch keeps the eight 16-byte reads of a lane group on different banks.
constexpr int GBK = 64;                  // k-chunk
constexpr int GPITCH = GBK * 2 + 16;     // bytes per staged row

__global__ __launch_bounds__(256) void gram_bf16_kernel(const uint16_t* __restrict__ H, int N, int K, int splitk,
                                                        float* __restrict__ raw) {
    __shared__ __attribute__((aligned(16))) char As[128 * GPITCH];
    __shared__ __attribute__((aligned(16))) char Bs[128 * GPITCH];
    // upper-triangular tile pair from blockIdx.x
    const int nt = (N + 127) / 128;
    int tm = 0, rem = blockIdx.x;
    while (rem >= nt - tm) { rem -= nt - tm; ++tm; }
    const int tn = tm + rem;
    const int b = blockIdx.y;
    const int zs = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int chunks = (K + GBK - 1) / GBK;
    const int per = (chunks + splitk - 1) / splitk;
    const int c0 = zs * per, c1 = min(chunks, c0 + per);
    const uint16_t* Hb = H + (long)b * N * K;
    const bool diag = tm == tn;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // staging map: 128 rows x 8 pieces of 16 B = 1024 pieces, 4 per thread
    u32x4 ra[4], rb[4];
    auto load = [&](int ck) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = tid + q * 256, row = pc >> 3, kp = (pc & 7) * 8;
            const int k = ck * GBK + kp;
            const int gi = tm * 128 + row, gj = tn * 128 + row;
            u32x4 z = {0u, 0u, 0u, 0u};
            // K is a multiple of 8 here (96 features per token), so a piece is either fully inside or outside
            ra[q] = (gi < N && k < K) ? *(const u32x4*)(Hb + (long)gi * K + k) : z;
            if (!diag) rb[q] = (gj < N && k < K) ? *(const u32x4*)(Hb + (long)gj * K + k) : z;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pc = tid + q * 256, row = pc >> 3, kp = (pc & 7) * 16;
            *(u32x4*)(As + row * GPITCH + kp) = ra[q];
            if (!diag) *(u32x4*)(Bs + row * GPITCH + kp) = rb[q];
        }
    };
    const char* Bsrc = diag ? As : Bs;
    const int r = lane & 31, h = lane >> 5;
    if (c0 < c1) {
        load(c0);
        for (int ck = c0; ck < c1; ++ck) {
            __syncthreads();
            store();
            __syncthreads();
            if (ck + 1 < c1) load(ck + 1);
#pragma unroll
            for (int ks = 0; ks < GBK / 16; ++ks) {
                bf16x8 a[2], bb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8*)(As + (wr * 64 + i * 32 + r) * GPITCH + ks * 32 + h * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) bb[j] = *(const bf16x8*)(Bsrc + (wc * 64 + j * 32 + r) * GPITCH + ks * 32 + h * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* out = raw + (long)b * N * N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gj = tn * 128 + wc * 64 + j * 32 + r;
            if (gj >= N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int gi = tm * 128 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (gi >= N) continue;
                const float v = acc[i][j][e];
                atomicAdd(out + (long)gi * N + gj, v);
                if (!diag) atomicAdd(out + (long)gj * N + gi, v);
            }
        }
}

// One workgroup (1024 threads) per sample.
__global__ __launch_bounds__(1024) void topk_mask_kernel(const float* __restrict__ sim, int N, int k_total,
                                                         float* __restrict__ adj) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_remaining, s_run;
    __shared__ uint32_t wave_cnt[16];
    const int b = blockIdx.x;
    const long E = (long)N * N;
    const float* v = sim + (long)b * E;
    float* out = adj + (long)b * E;
    const int tid = threadIdx.x;

    if (tid == 0) { s_prefix = 0u; s_remaining = (uint32_t)min((long)k_total, E); }
    __syncthreads();
    // radix select of the k-th largest key, most significant byte first
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (long e = tid; e < E; e += 1024) {
            uint32_t key = f32_order_key(v[e]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 0xffu], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t rem = s_remaining, cum = 0u;
            int bin = 255;
            for (; bin > 0; --bin) {
                if (cum + hist[bin] >= rem) break;
                cum += hist[bin];
            }
            s_prefix = prefix | ((uint32_t)bin << shift);
            s_remaining = rem - cum;           // how many still to take inside this bin
        }
        __syncthreads();
    }
    const uint32_t thr = s_prefix;             // key of the k-th largest element
    const uint32_t need_eq = s_remaining;      // number of threshold-valued entries to keep
    if (tid == 0) s_run = 0u;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (long base = 0; base < E; base += 1024) {
        const long e = base + tid;
        float val = 0.f;
        uint32_t key = 0u;
        bool in = e < E;
        if (in) { val = v[e]; key = f32_order_key(val); }
        const bool eq = in && key == thr;
        const unsigned long long bal = __ballot(eq);
        const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = s_run;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (in) {
            bool sel = key > thr || (eq && (off + before) < need_eq);
            const int i = (int)(e / N), j = (int)(e % N);
            // discrete_graph_learning.py:108 keeps scattered values != 0; :165-166 clears the diagonal
            out[e] = (sel && val != 0.f && i != j) ? 1.f : 0.f;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t t = 0;
            for (int w = 0; w < 16; ++w) t += wave_cnt[w];
            s_run += t;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" long step_knn_workspace_bytes(int B, int N, int F) {
    (void)B; (void)N; (void)F;
    return 256;   // the selection runs in LDS; kept for ABI stability
}

extern "C" int step_topk_mask(const float* sim, int B, int N, int k_total, float* adj, void* work, long work_bytes,
                              void* stream) {
    (void)work; (void)work_bytes;
    STEP_REQUIRE(sim && adj && B > 0 && N > 0 && k_total > 0, "topk_mask: bad arguments");
    topk_mask_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(sim, N, k_total, adj);
    STEP_LAUNCH_CHECK("step_topk_mask");
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
        const int nt = cdiv(N, 128);
        const int pairs = nt * (nt + 1) / 2;
        int split = (1024 + pairs * B - 1) / (pairs * B);
        const int chunks = cdiv(F, GBK);
        if (split > chunks / 4) split = chunks / 4;
        if (split < 1) split = 1;
        gram_bf16_kernel<<<dim3(pairs, B, split), 256, 0, st>>>(hidden, N, F, split, sim);
        STEP_LAUNCH_CHECK("gram_bf16");
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
