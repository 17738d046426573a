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
    StepGemm g;
    memset(&g, 0, sizeof(g));
    g.M = N; g.N = N; g.K = F; g.batch = B;
    g.A = hidden; g.sam = F; g.sak = 1; g.sab = (long)N * F; g.a_bf16 = 1;
    g.B = hidden; g.sbk = 1; g.sbn = F; g.sbb = (long)N * F; g.b_bf16 = 1;
    g.C = sim; g.ldc = N; g.scn = 1; g.scb = (long)N * N;
    g.alpha = 1.f; g.accumulate = 2;
    long tiles = (long)cdiv(N, 64) * cdiv(N, 64) * B;
    int split = (int)((2048 + tiles - 1) / tiles);
    int ksteps = cdiv(F, 16);
    if (split > ksteps / 8) split = ksteps / 8;
    if (split < 1) split = 1;
    g.splitk = split;
    STEP_TRY(step_gemm_launch(g, st));
    dim3 grid(cdiv(N, 256) > 4 ? 4 : cdiv(N, 256), N, B);
    cosine_finalize_kernel<<<grid, 256, 0, st>>>(sim, sqnorm_part, N);
    STEP_LAUNCH_CHECK("cosine_finalize");
    if (!sqnorm_part) {
        cosine_diag_kernel<<<dim3(cdiv(N, 256), B), 256, 0, st>>>(sim, N);
        STEP_LAUNCH_CHECK("cosine_diag");
    }
    return step_topk_mask(sim, B, N, k_total, adj, work, work_bytes, stream);
}
