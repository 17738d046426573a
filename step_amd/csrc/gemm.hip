// Generic strided-batched GEMM on the f32-input matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate, 157 TF peak).
//
//   C[b](m,n) (op)= alpha * sum_k A[b](m,k) * B[b](k,n)  (+ bias[n]) (relu)
//
// A and B are addressed by element strides, so every transpose / crop / channel-slice the
// STEP path needs (nconv over the first adjacency index, its two backward contractions,
// the DGL fc forward/backward, weight-gradient reductions over positions) is one call with
// no data movement.  Inputs may be f32 or bf16 (bf16 is widened on the way into LDS, which
// is how the cosine Gram matrix reads the TSFormer hidden states).
//
// Tiling: 256 threads = 4 waves; block tile BM x BN x 16, staged through LDS k-major so the
// MFMA operand read (lane l: A[i = l&31][k = l>>5]) is a conflict-free ds_read_b32.
// Split-K (grid.z = batch*splitk) accumulates with f32 atomics.
#include "common.h"
#include "step_internal.h"

namespace {

constexpr int BK = 16;

template <typename T>
__device__ __forceinline__ float ld_elem(const void* p, long idx) {
    if constexpr (sizeof(T) == 2) {
        return bf16_bits_to_f32(((const uint16_t*)p)[idx]);
    } else {
        return ((const float*)p)[idx];
    }
}

template <int BM, int BN, int WM, int WN, typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm_f32mfma_kernel(StepGemm g) {
    constexpr int TM = BM / WM / 32;      // 32x32 tiles per wave along m
    constexpr int TN = BN / WN / 32;
    constexpr int LDA = BM + 4;           // +4 words: k-rows land on different banks
    constexpr int LDB = BN + 4;
    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    const int zb = blockIdx.z / g.splitk;          // batch index
    const int zs = blockIdx.z % g.splitk;          // k-split index
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;

    // K range of this split (multiples of BK)
    const int ksteps = (g.K + BK - 1) / BK;
    const int per = (ksteps + g.splitk - 1) / g.splitk;
    const int kbeg = zs * per * BK;
    const int kend = min(g.K, (zs + 1) * per * BK);

    const char* Ab = (const char*)g.A + (long)zb * g.sab * (long)sizeof(TA);
    const char* Bb = (const char*)g.B + (long)zb * g.sbb * (long)sizeof(TB);

    // loader maps: make the thread index run along whichever dimension is contiguous
    constexpr int AE = BM * BK / 256;     // elements per thread
    constexpr int BE = BN * BK / 256;
    const bool a_kc = (g.sak == 1);
    const bool b_kc = (g.sbk == 1);

    float ra[AE], rb[BE];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int r = 0; r < AE; ++r) {
            int e = tid + r * 256;
            int kk, mm;
            if (a_kc) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
            int gm = m0 + mm, gk = k0 + kk;
            float v = 0.f;
            if (gm < g.M && gk < kend) {
                long ki = g.a_kblk ? (long)(gk / g.a_kblk) * g.a_kstride + (gk % g.a_kblk) : (long)gk;
                v = ld_elem<TA>(Ab, (long)gm * g.sam + ki * g.sak);
                if (g.a_kscale) { int c = gk / g.a_kperiod; v = v * g.a_kscale[c] + g.a_kshift[c]; }
            }
            ra[r] = v;
        }
#pragma unroll
        for (int r = 0; r < BE; ++r) {
            int e = tid + r * 256;
            int kk, nn;
            if (b_kc) { kk = e % BK; nn = e / BK; } else { nn = e % BN; kk = e / BN; }
            int gn = n0 + nn, gk = k0 + kk;
            float v = 0.f;
            if (gn < g.N && gk < kend) {
                long ki = g.b_kblk ? (long)(gk / g.b_kblk) * g.b_kstride + (gk % g.b_kblk) : (long)gk;
                long ni = g.b_nblk ? (long)(gn / g.b_nblk) * g.b_nstride + (gn % g.b_nblk) : (long)gn;
                v = ld_elem<TB>(Bb, ki * g.sbk + ni * g.sbn);
            }
            rb[r] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int r = 0; r < AE; ++r) {
            int e = tid + r * 256;
            int kk, mm;
            if (a_kc) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
            As[kk * LDA + mm] = ra[r];
        }
#pragma unroll
        for (int r = 0; r < BE; ++r) {
            int e = tid + r * 256;
            int kk, nn;
            if (b_kc) { kk = e % BK; nn = e / BK; } else { nn = e % BN; kk = e / BN; }
            Bs[kk * LDB + nn] = rb[r];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, lk = lane >> 5;
    if (kbeg < kend) {
        load_tiles(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            __syncthreads();                  // previous tile fully consumed
            store_tiles();
            __syncthreads();
            if (k0 + BK < kend) load_tiles(k0 + BK);     // prefetch next tile into registers
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = As[(kk + lk) * LDA + wr * (TM * 32) + i * 32 + li];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + lk) * LDB + wc * (TN * 32) + j * 32 + li];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // epilogue: D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    float* Cb = g.C + (long)zb * g.scb;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int gn = n0 + wc * (TN * 32) + j * 32 + li;
            if (gn >= g.N) continue;
            float bv = (g.bias != nullptr) ? g.bias[gn] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int gm = m0 + wr * (TM * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (gm >= g.M) continue;
                float v = g.alpha * acc[i][j][e];
                long ni = g.c_nblk ? (long)(gn / g.c_nblk) * g.c_nstride + (gn % g.c_nblk) : (long)gn;
                float* dst = Cb + (long)gm * g.ldc + ni * g.scn;
                if (g.accumulate == 2) {
                    atomicAdd(dst, v);
                } else {
                    if (g.accumulate == 1) v += *dst;
                    v += bv;
                    if (g.relu) v = fmaxf(v, 0.f);
                    *dst = v;
                }
            }
        }
}

template <int BM, int BN, int WM, int WN>
int launch(const StepGemm& g, hipStream_t st) {
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.batch * g.splitk);
    if (g.a_bf16 && g.b_bf16)
        gemm_f32mfma_kernel<BM, BN, WM, WN, uint16_t, uint16_t><<<grid, 256, 0, st>>>(g);
    else if (g.a_bf16)
        gemm_f32mfma_kernel<BM, BN, WM, WN, uint16_t, float><<<grid, 256, 0, st>>>(g);
    else if (g.b_bf16)
        gemm_f32mfma_kernel<BM, BN, WM, WN, float, uint16_t><<<grid, 256, 0, st>>>(g);
    else
        gemm_f32mfma_kernel<BM, BN, WM, WN, float, float><<<grid, 256, 0, st>>>(g);
    STEP_LAUNCH_CHECK("step_gemm");
    return STEP_OK;
}

}  // namespace

int step_gemm_launch(StepGemm g, hipStream_t st) {
    STEP_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0 && g.batch > 0, "step_gemm: bad sizes M=%d N=%d K=%d batch=%d", g.M, g.N, g.K, g.batch);
    STEP_REQUIRE(g.A && g.B && g.C, "step_gemm: null operand");
    if (g.scn == 0) g.scn = 1;
    if (g.splitk < 0) {            // auto: enough workgroups to fill 256 CUs, at least 4 k-steps each
        STEP_REQUIRE(g.accumulate == 2, "step_gemm: automatic split-K needs accumulate==2");
        const int bm = (g.N <= 32 && g.M > 64) ? 128 : (g.M <= 32 && g.N > 64) ? 32 : (g.M <= 64 || g.N <= 64) ? 64 : 128;
        const int bn = (g.N <= 32 && g.M > 64) ? 32 : (g.M <= 32 && g.N > 64) ? 128 : (g.M <= 64 || g.N <= 64) ? 64 : 128;
        long tiles = (long)cdiv(g.M, bm) * cdiv(g.N, bn) * g.batch;
        long want = (768 + tiles - 1) / tiles;
        long maxs = cdiv(g.K, 16) / 4;
        g.splitk = (int)(want < 1 ? 1 : (want > maxs ? (maxs < 1 ? 1 : maxs) : want));
    }
    if (g.splitk < 1) g.splitk = 1;
    if (g.splitk > 1) {
        STEP_REQUIRE(g.accumulate == 2, "step_gemm: split-K needs accumulate==2 (atomic) and a pre-zeroed/accumulating C");
    }
    STEP_REQUIRE(!(g.accumulate == 2 && (g.bias || g.relu)), "step_gemm: bias/relu epilogue not available with atomic accumulate");
    if (g.N <= 32 && g.M > 64) return launch<128, 32, 4, 1>(g, st);
    if (g.M <= 32 && g.N > 64) return launch<32, 128, 1, 4>(g, st);
    if (g.M <= 64 || g.N <= 64) return launch<64, 64, 2, 2>(g, st);
    // enough 128x128 tiles to fill the chip?  otherwise prefer 64x64 for occupancy
    long tiles128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * g.batch * g.splitk;
    if (tiles128 < 256) return launch<64, 64, 2, 2>(g, st);
    return launch<128, 128, 2, 2>(g, st);
}

extern "C" int step_gemm(const StepGemm* g, void* stream) {
    STEP_REQUIRE(g != nullptr, "step_gemm: null descriptor");
    return step_gemm_launch(*g, (hipStream_t)stream);
}
