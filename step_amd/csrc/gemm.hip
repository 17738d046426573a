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
// Tiling: 256 threads = 4 waves; block tile BM x BN x 32, staged through LDS k-major so the
// MFMA operand read (lane l: A[i = l&31][k = l>>5]) is a conflict-free ds_read_b32.
// Split-K (grid.z = batch*splitk) accumulates with f32 atomics.
//
// Dispatch (step_gemm_launch): compute_bf16 -> gemm_bf16.hip; otherwise the staged exact-f32 kernel of gemm_bf16.hip when the
// operands are 16-byte aligned, else the kernels of this file (the direct-fragment kernel for short TN contractions such as
// the f32-mode diffusion hops, the LDS-tiled kernel for the rest).
#include "common.h"
#include "step_internal.h"

namespace {

constexpr int BK = 32;

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
    __shared__ __attribute__((aligned(16))) float As[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

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

    // two-level batch: zb = i1 * batch0 + i0 (batch0 == 0: single level)
    const int i0 = g.batch0 ? zb % g.batch0 : zb, i1 = g.batch0 ? zb / g.batch0 : 0;
    const char* Ab = (const char*)g.A + ((long)i0 * g.sab + (long)i1 * g.sab1) * (long)sizeof(TA);
    const char* Bb = (const char*)g.B + ((long)i0 * g.sbb + (long)i1 * g.sbb1) * (long)sizeof(TB);

    // Loader: every thread moves groups of 4 elements that are consecutive along the operand's contiguous
    // dimension (k when the k-stride is 1, otherwise m / n); a group is one 16-byte global load when the
    // alignment conditions hold (checked once, wave-uniform) and 4 bounds-checked scalar loads otherwise.
    constexpr int GA = BM * BK / 4 / 256;     // groups per thread
    constexpr int GB = BN * BK / 4 / 256;
    const bool a_kc = (g.sak == 1);
    const bool b_kc = (g.sbk == 1);
    const bool f32a = sizeof(TA) == 4, f32b = sizeof(TB) == 4;
    const bool a_al = f32a && (((uintptr_t)g.A & 15) == 0) && g.sab % 4 == 0 && g.sab1 % 4 == 0;
    const bool b_al = f32b && (((uintptr_t)g.B & 15) == 0) && g.sbb % 4 == 0 && g.sbb1 % 4 == 0;
    const bool a_vec = a_al && (a_kc ? (g.sam % 4 == 0 && (g.a_kblk == 0 || (g.a_kblk % 4 == 0 && g.a_kstride % 4 == 0)))
                                     : (g.sam == 1 && g.sak % 4 == 0));
    const bool b_vec = b_al && (b_kc ? (g.sbn % 4 == 0 && (g.b_kblk == 0 || (g.b_kblk % 4 == 0 && g.b_kstride % 4 == 0)) &&
                                        (g.b_nblk == 0 || g.b_nstride % 4 == 0))
                                     : (g.sbn == 1 && g.sbk % 4 == 0 && (g.b_nblk == 0 || (g.b_nblk % 4 == 0 && g.b_nstride % 4 == 0))));

    float ra[GA][4], rb[GB][4];

    auto a_elem = [&](int gm, int gk) -> float {
        float v = 0.f;
        if (gm < g.M && gk < kend) {
            long ki = g.a_kblk ? (long)(gk / g.a_kblk) * g.a_kstride + (gk % g.a_kblk) : (long)gk;
            v = ld_elem<TA>(Ab, (long)gm * g.sam + ki * g.sak);
            if (g.a_kscale) { int c = gk / g.a_kperiod; v = v * g.a_kscale[c] + g.a_kshift[c]; }
        }
        return v;
    };
    auto b_elem = [&](int gk, int gn) -> float {
        float v = 0.f;
        if (gn < g.N && gk < kend) {
            long ki = g.b_kblk ? (long)(gk / g.b_kblk) * g.b_kstride + (gk % g.b_kblk) : (long)gk;
            long ni = g.b_nblk ? (long)(gn / g.b_nblk) * g.b_nstride + (gn % g.b_nblk) : (long)gn;
            v = ld_elem<TB>(Bb, ki * g.sbk + ni * g.sbn);
        }
        return v;
    };
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int r = 0; r < GA; ++r) {
            const int e4 = tid + r * 256;
            if (a_kc) {
                const int kk = (e4 % (BK / 4)) * 4, mm = e4 / (BK / 4);
                const int gm = m0 + mm, gk = k0 + kk;
                if (a_vec && gm < g.M && gk + 3 < kend) {
                    long ki = g.a_kblk ? (long)(gk / g.a_kblk) * g.a_kstride + (gk % g.a_kblk) : (long)gk;
                    float4 t = *(const float4*)((const float*)Ab + (long)gm * g.sam + ki);
                    ra[r][0] = t.x; ra[r][1] = t.y; ra[r][2] = t.z; ra[r][3] = t.w;
                    if (g.a_kscale) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { int c = (gk + i) / g.a_kperiod; ra[r][i] = ra[r][i] * g.a_kscale[c] + g.a_kshift[c]; }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ra[r][i] = a_elem(gm, gk + i);
                }
            } else {
                const int mm = (e4 % (BM / 4)) * 4, kk = e4 / (BM / 4);
                const int gm = m0 + mm, gk = k0 + kk;
                if (a_vec && gm + 3 < g.M && gk < kend) {
                    long ki = g.a_kblk ? (long)(gk / g.a_kblk) * g.a_kstride + (gk % g.a_kblk) : (long)gk;
                    float4 t = *(const float4*)((const float*)Ab + gm + ki * g.sak);
                    ra[r][0] = t.x; ra[r][1] = t.y; ra[r][2] = t.z; ra[r][3] = t.w;
                    if (g.a_kscale) {
                        int c = gk / g.a_kperiod;
#pragma unroll
                        for (int i = 0; i < 4; ++i) ra[r][i] = ra[r][i] * g.a_kscale[c] + g.a_kshift[c];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ra[r][i] = a_elem(gm + i, gk);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < GB; ++r) {
            const int e4 = tid + r * 256;
            if (b_kc) {
                const int kk = (e4 % (BK / 4)) * 4, nn = e4 / (BK / 4);
                const int gn = n0 + nn, gk = k0 + kk;
                if (b_vec && gn < g.N && gk + 3 < kend) {
                    long ki = g.b_kblk ? (long)(gk / g.b_kblk) * g.b_kstride + (gk % g.b_kblk) : (long)gk;
                    long ni = g.b_nblk ? (long)(gn / g.b_nblk) * g.b_nstride + (gn % g.b_nblk) : (long)gn;
                    float4 t = *(const float4*)((const float*)Bb + ki + ni * g.sbn);
                    rb[r][0] = t.x; rb[r][1] = t.y; rb[r][2] = t.z; rb[r][3] = t.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) rb[r][i] = b_elem(gk + i, gn);
                }
            } else {
                const int nn = (e4 % (BN / 4)) * 4, kk = e4 / (BN / 4);
                const int gn = n0 + nn, gk = k0 + kk;
                if (b_vec && gn + 3 < g.N && gk < kend) {
                    long ki = g.b_kblk ? (long)(gk / g.b_kblk) * g.b_kstride + (gk % g.b_kblk) : (long)gk;
                    long ni = g.b_nblk ? (long)(gn / g.b_nblk) * g.b_nstride + (gn % g.b_nblk) : (long)gn;
                    float4 t = *(const float4*)((const float*)Bb + ki * g.sbk + ni);
                    rb[r][0] = t.x; rb[r][1] = t.y; rb[r][2] = t.z; rb[r][3] = t.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) rb[r][i] = b_elem(gk, gn + i);
                }
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int r = 0; r < GA; ++r) {
            const int e4 = tid + r * 256;
            if (a_kc) {
                const int kk = (e4 % (BK / 4)) * 4, mm = e4 / (BK / 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) As[(kk + i) * LDA + mm] = ra[r][i];
            } else {
                const int mm = (e4 % (BM / 4)) * 4, kk = e4 / (BM / 4);
                *(float4*)&As[kk * LDA + mm] = make_float4(ra[r][0], ra[r][1], ra[r][2], ra[r][3]);
            }
        }
#pragma unroll
        for (int r = 0; r < GB; ++r) {
            const int e4 = tid + r * 256;
            if (b_kc) {
                const int kk = (e4 % (BK / 4)) * 4, nn = e4 / (BK / 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) Bs[(kk + i) * LDB + nn] = rb[r][i];
            } else {
                const int nn = (e4 % (BN / 4)) * 4, kk = e4 / (BN / 4);
                *(float4*)&Bs[kk * LDB + nn] = make_float4(rb[r][0], rb[r][1], rb[r][2], rb[r][3]);
            }
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
    float* Cb = g.C + (long)i0 * g.scb + (long)i1 * g.scb1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int gn = n0 + wc * (TN * 32) + j * 32 + li;
            if (gn >= g.N) continue;
            const float bv = (g.bias != nullptr) ? g.bias[gn] : 0.f;
            const long ni = g.c_nblk ? (long)(gn / g.c_nblk) * g.c_nstride + (gn % g.c_nblk) : (long)gn;
            gemm_store_tile(acc[i][j], Cb + ni * g.scn, m0 + wr * (TM * 32) + i * 32 + 4 * lk, g.M, g.ldc, g.alpha, g.accumulate, bv, g.relu, gemm_col_affine(g, gn));
        }
}

// ------------------------------------------------------------------------------------------
// Direct-fragment variant for short contractions (K <= 1024): one wave per 32x32 output tile,
// no LDS and no barriers -- each lane fetches its own MFMA operand words straight from global
// memory (the operands of these calls are L2-resident), so a [307 x 384 x 307] diffusion hop
// becomes ~1000 independent waves instead of 240 latency-bound workgroups.
// k-slot trick: a lane loads 4 consecutive k for its row/column (one dwordx4 when that operand is
// k-contiguous) and MFMA j of the group consumes element j, i.e. MFMA j contracts
// k in {k0 + j, k0 + 4 + j}; A and B use the same map, so the sum over the group is exact.
// Addressing: the k-dependent part of every address is wave-uniform (scalar unit), the lane part
// is a loop-invariant 32-bit offset; slot remaps must have a power-of-two block (shift/mask).
struct DirectArgs {
    int tiles_m, tiles_n;
    int a_sh, a_mask, b_sh, b_mask, bn_sh, bn_mask, cn_sh, cn_mask;   // remap(x) = (x >> sh) * stride + (x & mask)
    int a_vec, b_vec;
};

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm_direct_kernel(StepGemm g, DirectArgs d) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = (long)blockIdx.x * 4 + wave;
    const long per_batch = (long)d.tiles_m * d.tiles_n;
    if (tile >= per_batch * g.batch) return;
    const int zb = (int)(tile / per_batch);
    const int tr = (int)(tile % per_batch);
    const int m0 = (tr / d.tiles_n) * 32, n0 = (tr % d.tiles_n) * 32;
    const int li = lane & 31, lk = lane >> 5;
    const int i0 = g.batch0 ? zb % g.batch0 : zb, i1 = g.batch0 ? zb / g.batch0 : 0;
    const TA* Ab = (const TA*)g.A + (long)i0 * g.sab + (long)i1 * g.sab1;          // wave-uniform
    const TB* Bb = (const TB*)g.B + (long)i0 * g.sbb + (long)i1 * g.sbb1;
    const int gm = m0 + li, gn = n0 + li;
    const bool m_ok = gm < g.M, n_ok = gn < g.N;
    const int K = g.K;
    const int sak = (int)g.sak, sbk = (int)g.sbk;
    // loop-invariant lane offsets (elements); out-of-range rows/columns read element 0 and are zeroed
    const int a_row = m_ok ? (int)((long)gm * g.sam) : 0;
    const int b_col = n_ok ? (int)((((long)(gn >> d.bn_sh)) * g.b_nstride + (gn & d.bn_mask)) * g.sbn) : 0;
    const int a_lane = a_row + 4 * lk * sak;
    const int b_lane = b_col + 4 * lk * sbk;

    auto ldA = [&](const TA* p, int off) -> float {
        if constexpr (sizeof(TA) == 2) return bf16_bits_to_f32(((const uint16_t*)p)[off]); else return p[off];
    };
    auto ldB = [&](const TB* p, int off) -> float {
        if constexpr (sizeof(TB) == 2) return bf16_bits_to_f32(((const uint16_t*)p)[off]); else return p[off];
    };
    // unchecked group load at uniform k0 (multiple of 8, k0 + 8 <= K)
    auto load_a = [&](int k0, float (&v)[4]) {
        const TA* pu = Ab + (long)(((k0 >> d.a_sh) * (int)g.a_kstride + (k0 & d.a_mask)) * sak);
        if (sizeof(TA) == 4 && d.a_vec) {
            float4 t = *(const float4*)((const float*)pu + a_lane);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ldA(pu, a_lane + j * sak);
        }
    };
    auto load_b = [&](int k0, float (&v)[4]) {
        const TB* pu = Bb + (long)(((k0 >> d.b_sh) * (int)g.b_kstride + (k0 & d.b_mask)) * sbk);
        if (sizeof(TB) == 4 && d.b_vec) {
            float4 t = *(const float4*)((const float*)pu + b_lane);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ldB(pu, b_lane + j * sbk);
        }
    };
    // checked variant for the ragged tail
    auto load_a_tail = [&](int k0, float (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gk = k0 + 4 * lk + j;
            const int gkc = gk < K ? gk : 0;
            const long ki = (long)(gkc >> d.a_sh) * g.a_kstride + (gkc & d.a_mask);
            float x = ldA(Ab, a_row + (int)(ki * sak));
            v[j] = gk < K ? x : 0.f;
        }
    };
    auto load_b_tail = [&](int k0, float (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gk = k0 + 4 * lk + j;
            const int gkc = gk < K ? gk : 0;
            const long ki = (long)(gkc >> d.b_sh) * g.b_kstride + (gkc & d.b_mask);
            float x = ldB(Bb, b_col + (int)(ki * sbk));
            v[j] = gk < K ? x : 0.f;
        }
    };

    // Deep software pipeline: with ~1 wave per SIMD there is no thread-level parallelism to hide the
    // load latency, so a whole 64-deep k-chunk (8 groups of 4 MFMAs = 2048 matrix-pipe cycles) is kept
    // in flight in registers while the previous chunk is being consumed.
    // (the vmcnt counter saturates at 63 outstanding loads: two chunks in flight must stay below that)
    constexpr int G = 3, CH = 8 * G;
    // two independent accumulators: consecutive MFMAs never wait on each other's result
    f32x16 acc, acc2;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc2[e] = 0.f; }
    const int kfull = K / CH * CH;
    if (kfull > 0) {
        float a[G][4], b[G][4], an[G][4], bn[G][4];
#pragma unroll
        for (int q = 0; q < G; ++q) { load_a(8 * q, a[q]); load_b(8 * q, b[q]); }
        for (int k0 = 0; k0 < kfull; k0 += CH) {
            const int kn = k0 + CH;
            if (kn < kfull) {
#pragma unroll
                for (int q = 0; q < G; ++q) { load_a(kn + 8 * q, an[q]); load_b(kn + 8 * q, bn[q]); }
            }
#pragma unroll
            for (int q = 0; q < G; ++q)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(m_ok ? a[q][j] : 0.f, n_ok ? b[q][j] : 0.f, acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(m_ok ? a[q][j + 1] : 0.f, n_ok ? b[q][j + 1] : 0.f, acc2, 0, 0, 0);
                }
            if (kn < kfull) {
#pragma unroll
                for (int q = 0; q < G; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[q][j] = an[q][j]; b[q][j] = bn[q][j]; }
            }
        }
    }
    for (int k0 = kfull; k0 < K; k0 += 8) {          // tail: up to G-1 groups, bounds-checked
        float a[4], b[4];
        load_a_tail(k0, a); load_b_tail(k0, b);
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(m_ok ? a[j] : 0.f, n_ok ? b[j] : 0.f, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(m_ok ? a[j + 1] : 0.f, n_ok ? b[j + 1] : 0.f, acc2, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += acc2[e];

    float* Cb = g.C + (long)i0 * g.scb + (long)i1 * g.scb1;
    if (!n_ok) return;
    const float bv = (g.bias != nullptr) ? g.bias[gn] : 0.f;
    const long ni = (long)(gn >> d.cn_sh) * g.c_nstride + (gn & d.cn_mask);
    gemm_store_tile(acc, Cb + ni * g.scn, m0 + 4 * lk, g.M, g.ldc, g.alpha, g.accumulate, bv, g.relu, gemm_col_affine(g, gn));
}

static bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
static int ilog2(int x) { int s = 0; while ((1 << s) < x) ++s; return s; }

// the direct kernel takes: K <= 1024, no per-k affine, power-of-two remap blocks, 32-bit element offsets
bool direct_eligible(const StepGemm& g) {
    if (g.K > 1024 || g.a_kscale) return false;
    if (g.sak == 1 || g.sbk == 1) return false;   // k-contiguous operands: fragment-shaped loads thrash the TA; LDS-tiled kernel instead
    if ((g.a_kblk && (!pow2(g.a_kblk) || g.a_kblk < 8)) || (g.b_kblk && (!pow2(g.b_kblk) || g.b_kblk < 8))) return false;
    if ((g.b_nblk && !pow2(g.b_nblk)) || (g.c_nblk && !pow2(g.c_nblk))) return false;
    const long lim = 1L << 30;
    if ((long)g.M * (g.sam < 0 ? -g.sam : g.sam) + (long)g.K * g.sak >= lim) return false;
    if ((long)g.N * (g.sbn < 0 ? -g.sbn : g.sbn) * (g.b_nblk ? g.b_nstride / g.b_nblk + 1 : 1) + (long)g.K * g.sbk >= lim) return false;
    return true;
}

int launch_direct(const StepGemm& g, hipStream_t st) {
    DirectArgs d;
    d.tiles_m = cdiv(g.M, 32); d.tiles_n = cdiv(g.N, 32);
    auto sm = [](int blk, int& sh, int& mask) { if (blk) { sh = ilog2(blk); mask = blk - 1; } else { sh = 31; mask = 0x7fffffff; } };
    sm(g.a_kblk, d.a_sh, d.a_mask); sm(g.b_kblk, d.b_sh, d.b_mask); sm(g.b_nblk, d.bn_sh, d.bn_mask); sm(g.c_nblk, d.cn_sh, d.cn_mask);
    // 16-byte operand loads: f32, unit k-stride, everything that moves the address a multiple of 4 elements
    d.a_vec = !g.a_bf16 && g.sak == 1 && (g.sam % 4 == 0) && (g.sab % 4 == 0) && (((uintptr_t)g.A & 15) == 0) &&
              (g.a_kblk == 0 || g.a_kstride % 4 == 0) && (g.sab1 % 4 == 0);
    d.b_vec = !g.b_bf16 && g.sbk == 1 && (g.sbn % 4 == 0) && (g.sbb % 4 == 0) && (((uintptr_t)g.B & 15) == 0) &&
              (g.b_kblk == 0 || g.b_kstride % 4 == 0) && (g.b_nblk == 0 || g.b_nstride % 4 == 0) && (g.sbb1 % 4 == 0);
    const long waves = (long)d.tiles_m * d.tiles_n * g.batch;
    dim3 grid((unsigned)((waves + 3) / 4));
    if (g.a_bf16 && g.b_bf16) gemm_direct_kernel<uint16_t, uint16_t><<<grid, 256, 0, st>>>(g, d);
    else if (g.a_bf16) gemm_direct_kernel<uint16_t, float><<<grid, 256, 0, st>>>(g, d);
    else if (g.b_bf16) gemm_direct_kernel<float, uint16_t><<<grid, 256, 0, st>>>(g, d);
    else gemm_direct_kernel<float, float><<<grid, 256, 0, st>>>(g, d);
    STEP_LAUNCH_CHECK("step_gemm(direct)");
    return STEP_OK;
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

int step_gemm_rowsum_separate(StepGemm* g, hipStream_t st) {
    if (!g->a_rowsum) return STEP_OK;
    STEP_REQUIRE(g->sam == 1 && g->a_kblk == 0, "step_gemm: a_rowsum on the general kernels needs an m-contiguous A without k remap");
    STEP_TRY(step_colsum_launch((const float*)g->A, g->K, g->M, g->sak, g->a_rowsum, st));
    g->a_rowsum = nullptr;
    return STEP_OK;
}

int step_gemm_launch(StepGemm g, hipStream_t st) {
    STEP_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0 && g.batch > 0, "step_gemm: bad sizes M=%d N=%d K=%d batch=%d", g.M, g.N, g.K, g.batch);
    STEP_REQUIRE(g.A && g.B && g.C, "step_gemm: null operand");
    if (g.scn == 0) g.scn = 1;
    if (g.a_rowsum)
        STEP_REQUIRE(g.batch == 1 && g.alpha == 1.f && !g.a_bf16 && !g.a_kscale, "step_gemm: a_rowsum needs batch 1, alpha 1, f32 A without affine");
    if (g.c_nscale)
        STEP_REQUIRE(g.c_nshift && g.c_mvec && g.c_nperiod > 0, "step_gemm: c_nscale needs c_nshift, c_mvec and a positive c_nperiod");
    if (g.compute_bf16) return step_gemm_bf16_launch(g, st);
    {
        const int rc = step_gemm_f32_fast_launch(g, st);
        if (rc != -1) return rc;
    }
    STEP_TRY(step_gemm_rowsum_separate(&g, st));
    if (g.splitk < 0 && direct_eligible(g)) g.splitk = 1;      // short contraction: the direct kernel, no split
    if (g.splitk < 0) {            // auto: enough workgroups to fill 256 CUs, at least 4 k-steps each
        STEP_REQUIRE(g.accumulate == 2, "step_gemm: automatic split-K needs accumulate==2");
        const int bm = (g.N <= 32 && g.M > 64) ? 128 : (g.M <= 32 && g.N > 64) ? 32 : (g.M <= 64 || g.N <= 64) ? 64 : 128;
        const int bn = (g.N <= 32 && g.M > 64) ? 32 : (g.M <= 32 && g.N > 64) ? 128 : (g.M <= 64 || g.N <= 64) ? 64 : 128;
        long tiles = (long)cdiv(g.M, bm) * cdiv(g.N, bn) * g.batch;
        long want = (768 + tiles - 1) / tiles;
        long maxs = cdiv(g.K, BK) / 2;
        g.splitk = (int)(want < 1 ? 1 : (want > maxs ? (maxs < 1 ? 1 : maxs) : want));
    }
    if (g.splitk < 1) g.splitk = 1;
    if (g.splitk == 1 && direct_eligible(g)) return launch_direct(g, st);
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

int step_gemm_launch_fused(StepGemm g, const GemmFused& fused, hipStream_t st) {
    STEP_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0 && g.batch == 1 && g.A && g.B && g.C, "step_gemm(fused): bad descriptor");
    STEP_REQUIRE(fused.channels > 0 && fused.period > 0 && (long)fused.channels * fused.period == g.N, "step_gemm(fused): N != channels * period");
    if (g.scn == 0) g.scn = 1;
    if (g.splitk < 1) g.splitk = 1;
    if (g.c_nscale) STEP_REQUIRE(g.c_nshift && g.c_mvec && g.c_nperiod == fused.period, "step_gemm(fused): column affine must use the same period");
    if (g.compute_bf16) return step_gemm_bf16_launch(g, st, &fused);
    const int rc = step_gemm_f32_fast_launch(g, st, &fused);
    if (rc == -1) { step_set_error("step_gemm(fused): operands do not qualify for the staged path"); return STEP_ERR_ARG; }
    return rc;
}

extern "C" int step_gemm(const StepGemm* g, void* stream) {
    STEP_REQUIRE(g != nullptr, "step_gemm: null descriptor");
    return step_gemm_launch(*g, (hipStream_t)stream);
}
