// Matrix-core GEMM kernels of the generic strided-batched descriptor (StepGemm), two generations:
//
//  1. gemm_bf16mfma_kernel -- general bf16 variant (compute_bf16 = 1) for arbitrary strides: f32 / bf16 operands rounded to
//     bf16 on the way into LDS, v_mfma_f32_32x32x16_bf16, f32 accumulation and output.  LDS holds both operands row-major with
//     k contiguous (80-byte pitch); element-wise loads with runtime index arithmetic.  Now only the fallback for operands that
//     are not 16-byte aligned.
//  2. gemm_fast_kernel -- the staged pipeline every hot contraction of the training step runs on (see the block comment
//     further down): 16-byte loads, BK = 64, double-buffered LDS, in bf16 (F32C = false) and exact-f32 (F32C = true) form.
//     step_gemm_bf16_launch / step_gemm_f32_fast_launch pick it whenever the operand layouts qualify.
#include "common.h"
#include "step_internal.h"

namespace {

constexpr int BK = 32;
constexpr int PITCH = BK * 2 + 16;        // bytes per staged row

typedef __attribute__((ext_vector_type(4))) float f32x4v;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

typedef __attribute__((ext_vector_type(2))) float f32x2v;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
// one v_cvt_pk_bf16_f32 on two arbitrary registers (RNE); written as asm so that the register-level 4x4 transposition of
// the MC loader stays a choice of source operands (the vectoriser otherwise builds it through scratch memory)
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack2(a, b), pack2(c, d)); }
__device__ __forceinline__ uint2 pack4(const float (&v)[4]) { return pack4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ uint16_t bf16_of(float x) {
    return __builtin_bit_cast(uint16_t, (__bf16)x);
}
template <typename T>
__device__ __forceinline__ float ld1(const void* p, long idx) {
    if constexpr (sizeof(T) == 2) return bf16_bits_to_f32(((const uint16_t*)p)[idx]); else return ((const float*)p)[idx];
}

template <int BM, int BN, typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm_bf16mfma_kernel(StepGemm g) {
    constexpr int TM = BM / 64, TN = BN / 64;         // 32x32 tiles per wave (2 x 2 waves)
    constexpr int GA = BM * BK / 4 / 256, GB = BN * BK / 4 / 256;
    __shared__ __attribute__((aligned(16))) char As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) char Bs[BN * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int zb = blockIdx.z / g.splitk, zs = blockIdx.z % g.splitk;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int ksteps = (g.K + BK - 1) / BK;
    const int per = (ksteps + g.splitk - 1) / g.splitk;
    const int kbeg = zs * per * BK;
    const int kend = min(g.K, (zs + 1) * per * BK);
    const int i0 = g.batch0 ? zb % g.batch0 : zb, i1 = g.batch0 ? zb / g.batch0 : 0;
    const char* Ab = (const char*)g.A + ((long)i0 * g.sab + (long)i1 * g.sab1) * (long)sizeof(TA);
    const char* Bb = (const char*)g.B + ((long)i0 * g.sbb + (long)i1 * g.sbb1) * (long)sizeof(TB);

    const bool a_kc = (g.sak == 1), b_kc = (g.sbk == 1);
    const bool a_al = sizeof(TA) == 4 && (((uintptr_t)g.A & 15) == 0) && g.sab % 4 == 0 && g.sab1 % 4 == 0;
    const bool b_al = sizeof(TB) == 4 && (((uintptr_t)g.B & 15) == 0) && g.sbb % 4 == 0 && g.sbb1 % 4 == 0;
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
            v = ld1<TA>(Ab, (long)gm * g.sam + ki * g.sak);
            if (g.a_kscale) { int c = gk / g.a_kperiod; v = v * g.a_kscale[c] + g.a_kshift[c]; }
        }
        return v;
    };
    auto b_elem = [&](int gk, int gn) -> float {
        float v = 0.f;
        if (gn < g.N && gk < kend) {
            long ki = g.b_kblk ? (long)(gk / g.b_kblk) * g.b_kstride + (gk % g.b_kblk) : (long)gk;
            long ni = g.b_nblk ? (long)(gn / g.b_nblk) * g.b_nstride + (gn % g.b_nblk) : (long)gn;
            v = ld1<TB>(Bb, ki * g.sbk + ni * g.sbn);
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
                *(uint2*)(As + mm * PITCH + kk * 2) = pack4(ra[r]);
            } else {
                const int mm = (e4 % (BM / 4)) * 4, kk = e4 / (BM / 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint16_t*)(As + (mm + i) * PITCH + kk * 2) = bf16_of(ra[r][i]);
            }
        }
#pragma unroll
        for (int r = 0; r < GB; ++r) {
            const int e4 = tid + r * 256;
            if (b_kc) {
                const int kk = (e4 % (BK / 4)) * 4, nn = e4 / (BK / 4);
                *(uint2*)(Bs + nn * PITCH + kk * 2) = pack4(rb[r]);
            } else {
                const int nn = (e4 % (BN / 4)) * 4, kk = e4 / (BN / 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint16_t*)(Bs + (nn + i) * PITCH + kk * 2) = bf16_of(rb[r][i]);
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

    const int r = lane & 31, h = lane >> 5;
    if (kbeg < kend) {
        load_tiles(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            __syncthreads();
            store_tiles();
            __syncthreads();
            if (k0 + BK < kend) load_tiles(k0 + BK);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(As + (wr * (TM * 32) + i * 32 + r) * PITCH + ks * 32 + h * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8*)(Bs + (wc * (TN * 32) + j * 32 + r) * PITCH + ks * 32 + h * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    float* Cb = g.C + (long)i0 * g.scb + (long)i1 * g.scb1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + wc * (TN * 32) + j * 32 + r;
            if (gn >= g.N) continue;
            const float bv = (g.bias != nullptr) ? g.bias[gn] : 0.f;
            const long ni = g.c_nblk ? (long)(gn / g.c_nblk) * g.c_nstride + (gn % g.c_nblk) : (long)gn;
            gemm_store_tile(acc[i][j], Cb + ni * g.scn, m0 + wr * (TM * 32) + i * 32 + 4 * h, g.M, g.ldc, g.alpha, g.accumulate, bv, g.relu, gemm_col_affine(g, gn));
        }
}


// ---------------------------------------------------------------------------------------------------------
// Fast path: 16-byte global loads only, no integer division in the k loop, BK = 64, LDS double-buffered (one
// barrier per k step), register prefetch one step ahead.  Each operand is staged by one of three loaders:
//   KC_F32   rows with k contiguous, f32   : float4 along k -> cvt_pk_bf16 x2 -> one 8-byte LDS write
//   KC_BF16  rows with k contiguous, bf16  : 8 bf16 (16 B) -> two 8-byte LDS writes
//   MC_F32   k rows with m (n) contiguous  : a 4(k) x 4(m) register block from four float4 loads, transposed in
//            registers -> four 8-byte LDS writes (2-way bank conflict at the 136-byte pitch)
// LDS rows are k-contiguous with a 136-byte pitch: fragment reads (two ds_read_b64 per 32x16 operand) are
// conflict-free.  Rows / columns past M / N are loaded as whatever the padded pitch holds (they only feed
// outputs that are never stored); k past the end is zeroed in both operands.
enum { KC_F32 = 0, KC_BF16 = 1, MC_F32 = 2, MC_BF16 = 3 };       // MC_BF16: B operand only (8(n) x 4(k) register blocks)
constexpr int FBK = 64;
constexpr int FPITCH = FBK * 2 + 8;          // bf16 rows: 64 x 2 B + 8
constexpr int FPITCH32 = FBK * 4 + 16;       // f32 rows (exact-f32 variant): 64 x 4 B + 16

struct FastArgs {
    const GemmKSeg* ktab;              // segmented contraction (KTAB kernels): one entry per 32-wide k block and outer batch index
    int ktab_per;                      // entries per outer batch index i1
    int a_klog, b_klog, b_nlog;        // log2 of the remap block along k / n, 30 = no remap
    int xcd_swizzle;                   // remap block ids so that the n tiles of an m tile share an XCD (see gemm_fast_kernel)
    int wide_store;                    // epilogue through LDS with 16-byte row pieces (plain store / += of a dense, aligned C)
    int m_fast;                        // block order with the m tiles of one n tile adjacent (they share that n tile's B columns), see launch_fast_fused
    long split_stride;                 // split-K through a workspace: split zs writes its partial tiles at C + zs * split_stride (0: off)
    GemmFused fu;                      // fused epilogues of the DGL fc backward (step_internal.h); all-null = off
};

// the LDS-staged epilogue pays off for large dense outputs only (two extra barriers and an LDS round trip per tile)
static int wide_store_ok(const StepGemm& g, bool force = false) {
    return g.accumulate != 2 && !g.a_rowsum && g.c_nblk == 0 && g.scn == 1 && g.N % 4 == 0 && g.ldc % 4 == 0 && g.scb % 4 == 0 &&
           g.scb1 % 4 == 0 && ((uintptr_t)g.C & 15) == 0 && (force || (long)g.M * g.N * g.batch >= (4L << 20));
}

// no remap is encoded as lg = 30, stride = 0 (i >> 30 == 0): branch-free
__device__ __forceinline__ long remap(int i, int lg, long stride) {
    return (long)(i >> lg) * stride + (i & ((1 << lg) - 1));
}

template <int MODE, int BR>
struct Stage {
    static constexpr int NV = MODE == KC_F32 ? BR * FBK / 4 / 256 : (MODE == KC_BF16 ? BR * FBK / 8 / 256 : (MODE == MC_F32 ? BR * FBK / 16 / 256 : (BR * FBK / 32 + 255) / 256));
    static constexpr int NR = (MODE == MC_F32 || MODE == MC_BF16) ? NV * 4 : NV;
    float f[NR][4];                    // plain scalars: every index below is a compile-time constant after unrolling
};

// rows: extent of the non-k dimension (M or N); srow / sk: element strides of that dimension and of k.
// stage_load only issues the loads (out-of-range lanes read the operand's first 16 bytes): straight-line code, every load
// of a k step is in flight before anything waits.  stage_fix, called after the MFMAs of the previous step, zeroes the
// out-of-range elements and applies the per-channel affine.
template <int MODE, int BR>
__device__ __forceinline__ void stage_load(Stage<MODE, BR>& s, const char* base, int row0, int rows, long srow, long sk, int k0,
                                           int kend, int klog, long kstride, int rlog, long rstride, int tid) {
    if constexpr (MODE == KC_F32) {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            const int k = k0 + (e % (FBK / 4)) * 4, row = row0 + e / (FBK / 4);
            const bool ok = row < rows && k < kend;
            const long off = ok ? remap(row, rlog, rstride) * srow + remap(k, klog, kstride) : 0;
            const float4 t = *(const float4*)((const float*)base + off);
            s.f[r][0] = t.x; s.f[r][1] = t.y; s.f[r][2] = t.z; s.f[r][3] = t.w;
        }
    } else if constexpr (MODE == KC_BF16) {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            const int k = k0 + (e % (FBK / 8)) * 8, row = row0 + e / (FBK / 8);
            const bool ok = row < rows && k < kend;
            const long off = ok ? remap(row, rlog, rstride) * srow + remap(k, klog, kstride) : 0;
            const float4 t = *(const float4*)((const uint16_t*)base + off);
            s.f[r][0] = t.x; s.f[r][1] = t.y; s.f[r][2] = t.z; s.f[r][3] = t.w;
        }
    } else if constexpr (MODE == MC_BF16) {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;                                                  // (64-wide tiles: threads 128.. have no block)
            const int row = row0 + (e % (BR / 8)) * 8, kb = k0 + (e / (BR / 8)) * 4;      // 8 rows (one 16-byte load) x 4 k
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = row < rows && kb + j < kend && e < BR * FBK / 32;
                const long off = ok ? (long)row * srow + remap(kb + j, klog, kstride) * sk : 0;
                const float4 t = *(const float4*)((const uint16_t*)base + off);
                s.f[r * 4 + j][0] = t.x; s.f[r * 4 + j][1] = t.y; s.f[r * 4 + j][2] = t.z; s.f[r * 4 + j][3] = t.w;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            const int row = row0 + (e % (BR / 4)) * 4, kb = k0 + (e / (BR / 4)) * 4;
            const long roff = remap(row, rlog, rstride) * srow;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = row < rows && kb + j < kend;
                const long off = ok ? roff + remap(kb + j, klog, kstride) * sk : 0;
                const float4 t = *(const float4*)((const float*)base + off);
                s.f[r * 4 + j][0] = t.x; s.f[r * 4 + j][1] = t.y; s.f[r * 4 + j][2] = t.z; s.f[r * 4 + j][3] = t.w;
            }
        }
    }
}

// Segmented contraction: k block kb = k >> 5 of operand `which` lives at base + seg[kb].off + row * seg[kb].rs + i0 * seg[kb].bs (+ k & 31):
// blocks of 32 contiguous k values scattered over several buffers (the per-layer gcn buffers of the GraphWaveNet, gwnet.hip).
template <int BR>
__device__ __forceinline__ void stage_load_seg(Stage<KC_F32, BR>& s, const float* base, int row0, int rows, int k0, int kend,
                                               const GemmKSeg* seg, int which, long i0, int tid) {
#pragma unroll
    for (int r = 0; r < s.NV; ++r) {
        const int e = tid + r * 256;
        const int k = k0 + (e % (FBK / 4)) * 4, row = row0 + e / (FBK / 4);
        const bool ok = row < rows && k < kend;
        const GemmKSeg sg = seg[(ok ? k : k0) >> 5];
        const long off = ok ? (which ? sg.b_off + (long)row * sg.b_rs + i0 * sg.b_bs : sg.a_off + (long)row * sg.a_rs + i0 * sg.a_bs) + (k & 31)
                            : (which ? sg.b_off : sg.a_off);
        const float4 t = *(const float4*)(base + off);
        s.f[r][0] = t.x; s.f[r][1] = t.y; s.f[r][2] = t.z; s.f[r][3] = t.w;
    }
}

template <int MODE, int BR>
__device__ __forceinline__ void stage_fix(Stage<MODE, BR>& s, int row0, int rows, int k0, int kend, const float* kscale,
                                          const float* kshiftv, int kperiod, int ones_row, int tid) {
    if constexpr (MODE == KC_F32) {
        float ks0 = 1.f, kh0 = 0.f, ks1 = 1.f, kh1 = 0.f;
        int kbnd = 0x7fffffff;
        if (kscale) {      // the 64-wide k tile spans at most two channels (kperiod >= 64): uniform scalars, loaded once per tile
            const int c0 = k0 / kperiod, c1 = (c0 + 1) * kperiod < kend ? c0 + 1 : c0;
            ks0 = kscale[c0]; kh0 = kshiftv[c0]; ks1 = kscale[c1]; kh1 = kshiftv[c1];
            kbnd = (c0 + 1) * kperiod;
        }
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            const int k = k0 + (e % (FBK / 4)) * 4, row = row0 + e / (FBK / 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = k + i < kbnd ? s.f[r][i] * ks0 + kh0 : s.f[r][i] * ks1 + kh1;
                s.f[r][i] = k + i < kend ? (row < rows ? a : (row == ones_row ? 1.f : 0.f)) : 0.f;
            }
        }
    } else if constexpr (MODE == KC_BF16) {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            const int k = k0 + (e % (FBK / 8)) * 8, row = row0 + e / (FBK / 8);
            const int left = row < rows ? kend - k : 0;          // valid elements in this 8-wide group
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t m = 2 * j + 1 < left ? 0xffffffffu : (2 * j < left ? 0xffffu : 0u);
                s.f[r][j] = __uint_as_float(__float_as_uint(s.f[r][j]) & m);
            }
        }
    } else if constexpr (MODE == MC_BF16) {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            const int row = row0 + (e % (BR / 8)) * 8, kb = k0 + (e / (BR / 8)) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = row < rows && kb + j < kend && e < BR * FBK / 32;          // rows come in whole groups of 8 (the row extent is a multiple of 8)
#pragma unroll
                for (int i = 0; i < 4; ++i) s.f[r * 4 + j][i] = ok ? s.f[r * 4 + j][i] : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            const int row = row0 + (e % (BR / 4)) * 4, kb = k0 + (e / (BR / 4)) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool kok = kb + j < kend;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    s.f[r * 4 + j][i] = kok ? (row + i == ones_row ? 1.f : (row < rows ? s.f[r * 4 + j][i] : 0.f)) : 0.f;
            }
        }
    }
}

template <int MODE, int BR, bool F32C>
__device__ __forceinline__ void stage_store(const Stage<MODE, BR>& s, char* lds, int tid) {
    if constexpr (F32C) {                 // exact-f32 variant: rows of 64 f32, 16-byte writes
        static_assert(MODE != KC_BF16 && MODE != MC_BF16, "bf16 operands only feed the bf16 variant");
        if constexpr (MODE == KC_F32) {
#pragma unroll
            for (int r = 0; r < s.NV; ++r) {
                const int e = tid + r * 256;
                *(float4*)(lds + (e / (FBK / 4)) * FPITCH32 + (e % (FBK / 4)) * 16) = make_float4(s.f[r][0], s.f[r][1], s.f[r][2], s.f[r][3]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < s.NV; ++r) {
                const int e = tid + r * 256;
                char* d = lds + ((e % (BR / 4)) * 4) * FPITCH32 + (e / (BR / 4)) * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(float4*)(d + i * FPITCH32) = make_float4(s.f[r * 4][i], s.f[r * 4 + 1][i], s.f[r * 4 + 2][i], s.f[r * 4 + 3][i]);
            }
        }
    } else if constexpr (MODE == KC_F32) {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            *(uint2*)(lds + (e / (FBK / 4)) * FPITCH + (e % (FBK / 4)) * 8) = pack4(s.f[r][0], s.f[r][1], s.f[r][2], s.f[r][3]);
        }
    } else if constexpr (MODE == KC_BF16) {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            char* d = lds + (e / (FBK / 8)) * FPITCH + (e % (FBK / 8)) * 16;
            *(uint2*)d = make_uint2(__float_as_uint(s.f[r][0]), __float_as_uint(s.f[r][1]));
            *(uint2*)(d + 8) = make_uint2(__float_as_uint(s.f[r][2]), __float_as_uint(s.f[r][3]));
        }
    } else if constexpr (MODE == MC_BF16) {
        // register block: dword p of load j holds rows 2p (low half) and 2p+1 (high half) at k = kb + j; LDS rows want 4 consecutive k
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            if (e >= BR * FBK / 32) continue;
            char* d = lds + ((e % (BR / 8)) * 8) * FPITCH + (e / (BR / 8)) * 8;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t d0 = __float_as_uint(s.f[r * 4][p]), d1 = __float_as_uint(s.f[r * 4 + 1][p]);
                const uint32_t d2 = __float_as_uint(s.f[r * 4 + 2][p]), d3 = __float_as_uint(s.f[r * 4 + 3][p]);
                *(uint2*)(d + (2 * p) * FPITCH) = make_uint2((d0 & 0xffffu) | (d1 << 16), (d2 & 0xffffu) | (d3 << 16));
                *(uint2*)(d + (2 * p + 1) * FPITCH) = make_uint2((d0 >> 16) | (d1 & 0xffff0000u), (d2 >> 16) | (d3 & 0xffff0000u));
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < s.NV; ++r) {
            const int e = tid + r * 256;
            char* d = lds + ((e % (BR / 4)) * 4) * FPITCH + (e / (BR / 4)) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(uint2*)(d + i * FPITCH) = pack4(s.f[r * 4][i], s.f[r * 4 + 1][i], s.f[r * 4 + 2][i], s.f[r * 4 + 3][i]);
        }
    }
}

// F32C = true: same staging, but LDS keeps f32 and the products run on v_mfma_f32_32x32x2_f32 (exact f32).  A lane of that
// instruction supplies one k per operand (k = lane >> 5); it reads 16-byte chunk 2*ks + (lane >> 5) of its row and feeds the
// four values to four successive MFMAs -- both operands permute k identically, so the contraction is unchanged.
// FUSED: the DGL fc-backward epilogues (GemmFused).  A separate instantiation: their batched loads would otherwise set the register
// allocation (and with it the occupancy) of every staged GEMM of the step.
template <int BM, int BN, int AMODE, int BMODE, bool F32C, bool KTAB = false, bool FUSED = false>
__global__ __launch_bounds__(256) void gemm_fast_kernel(StepGemm g, FastArgs fa) {
    static_assert(!KTAB || (AMODE == KC_F32 && BMODE == KC_F32), "segmented contraction: k-contiguous f32 blocks on both sides");
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PITCH_ = F32C ? FPITCH32 : FPITCH;
    constexpr int BUF = (BM + BN) * PITCH_;
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order (fa.xcd_swizzle, host-checked: gridDim.y * gridDim.z % 8 == 0): flattened block b is observed to run on XCD
    // b % 8, so the n tiles of one (m tile, batch) group are given ids 8 apart -- they share that group's A rows in ONE XCD's L2 instead
    // of fetching them into up to 8.  Bijective; a pure placement choice.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (fa.xcd_swizzle) {
        const unsigned gx = gridDim.x, lin = bx + gx * (by + gridDim.y * bz);
        const unsigned q = lin >> 3, j = q / gx, grp = (lin & 7u) + 8u * j;
        bx = q - j * gx; by = grp % gridDim.y; bz = grp / gridDim.y;
    } else if (fa.m_fast) {
        // few m tiles over a B operand far larger than the caches (d_a2 of the graph learner: 7 row tiles x 2114 column tiles of a 54 MB
        // weight copy at PEMS07): in the default order the m tiles of one n tile are a whole grid row apart and each re-reads its B columns
        // from HBM; adjacent, they meet in the memory-side cache
        const unsigned lin = bx + gridDim.x * by;
        by = lin % gridDim.y; bx = lin / gridDim.y;
    }
    const int zb = bz / g.splitk, zs = bz % g.splitk;
    const int m0 = by * BM, n0 = bx * BN;
    const int ksteps = (g.K + FBK - 1) / FBK;
    const int per = (ksteps + g.splitk - 1) / g.splitk;
    const int kbeg = zs * per * FBK;
    const int kend = min(g.K, (zs + 1) * per * FBK);
    const int i0 = g.batch0 ? zb % g.batch0 : zb, i1 = g.batch0 ? zb / g.batch0 : 0;
    const char* Ab = (const char*)g.A + ((long)i0 * g.sab + (long)i1 * g.sab1) * (AMODE == KC_BF16 ? 2 : 4);
    const char* Bb = (const char*)g.B + ((long)i0 * g.sbb + (long)i1 * g.sbb1) * ((BMODE == KC_BF16 || BMODE == MC_BF16) ? 2 : 4);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    const int ones_row = (BMODE != KC_BF16 && BMODE != MC_BF16 && g.a_rowsum) ? g.N : -1;      // virtual all-ones column of B -> row sums of A
    if (kbeg < kend) {
        Stage<AMODE, BM> sa;
        Stage<BMODE, BN> sb;
        __shared__ GemmKSeg segs[KTAB ? GEMM_KSEG_MAX : 1];
        if constexpr (KTAB) {
            const int nseg = (g.K + 31) >> 5;
            for (int i = tid; i < nseg; i += 256) segs[i] = fa.ktab[(long)i1 * fa.ktab_per + i];
            __syncthreads();
        }
        auto fetch = [&](int k0) {
            if constexpr (KTAB) {
                stage_load_seg<BM>(sa, (const float*)g.A, m0, g.M, k0, kend, segs, 0, i0, tid);
                stage_load_seg<BN>(sb, (const float*)g.B, n0, g.N, k0, kend, segs, 1, i0, tid);
            } else {
                stage_load<AMODE, BM>(sa, Ab, m0, g.M, g.sam, g.sak, k0, kend, fa.a_klog, g.a_kstride, 30, 0, tid);
                stage_load<BMODE, BN>(sb, Bb, n0, g.N, g.sbn, g.sbk, k0, kend, fa.b_klog, g.b_kstride, fa.b_nlog, g.b_nstride, tid);
            }
        };
        auto commit = [&](int k0, char* buf) {          // masks / affine, bf16 rounding, LDS writes
            stage_fix<AMODE, BM>(sa, m0, g.M, k0, kend, g.a_kscale, g.a_kshift, g.a_kperiod, -1, tid);
            stage_fix<BMODE, BN>(sb, n0, g.N, k0, kend, nullptr, nullptr, 1, ones_row, tid);
            stage_store<AMODE, BM, F32C>(sa, buf, tid);
            stage_store<BMODE, BN, F32C>(sb, buf + BM * PITCH_, tid);
        };
        fetch(kbeg);
        commit(kbeg, lds);
        __syncthreads();
        int cur = 0;
        for (int k0 = kbeg; k0 < kend; k0 += FBK) {
            const bool more = k0 + FBK < kend;
            if (more) fetch(k0 + FBK);
            const char* As = lds + cur * BUF;
            const char* Bs = As + BM * PITCH_;
            if constexpr (F32C) {
#pragma unroll
                for (int ks = 0; ks < FBK / 8; ++ks) {
                    float4 a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = *(const float4*)(As + (wr * (TM * 32) + i * 32 + r) * PITCH_ + (2 * ks + h) * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = *(const float4*)(Bs + (wc * (TN * 32) + j * 32 + r) * PITCH_ + (2 * ks + h) * 16);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                        }
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < FBK / 16; ++ks) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const char* q = As + (wr * (TM * 32) + i * 32 + r) * FPITCH + ks * 32 + h * 16;
                    const uint2 lo = *(const uint2*)q, hi = *(const uint2*)(q + 8);
                    a[i] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const char* q = Bs + (wc * (TN * 32) + j * 32 + r) * FPITCH + ks * 32 + h * 16;
                    const uint2 lo = *(const uint2*)q, hi = *(const uint2*)(q + 8);
                    b[j] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            }
            if (more) commit(k0 + FBK, lds + (cur ^ 1) * BUF);
            __syncthreads();
            cur ^= 1;
        }
    }

    float* Cb = g.C + (long)i0 * g.scb + (long)i1 * g.scb1 + (long)zs * fa.split_stride;
    if constexpr (FUSED) {
        // Wide-store epilogue with the fused pieces of the DGL BatchNorm2 backward (GemmFused, step_internal.h).  The RAW
        // alpha * A.B tile is staged; the column-block affine, the per-channel reductions against W and the BatchNorm-backward
        // transform run in the piece loop, where a thread holds 4 consecutive columns of one row.  A tile spans at most two
        // channels (period >= BN, checked on the host).
        constexpr int TP = BN + 4;
        static_assert(BM * TP * 4 <= 2 * BUF, "staging tile must fit in the operand buffers");
        __shared__ float fred[4][4];
        float* tile = (float*)lds;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cn = wc * (TN * 32) + j * 32 + r;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rm = wr * (TM * 32) + i * 32 + 4 * h + (e & 3) + 8 * (e >> 2);
                    tile[rm * TP + cn] = g.alpha * acc[i][j][e];
                }
            }
        __syncthreads();
        const GemmFused& fu = fa.fu;
        constexpr int PIECES = BM * BN / 4, NPC = PIECES / 256, NB = NPC < 8 ? NPC : 8;
        static_assert(PIECES % 256 == 0 && NPC % NB == 0, "piece loop shape");
        if (fu.flags & (GEMM_FUSED_FFN_FWD | GEMM_FUSED_MASKNZ | GEMM_FUSED_BF16OUT)) {
            // GEMM_FUSED_BF16OUT: v + bias stored as bf16.  The other two: bf16 hidden layer of the pre-training feed-forward block: a thread's piece = 4 consecutive columns of one row = one
            // Philox call of the step_pt_dropout stream (element index m * N + n, N % 4 == 0)
            uint16_t* Ch = (uint16_t*)Cb;
            const uint16_t* xh = (const uint16_t*)fu.maskx;
            const bool fwd = (fu.flags & GEMM_FUSED_FFN_FWD) != 0, plain = (fu.flags & GEMM_FUSED_BF16OUT) != 0;
            const float ks = fu.p > 0.f ? 1.f / (1.f - fu.p) : 1.f;
            const int c4t = (tid % (BN / 4)) * 4;                    // the same 4 columns for all of this thread's pieces
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if ((fwd || plain) && fu.ffn_bias && n0 + c4t < g.N) {
                const float4 t = *(const float4*)(fu.ffn_bias + n0 + c4t);
                b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
            }
            for (int pb = 0; pb < NPC; pb += NB) {
                uint2 x2[NB];
                long off[NB];
                bool ok[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int pc = tid + (pb + u) * 256;
                    const int gm = m0 + pc / (BN / 4), gn = n0 + (pc % (BN / 4)) * 4;
                    ok[u] = gm < g.M && gn < g.N;
                    off[u] = ok[u] ? (long)gm * g.ldc + gn : 0;
                    if (!fwd && !plain) x2[u] = *(const uint2*)(xh + off[u]);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (!ok[u]) continue;
                    const int pc = tid + (pb + u) * 256;
                    const float4 t4 = *(const float4*)(tile + (pc / (BN / 4)) * TP + (pc % (BN / 4)) * 4);
                    const float v[4] = {t4.x, t4.y, t4.z, t4.w};
                    float o[4];
                    if (plain) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = v[i] + b4[i];
                    } else if (fwd) {
                        float m[4] = {1.f, 1.f, 1.f, 1.f};
                        if (fu.p > 0.f) {
                            const long gm = m0 + pc / (BN / 4), gn = n0 + (pc % (BN / 4)) * 4;
                            const long blk = (gm * (long)g.N + gn) >> 2;
                            uint32_t rr[4];
                            philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), fu.site, 0xD20Fu, fu.seed_lo, fu.seed_hi, rr);
#pragma unroll
                            for (int i = 0; i < 4; ++i) m[i] = u32_to_unit(rr[i]) >= fu.p ? ks : 0.f;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = fmaxf(v[i] + b4[i], 0.f) * m[i];
                    } else {
                        const uint32_t w0 = x2[u].x, w1 = x2[u].y;
                        // (a stored value of +-0 means "closed": compare the magnitude bits)
                        o[0] = (w0 & 0x7fffu) ? v[0] * ks : 0.f;
                        o[1] = (w0 & 0x7fff0000u) ? v[1] * ks : 0.f;
                        o[2] = (w1 & 0x7fffu) ? v[2] * ks : 0.f;
                        o[3] = (w1 & 0x7fff0000u) ? v[3] * ks : 0.f;
                    }
                    *(uint2*)(Ch + off[u]) = pack4(o[0], o[1], o[2], o[3]);
                }
            }
            return;
        }
        if (fu.flags & GEMM_FUSED_INTERLEAVED) {
            // channels-last bf16 activations (dgl_conv_mfma.hip): column n belongs to channel n % C, the result and the BatchNorm input
            // are bf16.  Stored: x > 0 ? v - kc (m1 + (x - mean) rstd m2) : 0 with v = alpha A.B (the BatchNorm scale kc is already in B).
            // A thread's 4 columns (and channels) are the same for all of its pieces: the coefficients live in registers.
            // A thread's piece: EIGHT consecutive columns of one row (16 bytes of bf16 in, 16 bytes out; host-checked: channels % 8 == 0).
            // All pieces of a thread are requested before the first store: 128 bytes in flight per thread -- with 4-column pieces
            // (8-byte loads, round 4) the launch moved 2.5-2.8 TB/s, Little's law on 64 bytes per thread.
            const int C = fu.channels, cb = (n0 + (tid % (BN / 8)) * 8) % C;
            float km1[8], km2[8], mu8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cb + i;
                km1[i] = fu.bncoef[2 * C + c] * fu.bncoef[c];
                km2[i] = fu.bncoef[2 * C + c] * fu.bncoef[C + c] * fu.bnstat[3 * C + c];
                mu8[i] = fu.bnstat[2 * C + c];
            }
            uint16_t* Ch = (uint16_t*)Cb;
            const uint16_t* xh = (const uint16_t*)fu.bnx;
            constexpr int PIECES8 = BM * BN / 8, NPC8 = PIECES8 / 256, NB8 = NPC8 < 8 ? NPC8 : 8;
            static_assert(PIECES8 % 256 == 0 && NPC8 % NB8 == 0, "piece loop shape (8-column pieces)");
            for (int pb = 0; pb < NPC8; pb += NB8) {
                uint4 x4[NB8];
                long off[NB8];
                bool ok[NB8];
#pragma unroll
                for (int u = 0; u < NB8; ++u) {
                    const int pc = tid + (pb + u) * 256;
                    const int gm = m0 + pc / (BN / 8), gn = n0 + (pc % (BN / 8)) * 8;
                    ok[u] = gm < g.M && gn < g.N;                       // (N % 8 == 0: a piece is inside or outside as a whole)
                    off[u] = ok[u] ? (long)gm * g.ldc + gn : 0;
                    x4[u] = *(const uint4*)(xh + off[u]);
                }
#pragma unroll
                for (int u = 0; u < NB8; ++u) {
                    if (!ok[u]) continue;
                    const int pc = tid + (pb + u) * 256;
                    const float* tp = tile + (pc / (BN / 8)) * TP + (pc % (BN / 8)) * 8;
                    const float4 t0 = *(const float4*)tp, t1 = *(const float4*)(tp + 4);
                    const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                    const uint32_t xw[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float x = __uint_as_float((i & 1) ? (xw[i >> 1] & 0xffff0000u) : (xw[i >> 1] << 16));
                        o[i] = x > 0.f ? v[i] - km1[i] - (x - mu8[i]) * km2[i] : 0.f;
                    }
                    const uint2 lo = pack4(o[0], o[1], o[2], o[3]), hi = pack4(o[4], o[5], o[6], o[7]);
                    *(uint4*)(Ch + off[u]) = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            return;
        }
        const int c_lo = n0 / fu.period, nb = (c_lo + 1) * fu.period, C = fu.channels;
        float cs[2] = {1.f, 1.f}, csh[2] = {0.f, 0.f}, kk[2], m1[2], m2r[2], mu[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c_lo + j < C ? c_lo + j : C - 1;
            if (g.c_nscale) { cs[j] = g.c_nscale[c]; csh[j] = g.c_nshift[c]; }
            if (fu.bnx) { m1[j] = fu.bncoef[c]; m2r[j] = fu.bncoef[C + c] * fu.bnstat[3 * C + c]; kk[j] = fu.bncoef[2 * C + c]; mu[j] = fu.bnstat[2 * C + c]; }
        }
        float dw[2] = {0.f, 0.f}, dm[2] = {0.f, 0.f};
        // batches of NB pieces per thread: all global reads of a batch (W / x / old C) are issued before the first store, so the
        // epilogue -- which is most of the d_a2 GEMM (K = 100) -- streams instead of paying one memory latency per piece
        for (int pb = 0; pb < NPC; pb += NB) {
            float4 w4[NB], x4[NB], o4[NB];
            long off[NB];
            bool ok[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int pc = tid + (pb + u) * 256;
                const int rm = pc / (BN / 4), c4 = (pc % (BN / 4)) * 4;
                const int gm = m0 + rm, gn = n0 + c4;
                ok[u] = gm < g.M && gn < g.N;
                off[u] = ok[u] ? (long)gm * g.ldc + gn : 0;
                if (fu.dotw) w4[u] = *(const float4*)(fu.dotw + off[u]);
                if (fu.bnx) x4[u] = *(const float4*)(fu.bnx + off[u]);
                if (g.accumulate == 1) o4[u] = *(const float4*)(Cb + off[u]);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (!ok[u]) continue;
                const int pc = tid + (pb + u) * 256;
                const int rm = pc / (BN / 4), c4 = (pc % (BN / 4)) * 4;
                const int gm = m0 + rm, gn = n0 + c4;
                const float4 t4 = *(const float4*)(tile + rm * TP + c4);
                float v[4] = {t4.x, t4.y, t4.z, t4.w};
                const float mv = g.c_mvec ? g.c_mvec[gm] : 0.f;
                if (fu.dotw) {
                    const float w[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int j = gn + i >= nb; dw[j] += w[i] * v[i]; dm[j] += w[i] * mv; }
                }
                if (g.c_nscale) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int j = gn + i >= nb; v[i] = v[i] * cs[j] + csh[j] * mv; }
                }
                if (fu.bnx) {
                    const float x[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int j = gn + i >= nb;
                        v[i] = x[i] > 0.f ? kk[j] * (v[i] - m1[j] - (x[i] - mu[j]) * m2r[j]) : 0.f;
                    }
                }
                float4 r4 = make_float4(v[0], v[1], v[2], v[3]);
                if (g.accumulate == 1) { r4.x += o4[u].x; r4.y += o4[u].y; r4.z += o4[u].z; r4.w += o4[u].w; }
                *(float4*)(Cb + off[u]) = r4;
            }
        }
        if (fu.dotw) {
            float q4[4] = {dw[0], dm[0], dw[1], dm[1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) q4[i] = wave_sum(q4[i]);
            if (lane == 0) { fred[wave][0] = q4[0]; fred[wave][1] = q4[1]; fred[wave][2] = q4[2]; fred[wave][3] = q4[3]; }
            __syncthreads();
            if (tid < 4) {
                const int c = c_lo + (tid >> 1);
                if (c < C) atomicAdd(fu.dots + 2 * c + (tid & 1), fred[0][tid] + fred[1][tid] + fred[2][tid] + fred[3][tid]);
            }
        }
        return;
    }
    if constexpr (BM * (BN + 4) * 4 <= 2 * BUF)
    if (fa.wide_store) {
        // Store / read-modify-write epilogue through LDS: the accumulator layout has 32 consecutive columns per wave
        // instruction (128-byte runs); staged as a [BM][BN] f32 tile, every thread instead moves 16-byte pieces of whole rows
        // (512-byte runs per wave instruction).  Used for the outputs that matter: the 267 MB d_a2 and the 87 MB fc gradient.
        constexpr int TP = BN + 4;                          // padded row of the staging tile (floats)
        static_assert(BM * TP * 4 <= 2 * BUF, "staging tile must fit in the operand buffers");
        float* tile = (float*)lds;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cn = wc * (TN * 32) + j * 32 + r, gn = n0 + cn;
                const GemmColAffine ca = gemm_col_affine(g, gn < g.N ? gn : 0);
                const float bv = (g.bias != nullptr && gn < g.N) ? g.bias[gn] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rm = wr * (TM * 32) + i * 32 + 4 * h + (e & 3) + 8 * (e >> 2), gm = m0 + rm;
                    float v = g.alpha * acc[i][j][e] * ca.cs + bv;
                    if (ca.mvec) v += ca.csh * (gm < g.M ? ca.mvec[gm] : 0.f);
                    tile[rm * TP + cn] = v;
                }
            }
        __syncthreads();
        constexpr int PIECES = BM * BN / 4;
#pragma unroll 4
        for (int pc = tid; pc < PIECES; pc += 256) {
            const int rm = pc / (BN / 4), c4 = (pc % (BN / 4)) * 4;
            const int gm = m0 + rm, gn = n0 + c4;
            if (gm >= g.M || gn >= g.N) continue;                       // N % 4 == 0: a piece is inside or outside as a whole
            float4 v = *(const float4*)(tile + rm * TP + c4);
            float4* dst = (float4*)(Cb + (long)gm * g.ldc + gn);
            if (g.accumulate == 1) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *dst = v;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + wc * (TN * 32) + j * 32 + r;
            const int mb = m0 + wr * (TM * 32) + i * 32 + 4 * h;
            if (gn == ones_row) gemm_store_tile(acc[i][j], g.a_rowsum, mb, g.M, 1, g.alpha, 2, 0.f, 0);
            if (gn >= g.N) continue;
            const float bv = (g.bias != nullptr) ? g.bias[gn] : 0.f;
            const long ni = g.c_nblk ? (long)(gn / g.c_nblk) * g.c_nstride + (gn % g.c_nblk) : (long)gn;
            gemm_store_tile(acc[i][j], Cb + ni * g.scn, mb, g.M, g.ldc, g.alpha, g.accumulate, bv, g.relu, gemm_col_affine(g, gn));
        }
}

static int ilog2_exact(int v) {        // log2 for powers of two >= 4, -2 otherwise
    for (int l = 2; l < 31; ++l) if ((1 << l) == v) return l;
    return -2;
}

// operand mode for the fast path, or -1 when the operand does not qualify
static int fast_mode(const void* base, int is_bf16, long srow, long sk, long sb0, long sb1, int kblk, long kstride, int rblk,
                     long rstride, int* klog, int* rlog) {
    const int vec = is_bf16 ? 8 : 4;
    if (((uintptr_t)base & 15) || sb0 % vec || sb1 % vec) return -1;
    *klog = 30; *rlog = 30;
    if (sk == 1) {                                   // k contiguous
        if (srow % vec) return -1;
        if (kblk) { *klog = ilog2_exact(kblk); if (*klog < 0 || kblk % vec || kstride % vec) return -1; }
        if (rblk) { *rlog = ilog2_exact(rblk); if (*rlog < 0) return -1; }
        return is_bf16 ? KC_BF16 : KC_F32;
    }
    if (srow == 1 && is_bf16) {                      // n contiguous bf16 (B operand of the 128-wide tiles only, no remaps)
        if (sk % 8 || kblk || rblk) return -1;
        return MC_BF16;
    }
    if (srow == 1 && !is_bf16) {                     // m / n contiguous
        if (sk % 4) return -1;
        if (kblk) { *klog = ilog2_exact(kblk); if (*klog < 0) return -1; }
        if (rblk) { *rlog = ilog2_exact(rblk); if (*rlog < 0 || rstride % 4) return -1; }
        return MC_F32;
    }
    return -1;
}

template <int BM, int BN, int AMODE>
int launch_fast_b(const StepGemm& g, const FastArgs& fa, int bmode, dim3 grid, hipStream_t st) {
    if (bmode == KC_F32) gemm_fast_kernel<BM, BN, AMODE, KC_F32, false><<<grid, 256, 0, st>>>(g, fa);
    else if (bmode == KC_BF16) gemm_fast_kernel<BM, BN, AMODE, KC_BF16, false><<<grid, 256, 0, st>>>(g, fa);
    else if (bmode == MC_F32) gemm_fast_kernel<BM, BN, AMODE, MC_F32, false><<<grid, 256, 0, st>>>(g, fa);
    else {
        if constexpr (AMODE != KC_BF16) gemm_fast_kernel<BM, BN, AMODE, MC_BF16, false><<<grid, 256, 0, st>>>(g, fa);
        else { step_set_error("step_gemm: an n-contiguous bf16 B operand needs an f32 A operand"); return STEP_ERR_ARG; }
    }
    STEP_LAUNCH_CHECK("step_gemm(bf16 fast)");
    return STEP_OK;
}
template <int BM, int BN>
int launch_fast(const StepGemm& g, const FastArgs& fa_in, int amode, int bmode, hipStream_t st) {
    dim3 grid(cdiv(g.N + (g.a_rowsum ? 1 : 0), BN), cdiv(g.M, BM), g.batch * g.splitk);
    FastArgs fa = fa_in;
    // worth it when A is re-read by several n tiles and is too big for one L2 (the adjacency of a large graph, the Gram operand)
    fa.xcd_swizzle = grid.x >= 2 && (grid.y * grid.z) % 8 == 0 && (long)g.M * g.K * g.batch >= (8L << 20);
    if (amode == KC_F32) return launch_fast_b<BM, BN, KC_F32>(g, fa, bmode, grid, st);
    if (amode == KC_BF16) return launch_fast_b<BM, BN, KC_BF16>(g, fa, bmode, grid, st);
    return launch_fast_b<BM, BN, MC_F32>(g, fa, bmode, grid, st);
}
// the fused DGL epilogues: only the operand combinations dgl.hip uses
template <int BM, int BN>
int launch_fast_fused(const StepGemm& g, const FastArgs& fa_in, int amode, int bmode, hipStream_t st) {
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), 1);
    FastArgs fa = fa_in;
    static const bool m_fast_on = []() { const char* e = getenv("STEP_GEMM_M_FAST"); return !e || atoi(e) != 0; }();      // (A/B knob)
    fa.m_fast = m_fast_on && grid.y >= 2 && grid.y <= 64 && (long)g.N * g.K * (g.b_bf16 ? 2 : 4) >= (16L << 20);
    if (amode == KC_F32 && bmode == MC_BF16) gemm_fast_kernel<BM, BN, KC_F32, MC_BF16, false, false, true><<<grid, 256, 0, st>>>(g, fa);
    else if (amode == KC_F32 && bmode == KC_F32) gemm_fast_kernel<BM, BN, KC_F32, KC_F32, false, false, true><<<grid, 256, 0, st>>>(g, fa);
    else if (amode == KC_F32 && bmode == MC_F32) gemm_fast_kernel<BM, BN, KC_F32, MC_F32, false, false, true><<<grid, 256, 0, st>>>(g, fa);
    else if (amode == MC_F32 && bmode == MC_F32) gemm_fast_kernel<BM, BN, MC_F32, MC_F32, false, false, true><<<grid, 256, 0, st>>>(g, fa);
    else { step_set_error("step_gemm(fused): operand layout not instantiated"); return STEP_ERR_ARG; }
    STEP_LAUNCH_CHECK("step_gemm(bf16 fused)");
    return STEP_OK;
}
int launch_fast_f32_fused(const StepGemm& g, const FastArgs& fa, int amode, int bmode, hipStream_t st) {
    dim3 grid(cdiv(g.N, 64), cdiv(g.M, 64), 1);
    if (amode == KC_F32 && bmode == MC_F32) gemm_fast_kernel<64, 64, KC_F32, MC_F32, true, false, true><<<grid, 256, 0, st>>>(g, fa);
    else if (amode == MC_F32 && bmode == MC_F32) gemm_fast_kernel<64, 64, MC_F32, MC_F32, true, false, true><<<grid, 256, 0, st>>>(g, fa);
    else { step_set_error("step_gemm(fused): operand layout not instantiated"); return STEP_ERR_ARG; }
    STEP_LAUNCH_CHECK("step_gemm(f32 fused)");
    return STEP_OK;
}
int launch_fast_f32(const StepGemm& g, const FastArgs& fa, int amode, int bmode, hipStream_t st) {
    dim3 grid(cdiv(g.N + (g.a_rowsum ? 1 : 0), 64), cdiv(g.M, 64), g.batch * g.splitk);
    if (amode == KC_F32 && bmode == KC_F32) gemm_fast_kernel<64, 64, KC_F32, KC_F32, true><<<grid, 256, 0, st>>>(g, fa);
    else if (amode == KC_F32) gemm_fast_kernel<64, 64, KC_F32, MC_F32, true><<<grid, 256, 0, st>>>(g, fa);
    else if (bmode == KC_F32) gemm_fast_kernel<64, 64, MC_F32, KC_F32, true><<<grid, 256, 0, st>>>(g, fa);
    else gemm_fast_kernel<64, 64, MC_F32, MC_F32, true><<<grid, 256, 0, st>>>(g, fa);
    STEP_LAUNCH_CHECK("step_gemm(f32 fast)");
    return STEP_OK;
}

template <int BM, int BN>
int launch_bf16(const StepGemm& g, hipStream_t st) {
    dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), g.batch * g.splitk);
    if (g.a_bf16 && g.b_bf16) gemm_bf16mfma_kernel<BM, BN, uint16_t, uint16_t><<<grid, 256, 0, st>>>(g);
    else if (g.a_bf16) gemm_bf16mfma_kernel<BM, BN, uint16_t, float><<<grid, 256, 0, st>>>(g);
    else if (g.b_bf16) gemm_bf16mfma_kernel<BM, BN, float, uint16_t><<<grid, 256, 0, st>>>(g);
    else gemm_bf16mfma_kernel<BM, BN, float, float><<<grid, 256, 0, st>>>(g);
    STEP_LAUNCH_CHECK("step_gemm(bf16)");
    return STEP_OK;
}

}  // namespace

namespace {
// second launch of a split-K product that went through a workspace: C(b, m, n) += sum over `chunk` splits of ws[z][b][m][n]
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int chunk, long mn, int N, int batch,
                                                            int batch0, float* __restrict__ C, long ldc, long scn, long scb, long scb1) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x, total = mn * batch;
    if (e >= total) return;
    const int z0 = blockIdx.y * chunk, z1 = z0 + chunk < splits ? z0 + chunk : splits;
    float s = 0.f;
#pragma unroll 8
    for (int z = z0; z < z1; ++z) s += ws[(long)z * total + e];
    const long b = e / mn, r = e - b * mn;
    const long m = r / N, n = r - m * N;
    const long i0 = batch0 ? b % batch0 : b, i1 = batch0 ? b / batch0 : 0;
    atomicAdd(C + i0 * scb + i1 * scb1 + m * ldc + n * scn, s);
}
}  // namespace

static int auto_splitk(int M, int N, int K, int batch, bool fast) {
    const int bk = fast ? FBK : BK;
    long tiles = fast ? (long)cdiv(M, 128) * cdiv(N, 128) * batch : (long)cdiv(M, 64) * cdiv(N, 64) * batch;
    static const int split_target = []() { const char* e = getenv("STEP_GEMM_SPLIT_TARGET"); return e ? atoi(e) : 768; }();      // (A/B knob)
    long want = ((fast ? split_target : 1024) + tiles - 1) / tiles;
    long maxs = cdiv(K, bk) / 4;
    return (int)(want < 1 ? 1 : (want > maxs ? (maxs < 1 ? 1 : maxs) : want));
}
// the number of splits step_gemm chooses for splitk = -1 on the staged path (callers size StepGemm.splitk_ws with it)
int step_gemm_auto_splitk(int M, int N, int K, int batch) { return auto_splitk(M, N, K, batch, true); }

int step_gemm_bf16_launch(StepGemm g, hipStream_t st, const GemmFused* fused) {
    FastArgs fa;
    memset(&fa, 0, sizeof(fa));
    if (fused) fa.fu = *fused;
    int dummy;
    const int amode = fast_mode(g.A, g.a_bf16, g.sam, g.sak, g.sab, g.sab1, g.a_kblk, g.a_kstride, 0, 0, &fa.a_klog, &dummy);
    const int bmode = fast_mode(g.B, g.b_bf16, g.sbn, g.sbk, g.sbb, g.sbb1, g.b_kblk, g.b_kstride, g.b_nblk, g.b_nstride, &fa.b_klog,
                                &fa.b_nlog);
    const bool fast = amode >= 0 && amode != MC_BF16 && bmode >= 0 && !(g.a_kscale && (amode != KC_F32 || g.a_kperiod < FBK)) &&
                !(g.a_rowsum && (bmode == KC_BF16 || bmode == MC_BF16)) && !(bmode == MC_BF16 && g.N % 8);
    fa.wide_store = wide_store_ok(g, fused != nullptr);
    if (fused) STEP_REQUIRE(fast && fa.wide_store && !g.bias && !g.relu && g.batch == 1 && g.splitk <= 1 &&
                            ((fused->flags & (GEMM_FUSED_FFN_FWD | GEMM_FUSED_MASKNZ | GEMM_FUSED_BF16OUT)) ? (g.accumulate == 0 && !fused->dotw && !fused->bnx && !g.c_nscale) :
                             (fused->flags & GEMM_FUSED_INTERLEAVED) ? (128 % fused->channels == 0 && fused->channels % 8 == 0 && g.N % 8 == 0 && g.ldc % 8 == 0 &&
                                                                       ((((uintptr_t)fused->bnx) | ((uintptr_t)g.C)) & 15) == 0 && !fused->dotw) : fused->period >= 128),
                            "step_gemm: the fused DGL epilogues need the staged path with a wide-store result (aligned dense C, period >= 128)");
    if (g.splitk < 0) {
        STEP_REQUIRE(g.accumulate == 2, "step_gemm: automatic split-K needs accumulate==2");
        g.splitk = auto_splitk(g.M, g.N, g.K, g.batch, fast);
    }
    if (g.splitk < 1) g.splitk = 1;
    if (g.splitk > 1) STEP_REQUIRE(g.accumulate == 2, "step_gemm: split-K needs accumulate==2");
    STEP_REQUIRE(!(g.accumulate == 2 && (g.bias || g.relu)), "step_gemm: bias/relu epilogue not available with atomic accumulate");
    long tiles128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * g.batch * g.splitk;
    const bool big = g.M > 64 && g.N > 64 && tiles128 >= 512;
    // split-K through the caller's workspace: partial tiles stored, summed into C by splitk_reduce_kernel
    static const bool use_ws = []() { const char* e = getenv("STEP_GEMM_SPLITK_WS"); return !e || atoi(e) != 0; }();      // (A/B knob)
    const StepGemm orig = g;
    bool ws_reduce = false;
    if (use_ws && fast && !fused && g.splitk > 1 && g.splitk_ws && !g.c_nblk && !g.c_nscale && (((uintptr_t)g.splitk_ws) & 15) == 0 &&
        (long)g.M * g.N * g.batch * g.splitk <= g.splitk_ws_floats) {
        const long mn = (long)g.M * g.N;
        ws_reduce = true;
        fa.split_stride = mn * g.batch;
        g.C = g.splitk_ws; g.ldc = g.N; g.scn = 1; g.accumulate = 0;
        g.scb = mn; g.scb1 = (long)g.batch0 * mn;
        fa.wide_store = wide_store_ok(g, true);
    }
    auto finish = [&](int rc) -> int {
        if (rc != STEP_OK || !ws_reduce) return rc;
        const long total = (long)orig.M * orig.N * orig.batch;
        const int chunk = 16;
        dim3 grid((unsigned)((total + 255) / 256), (unsigned)cdiv(orig.splitk, chunk));
        splitk_reduce_kernel<<<grid, 256, 0, st>>>(orig.splitk_ws, orig.splitk, chunk, (long)orig.M * orig.N, orig.N, orig.batch, orig.batch0, orig.C,
                                                  orig.ldc, orig.scn ? orig.scn : 1, orig.scb, orig.scb1);
        STEP_LAUNCH_CHECK("step_gemm(split-K reduce)");
        return STEP_OK;
    };
    if (fast && fused) return big ? launch_fast_fused<128, 128>(g, fa, amode, bmode, st) : launch_fast_fused<64, 64>(g, fa, amode, bmode, st);
    // both operands k-contiguous bf16 on a large graph (the diffusion hops with their transposed source copy): 128 x 128 tiles when they fill
    // the chip -- the support is then re-read by 3 column tiles instead of 6 (STEP_GEMM_HOP128=0: the 128 x 64 rule below, A/B measurements)
    static const bool hop128 = []() { const char* e = getenv("STEP_GEMM_HOP128"); return !e || atoi(e) != 0; }();
    if (hop128 && fast && !big && amode == KC_BF16 && bmode == KC_BF16 && g.M >= 2048 && g.N >= 256 && tiles128 >= 256)
        return finish(launch_fast<128, 128>(g, fa, amode, bmode, st));
    static const int tall_m = []() { const char* e = getenv("STEP_GEMM_TALL_M"); return e ? atoi(e) : 1024; }();      // (A/B knob)
    if (fast && !big && bmode != MC_BF16 && g.M >= tall_m && g.N > 64) {
        // tall products with a short n axis (the diffusion hops of large graphs: 4096 x 32 T x 4096): 128 x 64 tiles -- three workgroups per
        // compute unit and half the A-operand LDS traffic of 64 x 64 -- when they still fill the chip
        const long t = (long)cdiv(g.M, 128) * cdiv(g.N, 64) * g.batch * g.splitk;
        if (t >= 256) return finish(launch_fast<128, 64>(g, fa, amode, bmode, st));
    }
    if (fast) return finish(big ? launch_fast<128, 128>(g, fa, amode, bmode, st) : launch_fast<64, 64>(g, fa, amode, bmode, st));
    STEP_TRY(step_gemm_rowsum_separate(&g, st));
    if (big) return launch_bf16<128, 128>(g, st);
    return launch_bf16<64, 64>(g, st);
}

// Exact-f32 GEMM through the same staged pipeline (64 x 64 tiles).  Returns -1 when the operands do not qualify
// (alignment / layout), in which case the caller falls back to the general kernels of gemm.hip.
// C[i1][i0] (M x N, row-major ldc, batch strides scb1 / scb) = sum over the K = 32 * nseg segmented k values (see stage_load_seg);
// g.A / g.B are the base pointers the table offsets refer to, g.batch0 = number of inner batch indices i0, g.batch = total.
int step_gemm_segmented_launch(StepGemm g, const GemmKSeg* ktab, int ktab_per, hipStream_t st) {
    STEP_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.K % 32 == 0 && g.K / 32 <= GEMM_KSEG_MAX && ktab && g.A && g.B && g.C && g.batch >= 1,
                 "step_gemm(segmented): bad descriptor");
    STEP_REQUIRE(g.accumulate != 2 && !g.bias && !g.relu && !g.a_rowsum && !g.a_kscale && g.scn <= 1 && !g.c_nblk,
                 "step_gemm(segmented): plain store / += only");
    STEP_REQUIRE((((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0, "step_gemm(segmented): 16-byte aligned bases");
    g.scn = 1; g.splitk = 1;
    if (g.batch0 == 0) g.batch0 = g.batch;
    FastArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.ktab = ktab; fa.ktab_per = ktab_per;
    fa.a_klog = fa.b_klog = fa.b_nlog = 30;
    fa.wide_store = 0;
    const long tiles128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * g.batch;
    if (g.M > 64 && g.N > 64 && tiles128 >= 256) {
        dim3 grid(cdiv(g.N, 128), cdiv(g.M, 128), g.batch);
        if (g.compute_bf16) gemm_fast_kernel<128, 128, KC_F32, KC_F32, false, true><<<grid, 256, 0, st>>>(g, fa);
        else { dim3 g64(cdiv(g.N, 64), cdiv(g.M, 64), g.batch); gemm_fast_kernel<64, 64, KC_F32, KC_F32, true, true><<<g64, 256, 0, st>>>(g, fa); }
    } else {
        dim3 grid(cdiv(g.N, 64), cdiv(g.M, 64), g.batch);
        if (g.compute_bf16) gemm_fast_kernel<64, 64, KC_F32, KC_F32, false, true><<<grid, 256, 0, st>>>(g, fa);
        else gemm_fast_kernel<64, 64, KC_F32, KC_F32, true, true><<<grid, 256, 0, st>>>(g, fa);
    }
    STEP_LAUNCH_CHECK("step_gemm(segmented)");
    return STEP_OK;
}

int step_gemm_f32_fast_launch(StepGemm g, hipStream_t st, const GemmFused* fused) {
    if (g.a_bf16 || g.b_bf16) return -1;
    FastArgs fa;
    memset(&fa, 0, sizeof(fa));
    if (fused) fa.fu = *fused;
    int dummy;
    const int amode = fast_mode(g.A, 0, g.sam, g.sak, g.sab, g.sab1, g.a_kblk, g.a_kstride, 0, 0, &fa.a_klog, &dummy);
    const int bmode = fast_mode(g.B, 0, g.sbn, g.sbk, g.sbb, g.sbb1, g.b_kblk, g.b_kstride, g.b_nblk, g.b_nstride, &fa.b_klog, &fa.b_nlog);
    if (amode < 0 || bmode < 0 || amode == MC_BF16 || bmode == MC_BF16 || (g.a_kscale && (amode != KC_F32 || g.a_kperiod < FBK))) return -1;
    fa.wide_store = wide_store_ok(g, fused != nullptr);
    if (fused && !(fa.wide_store && !g.bias && !g.relu && g.batch == 1 && g.splitk <= 1 && fused->period >= 128)) return -1;
    if (g.splitk < 0) {
        STEP_REQUIRE(g.accumulate == 2, "step_gemm: automatic split-K needs accumulate==2");
        long tiles = (long)cdiv(g.M, 64) * cdiv(g.N, 64) * g.batch;
        long want = (512 + tiles - 1) / tiles;
        long maxs = cdiv(g.K, FBK) / 2;
        g.splitk = (int)(want < 1 ? 1 : (want > maxs ? (maxs < 1 ? 1 : maxs) : want));
    }
    if (g.splitk < 1) g.splitk = 1;
    if (g.splitk > 1) STEP_REQUIRE(g.accumulate == 2, "step_gemm: split-K needs accumulate==2");
    STEP_REQUIRE(!(g.accumulate == 2 && (g.bias || g.relu)), "step_gemm: bias/relu epilogue not available with atomic accumulate");
    if (fused) return launch_fast_f32_fused(g, fa, amode, bmode, st);
    return launch_fast_f32(g, fa, amode, bmode, st);
}
