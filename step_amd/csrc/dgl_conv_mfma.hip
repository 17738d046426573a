// bf16 matrix-core versions of the DiscreteGraphLearning conv2 stage (8 -> 16 channels, 10 taps, valid) and its two
// adjoints -- reference discrete_graph_learning.py:133-134 (conv2 + relu, bn1 folded into the read) and their autograd.
// Used when StepDglParams.gemm_bf16 is set; the exact-f32 VALU kernels of dgl.hip stay the parity path.
//
// All three are HBM-bound streams over the [N][C][T] activations (a1: 133 MB, a2 / d_a2: 267 MB at PEMS04):
//   forward   reads a1, writes a2                        400 MB
//   dgrad     reads dz2, writes d_a1                     400 MB
//   wgrad     reads dz2 and a1, writes 16x8x10 numbers   400 MB
// The arithmetic (10.7 GFLOP each) is mapped on v_mfma_f32_16x16x32_bf16 with the time axis as the n (forward, dgrad) or
// k (wgrad) dimension; the im2col operand is never materialised: the input window lives in LDS as [t][channels] bf16
// rows (one 16-byte row per time step) and the MFMA operand of lane (t, tap group) is one ds_read_b128 at row t + tap.
//
// Lane maps of v_mfma_f32_16x16x32_bf16 (guide "Fragment layout"): A[m = l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][n = l&15],
// D[m = 4(l>>4)+e][n = l&15].
#include "common.h"
#include "step_internal.h"

namespace {

constexpr int KW = 10, CI = 8, CO = 16, C1_ = 8;
constexpr int FW_TT = 1024;             // outputs per workgroup (forward, dgrad)

__device__ __forceinline__ uint32_t cvt2(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// ------------------------------------------------------------------------------------------------ forward
// a2[n][co][t] = relu(b[co] + sum_{ci,kk} w[co][ci][kk] * (a1[n][ci][t+kk] * sc[ci] + sh[ci])),  partial BN2 sums per block.
// k index of the contraction = kk*8 + ci (three 32-wide steps, taps 10 and 11 carry zero weights).
__global__ __launch_bounds__(256) void conv2_fwd_mfma_kernel(const float* __restrict__ a1, const float* __restrict__ w,
                                                             const float* __restrict__ b, const float* __restrict__ sc,
                                                             const float* __restrict__ sh, float* __restrict__ a2,
                                                             float* __restrict__ partial, int T1) {
    __shared__ uint4 xs[FW_TT + 16];              // [t][8 ci] bf16
    __shared__ float red[4][2 * CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
    const int T2 = T1 - (KW - 1), t0 = blockIdx.x * FW_TT;
    const int ln = lane & 15, q = lane >> 4;

    float s8[CI], h8[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) { s8[ci] = sc[ci]; h8[ci] = sh[ci]; }
    for (int tt = tid; tt < FW_TT + 16; tt += 256) {
        const int t = t0 + tt;
        float v[CI];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) v[ci] = t < T1 ? a1[((long)n * CI + ci) * T1 + t] : 0.f;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) v[ci] = t < T1 ? v[ci] * s8[ci] + h8[ci] : 0.f;
        xs[tt] = make_uint4(cvt2(v[0], v[1]), cvt2(v[2], v[3]), cvt2(v[4], v[5]), cvt2(v[6], v[7]));
    }
    bf16x8 wf[3];                                 // weights: lane (co = ln, tap = 4s + q), 8 input channels
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int kk = 4 * s + q;
        float v[CI];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) v[ci] = kk < KW ? w[(ln * CI + ci) * KW + kk] : 0.f;
        wf[s] = as_frag(make_uint4(cvt2(v[0], v[1]), cvt2(v[2], v[3]), cvt2(v[4], v[5]), cvt2(v[6], v[7])));
    }
    float bias[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[e] = b[4 * q + e];
    __syncthreads();

    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tile = wave; tile < FW_TT / 16; tile += 4) {
        const int tb = tile * 16;
        if (t0 + tb >= T2) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], as_frag(xs[tb + ln + 4 * s + q]), acc, 0, 0, 0);
        const int t = t0 + tb + ln;
        if (t < T2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = fmaxf(acc[e] + bias[e], 0.f);
                a2[((long)n * CO + 4 * q + e) * T2 + t] = v;
                s1[e] += v; s2[e] += v * v;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        if (ln == 0) { red[wave][4 * q + e] = s1[e]; red[wave][CO + 4 * q + e] = s2[e]; }
    }
    __syncthreads();
    if (tid < 2 * CO)
        partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * (2 * CO) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// ------------------------------------------------------------------------------------------------ dgrad
// d_a1[n][ci][t] = sum_{co,kk} w[co][ci][kk] * dz[n][co][t-kk]      (dz = 0 outside [0, T2))
// k index = kk*16 + co (five 32-wide steps); rows m = ci (8 of the 16 MFMA rows are used).
// bnx != nullptr: fused BatchNorm1 backward -- what is stored is dz1 = [x > 0] * coef[2C+c] * (d_a1 - coef[c] - (x - mean_c) * rstd_c * coef[C+c])
// with x = bnx (= a1, laid out like din), coef = [m1 | m2 | gamma*rstd], stat = [scale | shift | mean | rstd], C = 8.
__global__ __launch_bounds__(256) void conv2_dgrad_mfma_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                               float* __restrict__ din, int T1, const float* __restrict__ bnx,
                                                               const float* __restrict__ coef, const float* __restrict__ stat, int own) {
    __shared__ uint4 zs[FW_TT + 16][2];           // [t - (t0 - 9)][16 co] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
    const int T2 = T1 - (KW - 1), t0 = blockIdx.x * FW_TT;
    const int ln = lane & 15, q = lane >> 4;
    // the BatchNorm input of ALL of this wave's tiles is requested up front (16 tiles x 4 channels per lane): the loads overlap the
    // staging of dz and each other instead of costing one memory latency per tile
    constexpr int NT = FW_TT / 16 / 4;
    float xall[NT][4];
    if (bnx && q < 2) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = t0 + (wave + 4 * i) * 16 + ln;
#pragma unroll
            for (int e = 0; e < 4; ++e) xall[i][e] = t < T1 ? bnx[((long)n * CI + 4 * q + e) * T1 + t] : 0.f;
        }
    }
    for (int tt = tid; tt < FW_TT + 16; tt += 256) {
        const int t = t0 - (KW - 1) + tt;
        const bool ok = t >= 0 && t < T2;
        float v[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) v[co] = ok ? dz[((long)n * CO + co) * T2 + t] : 0.f;
        zs[tt][0] = make_uint4(cvt2(v[0], v[1]), cvt2(v[2], v[3]), cvt2(v[4], v[5]), cvt2(v[6], v[7]));
        zs[tt][1] = make_uint4(cvt2(v[8], v[9]), cvt2(v[10], v[11]), cvt2(v[12], v[13]), cvt2(v[14], v[15]));
    }
    bf16x8 wf[5];                                 // lane (ci = ln, tap = 2s + (q>>1), co = 8(q&1) + j)
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int kk = 2 * s + (q >> 1), c0 = 8 * (q & 1);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ln < CI ? w[((c0 + j) * CI + ln) * KW + kk] : 0.f;
        wf[s] = as_frag(make_uint4(cvt2(v[0], v[1]), cvt2(v[2], v[3]), cvt2(v[4], v[5]), cvt2(v[6], v[7])));
    }
    float bk[4] = {1.f, 1.f, 1.f, 1.f}, bm1[4] = {0.f, 0.f, 0.f, 0.f}, bmu[4] = {0.f, 0.f, 0.f, 0.f}, bm2r[4] = {0.f, 0.f, 0.f, 0.f};
    if (bnx && q < 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * q + e;
            bk[e] = coef[2 * CI + c]; bm1[e] = coef[c]; bmu[e] = stat[2 * CI + c]; bm2r[e] = stat[3 * CI + c] * coef[CI + c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tb = (wave + 4 * i) * 16;
        if (t0 + tb < T1) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int kk = 2 * s + (q >> 1);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], as_frag(zs[tb + ln + (KW - 1) - kk][q & 1]), acc, 0, 0, 0);
            }
            const int t = t0 + tb + ln;
            if (t < T1 && q < 2) {
                if (bnx) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = xall[i][e];
                        const float cst = t < own ? bm1[e] + (x - bmu[e]) * bm2r[e] : 0.f;
                        din[((long)n * CI + 4 * q + e) * T1 + t] = x > 0.f ? bk[e] * (acc[e] - cst) : 0.f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) din[((long)n * CI + 4 * q + e) * T1 + t] = acc[e];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ wgrad
// dw[co][ci][kk] += sum_{n,t} dz[n][co][t] * (a1[n][ci][t+kk] * sc[ci] + sh[ci]),   db[co] += sum_{n,t} dz[n][co][t]
// Contraction over time: m = co, n = kk*8 + ci (five 16-wide tiles), k = 32 time steps per MFMA.  The shifted input rows are
// read with 4-byte LDS loads; a second copy of the window shifted by one element serves the odd taps.
constexpr int WG_SC = 256;               // time steps staged per pass
constexpr int WG_PASSES = 14;            // passes per workgroup (3584 steps)
constexpr int WG_OUT = CO * CI * KW + CO;

// xhat != 0: the input is read NORMALISED (sc / sh then point at BatchNorm1's [scale | shift | mean | rstd] block and
// (a1 - mean) * rstd is used): the result is G'[co][ci][kk] = sum dz * xhat, from which conv2_wgrad_finish_kernel forms both the
// weight gradient gamma * G' + beta * db and BatchNorm1's backward sums.
__global__ __launch_bounds__(256) void conv2_wgrad_mfma_kernel(const float* __restrict__ dz, const float* __restrict__ a1,
                                                               const float* __restrict__ sc, const float* __restrict__ sh,
                                                               float* __restrict__ partial, int T1, int xhat) {
    constexpr int ZP = WG_SC + 8;        // bf16 elements per row (16-byte multiple)
    constexpr int XP = WG_SC + 16;
    __shared__ __attribute__((aligned(16))) uint16_t zs[CO][ZP];
    __shared__ __attribute__((aligned(16))) uint16_t x0[CI][XP], x1[CI][XP];        // x1[c][i] = x0[c][i + 1]
    __shared__ float red[3][5][64][4];
    __shared__ float redb[4][CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
    const int T2 = T1 - (KW - 1), tbeg = blockIdx.x * (WG_SC * WG_PASSES);
    const int ln = lane & 15, q = lane >> 4;
    float s8[CI], h8[CI];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
        if (xhat) { s8[ci] = sc[3 * CI + ci]; h8[ci] = -sc[2 * CI + ci] * sc[3 * CI + ci]; }
        else { s8[ci] = sc[ci]; h8[ci] = sh[ci]; }
    }

    f32x4 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) bsum[co] = 0.f;

    // software pipeline: the global loads of pass p+1 are in flight while pass p is converted, staged and multiplied
    float v[CO], xa[CI], xb[CI];
    auto fetch = [&](int tp) {           // dz (zero past T2) and the input window [tp, tp + 256 + 10) of this thread
        const int t = tp + tid;
#pragma unroll
        for (int co = 0; co < CO; ++co) v[co] = t < T2 ? dz[((long)n * CO + co) * T2 + t] : 0.f;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) xa[ci] = t < T1 ? a1[((long)n * CI + ci) * T1 + t] : 0.f;
        const int t2 = tp + 256 + tid;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) xb[ci] = (tid < 10 && t2 < T1) ? a1[((long)n * CI + ci) * T1 + t2] : 0.f;
    };
    fetch(tbeg);
    for (int pass = 0; pass < WG_PASSES; ++pass) {
        const int tp = tbeg + pass * WG_SC;
        if (tp >= T2) break;
        __syncthreads();
        {
            const int t = tp + tid;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                bsum[co] += v[co];
                zs[co][tid] = (uint16_t)(cvt2(v[co], 0.f) & 0xffffu);
            }
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const uint16_t hb = (uint16_t)(cvt2(t < T1 ? xa[ci] * s8[ci] + h8[ci] : 0.f, 0.f) & 0xffffu);
                x0[ci][tid] = hb;
                if (tid > 0) x1[ci][tid - 1] = hb;
            }
            if (tid < 10) {
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const uint16_t hb = (uint16_t)(cvt2(t + 256 < T1 ? xb[ci] * s8[ci] + h8[ci] : 0.f, 0.f) & 0xffffu);
                    x0[ci][tid + 256] = hb;
                    x1[ci][tid + 255] = hb;
                }
            }
        }
        __syncthreads();
        if (pass + 1 < WG_PASSES && tp + WG_SC < T2) fetch(tp + WG_SC);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tb = 64 * wave + 32 * u + 8 * q;          // first of this lane's 8 time steps
            const bf16x8 a = as_frag(*(const uint4*)&zs[ln][tb]);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int kk = 2 * j + (ln >> 3), ci = ln & 7;
                const uint16_t* row = (kk & 1) ? &x1[ci][0] : &x0[ci][0];
                const uint32_t* p = (const uint32_t*)(row + tb + (kk & ~1));
                const bf16x8 bfr = as_frag(make_uint4(p[0], p[1], p[2], p[3]));
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr, acc[j], 0, 0, 0);
            }
        }
    }
    // combine the four waves, then one row of partial results per workgroup: [co][ci][kk] then db[co]
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave - 1][j][lane][e] = acc[j][e];
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float s = wave_sum(bsum[co]);
        if (lane == 0) redb[wave][co] = s;
    }
    __syncthreads();
    float* out = partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * WG_OUT;
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int kk = 2 * j + (ln >> 3), ci = ln & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 4 * q + e;
                out[(co * CI + ci) * KW + kk] = acc[j][e] + red[0][j][lane][e] + red[1][j][lane][e] + red[2][j][lane][e];
            }
        }
    }
    if (tid < CO) out[CO * CI * KW + tid] = redb[0][tid] + redb[1][tid] + redb[2][tid] + redb[3][tid];
}

// dw / db += column sums of the per-workgroup partial rows (grid.y row slices, one atomic per slice and output)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int nrows, int nout, int nw,
                                                                float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[4][64];
    const int o = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (o < nout)
        for (int r = blockIdx.y * 4 + w; r < nrows; r += gridDim.y * 4) s += partial[(long)r * nout + o];
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && o < nout) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(o < nw ? &dw[o] : &db[o - nw], v);
    }
}

// Fused BatchNorm1 backward, coefficient step (one block).  graw = this step's sums over nodes and time
// G'[co][ci][kk] = sum dz2 * xhat1 (1280 numbers) followed by db2[co] = sum dz2 (16).  With y1 = gamma xhat1 + beta the input of conv2:
//     d conv2_w = gamma[ci] G' + beta[ci] db2[co],  d conv2_b = db2,
//     S1[ci] = sum d_y1 = sum_{co,kk} w[co][ci][kk] db2[co],   S2[ci] = sum d_y1 xhat1 = sum_{co,kk} w[co][ci][kk] G'[co][ci][kk]
// (exact: d_y1 is the full correlation of dz2 with w, every dz2 element meets every tap inside the valid range), hence
// BatchNorm1's dbeta += S1, dgamma += S2 and coef = [S1/count | S2/count | gamma*rstd] without a pass over d_a1 and a1.
// raw != 0: graw holds Graw = sum dz2 * a1 on the un-normalised activations (channels-last bf16 path): G' = rstd (Graw - mean db2).
__global__ __launch_bounds__(256) void conv2_wgrad_finish_kernel(const float* __restrict__ graw, const float* __restrict__ w,
                                                                 const float* __restrict__ stat, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, double count, float* __restrict__ dw,
                                                                 float* __restrict__ db, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ coef, int raw) {
    __shared__ float r1[CI][32], r2[CI][32];
    const int tid = threadIdx.x;
    const int ci = tid >> 5, sub = tid & 31;                 // 8 channels x 32 threads, 5 (co, kk) pairs each
    float s1 = 0.f, s2 = 0.f;
    for (int i = sub; i < CO * KW; i += 32) {
        const int co = i / KW, kk = i % KW;
        const int idx = (co * CI + ci) * KW + kk;
        const float d = graw[CO * CI * KW + co], wv = w[idx];
        const float g = raw ? stat[3 * CI + ci] * (graw[idx] - stat[2 * CI + ci] * d) : graw[idx];
        dw[idx] += gamma[ci] * g + beta[ci] * d;
        s1 += wv * d;
        s2 += wv * g;
    }
    r1[ci][sub] = s1; r2[ci][sub] = s2;
    if (tid < CO) db[tid] += graw[CO * CI * KW + tid];
    __syncthreads();
    if (tid < CI) {
        float a = 0.f, b = 0.f;
        for (int i = 0; i < 32; ++i) { a += r1[tid][i]; b += r2[tid][i]; }
        dbeta[tid] += a;
        dgamma[tid] += b;
        coef[tid] = (float)(a / count);
        coef[CI + tid] = (float)(b / count);
        coef[2 * CI + tid] = gamma[tid] * stat[3 * CI + tid];
    }
}

// ------------------------------------------------------------------------------------------------ conv1 wgrad
// dw1[co][kk] += sum_{n,t} dz1[n][co][t] * x[n][t+kk],  db1[co] += sum dz1     (conv1: 1 -> 8 channels, no input affine)
// Same scheme as conv2: m = co (8 of 16 rows), n = tap (10 of 16 columns), k = 32 time steps per MFMA.
constexpr int C1 = 8;
constexpr int W1_OUT = C1 * KW + C1;
__global__ __launch_bounds__(256) void conv1_wgrad_mfma_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                               float* __restrict__ partial, int T) {
    constexpr int ZP = WG_SC + 8, XP = WG_SC + 16;
    __shared__ __attribute__((aligned(16))) uint16_t zs[C1][ZP];
    __shared__ __attribute__((aligned(16))) uint16_t x0[XP], x1[XP];        // x1[i] = x0[i + 1]
    __shared__ float red[3][64][4];
    __shared__ float redb[4][C1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
    const int T1 = T - (KW - 1), tbeg = blockIdx.x * (WG_SC * WG_PASSES);
    const int ln = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum[C1];
#pragma unroll
    for (int co = 0; co < C1; ++co) bsum[co] = 0.f;
    float v[C1], xa, xb;
    auto fetch = [&](int tp) {
        const int t = tp + tid;
#pragma unroll
        for (int co = 0; co < C1; ++co) v[co] = t < T1 ? dz[((long)n * C1 + co) * T1 + t] : 0.f;
        xa = t < T ? x[(long)n * T + t] : 0.f;
        xb = (tid < 10 && t + 256 < T) ? x[(long)n * T + t + 256] : 0.f;
    };
    fetch(tbeg);
    for (int pass = 0; pass < WG_PASSES; ++pass) {
        const int tp = tbeg + pass * WG_SC;
        if (tp >= T1) break;
        __syncthreads();
        {
#pragma unroll
            for (int co = 0; co < C1; ++co) {
                bsum[co] += v[co];
                zs[co][tid] = (uint16_t)(cvt2(v[co], 0.f) & 0xffffu);
            }
            const uint16_t ha = (uint16_t)(cvt2(xa, 0.f) & 0xffffu);
            x0[tid] = ha;
            if (tid > 0) x1[tid - 1] = ha;
            if (tid < 10) {
                const uint16_t hb = (uint16_t)(cvt2(xb, 0.f) & 0xffffu);
                x0[tid + 256] = hb;
                x1[tid + 255] = hb;
            }
        }
        __syncthreads();
        if (pass + 1 < WG_PASSES && tp + WG_SC < T1) fetch(tp + WG_SC);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tb = 64 * wave + 32 * u + 8 * q;
            uint4 av = *(const uint4*)&zs[ln & 7][tb];
            if (ln >= C1) av = make_uint4(0u, 0u, 0u, 0u);
            const int kk = ln < KW ? ln : 0;
            const uint16_t* row = (kk & 1) ? x1 : x0;
            const uint32_t* p = (const uint32_t*)(row + tb + (kk & ~1));
            uint4 bv = make_uint4(p[0], p[1], p[2], p[3]);
            if (ln >= KW) bv = make_uint4(0u, 0u, 0u, 0u);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(av), as_frag(bv), acc, 0, 0, 0);
        }
    }
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave - 1][lane][e] = acc[e];
    }
#pragma unroll
    for (int co = 0; co < C1; ++co) {
        const float sm = wave_sum(bsum[co]);
        if (lane == 0) redb[wave][co] = sm;
    }
    __syncthreads();
    float* out = partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * W1_OUT;
    if (wave == 0 && q < 2 && ln < KW) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            out[(4 * q + e) * KW + ln] = acc[e] + red[0][lane][e] + red[1][lane][e] + red[2][lane][e];
    }
    if (tid < C1) out[C1 * KW + tid] = redb[0][tid] + redb[1][tid] + redb[2][tid] + redb[3][tid];
}


// =====================================================================================================================
// bf16 storage, channels-last (bf16 contraction mode): a1h [N][T1][8], a2h [N][T2][16], dz2h [N][T2][16], dz1h [N][T1][8] --
// one 16 / 32-byte row per time step, which IS the LDS row / MFMA operand of the kernels above: staging is a plain copy,
// outputs leave as 8-byte pieces (4 channels of one time step per lane, 16 consecutive time steps per lane group).
// Halves the bytes of every stream of the stage (the stage is HBM-bound).  BatchNorm affines never touch the activations:
// BatchNorm1's scale / shift are folded into conv2's weights / bias, BatchNorm2's into the fc weight copy (dgl.hip).
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// conv1 (1 -> 8 channels, 10 taps, valid) + ReLU on the f32 series -> a1h, BatchNorm1 partial sums (from the f32 values, columns
// < stat_limit only: the rest is halo of a time slice).  4 consecutive time steps per thread.
__global__ __launch_bounds__(256) void conv1_fwd_cl_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, uint4* __restrict__ a1h,
                                                           float* __restrict__ partial, int T, int stat_limit) {
    __shared__ float ws[KW][C1_];
    __shared__ float red[4][2 * C1_];
    __shared__ __attribute__((aligned(16))) float xs[1024 + 16];
    const int tid = threadIdx.x, n = blockIdx.y, T1 = T - (KW - 1);
    if (tid < KW * C1_) ws[tid / C1_][tid % C1_] = w[(tid % C1_) * KW + tid / C1_];
    // the workgroup's 1024 + 9 series values through LDS: coalesced dword loads (a series row of odd length has no 16-byte alignment),
    // then four aligned 16-byte LDS reads per thread -- instead of 13 stride-16-byte global loads per thread (22 % of the HBM rate)
    {
        const int tb = blockIdx.x * 1024;
        const float* xr = x + (long)n * T + tb;
        for (int i = tid; i < 1024 + 16; i += 256) xs[i] = tb + i < T ? xr[i] : 0.f;
    }
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + tid) * 4;
    float s1[C1_], s2[C1_];
#pragma unroll
    for (int c = 0; c < C1_; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
    if (t0 < T1) {
        float xv[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 v4 = *(const float4*)(xs + 4 * tid + 4 * q4);
            xv[4 * q4] = v4.x; xv[4 * q4 + 1] = v4.y; xv[4 * q4 + 2] = v4.z; xv[4 * q4 + 3] = v4.w;
        }
        float acc[4][C1_];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < C1_; ++c) acc[j][c] = b[c];
#pragma unroll
        for (int k = 0; k < KW; ++k)
#pragma unroll
            for (int c = 0; c < C1_; ++c) {
                const float wv = ws[k][c];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][c] += wv * xv[k + j];
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (t0 + j >= T1) break;
            float v[C1_];
#pragma unroll
            for (int c = 0; c < C1_; ++c) {
                v[c] = fmaxf(acc[j][c], 0.f);
                if (t0 + j < stat_limit) { s1[c] += v[c]; s2[c] += v[c] * v[c]; }
            }
            a1h[(long)n * T1 + t0 + j] = make_uint4(cvt2(v[0], v[1]), cvt2(v[2], v[3]), cvt2(v[4], v[5]), cvt2(v[6], v[7]));
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int c = 0; c < C1_; ++c) {
        const float a = wave_sum(s1[c]), q = wave_sum(s2[c]);
        if (lane == 0) { red[wave][c] = a; red[wave][C1_ + c] = q; }
    }
    __syncthreads();
    if (tid < 2 * C1_)
        partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * (2 * C1_) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// conv2's weights with BatchNorm1's scale folded in, as the MFMA operand fragments of the forward (3 x 64 lanes) and of the data
// gradient (5 x 64 lanes), plus the folded bias [16]: built once per launch instead of by every workgroup (a workgroup streams
// only 48 KB, ~100 scalar weight loads per lane were a visible part of its life).  pack: uint4[8 * 64] then float[16].
__global__ void conv2_pack_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ sc,
                                  const float* __restrict__ sh, uint4* __restrict__ pack) {
    const int lane = threadIdx.x & 63, s = threadIdx.x >> 6, ln = lane & 15, q = lane >> 4;      // 8 waves: s = 0..2 forward, 3..7 dgrad
    float v[8];
    if (s < 3) {
        const int kk = 4 * s + q;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) v[ci] = kk < KW ? w[(ln * CI + ci) * KW + kk] * sc[ci] : 0.f;
    } else {
        const int kk = 2 * (s - 3) + (q >> 1), c0 = 8 * (q & 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ln < CI ? w[((c0 + j) * CI + ln) * KW + kk] * sc[ln] : 0.f;
    }
    pack[s * 64 + lane] = make_uint4(cvt2(v[0], v[1]), cvt2(v[2], v[3]), cvt2(v[4], v[5]), cvt2(v[6], v[7]));
    if (threadIdx.x < CO) {
        float a = b[threadIdx.x];
        for (int ci = 0; ci < CI; ++ci) {
            float ws = 0.f;
            for (int kk = 0; kk < KW; ++kk) ws += w[(threadIdx.x * CI + ci) * KW + kk];
            a += ws * sh[ci];
        }
        ((float*)(pack + 8 * 64))[threadIdx.x] = a;
    }
}

// conv2 forward: a2h[n][t][co] = relu(b'[co] + sum_{ci,kk} w'[co][ci][kk] a1h[n][t+kk][ci]) with w' = w * sc1[ci] and
// b' = b + sum w * sh1[ci] (BatchNorm1 folded: the conv is valid, every output meets all taps); BatchNorm2 partial sums.
__global__ __launch_bounds__(256) void conv2_fwd_cl_kernel(const uint4* __restrict__ a1h, const uint4* __restrict__ pack,
                                                           uint2* __restrict__ a2h, float* __restrict__ partial, int T1) {
    __shared__ uint4 xs[FW_TT + 16];
    __shared__ float red[4][2 * CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
    const int T2 = T1 - (KW - 1), t0 = blockIdx.x * FW_TT;
    const int ln = lane & 15, q = lane >> 4;
    for (int tt = tid; tt < FW_TT + 16; tt += 256) {
        const int t = t0 + tt;
        xs[tt] = t < T1 ? a1h[(long)n * T1 + t] : make_uint4(0u, 0u, 0u, 0u);
    }
    bf16x8 wf[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) wf[s] = as_frag(pack[s * 64 + lane]);
    float bias[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[e] = ((const float*)(pack + 8 * 64))[4 * q + e];
    __syncthreads();
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tile = wave; tile < FW_TT / 16; tile += 4) {
        const int tb = tile * 16;
        if (t0 + tb >= T2) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], as_frag(xs[tb + ln + 4 * s + q]), acc, 0, 0, 0);
        const int t = t0 + tb + ln;
        if (t < T2) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = fmaxf(acc[e] + bias[e], 0.f); s1[e] += v[e]; s2[e] += v[e] * v[e]; }
            a2h[((long)n * T2 + t) * 4 + q] = make_uint2(cvt2(v[0], v[1]), cvt2(v[2], v[3]));
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        if (ln == 0) { red[wave][4 * q + e] = s1[e]; red[wave][CO + 4 * q + e] = s2[e]; }
    }
    __syncthreads();
    if (tid < 2 * CO)
        partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * (2 * CO) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// conv2 data gradient + fused BatchNorm1 backward: dz1h = [x > 0] kc (d_a1 - [t < own] (m1 + (x - mean) rstd m2)),  x = a1h;
// d_a1 = full correlation of dz2h with w' = w * sc1 -- the BatchNorm1 scale kc = gamma rstd = sc1 is already in the weights.
__global__ __launch_bounds__(256) void conv2_dgrad_cl_kernel(const uint4* __restrict__ dz2h, const uint4* __restrict__ pack,
                                                             uint2* __restrict__ dz1h, int T1,
                                                             const uint2* __restrict__ a1h, const float* __restrict__ coef,
                                                             const float* __restrict__ stat, int own) {
    __shared__ uint4 zs[FW_TT + 16][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
    const int T2 = T1 - (KW - 1), t0 = blockIdx.x * FW_TT;
    const int ln = lane & 15, q = lane >> 4;
    constexpr int NT = FW_TT / 16 / 4;
    uint2 xall[NT];
    if (q < 2) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = t0 + (wave + 4 * i) * 16 + ln;
            xall[i] = t < T1 ? a1h[((long)n * T1 + t) * 2 + q] : make_uint2(0u, 0u);
        }
    }
    for (int tt = tid; tt < FW_TT + 16; tt += 256) {
        const int t = t0 - (KW - 1) + tt;
        const bool ok = t >= 0 && t < T2;
        zs[tt][0] = ok ? dz2h[((long)n * T2 + t) * 2] : make_uint4(0u, 0u, 0u, 0u);
        zs[tt][1] = ok ? dz2h[((long)n * T2 + t) * 2 + 1] : make_uint4(0u, 0u, 0u, 0u);
    }
    bf16x8 wf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) wf[s] = as_frag(pack[(3 + s) * 64 + lane]);
    float km1[4] = {0.f, 0.f, 0.f, 0.f}, km2[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f};
    if (q < 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * q + e;
            km1[e] = coef[2 * CI + c] * coef[c]; km2[e] = coef[2 * CI + c] * coef[CI + c] * stat[3 * CI + c]; mu[e] = stat[2 * CI + c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int tb = (wave + 4 * i) * 16;
        if (t0 + tb < T1) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int kk = 2 * s + (q >> 1);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], as_frag(zs[tb + ln + (KW - 1) - kk][q & 1]), acc, 0, 0, 0);
            }
            const int t = t0 + tb + ln;
            if (t < T1 && q < 2) {
                const float x[4] = {bf16lo(xall[i].x), bf16hi(xall[i].x), bf16lo(xall[i].y), bf16hi(xall[i].y)};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = x[e] > 0.f ? acc[e] - (t < own ? km1[e] + (x[e] - mu[e]) * km2[e] : 0.f) : 0.f;
                dz1h[((long)n * T1 + t) * 2 + q] = make_uint2(cvt2(o[0], o[1]), cvt2(o[2], o[3]));
            }
        }
    }
}

// conv2 weight-gradient sums on the RAW activations: partial rows [Graw[co][ci][kk] = sum dz2 * a1 | db2[co] = sum dz2]; the
// normalisation is applied to the 1296 sums afterwards (conv2_wgrad_finish_kernel, raw = 1):  G' = rstd (Graw - mean db2).
__global__ __launch_bounds__(256) void conv2_wgrad_cl_kernel(const uint4* __restrict__ dz2h, const uint4* __restrict__ a1h,
                                                             float* __restrict__ partial, int T1) {
    constexpr int ZP = WG_SC + 8, XP = WG_SC + 16;
    __shared__ __attribute__((aligned(16))) uint16_t zs[CO][ZP];
    __shared__ __attribute__((aligned(16))) uint16_t x0[CI][XP], x1[CI][XP];
    __shared__ float red[3][5][64][4];
    __shared__ float redb[4][CO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
    const int T2 = T1 - (KW - 1), tbeg = blockIdx.x * (WG_SC * WG_PASSES);
    const int ln = lane & 15, q = lane >> 4;
    f32x4 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) bsum[co] = 0.f;
    uint4 z0, z1, xa, xb;
    auto fetch = [&](int tp) {
        const int t = tp + tid;
        const bool zok = t < T2;
        z0 = zok ? dz2h[((long)n * T2 + t) * 2] : make_uint4(0u, 0u, 0u, 0u);
        z1 = zok ? dz2h[((long)n * T2 + t) * 2 + 1] : make_uint4(0u, 0u, 0u, 0u);
        xa = t < T1 ? a1h[(long)n * T1 + t] : make_uint4(0u, 0u, 0u, 0u);
        const int t2 = tp + 256 + tid;
        xb = (tid < 10 && t2 < T1) ? a1h[(long)n * T1 + t2] : make_uint4(0u, 0u, 0u, 0u);
    };
    fetch(tbeg);
    for (int pass = 0; pass < WG_PASSES; ++pass) {
        const int tp = tbeg + pass * WG_SC;
        if (tp >= T2) break;
        __syncthreads();
        {
            const uint32_t zw[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                bsum[2 * p] += bf16lo(zw[p]); bsum[2 * p + 1] += bf16hi(zw[p]);
                zs[2 * p][tid] = (uint16_t)(zw[p] & 0xffffu); zs[2 * p + 1][tid] = (uint16_t)(zw[p] >> 16);
            }
            const uint32_t xw[4] = {xa.x, xa.y, xa.z, xa.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint16_t lo = (uint16_t)(xw[p] & 0xffffu), hi = (uint16_t)(xw[p] >> 16);
                x0[2 * p][tid] = lo; x0[2 * p + 1][tid] = hi;
                if (tid > 0) { x1[2 * p][tid - 1] = lo; x1[2 * p + 1][tid - 1] = hi; }
            }
            if (tid < 10) {
                const uint32_t yw[4] = {xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint16_t lo = (uint16_t)(yw[p] & 0xffffu), hi = (uint16_t)(yw[p] >> 16);
                    x0[2 * p][tid + 256] = lo; x0[2 * p + 1][tid + 256] = hi;
                    x1[2 * p][tid + 255] = lo; x1[2 * p + 1][tid + 255] = hi;
                }
            }
        }
        __syncthreads();
        if (pass + 1 < WG_PASSES && tp + WG_SC < T2) fetch(tp + WG_SC);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tb = 64 * wave + 32 * u + 8 * q;
            const bf16x8 a = as_frag(*(const uint4*)&zs[ln][tb]);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int kk = 2 * j + (ln >> 3), ci = ln & 7;
                const uint16_t* row = (kk & 1) ? &x1[ci][0] : &x0[ci][0];
                const uint32_t* p = (const uint32_t*)(row + tb + (kk & ~1));
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, as_frag(make_uint4(p[0], p[1], p[2], p[3])), acc[j], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave - 1][j][lane][e] = acc[j][e];
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float sm = wave_sum(bsum[co]);
        if (lane == 0) redb[wave][co] = sm;
    }
    __syncthreads();
    float* out = partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * WG_OUT;
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int kk = 2 * j + (ln >> 3), ci = ln & 7;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[((4 * q + e) * CI + ci) * KW + kk] = acc[j][e] + red[0][j][lane][e] + red[1][j][lane][e] + red[2][j][lane][e];
        }
    }
    if (tid < CO) out[CO * CI * KW + tid] = redb[0][tid] + redb[1][tid] + redb[2][tid] + redb[3][tid];
}

// conv1 weight gradient from dz1h and the f32 series
__global__ __launch_bounds__(256) void conv1_wgrad_cl_kernel(const uint4* __restrict__ dz1h, const float* __restrict__ x,
                                                             float* __restrict__ partial, int T) {
    constexpr int ZP = WG_SC + 8, XP = WG_SC + 16;
    __shared__ __attribute__((aligned(16))) uint16_t zs[C1][ZP];
    __shared__ __attribute__((aligned(16))) uint16_t x0[XP], x1[XP];
    __shared__ float red[3][64][4];
    __shared__ float redb[4][C1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.y;
    const int T1 = T - (KW - 1), tbeg = blockIdx.x * (WG_SC * WG_PASSES);
    const int ln = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum[C1];
#pragma unroll
    for (int co = 0; co < C1; ++co) bsum[co] = 0.f;
    uint4 z;
    float xa, xb;
    auto fetch = [&](int tp) {
        const int t = tp + tid;
        z = t < T1 ? dz1h[(long)n * T1 + t] : make_uint4(0u, 0u, 0u, 0u);
        xa = t < T ? x[(long)n * T + t] : 0.f;
        xb = (tid < 10 && t + 256 < T) ? x[(long)n * T + t + 256] : 0.f;
    };
    fetch(tbeg);
    for (int pass = 0; pass < WG_PASSES; ++pass) {
        const int tp = tbeg + pass * WG_SC;
        if (tp >= T1) break;
        __syncthreads();
        {
            const uint32_t zw[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                bsum[2 * p] += bf16lo(zw[p]); bsum[2 * p + 1] += bf16hi(zw[p]);
                zs[2 * p][tid] = (uint16_t)(zw[p] & 0xffffu); zs[2 * p + 1][tid] = (uint16_t)(zw[p] >> 16);
            }
            const uint16_t ha = (uint16_t)(cvt2(xa, 0.f) & 0xffffu);
            x0[tid] = ha;
            if (tid > 0) x1[tid - 1] = ha;
            if (tid < 10) {
                const uint16_t hb = (uint16_t)(cvt2(xb, 0.f) & 0xffffu);
                x0[tid + 256] = hb;
                x1[tid + 255] = hb;
            }
        }
        __syncthreads();
        if (pass + 1 < WG_PASSES && tp + WG_SC < T1) fetch(tp + WG_SC);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tb = 64 * wave + 32 * u + 8 * q;
            uint4 av = *(const uint4*)&zs[ln & 7][tb];
            if (ln >= C1) av = make_uint4(0u, 0u, 0u, 0u);
            const int kk = ln < KW ? ln : 0;
            const uint16_t* row = (kk & 1) ? x1 : x0;
            const uint32_t* p = (const uint32_t*)(row + tb + (kk & ~1));
            uint4 bv = make_uint4(p[0], p[1], p[2], p[3]);
            if (ln >= KW) bv = make_uint4(0u, 0u, 0u, 0u);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(av), as_frag(bv), acc, 0, 0, 0);
        }
    }
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave - 1][lane][e] = acc[e];
    }
#pragma unroll
    for (int co = 0; co < C1; ++co) {
        const float sm = wave_sum(bsum[co]);
        if (lane == 0) redb[wave][co] = sm;
    }
    __syncthreads();
    float* out = partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * W1_OUT;
    if (wave == 0 && q < 2 && ln < KW) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[(4 * q + e) * KW + ln] = acc[e] + red[0][lane][e] + red[1][lane][e] + red[2][lane][e];
    }
    if (tid < C1) out[C1 * KW + tid] = redb[0][tid] + redb[1][tid] + redb[2][tid] + redb[3][tid];
}


}  // namespace

// row slices of the partial-row reduction: ~64 rows per thread at least, at most 128 slices (one atomic per slice and output)
static int wgrad_slices(long nrows) {
    long s = nrows / 256;
    return (int)(s < 16 ? 16 : (s > 128 ? 128 : s));
}

int dgl_conv2_fwd_mfma(const float* a1, const float* w, const float* b, const float* sc, const float* sh, float* a2, float* partial,
                       int N, int T1, int* nblk, hipStream_t st) {
    const int T2 = T1 - (KW - 1);
    dim3 grid(cdiv(T2, FW_TT), N);
    conv2_fwd_mfma_kernel<<<grid, 256, 0, st>>>(a1, w, b, sc, sh, a2, partial, T1);
    STEP_LAUNCH_CHECK("conv2_fwd_mfma");
    *nblk = grid.x * grid.y;
    return STEP_OK;
}

int dgl_conv2_dgrad_mfma(const float* dz, const float* w, float* din, int N, int T1, const float* bnx, const float* coef, const float* stat,
                         int own, hipStream_t st) {
    conv2_dgrad_mfma_kernel<<<dim3(cdiv(T1, FW_TT), N), 256, 0, st>>>(dz, w, din, T1, bnx, coef, stat, own);
    STEP_LAUNCH_CHECK("conv2_dgrad_mfma");
    return STEP_OK;
}

long dgl_conv2_wgrad_scratch_floats(int N, int T1) {
    const int T2 = T1 - (KW - 1);
    return (long)N * cdiv(T2, WG_SC * WG_PASSES) * WG_OUT;
}

int dgl_conv2_wgrad_mfma(const float* dz, const float* a1, const float* sc, const float* sh, float* scratch, float* dw, float* db, int N,
                         int T1, hipStream_t st) {
    const int T2 = T1 - (KW - 1);
    dim3 grid(cdiv(T2, WG_SC * WG_PASSES), N);
    conv2_wgrad_mfma_kernel<<<grid, 256, 0, st>>>(dz, a1, sc, sh, scratch, T1, 0);
    STEP_LAUNCH_CHECK("conv2_wgrad_mfma");
    conv_wgrad_reduce_kernel<<<dim3(cdiv(WG_OUT, 64), wgrad_slices(grid.x * grid.y)), 256, 0, st>>>(scratch, grid.x * grid.y, WG_OUT, CO * CI * KW, dw, db);
    STEP_LAUNCH_CHECK("conv2_wgrad_reduce");
    return STEP_OK;
}

// conv2 weight / bias gradient together with the fused BatchNorm1 backward's coefficient step (see conv2_wgrad_finish_kernel), in
// two calls so that a time-sliced caller can sum graw over the ranks in between:
//   xhat:   graw (1296 floats) = [G'[co][ci][kk] = sum dz * xhat1 | db2[co] = sum dz]   (stat1 = BatchNorm1's [scale | shift | mean | rstd])
//   finish: dw / db / dgamma1 / dbeta1 += ..., coef1 out (24 floats); count = nodes x conv1 columns of the WHOLE series
int dgl_conv2_wgrad_xhat_mfma(const float* dz, const float* a1, const float* stat1, float* scratch, float* graw, int N, int T1,
                              hipStream_t st) {
    const int T2 = T1 - (KW - 1);
    dim3 grid(cdiv(T2, WG_SC * WG_PASSES), N);
    if (hipMemsetAsync(graw, 0, WG_OUT * sizeof(float), st) != hipSuccess) { step_set_error("memset failed"); return STEP_ERR_HIP; }
    conv2_wgrad_mfma_kernel<<<grid, 256, 0, st>>>(dz, a1, stat1, nullptr, scratch, T1, 1);
    STEP_LAUNCH_CHECK("conv2_wgrad_mfma(xhat)");
    conv_wgrad_reduce_kernel<<<dim3(cdiv(WG_OUT, 64), wgrad_slices(grid.x * grid.y)), 256, 0, st>>>(scratch, grid.x * grid.y, WG_OUT, CO * CI * KW, graw, graw + CO * CI * KW);
    STEP_LAUNCH_CHECK("conv2_wgrad_reduce");
    return STEP_OK;
}
int dgl_conv2_wgrad_finish(const float* graw, const float* w, const float* stat1, const float* gamma1, const float* beta1, double count,
                           float* dw, float* db, float* dgamma1, float* dbeta1, float* coef1, int raw, hipStream_t st) {
    conv2_wgrad_finish_kernel<<<1, 256, 0, st>>>(graw, w, stat1, gamma1, beta1, count, dw, db, dgamma1, dbeta1, coef1, raw);
    STEP_LAUNCH_CHECK("conv2_wgrad_finish");
    return STEP_OK;
}

// conv1 weight / bias gradient; scratch as for conv2 (dgl_conv2_wgrad_scratch_floats covers it: same grid, fewer outputs)
int dgl_conv1_wgrad_mfma(const float* dz, const float* x, float* scratch, float* dw, float* db, int N, int T, hipStream_t st) {
    const int T1 = T - (KW - 1);
    dim3 grid(cdiv(T1, WG_SC * WG_PASSES), N);
    conv1_wgrad_mfma_kernel<<<grid, 256, 0, st>>>(dz, x, scratch, T);
    STEP_LAUNCH_CHECK("conv1_wgrad_mfma");
    conv_wgrad_reduce_kernel<<<dim3(cdiv(W1_OUT, 64), wgrad_slices(grid.x * grid.y)), 256, 0, st>>>(scratch, grid.x * grid.y, W1_OUT, C1 * KW, dw, db);
    STEP_LAUNCH_CHECK("conv1_wgrad_reduce");
    return STEP_OK;
}

// ---- channels-last bf16 storage (see the kernels): host wrappers.  a1h / a2h / dz2h / dz1h are passed as void* (bf16 rows).
int dgl_conv1_fwd_cl(const float* x, const float* w, const float* b, void* a1h, float* partial, int N, int T, int stat_limit, int* nblk,
                     hipStream_t st) {
    const int T1 = T - (KW - 1);
    dim3 grid(cdiv(T1, 1024), N);
    conv1_fwd_cl_kernel<<<grid, 256, 0, st>>>(x, w, b, (uint4*)a1h, partial, T, stat_limit);
    STEP_LAUNCH_CHECK("conv1_fwd_cl");
    *nblk = grid.x * grid.y;
    return STEP_OK;
}
// pack: 8 * 64 * 16 + 64 bytes of scratch (dgl_conv2_pack_floats())
long dgl_conv2_pack_floats() { return 8 * 64 * 4 + 16; }
int dgl_conv2_fwd_cl(const void* a1h, const float* w, const float* b, const float* sc, const float* sh, void* a2h, float* partial, int N,
                     int T1, int* nblk, float* pack, hipStream_t st) {
    const int T2 = T1 - (KW - 1);
    dim3 grid(cdiv(T2, FW_TT), N);
    conv2_pack_kernel<<<1, 512, 0, st>>>(w, b, sc, sh, (uint4*)pack);
    conv2_fwd_cl_kernel<<<grid, 256, 0, st>>>((const uint4*)a1h, (const uint4*)pack, (uint2*)a2h, partial, T1);
    STEP_LAUNCH_CHECK("conv2_fwd_cl");
    *nblk = grid.x * grid.y;
    return STEP_OK;
}
int dgl_conv2_dgrad_cl(const void* dz2h, const float* w, const float* sc, void* dz1h, int N, int T1, const void* a1h, const float* coef,
                       const float* stat, int own, float* pack, hipStream_t st) {
    conv2_pack_kernel<<<1, 512, 0, st>>>(w, w, sc, sc, (uint4*)pack);       // (the bias part is not used by the data gradient)
    conv2_dgrad_cl_kernel<<<dim3(cdiv(T1, FW_TT), N), 256, 0, st>>>((const uint4*)dz2h, (const uint4*)pack, (uint2*)dz1h, T1, (const uint2*)a1h, coef, stat, own);
    STEP_LAUNCH_CHECK("conv2_dgrad_cl");
    return STEP_OK;
}
int dgl_conv2_wgrad_cl(const void* dz2h, const void* a1h, float* scratch, float* graw, int N, int T1, hipStream_t st) {
    const int T2 = T1 - (KW - 1);
    dim3 grid(cdiv(T2, WG_SC * WG_PASSES), N);
    if (hipMemsetAsync(graw, 0, WG_OUT * sizeof(float), st) != hipSuccess) { step_set_error("memset failed"); return STEP_ERR_HIP; }
    conv2_wgrad_cl_kernel<<<grid, 256, 0, st>>>((const uint4*)dz2h, (const uint4*)a1h, scratch, T1);
    STEP_LAUNCH_CHECK("conv2_wgrad_cl");
    conv_wgrad_reduce_kernel<<<dim3(cdiv(WG_OUT, 64), wgrad_slices(grid.x * grid.y)), 256, 0, st>>>(scratch, grid.x * grid.y, WG_OUT, CO * CI * KW, graw, graw + CO * CI * KW);
    STEP_LAUNCH_CHECK("conv2_wgrad_reduce");
    return STEP_OK;
}
int dgl_conv1_wgrad_cl(const void* dz1h, const float* x, float* scratch, float* dw, float* db, int N, int T, hipStream_t st) {
    const int T1 = T - (KW - 1);
    dim3 grid(cdiv(T1, WG_SC * WG_PASSES), N);
    conv1_wgrad_cl_kernel<<<grid, 256, 0, st>>>((const uint4*)dz1h, x, scratch, T);
    STEP_LAUNCH_CHECK("conv1_wgrad_cl");
    conv_wgrad_reduce_kernel<<<dim3(cdiv(W1_OUT, 64), wgrad_slices(grid.x * grid.y)), 256, 0, st>>>(scratch, grid.x * grid.y, W1_OUT, C1 * KW, dw, db);
    STEP_LAUNCH_CHECK("conv1_wgrad_reduce");
    return STEP_OK;
}
