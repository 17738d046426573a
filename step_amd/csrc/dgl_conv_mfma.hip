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

constexpr int KW = 10, CI = 8, CO = 16;
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
    __syncthreads();
    for (int tile = wave; tile < FW_TT / 16; tile += 4) {
        const int tb = tile * 16;
        if (t0 + tb >= T1) break;
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
                    const int c = 4 * q + e;
                    const long idx = ((long)n * CI + c) * T1 + t;
                    const float x = bnx[idx];
                    const float cst = t < own ? coef[c] + (x - stat[2 * CI + c]) * stat[3 * CI + c] * coef[CI + c] : 0.f;
                    const float v = coef[2 * CI + c] * (acc[e] - cst);
                    din[idx] = x > 0.f ? v : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) din[((long)n * CI + 4 * q + e) * T1 + t] = acc[e];
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
__global__ __launch_bounds__(256) void conv2_wgrad_finish_kernel(const float* __restrict__ graw, const float* __restrict__ w,
                                                                 const float* __restrict__ stat, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, double count, float* __restrict__ dw,
                                                                 float* __restrict__ db, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ float r1[CI][32], r2[CI][32];
    const int tid = threadIdx.x;
    const int ci = tid >> 5, sub = tid & 31;                 // 8 channels x 32 threads, 5 (co, kk) pairs each
    float s1 = 0.f, s2 = 0.f;
    for (int i = sub; i < CO * KW; i += 32) {
        const int co = i / KW, kk = i % KW;
        const int idx = (co * CI + ci) * KW + kk;
        const float g = graw[idx], d = graw[CO * CI * KW + co], wv = w[idx];
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

}  // namespace

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
    conv_wgrad_reduce_kernel<<<dim3(cdiv(WG_OUT, 64), 16), 256, 0, st>>>(scratch, grid.x * grid.y, WG_OUT, CO * CI * KW, dw, db);
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
    conv_wgrad_reduce_kernel<<<dim3(cdiv(WG_OUT, 64), 16), 256, 0, st>>>(scratch, grid.x * grid.y, WG_OUT, CO * CI * KW, graw, graw + CO * CI * KW);
    STEP_LAUNCH_CHECK("conv2_wgrad_reduce");
    return STEP_OK;
}
int dgl_conv2_wgrad_finish(const float* graw, const float* w, const float* stat1, const float* gamma1, const float* beta1, double count,
                           float* dw, float* db, float* dgamma1, float* dbeta1, float* coef1, hipStream_t st) {
    conv2_wgrad_finish_kernel<<<1, 256, 0, st>>>(graw, w, stat1, gamma1, beta1, count, dw, db, dgamma1, dbeta1, coef1);
    STEP_LAUNCH_CHECK("conv2_wgrad_finish");
    return STEP_OK;
}

// conv1 weight / bias gradient; scratch as for conv2 (dgl_conv2_wgrad_scratch_floats covers it: same grid, fewer outputs)
int dgl_conv1_wgrad_mfma(const float* dz, const float* x, float* scratch, float* dw, float* db, int N, int T, hipStream_t st) {
    const int T1 = T - (KW - 1);
    dim3 grid(cdiv(T1, WG_SC * WG_PASSES), N);
    conv1_wgrad_mfma_kernel<<<grid, 256, 0, st>>>(dz, x, scratch, T);
    STEP_LAUNCH_CHECK("conv1_wgrad_mfma");
    conv_wgrad_reduce_kernel<<<dim3(cdiv(W1_OUT, 64), 16), 256, 0, st>>>(scratch, grid.x * grid.y, W1_OUT, C1 * KW, dw, db);
    STEP_LAUNCH_CHECK("conv1_wgrad_reduce");
    return STEP_OK;
}
