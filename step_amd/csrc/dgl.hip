// DiscreteGraphLearning on device (reference: step/step_arch/discrete_graph_learning.py).
//
//  global feature  (:131-136)  Conv1d(1->8,k10) ReLU BN1 Conv1d(8->16,k10) ReLU BN2 flatten fc ReLU BN3
//  edge logits     (:148-153)  factorised: the one-hot gathers rel_rec/rel_send (:81-89) mean
//                              recv = g[e / N], send = g[e % N], so
//                              fc_out([send, recv]) = W[:, :100] g_j + W[:, 100:] g_i + b
//  Gumbel sampling (:11-45,157-161), straight-through backward
//
// The batch-norms are folded into their consumers (BN1 into conv2's input read, BN2 into the fc
// contraction via the GEMM's per-k affine) so the 267 MB conv2 activation is written once and
// read once per pass.  Everything here is HBM-bound f32 VALU work except the fc, which runs on
// the f32 matrix cores through step_gemm.
#include "common.h"
#include "step_internal.h"

namespace {

constexpr int KW = 10;      // conv kernel width
constexpr int EMB = 100;    // embedding_dim

// ------------------------------------------------------------------------------------------
// conv1d (valid) + ReLU, optional per-input-channel affine (a folded BatchNorm), per-block
// partial sums for the following BatchNorm.  in [N][CI][Tin] -> out [N][CO][Tin-9]
template <int CI, int CO, int TPT>
__global__ __launch_bounds__(256) void conv_relu_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ b, const float* __restrict__ in_sc,
                                                            const float* __restrict__ in_sh, float* __restrict__ out,
                                                            float* __restrict__ partial, int Tin, int stat_limit) {
    __shared__ __attribute__((aligned(16))) float ws[CI * KW * CO];   // [ci][k][co]
    __shared__ float red[4][2 * CO];
    const int tid = threadIdx.x, n = blockIdx.y;
    const int Tout = Tin - (KW - 1);
    for (int e = tid; e < CI * KW * CO; e += 256) {
        int co = e % CO, k = (e / CO) % KW, ci = e / (CO * KW);
        ws[e] = w[(co * CI + ci) * KW + k];
    }
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + tid) * TPT;
    float acc[CO][TPT];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int j = 0; j < TPT; ++j) acc[co][j] = b[co];
    if (t0 < Tout) {
#pragma unroll 1
        for (int ci = 0; ci < CI; ++ci) {
            const float* src = in + ((long)n * CI + ci) * Tin + t0;
            const float sc = in_sc ? in_sc[ci] : 1.f, sh = in_sh ? in_sh[ci] : 0.f;
            float x[TPT + KW - 1];
#pragma unroll
            for (int q = 0; q < TPT + KW - 1; ++q) x[q] = (t0 + q < Tin) ? src[q] * sc + sh : 0.f;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const float* wk = ws + (ci * KW + k) * CO;
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float wv = wk[co];
#pragma unroll
                    for (int j = 0; j < TPT; ++j) acc[co][j] += wv * x[k + j];
                }
            }
        }
    }
    float s1[CO], s2[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        s1[co] = 0.f; s2[co] = 0.f;
        float* dst = out + ((long)n * CO + co) * Tout + t0;
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            if (t0 + j < Tout) {
                float v = fmaxf(acc[co][j], 0.f);
                dst[j] = v;
                if (t0 + j < stat_limit) { s1[co] += v; s2[co] += v * v; }     // columns >= stat_limit: halo owned by the next time slice
            }
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float a = wave_sum(s1[co]), q = wave_sum(s2[co]);
        if (lane == 0) { red[wave][co] = a; red[wave][CO + co] = q; }
    }
    __syncthreads();
    if (tid < 2 * CO)
        partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * (2 * CO) + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// BatchNorm statistics from per-block partials -> folded scale/shift, saved mean/rstd, running stats.
// One block per channel.  stat layout: [4][C] = scale, shift, mean, rstd
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ rmean, float* __restrict__ rvar, int training,
                                                          float momentum, float* __restrict__ stat) {
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double mean, var;
    if (training) {
        double a = 0.0, q = 0.0;
        for (int i = tid; i < nblk; i += 256) { a += partial[(long)i * 2 * C + c]; q += partial[(long)i * 2 * C + C + c]; }
        r1[tid] = a; r2[tid] = q;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) { r1[tid] += r1[tid + s]; r2[tid] += r2[tid + s]; }
            __syncthreads();
        }
        mean = r1[0] / count;
        var = fmax(r2[0] / count - mean * mean, 0.0);
        if (tid == 0) {
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * count / fmax(count - 1.0, 1.0));
        }
    } else {
        mean = rmean[c];
        var = rvar[c];
    }
    if (tid == 0) {
        float rstd = (float)(1.0 / sqrt(var + 1e-5));
        float sc = gamma[c] * rstd;
        stat[c] = sc;
        stat[C + c] = beta[c] - (float)mean * sc;
        stat[2 * C + c] = (float)mean;
        stat[3 * C + c] = rstd;
    }
}

// Time-sliced variant (data-parallel ranks each own a time slice, see step_dgl_global_forward_shard): the per-block partials
// are first reduced to f64 sums [2C] -- summed over the ranks by the caller -- and the statistics come from those sums.
__global__ __launch_bounds__(256) void bn_sums_kernel(const float* __restrict__ partial, int nblk, int C, double* __restrict__ sums) {
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double a = 0.0, q = 0.0;
    for (int i = tid; i < nblk; i += 256) { a += partial[(long)i * 2 * C + c]; q += partial[(long)i * 2 * C + C + c]; }
    r1[tid] = a; r2[tid] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { r1[tid] += r1[tid + s]; r2[tid] += r2[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { sums[c] = r1[0]; sums[C + c] = r2[0]; }
}
// the same sums with the partial rows split over grid.y blocks per channel (f64 atomics into zeroed sums): at N = 4096 one block per
// channel walked 70 000 partial rows
__global__ __launch_bounds__(256) void bn_sums_sliced_kernel(const float* __restrict__ partial, int nblk, int C, double* __restrict__ sums) {
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int per = (nblk + gridDim.y - 1) / gridDim.y, beg = blockIdx.y * per, end = min(nblk, beg + per);
    double a = 0.0, q = 0.0;
    for (int i = beg + tid; i < end; i += 256) { a += partial[(long)i * 2 * C + c]; q += partial[(long)i * 2 * C + C + c]; }
    r1[tid] = a; r2[tid] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { r1[tid] += r1[tid + s]; r2[tid] += r2[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) { unsafeAtomicAdd(sums + c, r1[0]); unsafeAtomicAdd(sums + C + c, r2[0]); }
}
__global__ void bn_finalize_sums_kernel(const double* __restrict__ sums, int C, double count, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                        int training, float momentum, float* __restrict__ stat) {
    const int c = threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) {
        mean = sums[c] / count;
        var = fmax(sums[C + c] / count - mean * mean, 0.0);
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * count / fmax(count - 1.0, 1.0));
    } else {
        mean = rmean[c]; var = rvar[c];
    }
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    const float sc = gamma[c] * rstd;
    stat[c] = sc; stat[C + c] = beta[c] - (float)mean * sc; stat[2 * C + c] = (float)mean; stat[3 * C + c] = rstd;
}

// gpre[n][j] += bias[j]   (in place; the value saved for backward is fc(x)+b)
// g[n][j] = BN3(relu(gpre[n][j]))    one block per feature column j
__global__ __launch_bounds__(256) void fc_post_bn3_kernel(float* __restrict__ gpre, const float* __restrict__ bias, int N,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ rmean, float* __restrict__ rvar, int training,
                                                          float momentum, float* __restrict__ stat, float* __restrict__ g) {
    __shared__ double r1[256], r2[256];
    const int j = blockIdx.x, tid = threadIdx.x;
    const float bj = bias[j];
    double a = 0.0, q = 0.0;
    for (int n = tid; n < N; n += 256) {
        float v = gpre[(long)n * EMB + j] + bj;
        gpre[(long)n * EMB + j] = v;
        v = fmaxf(v, 0.f);
        a += v; q += (double)v * v;
    }
    r1[tid] = a; r2[tid] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { r1[tid] += r1[tid + s]; r2[tid] += r2[tid + s]; }
        __syncthreads();
    }
    double mean, var;
    if (training) {
        mean = r1[0] / N;
        var = fmax(r2[0] / N - mean * mean, 0.0);
        if (tid == 0) {
            rmean[j] = (1.f - momentum) * rmean[j] + momentum * (float)mean;
            rvar[j] = (1.f - momentum) * rvar[j] + momentum * (float)(var * N / fmax((double)N - 1.0, 1.0));
        }
    } else {
        mean = rmean[j]; var = rvar[j];
    }
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    if (tid == 0) { stat[2 * EMB + j] = (float)mean; stat[3 * EMB + j] = rstd; }
    const float sc = gamma[j] * rstd, sh = beta[j] - (float)mean * sc;
    for (int n = tid; n < N; n += 256) g[(long)n * EMB + j] = fmaxf(gpre[(long)n * EMB + j], 0.f) * sc + sh;
}

// backward of g = BN3(relu(gpre)) (training): one block per feature.
__global__ __launch_bounds__(256) void bn3_relu_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ gpre, int N,
                                                           const float* __restrict__ gamma, const float* __restrict__ stat,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dbias, float* __restrict__ dgpre,
                                                           float* __restrict__ dgpreT, float* __restrict__ colsum) {
    __shared__ double r1[256], r2[256];
    const int j = blockIdx.x, tid = threadIdx.x;
    const float mean = stat[2 * EMB + j], rstd = stat[3 * EMB + j];
    double a = 0.0, q = 0.0;
    for (int n = tid; n < N; n += 256) {
        float d = dg[(long)n * EMB + j];
        float xh = (fmaxf(gpre[(long)n * EMB + j], 0.f) - mean) * rstd;
        a += d; q += (double)d * xh;
    }
    r1[tid] = a; r2[tid] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { r1[tid] += r1[tid + s]; r2[tid] += r2[tid + s]; }
        __syncthreads();
    }
    const float m1 = (float)(r1[0] / N), m2 = (float)(r2[0] / N);
    const float k = gamma[j] * rstd;
    __syncthreads();
    double sb = 0.0;
    for (int n = tid; n < N; n += 256) {
        float pre = gpre[(long)n * EMB + j];
        float xh = (fmaxf(pre, 0.f) - mean) * rstd;
        float dh = k * (dg[(long)n * EMB + j] - m1 - xh * m2);
        float v = pre > 0.f ? dh : 0.f;
        dgpre[(long)n * EMB + j] = v;
        dgpreT[(long)j * N + n] = v;          // [EMB][N]: node index contiguous
        sb += v;
    }
    if (tid == 0) { dgamma[j] += (float)r2[0]; dbeta[j] += (float)r1[0]; }
    __syncthreads();
    r1[tid] = sb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) r1[tid] += r1[tid + s];
        __syncthreads();
    }
    if (tid == 0) { dbias[j] += (float)r1[0]; colsum[j] = (float)r1[0]; }
}

// BatchNorm (over n, t per channel) backward, pass 1: S1 = sum dy, S2 = sum dy * xhat, per block partials
template <int C>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ stat, int Tlen, float* __restrict__ partial) {
    __shared__ float red[4][2];
    const int n = blockIdx.z, c = blockIdx.y, tid = threadIdx.x;
    const float mean = stat[2 * C + c], rstd = stat[3 * C + c];
    const long base = ((long)n * C + c) * Tlen;
    float a = 0.f, q = 0.f;
    for (int t = blockIdx.x * 1024 + tid; t < min(Tlen, (int)(blockIdx.x + 1) * 1024); t += 256) {
        float d = dy[base + t];
        a += d; q += d * ((x[base + t] - mean) * rstd);
    }
    a = wave_sum(a); q = wave_sum(q);
    if ((tid & 63) == 0) { red[tid >> 6][0] = a; red[tid >> 6][1] = q; }
    __syncthreads();
    if (tid < 2) {
        long blk = (long)n * gridDim.x + blockIdx.x;
        partial[(blk * C + c) * 2 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
}
// pass 2: coefficients; coef layout [3][C]: m1 = S1/count, m2 = S2/count, k = gamma*rstd; also dgamma/dbeta
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ stat,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef) {
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double a = 0.0, q = 0.0;
    for (int i = tid; i < nblk; i += 256) { a += partial[((long)i * C + c) * 2]; q += partial[((long)i * C + c) * 2 + 1]; }
    r1[tid] = a; r2[tid] = q;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { r1[tid] += r1[tid + s]; r2[tid] += r2[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        dgamma[c] += (float)r2[0];
        dbeta[c] += (float)r1[0];
        coef[c] = (float)(r1[0] / count);
        coef[C + c] = (float)(r2[0] / count);
        coef[2 * C + c] = gamma[c] * stat[3 * C + c];
    }
}
// pass 3 (in place on dy): dz = [x > 0] * k * (dy - m1 - xhat * m2)      (x is the post-ReLU conv output)
template <int C>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ stat, const float* __restrict__ coef, int Tlen) {
    const int n = blockIdx.z, c = blockIdx.y;
    const float mean = stat[2 * C + c], rstd = stat[3 * C + c];
    const float m1 = coef[c], m2 = coef[C + c], k = coef[2 * C + c];
    const long base = ((long)n * C + c) * Tlen;
    for (int t = blockIdx.x * 1024 + threadIdx.x; t < min(Tlen, (int)(blockIdx.x + 1) * 1024); t += 256) {
        float xv = x[base + t];
        float d = k * (dy[base + t] - m1 - (xv - mean) * rstd * m2);
        dy[base + t] = xv > 0.f ? d : 0.f;
    }
}

// conv backward w.r.t. its input: din[n][ci][t'] = sum_{co,k} w[co][ci][k] * dz[n][co][t'-k]
template <int CI, int CO, int TPT>
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                            float* __restrict__ din, int Tin) {
    __shared__ float ws[CO * KW * CI];    // [co][k][ci]
    const int tid = threadIdx.x, n = blockIdx.y;
    const int Tout = Tin - (KW - 1);
    for (int e = tid; e < CO * KW * CI; e += 256) {
        int ci = e % CI, k = (e / CI) % KW, co = e / (CI * KW);
        ws[e] = w[(co * CI + ci) * KW + k];
    }
    __syncthreads();
    const int t0 = (blockIdx.x * 256 + tid) * TPT;
    if (t0 >= Tin) return;
    float acc[CI][TPT];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int j = 0; j < TPT; ++j) acc[ci][j] = 0.f;
#pragma unroll 2
    for (int co = 0; co < CO; ++co) {
        const float* src = dz + ((long)n * CO + co) * Tout;
        float d[TPT + KW - 1];           // d[q] = dz[t0 - 9 + q]
#pragma unroll
        for (int q = 0; q < TPT + KW - 1; ++q) {
            int t = t0 - (KW - 1) + q;
            d[q] = (t >= 0 && t < Tout) ? src[t] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const float* wk = ws + (co * KW + k) * CI;
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const float wv = wk[ci];
#pragma unroll
                for (int j = 0; j < TPT; ++j) acc[ci][j] += wv * d[j + (KW - 1) - k];
            }
        }
    }
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int j = 0; j < TPT; ++j)
            if (t0 + j < Tin) din[((long)n * CI + ci) * Tin + t0 + j] = acc[ci][j];
}

// conv backward w.r.t. weight and bias, one block per node:
//   dw[co][ci][k] += sum_t dz[n][co][t] * (in[n][ci][t+k] * sc[ci] + sh[ci]);  db[co] += sum_t dz[n][co][t]
// Each thread owns one (co, ci) pair and a slice of the time chunk; the 10-tap input window slides
// through registers, so one LDS read of dz and one of the input feed 10 FMAs.
template <int CI, int CO>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(const float* __restrict__ dz, const float* __restrict__ in,
                                                              const float* __restrict__ in_sc, const float* __restrict__ in_sh,
                                                              float* __restrict__ dw, float* __restrict__ db, int Tin) {
    constexpr int CH = 640;
    constexpr int NP = CO * CI;                 // pairs
    constexpr int SPL = 256 / NP;               // time slices per pair
    constexpr int SL = CH / SPL;                // slice length (multiple of 10)
    static_assert(256 % NP == 0 && CH % SPL == 0 && SL % 10 == 0, "tiling");
    __shared__ float sdz[CO][CH + 1];
    __shared__ float sin_[CI][CH + KW];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int Tout = Tin - (KW - 1);
    const int pr = tid % NP, sl = tid / NP;
    const int co = pr / CI, ci = pr % CI;
    float acc[KW], accb = 0.f;
#pragma unroll
    for (int k = 0; k < KW; ++k) acc[k] = 0.f;
    for (int t0 = 0; t0 < Tout; t0 += CH) {
        __syncthreads();
        for (int e = tid; e < CO * CH; e += 256) {
            int c2 = e / CH, t = e % CH;
            sdz[c2][t] = (t0 + t < Tout) ? dz[((long)n * CO + c2) * Tout + t0 + t] : 0.f;
        }
        for (int e = tid; e < CI * (CH + KW - 1); e += 256) {
            int c2 = e / (CH + KW - 1), t = e % (CH + KW - 1);
            float v = 0.f;
            if (t0 + t < Tin) {
                v = in[((long)n * CI + c2) * Tin + t0 + t];
                if (in_sc) v = v * in_sc[c2] + in_sh[c2];
            }
            sin_[c2][t] = v;
        }
        __syncthreads();
        const float* zr = &sdz[co][sl * SL];
        const float* xr = &sin_[ci][sl * SL];
        float w[KW];                               // w[q] = x[t + q]
#pragma unroll
        for (int q = 0; q < KW - 1; ++q) w[q] = xr[q];
        for (int tb = 0; tb < SL; tb += KW) {
#pragma unroll
            for (int u = 0; u < KW; ++u) {         // fully unrolled so the rotating window has static indices
                w[(u + KW - 1) % KW] = xr[tb + u + KW - 1];
                const float d = zr[tb + u];
                if (ci == 0) accb += d;
#pragma unroll
                for (int k = 0; k < KW; ++k) acc[k] += d * w[(u + k) % KW];
            }
        }
    }
    // combine the time slices inside the block (reusing sdz), then one atomic per output per block
    __syncthreads();
    float* red = &sdz[0][0];                     // [SPL][NP][KW + 1]
#pragma unroll
    for (int k = 0; k < KW; ++k) red[(sl * NP + pr) * (KW + 1) + k] = acc[k];
    red[(sl * NP + pr) * (KW + 1) + KW] = accb;
    __syncthreads();
    for (int o = tid; o < NP * (KW + 1); o += 256) {
        const int p2 = o / (KW + 1), k = o % (KW + 1);
        float s = 0.f;
        for (int q = 0; q < SPL; ++q) s += red[(q * NP + p2) * (KW + 1) + k];
        if (k < KW) atomicAdd(&dw[p2 * KW + k], s);
        else if (p2 % CI == 0) atomicAdd(&db[p2 / CI], s);
    }
}

// ------------------------------------------------------------------------------------------ edges
// z[i*N + j] = (wc0 - wc1) . relu(rcv[i] + snd[j]) + (bc0 - bc1),  theta = sigmoid(z)
// sndT is [EMB][N] (coalesced along j), rcv is [N][EMB] and already contains fc_out.bias.
__global__ __launch_bounds__(256) void edge_logit_kernel(const float* __restrict__ sndT, const float* __restrict__ rcv,
                                                         const float* __restrict__ wcat, const float* __restrict__ bcat, int N,
                                                         float* __restrict__ z, int vec4) {
    __shared__ float sr[EMB], sw[EMB];
    const int i = blockIdx.y;
    if (threadIdx.x < EMB) {
        sr[threadIdx.x] = rcv[(long)i * EMB + threadIdx.x];
        sw[threadIdx.x] = wcat[threadIdx.x] - wcat[EMB + threadIdx.x];
    }
    __syncthreads();
    const float b0 = bcat[0] - bcat[1];
    if (vec4) {        // four senders per thread: 16-byte loads of the feature rows (N % 4 == 0, aligned buffers: checked by the host)
        const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
        if (j >= N) return;
        float a0 = b0, a1 = b0, a2 = b0, a3 = b0;
#pragma unroll 4
        for (int f = 0; f < EMB; ++f) {
            const float4 s4 = *(const float4*)(sndT + (long)f * N + j);
            const float r = sr[f], w = sw[f];
            a0 += w * fmaxf(r + s4.x, 0.f); a1 += w * fmaxf(r + s4.y, 0.f); a2 += w * fmaxf(r + s4.z, 0.f); a3 += w * fmaxf(r + s4.w, 0.f);
        }
        *(float4*)(z + (long)i * N + j) = make_float4(a0, a1, a2, a3);
        return;
    }
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    float acc = b0;
#pragma unroll 4
    for (int f = 0; f < EMB; ++f) acc += sw[f] * fmaxf(sr[f] + sndT[(long)f * N + j], 0.f);
    z[(long)i * N + j] = acc;
}

// theta[b][e] = sigmoid(z[e]);  y0 = sigmoid((z + g0 - g1)/tau), adj = [y0 >= y1] with the diagonal cleared
__global__ __launch_bounds__(256) void gumbel_sample_kernel(const float* __restrict__ z, const float* __restrict__ u, int B, int N,
                                                            uint32_t seed_lo, uint32_t seed_hi, float inv_tau,
                                                            float* __restrict__ theta, float* __restrict__ y0, float* __restrict__ adj,
                                                            const StepDynState* __restrict__ dyn) {
    if (dyn) { const uint64_t x = dyn->seed_xor; seed_lo ^= (uint32_t)x; seed_hi ^= (uint32_t)(x >> 32); }      // replayed steps (step_hip.h)
    const long E = (long)N * N;
    const int b = blockIdx.y;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long)gridDim.x * 256) {
        const float zv = z[e];
        float u0, u1;
        if (u) {
            u0 = u[((long)b * E + e) * 2];
            u1 = u[((long)b * E + e) * 2 + 1];
        } else {
            uint32_t r[4];
            philox4x32((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)b, 0x5EEDu, seed_lo, seed_hi, r);
            u0 = u32_to_unit(r[0]); u1 = u32_to_unit(r[1]);
        }
        // sample_gumbel with eps = 1e-10 (discrete_graph_learning.py:11-18,36)
        const float g0 = -logf(-logf(u0 + 1e-10f) + 1e-10f);
        const float g1 = -logf(-logf(u1 + 1e-10f) + 1e-10f);
        const float a0 = (zv + g0 - g1) * inv_tau;         // (l0+g0)/tau - (l1+g1)/tau
        const float ys = 1.f / (1.f + __expf(-a0));
        const int i = (int)(e / N), j = (int)(e % N);
        theta[(long)b * E + e] = 1.f / (1.f + __expf(-zv));
        y0[(long)b * E + e] = ys;
        adj[(long)b * E + e] = (i != j && a0 >= 0.f) ? 1.f : 0.f;
    }
}

// dz[e] = sum_b dtheta*theta(1-theta) + [i != j] dadj * y0 (1-y0) / tau
__global__ __launch_bounds__(256) void edge_dz_kernel(const float* __restrict__ dtheta, const float* __restrict__ dadj,
                                                      const float* __restrict__ theta, const float* __restrict__ y0, int B, int N,
                                                      float inv_tau, float* __restrict__ dz) {
    const long E = (long)N * N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long)gridDim.x * 256) {
        const int i = (int)(e / N), j = (int)(e % N);
        float s = 0.f;
        for (int b = 0; b < B; ++b) {
            const long o = (long)b * E + e;
            float th = theta[o];
            if (dtheta) s += dtheta[o] * th * (1.f - th);
            if (dadj && i != j) { float y = y0[o]; s += dadj[o] * y * (1.f - y) * inv_tau; }
        }
        dz[e] = s;
    }
}

// Row pass (one block per receiver i): drcv[i][f] = sum_j dz[i][j] * wd[f] * [hid > 0];  also the
// fc_cat weight/bias gradients.  Column pass (one block per 64 senders): dsndT[f][j] = sum_i ...
__global__ __launch_bounds__(256) void edge_bwd_row_kernel(const float* __restrict__ dz, const float* __restrict__ sndT,
                                                           const float* __restrict__ rcv, const float* __restrict__ wcat, int N,
                                                           float* __restrict__ drcv, float* __restrict__ dwcat, float* __restrict__ dbcat) {
    extern __shared__ float sdz[];               // this receiver's dz row [N]
    __shared__ float sr[EMB], sw[EMB], ra[EMB], rh[EMB];
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < EMB) { sr[tid] = rcv[(long)i * EMB + tid]; sw[tid] = wcat[tid] - wcat[EMB + tid]; }
    float tot = 0.f;
    for (int j = tid; j < N; j += 256) { const float d = dz[(long)i * N + j]; sdz[j] = d; tot += d; }
    tot = wave_sum(tot);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    // FIVE features per wave at a time (25 features per wave = five rounds; round 6): a -> drcv, h -> dwcat (sum dz * hid).  With one feature
    // at a time a wave's life was 25 x (one load round trip + two wave reductions of 6 dependent shuffles) -- 48 us at 307 nodes with ~1 wave
    // per SIMD on the chip; five independent rows are requested together and their ten sums reduce interleaved.
    constexpr int FB = 5;
    static_assert(EMB % (4 * FB) == 0, "feature rounds");
    for (int f0 = wave * FB; f0 < EMB; f0 += 4 * FB) {
        float a[FB], h[FB], rf[FB];
        const float* srow[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) { a[u] = 0.f; h[u] = 0.f; rf[u] = sr[f0 + u]; srow[u] = sndT + (long)(f0 + u) * N; }
        // four senders per lane and iteration: one 16-byte load of each feature row (rows start 16-byte aligned when N % 4 == 0; otherwise
        // the scalar loop below takes everything), selects instead of branches
        const bool vec = (N & 3) == 0 && ((((uintptr_t)sndT) | ((uintptr_t)sdz)) & 15) == 0;
        const int n4 = vec ? N : 0;
        for (int j = 4 * lane; j < n4; j += 256) {
            const float4 d4 = *(const float4*)(sdz + j);
            float4 s4[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) s4[u] = *(const float4*)(srow[u] + j);
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const float h0 = rf[u] + s4[u].x, h1 = rf[u] + s4[u].y, h2 = rf[u] + s4[u].z, h3 = rf[u] + s4[u].w;
                const float m0 = h0 > 0.f ? d4.x : 0.f, m1 = h1 > 0.f ? d4.y : 0.f, m2 = h2 > 0.f ? d4.z : 0.f, m3 = h3 > 0.f ? d4.w : 0.f;
                a[u] += (m0 + m1) + (m2 + m3);
                h[u] += (m0 * h0 + m1 * h1) + (m2 * h2 + m3 * h3);
            }
        }
        for (int j = n4 + lane; j < N; j += 64) {                // (N % 4 != 0: one sender per lane)
            const float d = sdz[j];
            float sv[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) sv[u] = srow[u][j];
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const float hid = rf[u] + sv[u];
                const float m = hid > 0.f ? d : 0.f;
                a[u] += m; h[u] += m * hid;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < FB; ++u) { a[u] += __shfl_xor(a[u], o, 64); h[u] += __shfl_xor(h[u], o, 64); }
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < FB; ++u) { ra[f0 + u] = a[u]; rh[f0 + u] = h[u]; }
        }
    }
    __syncthreads();
    if (tid < EMB) {
        drcv[(long)i * EMB + tid] = ra[tid] * sw[tid];
        atomicAdd(&dwcat[tid], rh[tid]);
        atomicAdd(&dwcat[EMB + tid], -rh[tid]);
    }
    if (tid == 0) {
        const float s = red[0] + red[1] + red[2] + red[3];
        atomicAdd(&dbcat[0], s);
        atomicAdd(&dbcat[1], -s);
    }
}
// VJ consecutive senders per lane (VJ = 4: one 16-byte load of dz per receiver and lane)
template <int VJ>
__global__ __launch_bounds__(256) void edge_bwd_col_kernel(const float* __restrict__ dz, const float* __restrict__ sndT,
                                                           const float* __restrict__ rcv, const float* __restrict__ wcat, int N,
                                                           float* __restrict__ dsndT) {
    // block: 64 * VJ sender columns j (lanes) x 4 features (one per wave); grid (N / (64 VJ), EMB/4)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = (blockIdx.x * 64 + lane) * VJ;
    const int f = blockIdx.y * 4 + wave;
    if (j >= N || f >= EMB) return;
    const float wd = wcat[f] - wcat[EMB + f];
    if constexpr (VJ == 4) {
        const float4 s4 = *(const float4*)(sndT + (long)f * N + j);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
        for (int i = 0; i < N; ++i) {
            const float r = rcv[(long)i * EMB + f];
            const float4 d = *(const float4*)(dz + (long)i * N + j);
            a0 += r + s4.x > 0.f ? d.x : 0.f; a1 += r + s4.y > 0.f ? d.y : 0.f;
            a2 += r + s4.z > 0.f ? d.z : 0.f; a3 += r + s4.w > 0.f ? d.w : 0.f;
        }
        *(float4*)(dsndT + (long)f * N + j) = make_float4(a0 * wd, a1 * wd, a2 * wd, a3 * wd);
    } else {
        const float s = sndT[(long)f * N + j];
        float a = 0.f;
#pragma unroll 8
        for (int i = 0; i < N; ++i) {
            float hid = rcv[(long)i * EMB + f] + s;
            float d = dz[(long)i * N + j];
            a += hid > 0.f ? d : 0.f;
        }
        dsndT[(long)f * N + j] = a * wd;
    }
}

__global__ void colsum_kernel(const float* __restrict__ x, long rows, int cols, long ld, float* __restrict__ out) {
    // out[c] += sum_r x[r*ld + c]; grid.x = cols tiles of 64, grid.y = row chunks
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cols)
        for (long r = (long)blockIdx.y * 4 + w; r < rows; r += (long)gridDim.y * 4) s += x[r * ld + c];
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < cols) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

int step_colsum_launch(const float* x, long rows, int cols, long ld, float* out, hipStream_t st) {
    int chunks = (int)((rows + 255) / 256);
    if (chunks > 512) chunks = 512;
    if (chunks < 1) chunks = 1;
    colsum_kernel<<<dim3(cdiv(cols, 64), chunks), 256, 0, st>>>(x, rows, cols, ld, out);
    STEP_LAUNCH_CHECK("colsum");
    return STEP_OK;
}

// Small scratch block of the backward (floats): [0,128) BatchNorm coefficients, [128,256) column sums of dgpre, [256,288) the two
// per-channel sums of the fused BatchNorm2 backward, [288,288+1296) this step's raw conv2 weight-gradient sums (fused BatchNorm1)
constexpr int DGL_SMALL = 256 + 64 + 1344 + 100 * 32;       // ... + per-output-row BatchNorm2 sums of the channels-last path [EMB][32]

// Fused BatchNorm backward (no pass over the activations): coefficients from per-channel sums that come out of the
// weight-gradient contractions.  With dy = d(BatchNorm output), xhat the normalised input:
//     S1 = sum dy,  S2 = sum dy * xhat   ->   dbeta += S1, dgamma += S2, coef = [S1/count | S2/count | gamma*rstd]
// BatchNorm2 (behind the fc): dy = dgpre fc_w is never materialised; the dW GEMM epilogue (GemmFused.dotw) supplies
//     dots[2c] = sum_{o,k in c} fc_w[o,k] (dgpre^T a2)[o,k],  dots[2c+1] = sum_{o,k in c} fc_w[o,k] colsum(dgpre)[o] = S1,
//     S2 = rstd_c (dots[2c] - mean_c S1).
__global__ void bn2_fused_coef_kernel(const float* __restrict__ dots, const float* __restrict__ stat, const float* __restrict__ gamma,
                                      double count, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    constexpr int C = 16;
    const int c = threadIdx.x;
    if (c >= C) return;
    const float S1 = dots[2 * c + 1];
    const float S2 = stat[3 * C + c] * (dots[2 * c] - stat[2 * C + c] * S1);
    dgamma[c] += S2;
    dbeta[c] += S1;
    coef[c] = (float)(S1 / count);
    coef[C + c] = (float)(S2 / count);
    coef[2 * C + c] = gamma[c] * stat[3 * C + c];
}

// ---- channels-last bf16 storage of the conv activations (bf16 mode; kernels in dgl_conv_mfma.hip) ------------------------------
// The fc then contracts over k' = t*16 + c instead of the reference's k = c*T2 + t (:134 flattens [N,16,T2]).  Per step:
//   fc_prep:      wp[o][t*16+c] = bf16(fc_w[o][c][t] * sc2[c])  (BatchNorm2's scale folded in),  shiftdot[o] += sum fc_w[o][c][t] * sh2[c]
//   fc_unpermute: G = dgpre^T a2 (raw, [o][t*16+c]) ->  d fc_w[o][c][t] += sc2[c] G + sh2[c] colsum(dgpre)[o],  and the BatchNorm2
//                 backward sums per output row, dots_o[o][2c] += fc_w G, dots_o[o][2c+1] += fc_w colsum[o]  (see bn2_fused_coef_kernel)
// One thread owns one time step of one output row (16 channels in registers): every access is coalesced along t on the [c][t]
// side and along the row on the [t][c] side, no LDS.
__global__ __launch_bounds__(256) void fc_prep_kernel(const float* __restrict__ fc_w, const float* __restrict__ st2, uint4* __restrict__ wp,
                                                      float* __restrict__ shiftdot, int T2) {
    // one thread = one time step of one output row: 16 coalesced loads along t (one per channel), one 32-byte row out
    __shared__ float red[4];
    const int o = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    if (t < T2) {
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = fc_w[((long)o * 16 + c) * T2 + t];
#pragma unroll
        for (int c = 0; c < 16; ++c) { acc += v[c] * st2[16 + c]; v[c] *= st2[c]; }
        uint4* dst = wp + ((long)o * T2 + t) * 2;
        dst[0] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        dst[1] = make_uint4(pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15]));
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) atomicAdd(&shiftdot[o], red[0] + red[1] + red[2] + red[3]);
}
__global__ void gpre_init_kernel(float* __restrict__ gpre, const float* __restrict__ shiftdot, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) gpre[i] = shiftdot[i % EMB];
}
__global__ __launch_bounds__(256) void fc_unpermute_kernel(const float* __restrict__ graw, const float* __restrict__ fc_w,
                                                           const float* __restrict__ st2, const float* __restrict__ colsum,
                                                           float* __restrict__ dfc_w, float* __restrict__ dots_o, int T2, int overwrite) {
    // one thread = one time step of one output row: its 64-byte row of G, 16 coalesced loads of fc_w and of the gradient
    __shared__ float red[4][32];
    const int o = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x, tid = threadIdx.x;
    const float cso = colsum[o];
    float d[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) d[i] = 0.f;
    if (t < T2) {
        const float4* src = (const float4*)(graw + ((long)o * T2 + t) * 16);
        const float4 g0 = src[0], g1 = src[1], g2 = src[2], g3 = src[3];
        const float g[16] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w};
        float w[16], old[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) { const long idx = ((long)o * 16 + c) * T2 + t; w[c] = fc_w[idx]; old[c] = overwrite ? 0.f : dfc_w[idx]; }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            dfc_w[((long)o * 16 + c) * T2 + t] = old[c] + st2[c] * g[c] + st2[16 + c] * cso;
            d[2 * c] = w[c] * g[c]; d[2 * c + 1] = w[c] * cso;
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float v = wave_sum(d[i]);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < 32) atomicAdd(&dots_o[o * 32 + tid], red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}
__global__ void dots_reduce_kernel(const float* __restrict__ dots_o, float* __restrict__ dots) {
    const int i = threadIdx.x;
    if (i >= 32) return;
    double a = 0.0;
    for (int o = 0; o < EMB; ++o) a += dots_o[o * 32 + i];
    dots[i] = (float)a;
}

// bf16 mode keeps the conv activations and their gradients as channels-last bf16 rows; STEP_DGL_F32_STORAGE=1 keeps the round-1
// [N][C][T] f32 layout (A/B measurements).  Read on every call, forward and backward of a step must see the same value.
static bool dgl_channels_last(const StepDglParams* p) {
    const char* e = getenv("STEP_DGL_F32_STORAGE");
    return p->gemm_bf16 && !(e && e[0] == '1');
}

// STEP_DGL_LEGACY_BN=1 in the environment keeps the three-pass BatchNorm backward (A/B measurements, debugging)
static bool dgl_legacy_bn_backward() {
    const char* e = getenv("STEP_DGL_LEGACY_BN");
    return e && e[0] == '1';
}

// =========================================================================================== C ABI
extern "C" long step_dgl_global_saved_floats(int N, int T) {
    long T1 = T - 9, T2 = T - 18;
    // ... + the BatchNorm2-scaled, (t, c)-ordered f32 copy of fc.weight and its shift term (channels-last bf16 storage, bf16 mode)
    return (long)N * 8 * T1 + (long)N * 16 * T2 + (long)N * EMB + 4 * 8 + 4 * 16 + 4 * EMB + (long)EMB * 16 * T2 + 128;
}
// partial tiles of the split-K fc product (StepGemm.splitk_ws): N x 100 results from K = 16 (T - 18) -- 3 tiles at PEMS04, 256 splits, whose
// 7.9 M atomics into gpre were most of that launch
static long fc_splitk_ws_floats(int N, int T2) {
    const int splits = step_gemm_auto_splitk(N, EMB, 16 * T2, 1);
    return splits > 1 ? (long)splits * N * EMB + 4 : 0;
}
extern "C" long step_dgl_global_work_floats(int N, int T, int backward) {
    long T1 = T - 9, T2 = T - 18;
    long nb1 = (long)N * cdiv(T1, 1024), nb2 = (long)N * cdiv(T2, 1024);
    long part = (nb1 > nb2 ? nb1 : nb2) * 32 + 64;
    if (!backward) return part + dgl_conv2_pack_floats() + 64 + fc_splitk_ws_floats(N, (int)T2);
    return part + (long)N * 16 * T2 + (long)N * 8 * T1 + (long)EMB * 16 * T2 + 2L * N * EMB + DGL_SMALL + dgl_conv2_wgrad_scratch_floats(N, (int)T1);
}

static float* saved_wp(float* saved, int N, int T) {        // [EMB][16 (T-18)] scaled fc weight copy, then shiftdot [128]
    long T1 = T - 9, T2 = T - 18;
    return saved + (long)N * 8 * T1 + (long)N * 16 * T2 + (long)N * EMB + 4 * 8 + 4 * 16 + 4 * EMB;
}
static void carve_saved(float* saved, int N, int T, float** a1, float** a2, float** gpre, float** st1, float** st2, float** st3) {
    long T1 = T - 9, T2 = T - 18;
    *a1 = saved; saved += (long)N * 8 * T1;
    *a2 = saved; saved += (long)N * 16 * T2;
    *gpre = saved; saved += (long)N * EMB;
    *st1 = saved; saved += 4 * 8;
    *st2 = saved; saved += 4 * 16;
    *st3 = saved;
}

// BatchNorm statistics from the per-block partial rows: one block per channel for few rows, sliced f64-atomic sums for many
static int dgl_bn_stats(const float* partial, int nblk, int C, double count, const float* gamma, const float* beta, float* rm, float* rv,
                        int training, float momentum, float* stat, double* sums, hipStream_t st) {
    if (nblk < 8192) {
        bn_finalize_kernel<<<C, 256, 0, st>>>(partial, nblk, C, count, gamma, beta, rm, rv, training, momentum, stat);
    } else {
        if (hipMemsetAsync(sums, 0, 2 * C * sizeof(double), st) != hipSuccess) { step_set_error("memset failed"); return STEP_ERR_HIP; }
        bn_sums_sliced_kernel<<<dim3(C, 32), 256, 0, st>>>(partial, nblk, C, sums);
        bn_finalize_sums_kernel<<<1, 64, 0, st>>>(sums, C, count, gamma, beta, rm, rv, training, momentum, stat);
    }
    STEP_LAUNCH_CHECK("bn_stats");
    return STEP_OK;
}

// Forward in up to four phases.  shard == nullptr: the whole series on this device (phase must be 0).  With a shard the caller
// sums `sums` (after phases 1 and 2) and gpre (after phase 3) over the ranks between the calls:
//   1: conv1 + its BatchNorm sums -> sums[0..16)            2: BatchNorm1 statistics, conv2 + sums -> sums[16..48)
//   3: BatchNorm2 statistics, partial fc product -> gpre     4: + bias, ReLU, BatchNorm3 -> g
static int dgl_global_forward_impl(const float* series_nt, int N, int T, const StepDglParams* p, int training, float momentum,
                                   float* saved, float* work, double* sums, float* g, const StepDglShard* shard, int phase,
                                   hipStream_t st) {
    const int T1 = T - 9, T2 = T - 18;
    float *a1, *a2, *gpre, *st1, *st2, *st3;
    carve_saved(saved, N, T, &a1, &a2, &gpre, &st1, &st2, &st3);
    float* partial = work;
    const bool all = phase == 0;
    const double count1 = shard ? shard->count1 : (double)N * T1, count2 = shard ? shard->count2 : (double)N * T2;
    const bool cl = dgl_channels_last(p);
    const long part_floats = ((long)N * cdiv(T1, 1024) > (long)N * cdiv(T2, 1024) ? (long)N * cdiv(T1, 1024) : (long)N * cdiv(T2, 1024)) * 32 + 64;
    double* own_sums = (double*)(work + part_floats + dgl_conv2_pack_floats());       // 32 f64 (unsharded statistics of many partial rows)
    float* wp = saved_wp(saved, N, T);
    float* shiftdot = wp + (long)EMB * 16 * T2;
    if (all || phase == 1) {
        int nblk;
        if (cl) {
            STEP_TRY(dgl_conv1_fwd_cl(series_nt, p->conv1_w, p->conv1_b, a1, partial, N, T, shard ? shard->own1 : T1, &nblk, st));
        } else {
            dim3 grid(cdiv(T1, 256 * 4), N);
            conv_relu_fwd_kernel<1, 8, 4><<<grid, 256, 0, st>>>(series_nt, p->conv1_w, p->conv1_b, nullptr, nullptr, a1, partial, T,
                                                                shard ? shard->own1 : T1);
            STEP_LAUNCH_CHECK("conv1");
            nblk = grid.x * grid.y;
        }
        if (shard) bn_sums_kernel<<<8, 256, 0, st>>>(partial, nblk, 8, sums);
        else STEP_TRY(dgl_bn_stats(partial, nblk, 8, count1, p->bn1_w, p->bn1_b, p->bn1_rm, p->bn1_rv, training, momentum, st1, own_sums, st));
        STEP_LAUNCH_CHECK("bn1");
    }
    if (all || phase == 2) {
        if (shard) bn_finalize_sums_kernel<<<1, 64, 0, st>>>(sums, 8, count1, p->bn1_w, p->bn1_b, p->bn1_rm, p->bn1_rv, training, momentum, st1);
        int nblk;
        if (cl) {
            STEP_TRY(dgl_conv2_fwd_cl(a1, p->conv2_w, p->conv2_b, st1, st1 + 8, a2, partial, N, T1, &nblk,
                                      work + part_floats, st));
        } else if (p->gemm_bf16) {
            STEP_TRY(dgl_conv2_fwd_mfma(a1, p->conv2_w, p->conv2_b, st1, st1 + 8, a2, partial, N, T1, &nblk, st));
        } else {
            dim3 grid(cdiv(T2, 256 * 4), N);
            conv_relu_fwd_kernel<8, 16, 4><<<grid, 256, 0, st>>>(a1, p->conv2_w, p->conv2_b, st1, st1 + 8, a2, partial, T1, T2);
            STEP_LAUNCH_CHECK("conv2");
            nblk = grid.x * grid.y;
        }
        if (shard) bn_sums_kernel<<<16, 256, 0, st>>>(partial, nblk, 16, sums + 16);
        else STEP_TRY(dgl_bn_stats(partial, nblk, 16, count2, p->bn2_w, p->bn2_b, p->bn2_rm, p->bn2_rv, training, momentum, st2, own_sums, st));
        STEP_LAUNCH_CHECK("bn2");
    }
    if (all || phase == 3) {
        if (shard) bn_finalize_sums_kernel<<<1, 64, 0, st>>>(sums + 16, 16, count2, p->bn2_w, p->bn2_b, p->bn2_rm, p->bn2_rv, training, momentum, st2);
        const long K = 16L * T2;
        StepGemm gm = gemm_desc(N, EMB, (int)K, a2, K, 1, p->fc_w, 1, K, gpre, EMB);
        if (cl) {
            // bf16 rows of a2 against the BatchNorm2-scaled (t, c)-ordered weight copy; the shift term starts the accumulator
            if (hipMemsetAsync(shiftdot, 0, 128 * sizeof(float), st) != hipSuccess) { step_set_error("memset failed"); return STEP_ERR_HIP; }
            fc_prep_kernel<<<dim3(cdiv(T2, 256), EMB), 256, 0, st>>>(p->fc_w, st2, (uint4*)wp, shiftdot, T2);
            gpre_init_kernel<<<cdiv((long)N * EMB, 256), 256, 0, st>>>(gpre, shiftdot, (long)N * EMB);
            STEP_LAUNCH_CHECK("fc_prep");
            gm.a_bf16 = 1;
            gm.B = wp; gm.b_bf16 = 1;
        } else {
            if (hipMemsetAsync(gpre, 0, (size_t)N * EMB * sizeof(float), st) != hipSuccess) { step_set_error("memset failed"); return STEP_ERR_HIP; }
            gm.a_kscale = st2; gm.a_kshift = st2 + 16; gm.a_kperiod = T2;
        }
        gm.accumulate = 2;
        gm.splitk = -1;
        gm.compute_bf16 = p->gemm_bf16;
        if (fc_splitk_ws_floats(N, T2) > 0) {          // (the work buffer of step_dgl_global_work_floats(N, T, 0) ends with this scratch)
            float* ws = work + part_floats + dgl_conv2_pack_floats() + 64;
            gm.splitk_ws = (float*)(((uintptr_t)ws + 15) & ~(uintptr_t)15);
            gm.splitk_ws_floats = fc_splitk_ws_floats(N, T2) - 4;
        }
        STEP_TRY(step_gemm_launch(gm, st));
    }
    if (all || phase == 4) {
        fc_post_bn3_kernel<<<EMB, 256, 0, st>>>(gpre, p->fc_b, N, p->bn3_w, p->bn3_b, p->bn3_rm, p->bn3_rv, training, momentum, st3, g);
        STEP_LAUNCH_CHECK("fc_post_bn3");
    }
    return STEP_OK;
}

extern "C" int step_dgl_global_forward(const float* series_nt, int N, int T, const StepDglParams* p, int training,
                                       float momentum, float* saved, float* work, float* g, void* stream) {
    STEP_REQUIRE(series_nt && p && saved && work && g && N > 0 && T > 18, "dgl_global_forward: bad arguments");
    return dgl_global_forward_impl(series_nt, N, T, p, training, momentum, saved, work, nullptr, g, nullptr, 0, (hipStream_t)stream);
}

extern "C" int step_dgl_global_forward_shard(const float* series_slice, int N, int Ts, const StepDglParams* p, int training,
                                             float momentum, float* saved, float* work, double* sums, float* g,
                                             const StepDglShard* shard, int phase, void* stream) {
    STEP_REQUIRE(series_slice && p && saved && work && sums && g && shard && N > 0 && Ts > 18 && phase >= 1 && phase <= 4,
                 "dgl_global_forward_shard: bad arguments");
    STEP_REQUIRE(shard->own1 > 0 && shard->own1 <= Ts - 9 && shard->count1 > 0 && shard->count2 > 0, "dgl_global_forward_shard: bad shard");
    return dgl_global_forward_impl(series_slice, N, Ts, p, training, momentum, saved, work, sums, g, shard, phase, (hipStream_t)stream);
}

// float offset of an item inside `saved` (item 0: a1, 1: a2, 2: gpre [N,100]) / inside the backward `work` (item 10: the 32 BatchNorm2
// sums "dots", 11: the 1296 raw conv2 weight-gradient sums "graw") -- what a sharded caller reduces over the ranks
extern "C" long step_dgl_global_offset(int N, int T, int item) {
    const long T1 = T - 9, T2 = T - 18;
    if (item == 0) return 0;
    if (item == 1) return (long)N * 8 * T1;
    if (item == 2) return (long)N * 8 * T1 + (long)N * 16 * T2;
    const long nb1 = (long)N * cdiv(T1, 1024), nb2 = (long)N * cdiv(T2, 1024);
    const long coef = (nb1 > nb2 ? nb1 : nb2) * 32 + 64 + (long)N * 16 * T2 + (long)N * 8 * T1 + (long)EMB * 16 * T2 + 2L * N * EMB;
    if (item == 10) return coef + 256;
    if (item == 11) return coef + 320;
    return -1;
}

static int dgl_global_backward_impl(const float* series_nt, int N, int T, const StepDglParams* p, const float* saved, const float* dg,
                                    float* work, const StepDglParams* grads, const StepDglShard* shard, int phase, void* stream);

extern "C" int step_dgl_global_backward(const float* series_nt, int N, int T, const StepDglParams* p, const float* saved,
                                        const float* dg, float* work, const StepDglParams* grads, void* stream) {
    return step_dgl_global_backward_phase(series_nt, N, T, p, saved, dg, work, grads, 0, stream);
}

// phase 0: everything; phase 1: bn3 / fc backward up to the finished fc weight gradient (the 87 MB that dominate the
// data-parallel all-reduce, which the caller can start right away); phase 2: the rest, with the same `work` buffer
extern "C" int step_dgl_global_backward_phase(const float* series_nt, int N, int T, const StepDglParams* p, const float* saved,
                                              const float* dg, float* work, const StepDglParams* grads, int phase, void* stream) {
    STEP_REQUIRE(series_nt && p && saved && dg && work && grads && N > 0 && T > 18 && (phase & ~STEP_DGL_FRESH_FC_GRAD) >= 0 &&
                 (phase & ~STEP_DGL_FRESH_FC_GRAD) <= 2, "dgl_global_backward: bad arguments");
    return dgl_global_backward_impl(series_nt, N, T, p, saved, dg, work, grads, nullptr, phase, stream);
}

// Time slice of a data-parallel rank (see step_dgl_global_forward_shard): dg = the gradient of g averaged over the ranks.
//   phase 1: BatchNorm3 / fc backward up to the fc weight-slice gradient and the BatchNorm2 sums ("dots", offset item 10)
//   phase 3: BatchNorm2 coefficients from the (rank-summed) dots, fc input gradient, conv2 weight-gradient sums ("graw", item 11)
//   phase 4: conv2 / BatchNorm1 gradients from the (rank-summed) graw, conv2 data gradient, conv1 weight gradient (a partial sum:
//            conv1_w / conv1_b are summed, not averaged, over the ranks)
extern "C" int step_dgl_global_backward_shard(const float* series_slice, int N, int Ts, const StepDglParams* p, const float* saved,
                                              const float* dg, float* work, const StepDglParams* grads, const StepDglShard* shard,
                                              int phase, void* stream) {
    const int ph = phase & ~STEP_DGL_FRESH_FC_GRAD;
    STEP_REQUIRE(series_slice && p && saved && dg && work && grads && shard && N > 0 && Ts > 18 && (ph == 1 || ph == 3 || ph == 4),
                 "dgl_global_backward_shard: bad arguments");
    STEP_REQUIRE(p->gemm_bf16 && Ts - 18 >= 128, "dgl_global_backward_shard: needs the bf16 contraction mode and a slice of >= 128 conv2 columns");
    return dgl_global_backward_impl(series_slice, N, Ts, p, saved, dg, work, grads, shard, phase, stream);
}

static int dgl_global_backward_impl(const float* series_nt, int N, int T, const StepDglParams* p, const float* saved, const float* dg,
                                    float* work, const StepDglParams* grads, const StepDglShard* shard, int phase, void* stream) {
    const int fresh_fc = (phase & STEP_DGL_FRESH_FC_GRAD) ? 1 : 0;      // grads->fc_w holds no previous value: store, do not read-modify-write
    phase &= ~STEP_DGL_FRESH_FC_GRAD;
    const bool do_fc = phase == 0 || phase == 1, do_mid = phase == 0 || phase == 2 || phase == 3, do_tail = phase == 0 || phase == 2 || phase == 4;
    hipStream_t st = (hipStream_t)stream;
    const int T1 = T - 9, T2 = T - 18;
    const long K = 16L * T2;
    float *a1, *a2, *gpre, *st1, *st2, *st3;
    carve_saved((float*)saved, N, T, &a1, &a2, &gpre, &st1, &st2, &st3);
    long nb1 = (long)N * cdiv(T1, 1024), nb2 = (long)N * cdiv(T2, 1024);
    float* partial = work;
    float* d_a2 = work + (nb1 > nb2 ? nb1 : nb2) * 32 + 64;
    float* d_a1 = d_a2 + (long)N * 16 * T2;
    float* wraw = d_a1 + (long)N * 8 * T1;
    float* dgpre = wraw + (long)EMB * K;
    float* dgpreT = dgpre + (long)N * EMB;
    float* coef = dgpreT + (long)N * EMB;
    float* dots = coef + 256;            // written by phase 1 (dW GEMM epilogue), read by phase 2
    float* graw = coef + 320;
    float* wg_scratch = coef + DGL_SMALL;
    // BatchNorm2 backward without its two passes over d_a2 / a2 (1.3 GB at PEMS04): needs a tile to span at most two channels
    const bool fuse2 = T2 >= 128 && (shard || !dgl_legacy_bn_backward());
    const double count1 = shard ? shard->count1 : (double)N * T1, count2 = shard ? shard->count2 : (double)N * T2;
    if (dgl_channels_last(p)) {
        // channels-last bf16 storage: a1 / a2 / d_a2 / d_a1 hold bf16 rows [t][channels]; wp = scaled (t, c)-ordered fc weight of the forward
        const float* wp = saved_wp((float*)saved, N, T);
        float* dots_o = coef + 256 + 64 + 1344;
        float* colsum = coef + 128;
        if (do_fc) {
            bn3_relu_bwd_kernel<<<EMB, 256, 0, st>>>(dg, gpre, N, p->bn3_w, st3, grads->bn3_w, grads->bn3_b, grads->fc_b, dgpre, dgpreT, colsum);
            STEP_LAUNCH_CHECK("bn3_bwd");
            // G = dgpre^T a2 on the raw bf16 rows (n-contiguous bf16 B operand), then back to fc.weight's [c][t] order with BatchNorm2's
            // affine and the BatchNorm2 backward sums
            StepGemm gm = gemm_desc(EMB, (int)K, N, dgpre, 1, EMB, a2, K, 1, wraw, K);
            gm.b_bf16 = 1;
            gm.compute_bf16 = 1;
            STEP_TRY(step_gemm_launch(gm, st));
            if (hipMemsetAsync(dots_o, 0, EMB * 32 * sizeof(float), st) != hipSuccess) { step_set_error("memset failed"); return STEP_ERR_HIP; }
            fc_unpermute_kernel<<<dim3(cdiv(T2, 256), EMB), 256, 0, st>>>(wraw, p->fc_w, st2, colsum, grads->fc_w, dots_o, T2, fresh_fc);
            dots_reduce_kernel<<<1, 64, 0, st>>>(dots_o, dots);
            STEP_LAUNCH_CHECK("fc_unpermute");
        }
        if (do_mid) {
            bn2_fused_coef_kernel<<<1, 64, 0, st>>>(dots, st2, p->bn2_w, count2, grads->bn2_w, grads->bn2_b, coef);
            STEP_LAUNCH_CHECK("bn2_fused_coef");
            // dz2 = BatchNorm2 backward of dgpre @ fc_w, masked by conv2's ReLU: the scale rides in wp, the rest in the epilogue; bf16 rows out
            // (round 4: a column-strip kernel over all rows -- wp's slice as register-resident fragments, the gradient as pre-packed
            //  fragments, 1.6 instead of 4.7 B of operand traffic per output element -- was measured and is NOT faster than this tiled GEMM:
            //  16.38 vs 16.33 ms at 4096 nodes, 5.99 vs 5.92 at PEMS07, profiles/r04_l_fc_dgrad_strips_ab.log)
            StepGemm gm = gemm_desc(N, (int)K, EMB, dgpre, EMB, 1, wp, K, 1, d_a2, K);
            gm.b_bf16 = 1;
            gm.compute_bf16 = 1;
            GemmFused fu = {nullptr, nullptr, a2, coef, st2, 16, T2, GEMM_FUSED_INTERLEAVED};
            STEP_TRY(step_gemm_launch_fused(gm, fu, st));
            STEP_TRY(dgl_conv2_wgrad_cl(d_a2, a1, wg_scratch, graw, N, T1, st));
        }
        if (do_tail) {
            STEP_TRY(dgl_conv2_wgrad_finish(graw, p->conv2_w, st1, p->bn1_w, p->bn1_b, count1, grads->conv2_w, grads->conv2_b, grads->bn1_w,
                                            grads->bn1_b, coef + 64, 1, st));
            STEP_TRY(dgl_conv2_dgrad_cl(d_a2, p->conv2_w, st1, d_a1, N, T1, a1, coef + 64, st1, shard ? shard->own1 : T1, wraw, st));      // (wraw is free again: weight fragments)
            STEP_TRY(dgl_conv1_wgrad_cl(d_a1, series_nt, wg_scratch, grads->conv1_w, grads->conv1_b, N, T, st));
        }
        return STEP_OK;
    }
    if (do_fc) {
        // BN3 + ReLU backward, fc bias gradient
        float* colsum = coef + 128;          // column sums of dgpre [EMB] (coef holds 3 x 16 BatchNorm coefficients at most)
        bn3_relu_bwd_kernel<<<EMB, 256, 0, st>>>(dg, gpre, N, p->bn3_w, st3, grads->bn3_w, grads->bn3_b, grads->fc_b, dgpre, dgpreT, colsum);
        STEP_LAUNCH_CHECK("bn3_bwd");
        // fc weight gradient on the raw conv2 activation, then fold BN2's affine in
        {
            // d fc_w[o][k] += sc[c(k)] * (dgpre^T a2)[o][k] + sh[c(k)] * colsum(dgpre)[o]: BN2's affine rides in the GEMM epilogue
            StepGemm gm = gemm_desc(EMB, (int)K, N, dgpre, 1, EMB, a2, K, 1, grads->fc_w, K);
            gm.accumulate = fresh_fc ? 0 : 1;
            gm.c_nscale = st2; gm.c_nshift = st2 + 16; gm.c_nperiod = T2; gm.c_mvec = colsum;
            gm.compute_bf16 = p->gemm_bf16;
            if (fuse2) {
                if (hipMemsetAsync(dots, 0, 32 * sizeof(float), st) != hipSuccess) { step_set_error("memset failed"); return STEP_ERR_HIP; }
                GemmFused fu = {p->fc_w, dots, nullptr, nullptr, nullptr, 16, T2};
                STEP_TRY(step_gemm_launch_fused(gm, fu, st));
            } else {
                STEP_TRY(step_gemm_launch(gm, st));
            }
        }
    }
    if (!do_mid && !do_tail) return STEP_OK;
    // d(BN2 output) = dgpre @ fc_w
    if (do_mid) {
        StepGemm gm = gemm_desc(N, (int)K, EMB, dgpreT, 1, N, p->fc_w, K, 1, d_a2, K);     // A(m=n, k=o) = dgpreT[o][n]
        if (p->gemm_bf16 || fuse2) { gm.A = dgpre; gm.sam = EMB; gm.sak = 1; }             // LDS-staged path: k-contiguous rows
        gm.compute_bf16 = p->gemm_bf16;
        if (fuse2) {       // coefficients from the sums of phase 1, BatchNorm2 backward + ReLU mask in the GEMM's epilogue: writes dz2
            bn2_fused_coef_kernel<<<1, 64, 0, st>>>(dots, st2, p->bn2_w, count2, grads->bn2_w, grads->bn2_b, coef);
            STEP_LAUNCH_CHECK("bn2_fused_coef");
            GemmFused fu = {nullptr, nullptr, a2, coef, st2, 16, T2};
            STEP_TRY(step_gemm_launch_fused(gm, fu, st));
        } else {
            STEP_TRY(step_gemm_launch(gm, st));
        }
    }
    // BN2 backward (+ReLU mask) in place -> dz2
    if (do_mid && !fuse2) {
        dim3 grid(cdiv(T2, 1024), 16, N);
        bn_bwd_reduce_kernel<16><<<grid, 256, 0, st>>>(d_a2, a2, st2, T2, partial);
        STEP_LAUNCH_CHECK("bn2_bwd_reduce");
        bn_bwd_finalize_kernel<<<16, 256, 0, st>>>(partial, (int)nb2, 16, (double)N * T2, p->bn2_w, st2, grads->bn2_w, grads->bn2_b, coef);
        STEP_LAUNCH_CHECK("bn2_bwd_finalize");
        bn_bwd_apply_kernel<16><<<grid, 256, 0, st>>>(d_a2, a2, st2, coef, T2);
        STEP_LAUNCH_CHECK("bn2_bwd_apply");
    }
    // conv2 backward: weights (BN1 affine folded into the input read) and data
    // bf16 mode: BatchNorm1's backward is fused as well -- its sums come out of the conv2 weight-gradient contraction, its
    // transform rides in the epilogue of the conv2 data gradient (no pass over d_a1 / a1)
    const bool fuse1 = p->gemm_bf16 && (shard || !dgl_legacy_bn_backward());
    if (fuse1) {
        // mid: the raw sums G' = sum dz2 * xhat1 and db2 (summed over the ranks by a sharded caller before the tail)
        if (do_mid) STEP_TRY(dgl_conv2_wgrad_xhat_mfma(d_a2, a1, st1, wg_scratch, graw, N, T1, st));
        if (!do_tail) return STEP_OK;
        STEP_TRY(dgl_conv2_wgrad_finish(graw, p->conv2_w, st1, p->bn1_w, p->bn1_b, count1, grads->conv2_w, grads->conv2_b, grads->bn1_w,
                                        grads->bn1_b, coef + 64, 0, st));
        // columns >= own1 of a time slice are halo: the owner (the next rank) subtracts BatchNorm1's constant terms there
        STEP_TRY(dgl_conv2_dgrad_mfma(d_a2, p->conv2_w, d_a1, N, T1, a1, coef + 64, st1, shard ? shard->own1 : T1, st));
    } else if (p->gemm_bf16) {
        STEP_TRY(dgl_conv2_wgrad_mfma(d_a2, a1, st1, st1 + 8, wg_scratch, grads->conv2_w, grads->conv2_b, N, T1, st));
        STEP_TRY(dgl_conv2_dgrad_mfma(d_a2, p->conv2_w, d_a1, N, T1, nullptr, nullptr, nullptr, T1, st));
    } else {
        conv_bwd_weight_kernel<8, 16><<<N, 256, 0, st>>>(d_a2, a1, st1, st1 + 8, grads->conv2_w, grads->conv2_b, T1);
        STEP_LAUNCH_CHECK("conv2_bwd_weight");
        conv_bwd_data_kernel<8, 16, 4><<<dim3(cdiv(T1, 1024), N), 256, 0, st>>>(d_a2, p->conv2_w, d_a1, T1);
        STEP_LAUNCH_CHECK("conv2_bwd_data");
    }
    // BN1 backward in place -> dz1
    if (!fuse1) {
        dim3 grid(cdiv(T1, 1024), 8, N);
        bn_bwd_reduce_kernel<8><<<grid, 256, 0, st>>>(d_a1, a1, st1, T1, partial);
        STEP_LAUNCH_CHECK("bn1_bwd_reduce");
        bn_bwd_finalize_kernel<<<8, 256, 0, st>>>(partial, (int)nb1, 8, (double)N * T1, p->bn1_w, st1, grads->bn1_w, grads->bn1_b, coef);
        STEP_LAUNCH_CHECK("bn1_bwd_finalize");
        bn_bwd_apply_kernel<8><<<grid, 256, 0, st>>>(d_a1, a1, st1, coef, T1);
        STEP_LAUNCH_CHECK("bn1_bwd_apply");
    }
    if (p->gemm_bf16) {
        STEP_TRY(dgl_conv1_wgrad_mfma(d_a1, series_nt, wg_scratch, grads->conv1_w, grads->conv1_b, N, T, st));
    } else {
        conv_bwd_weight_kernel<1, 8><<<N, 256, 0, st>>>(d_a1, series_nt, nullptr, nullptr, grads->conv1_w, grads->conv1_b, T);
        STEP_LAUNCH_CHECK("conv1_bwd_weight");
    }
    return STEP_OK;
}

extern "C" long step_dgl_edges_saved_floats(int B, int N) { return 2L * N * EMB + (long)N * N + 2L * B * N * N; }
// where theta [B][N*N] lives inside `saved` (floats): pass saved + this as theta_out and the forward writes it once, in place
extern "C" long step_dgl_edges_theta_offset(int N) { return 2L * N * EMB + (long)N * N; }

// saved layout: sndT [EMB][N] | rcv [N][EMB] | z [N*N] | theta [B][N*N] | y0 [B][N*N]
extern "C" int step_dgl_edges_forward(const float* g, int N, int B, const StepDglParams* p, const float* u, uint64_t seed,
                                      float temperature, float* saved, float* theta_out, float* adj_out, void* stream) {
    return step_dgl_edges_forward_dyn(g, N, B, p, u, seed, temperature, saved, theta_out, adj_out, nullptr, stream);
}
extern "C" int step_dgl_edges_forward_dyn(const float* g, int N, int B, const StepDglParams* p, const float* u, uint64_t seed,
                                          float temperature, float* saved, float* theta_out, float* adj_out, const StepDynState* dyn,
                                          void* stream) {
    STEP_REQUIRE(g && p && saved && adj_out && N > 0 && B > 0 && temperature > 0.f, "dgl_edges_forward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* sndT = saved;
    float* rcv = sndT + (long)N * EMB;
    float* z = rcv + (long)N * EMB;
    float* theta = z + (long)N * N;
    float* y0 = theta + (long)B * N * N;
    // sndT[f][j] = sum_c g[j][c] * W[f][c]              (fc_out.weight[:, :100])
    StepGemm gs = gemm_desc(N, EMB, EMB, g, EMB, 1, p->fc_out_w, 1, 2 * EMB, sndT, 1);
    gs.scn = N;
    STEP_TRY(step_gemm_launch(gs, st));
    // rcv[i][f] = sum_c g[i][c] * W[f][100 + c] + b[f]
    StepGemm gr = gemm_desc(N, EMB, EMB, g, EMB, 1, p->fc_out_w + EMB, 1, 2 * EMB, rcv, EMB);
    gr.bias = p->fc_out_b;
    STEP_TRY(step_gemm_launch(gr, st));
    {
        const int vec4 = N % 4 == 0 && N >= 1024 && (((uintptr_t)sndT | (uintptr_t)z) & 15) == 0;
        edge_logit_kernel<<<dim3(cdiv(N, vec4 ? 1024 : 256), N), 256, 0, st>>>(sndT, rcv, p->fc_cat_w, p->fc_cat_b, N, z, vec4);
    }
    STEP_LAUNCH_CHECK("edge_logit");
    int gx = cdiv((long)N * N, 256);
    if (gx > 4096) gx = 4096;
    gumbel_sample_kernel<<<dim3(gx, B), 256, 0, st>>>(z, u, B, N, (uint32_t)seed, (uint32_t)(seed >> 32), 1.f / temperature, theta, y0, adj_out, dyn);
    STEP_LAUNCH_CHECK("gumbel_sample");
    if (theta_out && theta_out != theta) {
        if (hipMemcpyAsync(theta_out, theta, (size_t)B * N * N * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            step_set_error("dgl_edges_forward: theta copy failed");
            return STEP_ERR_HIP;
        }
    }
    return STEP_OK;
}

// work: dz [N*N] | drcv [N][EMB] | dsndT [EMB][N]
extern "C" long step_dgl_edges_work_floats(int N) { return (long)N * N + 2L * N * EMB; }

extern "C" int step_dgl_edges_backward(const float* g, int N, int B, const StepDglParams* p, const float* saved,
                                       const float* dtheta, const float* dadj, float temperature, float* work,
                                       const StepDglParams* grads, float* dg, void* aux_stream, void* stream) {
    STEP_REQUIRE(g && p && saved && work && grads && dg && N > 0 && B > 0, "dgl_edges_backward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    AuxLane lane(st, (hipStream_t)aux_stream);
    const float* sndT = saved;
    const float* rcv = sndT + (long)N * EMB;
    const float* theta = rcv + (long)N * EMB + (long)N * N;
    const float* y0 = theta + (long)B * N * N;
    float* dz = work;
    float* drcv = dz + (long)N * N;
    float* dsndT = drcv + (long)N * EMB;
    int gx = cdiv((long)N * N, 256);
    if (gx > 4096) gx = 4096;
    edge_dz_kernel<<<gx, 256, 0, st>>>(dtheta, dadj, theta, y0, B, N, 1.f / temperature, dz);
    STEP_LAUNCH_CHECK("edge_dz");
    edge_bwd_row_kernel<<<N, 256, (size_t)N * sizeof(float), st>>>(dz, sndT, rcv, p->fc_cat_w, N, drcv, grads->fc_cat_w, grads->fc_cat_b);
    STEP_LAUNCH_CHECK("edge_bwd_row");
    // (the column pass only shares its input with the row pass, but the auxiliary stream is the wrong place for it: the unjoined leaves of
    //  the WaveNet backward are still queued there -- 0.2 ms of them at PEMS07 -- and the main stream would wait behind them: measured
    //  6.13 vs 6.01 ms at PEMS07, 17.5 vs 17.0 at N = 4096, profiles/r03_ak_bench.log)
    if (N % 4 == 0 && N >= 1024 && (((uintptr_t)dz | (uintptr_t)sndT | (uintptr_t)dsndT) & 15) == 0)
        edge_bwd_col_kernel<4><<<dim3(cdiv(N, 256), cdiv(EMB, 4)), 256, 0, st>>>(dz, sndT, rcv, p->fc_cat_w, N, dsndT);
    else
        edge_bwd_col_kernel<1><<<dim3(cdiv(N, 64), cdiv(EMB, 4)), 256, 0, st>>>(dz, sndT, rcv, p->fc_cat_w, N, dsndT);
    STEP_LAUNCH_CHECK("edge_bwd_col");
    // dg = drcv @ W[:, 100:] + dsnd @ W[:, :100]
    StepGemm g1 = gemm_desc(N, EMB, EMB, drcv, EMB, 1, p->fc_out_w + EMB, 2 * EMB, 1, dg, EMB);
    STEP_TRY(step_gemm_launch(g1, st));
    StepGemm g2 = gemm_desc(N, EMB, EMB, dsndT, 1, N, p->fc_out_w, 2 * EMB, 1, dg, EMB);
    g2.accumulate = 1;
    STEP_TRY(step_gemm_launch(g2, st));
    // dW[:, 100:] += drcv^T @ g ;  dW[:, :100] += dsnd^T @ g ;  db += colsum(drcv): leaves (K = N products with 100 x 100 results, 57 us at
    // PEMS04 and 130 us at PEMS07 on four workgroups each) -- to the auxiliary stream, NOT joined here: the caller orders the first reader
    // of `grads` (and the release of `work`) after aux_stream
    static const bool tail_leaves = []() { const char* e = getenv("STEP_TAIL_LEAVES"); return !(e && e[0] == '0'); }();      // (A/B knob)
    hipStream_t leaf = tail_leaves ? lane.fork() : st;
    StepGemm g3 = gemm_desc(EMB, EMB, N, drcv, 1, EMB, g, EMB, 1, grads->fc_out_w + EMB, 2 * EMB);
    g3.accumulate = 1;
    STEP_TRY(step_gemm_launch(g3, leaf));
    StepGemm g4 = gemm_desc(EMB, EMB, N, dsndT, N, 1, g, EMB, 1, grads->fc_out_w, 2 * EMB);
    g4.accumulate = 1;
    STEP_TRY(step_gemm_launch(g4, leaf));
    STEP_TRY(step_colsum_launch(drcv, N, EMB, EMB, grads->fc_out_b, leaf));
    return STEP_OK;
}
