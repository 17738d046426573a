// GraphWaveNet backbone (STEP's fork, reference: step/step_arch/graphwavenet/model.py:132-224),
// forward and hand-written backward, orchestrated on the host side of the C ABI.
//
// Activation layout: X[b][n][t][c] with the 32 channels innermost ("position-major": one
// position = (b, n, t)).  Every channel contraction is then a plain row-major GEMM and the
// K-hop diffusion  (P x)[b,:,w,:] = sum_v x[b,:,v,:] P[b,v,w]  (model.py:10-16) is a batched
// GEMM  Out[b] (N x T*32) = P[b]^T X[b]  that reads/writes 32-channel slots of the concatenated
// gcn buffer cat[b][n][t][7*32] in place (slot order = torch.cat order of model.py:36-45:
// [x, P_f x, P_f^2 x, P_b x, P_b^2 x, P_a x, P_a^2 x]), so the 224->32 mix is one GEMM.
// All GEMMs run on the f32 matrix cores (step_gemm).  The remaining kernels are HBM-bound
// position-wise maps / reductions.
//
// Semantics kept from the reference: skip connections only matter at the last time index
// (every crop at model.py:196 keeps the tail, final T = 1); the last layer's gcn/bn output is
// dead code (model.py:202-213 for i = 7), so gconv.7 / bn.7 / residual_convs.* get no gradient.
#include "common.h"
#include "step_internal.h"

namespace {

constexpr int C = 32, CAT = 224, CS = 256, CE = 512, OUT = 12, NL = 8, HID = 96;
constexpr int TIN[NL] = {13, 12, 10, 9, 7, 6, 4, 3};
constexpr int DIL[NL] = {1, 2, 1, 2, 1, 2, 1, 2};
constexpr int TOUT[NL] = {12, 10, 9, 7, 6, 4, 3, 1};

// ---------------------------------------------------------------------------- pointwise kernels
// start conv on the left-padded 2-channel input (model.py:143-155): x0[bn][t][c], t = 0..12
__global__ void start_conv_kernel(const float* __restrict__ hist, int B, int N, int Cin, const float* __restrict__ w,
                                  const float* __restrict__ b, float* __restrict__ x0) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)B * N * 13 * C;
    if (idx >= total) return;
    int c = idx % C, t = (idx / C) % 13;
    long bn = idx / (C * 13);
    int n = bn % N, bb = bn / N;
    float v = b[c];
    if (t >= 1) {
        const float* src = hist + (((long)bb * 12 + (t - 1)) * N + n) * Cin;
        v += w[c * 2] * src[0] + w[c * 2 + 1] * src[1];
    }
    x0[idx] = v;
}
__global__ __launch_bounds__(256) void start_conv_bwd_kernel(const float* __restrict__ hist, int B, int N, int Cin,
                                                             const float* __restrict__ dx0, float* __restrict__ dw,
                                                             float* __restrict__ db) {
    __shared__ float red[8][96];
    const int c = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;                       // 8 position sub-rows per block
    long npos = (long)B * N * 13;
    float a0 = 0.f, a1 = 0.f, ab = 0.f;
    for (long p = (long)blockIdx.x * 8 + sub; p < npos; p += (long)gridDim.x * 8) {
        int t = p % 13;
        long bn = p / 13;
        int n = bn % N, bb = bn / N;
        float d = dx0[p * C + c];
        ab += d;
        if (t >= 1) {
            const float* src = hist + (((long)bb * 12 + (t - 1)) * N + n) * Cin;
            a0 += d * src[0]; a1 += d * src[1];
        }
    }
    red[sub][c] = a0; red[sub][32 + c] = a1; red[sub][64 + c] = ab;
    __syncthreads();
    if (threadIdx.x < 96) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
        const int cc = threadIdx.x & 31, kind = threadIdx.x >> 5;
        if (kind == 2) atomicAdd(&db[cc], s); else atomicAdd(&dw[cc * 2 + kind], s);
    }
}

// degree vectors: rs[b][i] = 1 + sum_j A[b][i][j] ; cs[b][j] = 1 + sum_i A[b][i][j]
__global__ __launch_bounds__(256) void row_sums_kernel(const float* __restrict__ A, int N, float* __restrict__ rs) {
    __shared__ float red[4];
    const long row = blockIdx.x;                            // b*N + i
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) s += A[row * N + j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rs[row] = 1.f + red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void col_sums_kernel(const float* __restrict__ A, int N, float* __restrict__ cs) {
    __shared__ float red[4][64];
    const int b = blockIdx.y;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float s = 0.f;
    if (j < N)
        for (int i = w; i < N; i += 4) s += A[((long)b * N + i) * N + j];
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && j < N) cs[(long)b * N + j] = 1.f + red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// P_f = D^-1 (A + I),  P_b = Dc^-1 (A^T + I)       (model.py:121-130,160); 32x32 tiles, grid (N/32, N/32, B)
// Writes both supports and their transposes (the adjoint hops read the transposed stack so that the
// contraction index is never the contiguous one).
__global__ __launch_bounds__(256) void rw_build_kernel(const float* __restrict__ A, int N, const float* __restrict__ rs,
                                                       const float* __restrict__ cs, float* __restrict__ Pf, float* __restrict__ Pb,
                                                       float* __restrict__ PfT, float* __restrict__ PbT) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long base = (long)b * N * N;
    for (int r = ty; r < 32; r += 8) {
        int i = i0 + r, j = j0 + tx;
        float a = (i < N && j < N) ? A[base + (long)i * N + j] : 0.f;
        tile[r][tx] = a;
        if (i < N && j < N) {
            const float v = a + (i == j ? 1.f : 0.f);
            Pf[base + (long)i * N + j] = v / rs[(long)b * N + i];
            PbT[base + (long)i * N + j] = v / cs[(long)b * N + j];        // P_b^T[i][j] = P_b[j][i]
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                       // transposed tile: element (i, j) = tile[tx][r]
        int j = j0 + r, i = i0 + tx;
        if (i < N && j < N) {
            const float v = tile[tx][r] + (i == j ? 1.f : 0.f);
            Pb[base + (long)j * N + i] = v / cs[(long)b * N + j];
            PfT[base + (long)j * N + i] = v / rs[(long)b * N + i];        // P_f^T[j][i] = P_f[i][j]
        }
    }
}
// adaptive support replicated per sample into stack slot 2 (and its transpose)
__global__ void replicate_adp_kernel(const float* __restrict__ Pa, int N, int B, float* __restrict__ P2, float* __restrict__ PT2) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * N) return;
    int i = idx / N, j = idx % N;
    float v = Pa[idx], vt = Pa[(long)j * N + i];
    for (int b = 0; b < B; ++b) { P2[(long)b * N * N + idx] = v; PT2[(long)b * N * N + idx] = vt; }
}
// bf16 copies of both stacks with the row pitch padded to N8 (pad = 0): the k-contiguous A operand of the bf16 hops
__global__ void stacks_to_bf16_kernel(const float* __restrict__ P, const float* __restrict__ PT, long rows, int N, int N8,
                                      uint16_t* __restrict__ P16, uint16_t* __restrict__ PT16) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * (N8 / 2)) return;
    const long row = idx / (N8 / 2);
    const int j = (int)(idx % (N8 / 2)) * 2;
    const float* src = blockIdx.y ? PT : P;
    uint16_t* dst = blockIdx.y ? PT16 : P16;
    const float a = j < N ? src[row * N + j] : 0.f, b = j + 1 < N ? src[row * N + j + 1] : 0.f;
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    f2 t = {a, b};
    *(uint32_t*)(dst + row * N8 + j) = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, b2));
}
// out[e] = sum_b x[b][e]
__global__ void sum_batches_kernel(const float* __restrict__ x, long n, int B, float* __restrict__ out) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += x[(long)b * n + idx];
    out[idx] = s;
}
// rdot[row] = sum_k dP[row][k] * P[row][k]
__global__ __launch_bounds__(256) void row_dot_kernel(const float* __restrict__ dP, const float* __restrict__ P, int N,
                                                      float* __restrict__ out) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) s += dP[row * N + j] * P[row * N + j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = red[0] + red[1] + red[2] + red[3];
}
// dA[b][i][j] = (dPf[i][j] - rf[i]) / rs[i] + (dPb[j][i] - rb[j]) / cs[j]
__global__ __launch_bounds__(256) void rw_bwd_kernel(const float* __restrict__ dPf, const float* __restrict__ dPb, int N,
                                                     const float* __restrict__ rs, const float* __restrict__ cs,
                                                     const float* __restrict__ rf, const float* __restrict__ rb, float* __restrict__ dA) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long base = (long)b * N * N;
    for (int r = ty; r < 32; r += 8) {                       // load dPb tile rows j, cols i
        int j = j0 + r, i = i0 + tx;
        tile[r][tx] = (i < N && j < N) ? (dPb[base + (long)j * N + i] - rb[(long)b * N + j]) / cs[(long)b * N + j] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int i = i0 + r, j = j0 + tx;
        if (i < N && j < N)
            dA[base + (long)i * N + j] = (dPf[base + (long)i * N + j] - rf[(long)b * N + i]) / rs[(long)b * N + i] + tile[tx][r];
    }
}
// adaptive adjacency: P_a[i][:] = softmax(relu(M[i][:]))   (model.py:165), one block per row
__global__ __launch_bounds__(256) void softmax_relu_rows_kernel(const float* __restrict__ M, int N, float* __restrict__ P) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    float mx = 0.f;                                          // relu output is >= 0
    for (int j = threadIdx.x; j < N; j += 256) mx = fmaxf(mx, M[row * N + j]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) s += __expf(fmaxf(M[row * N + j], 0.f) - mx);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    for (int j = threadIdx.x; j < N; j += 256) P[row * N + j] = __expf(fmaxf(M[row * N + j], 0.f) - mx) * inv;
}
// dM = [M > 0] * P * (dP - rowdot)
__global__ void softmax_relu_bwd_kernel(const float* __restrict__ M, const float* __restrict__ P, const float* __restrict__ dP,
                                        const float* __restrict__ rdot, int N, float* __restrict__ dM) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * N) return;
    int i = idx / N;
    dM[idx] = M[idx] > 0.f ? P[idx] * (dP[idx] - rdot[i]) : 0.f;
}

// gated-TCN weights [32,32,1,2] x2 -> Wcat[64 out][64 in] (in = tap*32 + c), bias[64]; all 8 layers in one launch (grid.y)
struct GatePtrs { float *wf[8], *bf[8], *wg[8], *bg[8]; };
__global__ void pack_gate_kernel(GatePtrs P, float* __restrict__ wcat_all, float* __restrict__ bcat_all) {
    const int L = blockIdx.y;
    const float *wf = P.wf[L], *bf = P.bf[L], *wg = P.wg[L], *bg = P.bg[L];
    float* wcat = wcat_all + L * 4096;
    float* bcat = bcat_all + L * 64;
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 64 * 64) {
        int o = idx / 64, k = idx % 64;
        const float* src = o < 32 ? wf : wg;
        wcat[idx] = src[(((o & 31) * 32) + (k & 31)) * 2 + (k >> 5)];
    }
    if (idx < 64) bcat[idx] = idx < 32 ? bf[idx] : bg[idx - 32];
}
__global__ void unpack_gate_grad_kernel(const float* __restrict__ dwcat_all, const float* __restrict__ dbcat_all, GatePtrs G) {
    const int L = blockIdx.y;
    const float* dwcat = dwcat_all + L * 4096;
    const float* dbcat = dbcat_all + L * 64;
    float *dwf = G.wf[L], *dbf = G.bf[L], *dwg = G.wg[L], *dbg = G.bg[L];
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 64 * 64) {
        int o = idx / 64, k = idx % 64;
        float* dst = o < 32 ? dwf : dwg;
        dst[(((o & 31) * 32) + (k & 31)) * 2 + (k >> 5)] += dwcat[idx];
    }
    if (idx < 64) { if (idx < 32) dbf[idx] += dbcat[idx]; else dbg[idx - 32] += dbcat[idx]; }
}
// xcat[(bn,t)][tap*32 + c] = x[bn][t + tap*dil][c]
__global__ void im2col_kernel(const float* __restrict__ x, long BN, int Tin, int Tout, int dil, float* __restrict__ xcat) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= BN * Tout * 64) return;
    int k = idx % 64, t = (idx / 64) % Tout;
    long bn = idx / (64L * Tout);
    xcat[idx] = x[(bn * Tin + t + (k >> 5) * dil) * C + (k & 31)];
}
// pre[pos][64] -> tf = tanh, sg = sigmoid, z = tf*sg into cat slot 0
__global__ void gate_act_kernel(const float* __restrict__ pre, long npos, float* __restrict__ tf, float* __restrict__ sg,
                                float* __restrict__ cat) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npos * C) return;
    long p = idx / C;
    int c = idx % C;
    float f = tanhf(pre[p * 64 + c]);
    float g = 1.f / (1.f + __expf(-pre[p * 64 + 32 + c]));
    tf[idx] = f; sg[idx] = g;
    cat[p * CAT + c] = f * g;
}
__global__ void gate_bwd_kernel(const float* __restrict__ dcat, const float* __restrict__ tf, const float* __restrict__ sg,
                                long npos, float* __restrict__ dpre) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npos * C) return;
    long p = idx / C;
    int c = idx % C;
    float dz = dcat[p * CAT + c], f = tf[idx], g = sg[idx];
    dpre[p * 64 + c] = dz * g * (1.f - f * f);
    dpre[p * 64 + 32 + c] = dz * f * g * (1.f - g);
}
// dx[bn][t'][c] = [t' < Tout] dxcat[(bn,t')][c] + [t' >= dil] (dxcat[(bn,t'-dil)][32+c] + dres[(bn,t'-dil)][c])
__global__ void col2im_kernel(const float* __restrict__ dxcat, const float* __restrict__ dres, long BN, int Tin, int Tout, int dil,
                              float* __restrict__ dx) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= BN * Tin * C) return;
    int c = idx % C, t = (idx / C) % Tin;
    long bn = idx / ((long)C * Tin);
    float v = 0.f;
    if (t < Tout) v += dxcat[(bn * Tout + t) * 64 + c];
    if (t >= dil) {
        long q = bn * Tout + (t - dil);
        v += dxcat[q * 64 + 32 + c];
        if (dres) v += dres[q * C + c];
    }
    dx[idx] = v;
}
// y = dropout(h) + x_in[t + dil];  BN partial sums.  block = 256 threads = 8 positions x 32 channels, grid-stride
__global__ __launch_bounds__(256) void mix_post_kernel(const float* __restrict__ h, const float* __restrict__ xin, long BN, int Tin,
                                                       int Tout, int dil, float drop_p, uint32_t seed_lo, uint32_t seed_hi,
                                                       uint32_t layer, float* __restrict__ mask, float* __restrict__ y,
                                                       float* __restrict__ partial) {
    __shared__ float red[8][64];
    const int c = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const long npos = BN * Tout;
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    float s1 = 0.f, s2 = 0.f;
    for (long p = (long)blockIdx.x * 8 + sub; p < npos; p += (long)gridDim.x * 8) {
        long bn = p / Tout;
        int t = p % Tout;
        float m = 1.f;
        if (drop_p > 0.f) {
            uint32_t r[4];
            long e = p * C + c;
            philox4x32((uint32_t)e, (uint32_t)(e >> 32), layer, 0xD409u, seed_lo, seed_hi, r);
            m = u32_to_unit(r[0]) >= drop_p ? keep_scale : 0.f;
            mask[p * C + c] = m;
        }
        float v = h[p * C + c] * m + xin[(bn * Tin + t + dil) * C + c];
        y[p * C + c] = v;
        s1 += v; s2 += v * v;
    }
    red[sub][c] = s1; red[sub][32 + c] = s2;
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = 0.f;
        for (int r = 0; r < 8; ++r) a += red[r][threadIdx.x];
        partial[(long)blockIdx.x * 64 + threadIdx.x] = a;
    }
}
// channel-last BN statistics -> stat [4][32] = scale, shift, mean, rstd (+ running stats)
__global__ __launch_bounds__(1024) void bn_cl_finalize_kernel(const float* __restrict__ partial, int nblk, double count,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ rm, float* __restrict__ rv, int training,
                                                             float momentum, float* __restrict__ stat) {
    __shared__ double ra[32][32], rq[32][32];
    const int c = threadIdx.x & 31, part = threadIdx.x >> 5;      // 1024 threads: 32 partial sums per channel
    double a = 0.0, q = 0.0;
    if (training)
        for (int i = part; i < nblk; i += 32) { a += partial[(long)i * 64 + c]; q += partial[(long)i * 64 + 32 + c]; }
    ra[part][c] = a; rq[part][c] = q;
    __syncthreads();
    if (part != 0) return;
    double mean, var;
    if (training) {
        for (int r = 1; r < 32; ++r) { a += ra[r][c]; q += rq[r][c]; }
        mean = a / count;
        var = fmax(q / count - mean * mean, 0.0);
        rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
        rv[c] = (1.f - momentum) * rv[c] + momentum * (float)(var * count / fmax(count - 1.0, 1.0));
    } else {
        mean = rm[c]; var = rv[c];
    }
    float rstd = (float)(1.0 / sqrt(var + 1e-5));
    float sc = gamma[c] * rstd;
    stat[c] = sc; stat[32 + c] = beta[c] - (float)mean * sc; stat[64 + c] = (float)mean; stat[96 + c] = rstd;
}
__global__ void bn_cl_apply_kernel(const float* __restrict__ y, long n, const float* __restrict__ stat, float* __restrict__ out) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    int c = idx % C;
    out[idx] = y[idx] * stat[c] + stat[32 + c];
}
// BN backward (channel-last): partial S1 = sum dy, S2 = sum dy*xhat
__global__ __launch_bounds__(256) void bn_cl_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y, long npos,
                                                               const float* __restrict__ stat, float* __restrict__ partial) {
    __shared__ float red[8][64];
    const int c = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const float mean = stat[64 + c], rstd = stat[96 + c];
    float s1 = 0.f, s2 = 0.f;
    for (long p = (long)blockIdx.x * 8 + sub; p < npos; p += (long)gridDim.x * 8) {
        float d = dy[p * C + c];
        s1 += d; s2 += d * (y[p * C + c] - mean) * rstd;
    }
    red[sub][c] = s1; red[sub][32 + c] = s2;
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = 0.f;
        for (int r = 0; r < 8; ++r) a += red[r][threadIdx.x];
        partial[(long)blockIdx.x * 64 + threadIdx.x] = a;
    }
}
__global__ __launch_bounds__(1024) void bn_cl_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, double count,
                                                                 const float* __restrict__ gamma, const float* __restrict__ stat,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ coef) {
    __shared__ double ra[32][32], rq[32][32];
    const int c = threadIdx.x & 31, part = threadIdx.x >> 5;
    double a = 0.0, q = 0.0;
    for (int i = part; i < nblk; i += 32) { a += partial[(long)i * 64 + c]; q += partial[(long)i * 64 + 32 + c]; }
    ra[part][c] = a; rq[part][c] = q;
    __syncthreads();
    if (part != 0) return;
    for (int r = 1; r < 32; ++r) { a += ra[r][c]; q += rq[r][c]; }
    dgamma[c] += (float)q; dbeta[c] += (float)a;
    coef[c] = (float)(a / count); coef[32 + c] = (float)(q / count); coef[64 + c] = gamma[c] * stat[96 + c];
}
// dpre = k (dy - m1 - xhat m2);  dh = dpre * mask
__global__ void bn_cl_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, long n, const float* __restrict__ stat,
                                       const float* __restrict__ coef, const float* __restrict__ mask, float* __restrict__ dpre,
                                       float* __restrict__ dh) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    int c = idx % C;
    float xh = (y[idx] - stat[64 + c]) * stat[96 + c];
    float d = coef[64 + c] * (dy[idx] - coef[c] - xh * coef[32 + c]);
    dpre[idx] = d;
    dh[idx] = mask ? d * mask[idx] : d;
}
// xh = relu(skip + bias_sum + h2)
__global__ void head_combine_kernel(const float* __restrict__ skip, const float* __restrict__ bsum, const float* __restrict__ h2,
                                    long n, float* __restrict__ xh) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    xh[idx] = fmaxf(skip[idx] + bsum[idx % CS] + h2[idx], 0.f);
}
struct Ptr8 { const float* p[8]; };
struct MPtr8 { float* p[8]; };
// out[i] = sum of the 8 skip-conv biases
__global__ void sum8_kernel(float* __restrict__ out, Ptr8 srcs, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += srcs.p[k][i];
    out[i] = s;
}
// dst_k[i] += v[i] for the 8 skip-conv bias gradients
__global__ void add_to8_kernel(MPtr8 dst, const float* __restrict__ v, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = v[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) dst.p[k][i] += x;
}
// in place: d *= (y > 0)
__global__ void relu_bwd_kernel(float* __restrict__ d, const float* __restrict__ y, long n) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n && !(y[idx] > 0.f)) d[idx] = 0.f;
}

// out[row % mod] += sum_j x[row*cols + j]   (bias gradient of the [B,12,N] prediction)
__global__ __launch_bounds__(256) void rowsum_mod_kernel(const float* __restrict__ x, int cols, int mod, float* __restrict__ out) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < cols; j += 256) s += x[row * cols + j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[row % mod], red[0] + red[1] + red[2] + red[3]);
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

// ---------------------------------------------------------------------------- buffer carving
struct Carver {
    float* base;
    long used = 0;
    explicit Carver(float* b) : base(b) {}
    float* take(long n) {
        float* p = base ? base + used : nullptr;
        used += (n + 3) & ~3L;            // keep 16-byte alignment
        return p;
    }
};

static inline int n8(int N) { return (N + 7) & ~7; }

struct Saved {
    float *x_in[NL], *cat[NL], *tf[NL], *sg[NL], *y[NL], *mask[NL], *bnstat[NL];
    float *Pstk, *PTstk, *Pa, *Madp, *rs, *cs;      // stacks: [3 supports f,b,a][B][N][N]
    uint16_t *P16, *PT16;                           // bf16 copies, rows zero-padded to N8 = roundup(N, 8) (bf16 hop operands)
    float *skip, *h1, *h2, *xh, *e1;
    long total;
};
Saved carve_saved(float* base, int B, int N, bool dropout) {
    Carver cv(base);
    Saved s;
    const long BN = (long)B * N;
    for (int i = 0; i < NL; ++i) {
        s.x_in[i] = cv.take(BN * TIN[i] * C);
        s.cat[i] = cv.take(BN * TOUT[i] * CAT);
        s.tf[i] = cv.take(BN * TOUT[i] * C);
        s.sg[i] = cv.take(BN * TOUT[i] * C);
        s.y[i] = i < NL - 1 ? cv.take(BN * TOUT[i] * C) : nullptr;
        s.mask[i] = (i < NL - 1 && dropout) ? cv.take(BN * TOUT[i] * C) : nullptr;
        s.bnstat[i] = cv.take(128);
    }
    s.Pstk = cv.take(3L * B * N * N);
    s.PTstk = cv.take(3L * B * N * N);
    s.P16 = (uint16_t*)cv.take(3L * B * N * n8(N) / 2);
    s.PT16 = (uint16_t*)cv.take(3L * B * N * n8(N) / 2);
    s.Pa = cv.take((long)N * N);
    s.Madp = cv.take((long)N * N);
    s.rs = cv.take(BN);
    s.cs = cv.take(BN);
    s.skip = cv.take(BN * CS);
    s.h1 = cv.take(BN * CE);
    s.h2 = cv.take(BN * CS);
    s.xh = cv.take(BN * CS);
    s.e1 = cv.take(BN * CE);
    s.total = cv.used;
    return s;
}
struct Work {
    float *wcat, *bcat, *dwcat, *dbcat;       // [8][64][64], [8][64]
    float *xcat, *pre, *h, *partial, *bsum;
    float *dcat, *dpre, *dxcat, *dh, *dres, *dxa, *dxb, *dskip, *dPstk, *dPa, *dM, *rf, *rb, *coef;
    float *d_e1, *d_xh, *d_h2, *d_h1;
    long total;
};
constexpr int BN_BLOCKS = 512;
Work carve_work(float* base, int B, int N, bool backward) {
    Carver cv(base);
    Work w;
    const long BN = (long)B * N;
    w.wcat = cv.take(NL * 64 * 64);
    w.bcat = cv.take(NL * 64);
    w.dwcat = cv.take(NL * 64 * 64);
    w.dbcat = cv.take(NL * 64);
    w.xcat = cv.take(BN * 12 * 64);
    w.pre = cv.take(BN * 12 * 64);
    w.h = cv.take(BN * 12 * C);
    w.partial = cv.take((long)BN_BLOCKS * 64);
    w.bsum = cv.take(CS);
    if (backward) {
        w.dcat = cv.take(BN * 12 * CAT);
        w.dpre = cv.take(BN * 12 * 64);
        w.dxcat = cv.take(BN * 12 * 64);
        w.dh = cv.take(BN * 12 * C);
        w.dres = cv.take(BN * 12 * C);
        w.dxa = cv.take(BN * 13 * C);
        w.dxb = cv.take(BN * 13 * C);
        w.dskip = cv.take(BN * CS);
        w.dPstk = cv.take(3L * B * N * N);
        w.dPa = cv.take((long)N * N);
        w.dM = cv.take((long)N * N);
        w.rf = cv.take(BN > N ? BN : N);
        w.rb = cv.take(BN);
        w.coef = cv.take(128);
        w.d_e1 = cv.take(BN * CE);
        w.d_xh = cv.take(BN * CS);
        w.d_h2 = cv.take(BN * CS);
        w.d_h1 = cv.take(BN * CE);
    }
    w.total = cv.used;
    return w;
}

int zero(float* p, long n, hipStream_t st) {
    if (hipMemsetAsync(p, 0, (size_t)n * sizeof(float), st) != hipSuccess) {
        step_set_error("gwnet: memset failed");
        return STEP_ERR_HIP;
    }
    return STEP_OK;
}

// One launch = the same diffusion hop for the three supports (two-level batch: i1 = support, i0 = sample).
// Forward hop  Out_s[b][w][n] = sum_v P_s[b][v][w] X_s[b][v][n]  reading slot src0 + sstep*s, writing dst0 + 2*s.
// bf16 mode: A(m=w, k=v) = PT16[w][v] (k contiguous, bf16) -- no transposition on the way into LDS.
int nconv_fwd3(const float* Pstk, const uint16_t* PT16, float* cat, int src0, int sstep, int dst0, int B, int N, int T, int bf16,
               hipStream_t st) {
    StepGemm g = gemm_desc(N, T * C, N, Pstk, 1, N, cat + src0 * C, (long)T * CAT, 1, cat + dst0 * C, (long)T * CAT);
    g.batch = 3 * B; g.batch0 = B;
    g.sab = (long)N * N; g.sab1 = (long)B * N * N;
    if (bf16) { g.A = PT16; g.a_bf16 = 1; g.sam = n8(N); g.sak = 1; g.sab = (long)N * n8(N); g.sab1 = (long)B * N * n8(N); }
    g.sbb = (long)N * T * CAT; g.sbb1 = (long)sstep * C;
    g.scb = (long)N * T * CAT; g.scb1 = 2L * C;
    g.b_nblk = C; g.b_nstride = CAT; g.c_nblk = C; g.c_nstride = CAT;
    g.compute_bf16 = bf16;
    return step_gemm_launch(g, st);
}
// Adjoint hop  dDst_s[b][v][n] += sum_w P_s[b][v][w] dSrc_s[b][w][n]  (reads the transposed stack: A(m=v,k=w) = PT[w][v]).
// dstep == 0: the three supports accumulate into the same slot -> atomics.
int nconv_bwd_data3(const float* PTstk, const uint16_t* P16, float* dcat, int src0, int dst0, int dstep, int B, int N, int T, int bf16,
                    hipStream_t st) {
    StepGemm g = gemm_desc(N, T * C, N, PTstk, 1, N, dcat + src0 * C, (long)T * CAT, 1, dcat + dst0 * C, (long)T * CAT);
    g.batch = 3 * B; g.batch0 = B;
    g.sab = (long)N * N; g.sab1 = (long)B * N * N;
    if (bf16) { g.A = P16; g.a_bf16 = 1; g.sam = n8(N); g.sak = 1; g.sab = (long)N * n8(N); g.sab1 = (long)B * N * n8(N); }
    g.sbb = (long)N * T * CAT; g.sbb1 = 2L * C;
    g.scb = (long)N * T * CAT; g.scb1 = (long)dstep * C;
    g.b_nblk = C; g.b_nstride = CAT; g.c_nblk = C; g.c_nstride = CAT;
    g.accumulate = dstep == 0 ? 2 : 1;
    g.compute_bf16 = bf16;
    return step_gemm_launch(g, st);
}
// dP_s[b][v][w] += sum_n X_s[b][v][n] * dOut_s[b][w][n]   (x from cat slot xs0 + xstep*s, dOut from dcat slot ds0 + 2*s)
int nconv_bwd_adj3(const float* cat, int xs0, int xstep, const float* dcat, int ds0, float* dPstk, int B, int N, int T,
                   int bf16, hipStream_t st) {
    StepGemm g = gemm_desc(N, N, T * C, cat + xs0 * C, (long)T * CAT, 1, dcat + ds0 * C, 1, (long)T * CAT, dPstk, N);
    g.batch = 3 * B; g.batch0 = B;
    g.sab = (long)N * T * CAT; g.sab1 = (long)xstep * C;
    g.sbb = (long)N * T * CAT; g.sbb1 = 2L * C;
    g.scb = (long)N * N; g.scb1 = (long)B * N * N;
    g.a_kblk = C; g.a_kstride = CAT; g.b_kblk = C; g.b_kstride = CAT;
    g.accumulate = 1;
    g.compute_bf16 = bf16;
    return step_gemm_launch(g, st);
}

}  // namespace

// =========================================================================================== C ABI
extern "C" long step_gwnet_saved_floats(int B, int N, int dropout) { return carve_saved(nullptr, B, N, dropout != 0).total; }
extern "C" long step_gwnet_work_floats(int B, int N, int backward) { return carve_work(nullptr, B, N, backward != 0).total; }
// element offset (in floats) of a saved item, for tests / the python side: item 0 = dropout mask, 1 = y (pre-BN),
// 2 = bnstat of layer `layer`
extern "C" long step_gwnet_saved_offset(int B, int N, int dropout, int item, int layer) {
    Saved s = carve_saved((float*)16, B, N, dropout != 0);
    float* p = item == 0 ? s.mask[layer] : item == 1 ? s.y[layer] : s.bnstat[layer];
    return p ? (long)(p - (float*)16) : -1;
}

extern "C" int step_gwnet_forward(const float* hist, int B, int N, int Cin, const float* hidden_last, const float* adj,
                                  const StepGwnetParams* p, int training, float dropout_p, uint64_t seed, float momentum,
                                  float* saved, float* work, float* pred, void* stream) {
    STEP_REQUIRE(hist && hidden_last && adj && p && saved && work && pred, "gwnet_forward: null argument");
    STEP_REQUIRE(B > 0 && N > 0 && Cin >= 2, "gwnet_forward: bad sizes B=%d N=%d C=%d", B, N, Cin);
    hipStream_t st = (hipStream_t)stream;
    const bool use_drop = training && dropout_p > 0.f;
    Saved S = carve_saved(saved, B, N, use_drop);
    Work W = carve_work(work, B, N, false);
    const long BN = (long)B * N;
    const int allbf16 = p->gemm_bf16;      // bf16 mode: every GEMM of this file on the bf16 matrix cores (the K=10 / dpred-transposed ones stay f32)

    start_conv_kernel<<<g1(BN * 13 * C), 256, 0, st>>>(hist, B, N, Cin, p->start_w, p->start_b, S.x_in[0]);
    STEP_LAUNCH_CHECK("start_conv");
    // supports (model.py:160-166)
    row_sums_kernel<<<(unsigned)BN, 256, 0, st>>>(adj, N, S.rs);
    col_sums_kernel<<<dim3(cdiv(N, 64), B), 256, 0, st>>>(adj, N, S.cs);
    const long NN = (long)N * N;
    rw_build_kernel<<<dim3(cdiv(N, 32), cdiv(N, 32), B), 256, 0, st>>>(adj, N, S.rs, S.cs, S.Pstk, S.Pstk + B * NN, S.PTstk,
                                                                       S.PTstk + B * NN);
    STEP_LAUNCH_CHECK("rw_build");
    {
        StepGemm g = gemm_desc(N, N, 10, p->nodevec1, 10, 1, p->nodevec2, N, 1, S.Madp, N);
        STEP_TRY(step_gemm_launch(g, st));
        softmax_relu_rows_kernel<<<N, 256, 0, st>>>(S.Madp, N, S.Pa);
        replicate_adp_kernel<<<g1(NN), 256, 0, st>>>(S.Pa, N, B, S.Pstk + 2 * B * NN, S.PTstk + 2 * B * NN);
        STEP_LAUNCH_CHECK("adp_softmax");
    }
    if (p->gemm_bf16) {
        const long rows = 3L * B * N;
        stacks_to_bf16_kernel<<<dim3((unsigned)cdiv(rows * (n8(N) / 2), 256), 2), 256, 0, st>>>(S.Pstk, S.PTstk, rows, N, n8(N), S.P16,
                                                                                               S.PT16);
        STEP_LAUNCH_CHECK("stacks_to_bf16");
    }
    {
        GatePtrs gp;
        for (int i = 0; i < NL; ++i) { gp.wf[i] = p->filter_w[i]; gp.bf[i] = p->filter_b[i]; gp.wg[i] = p->gate_w[i]; gp.bg[i] = p->gate_b[i]; }
        pack_gate_kernel<<<dim3(16, NL), 256, 0, st>>>(gp, W.wcat, W.bcat);
        STEP_LAUNCH_CHECK("pack_gate");
    }

    for (int i = 0; i < NL; ++i) {
        const int Tin = TIN[i], Tout = TOUT[i], dil = DIL[i];
        const long npos = BN * Tout;
        im2col_kernel<<<g1(npos * 64), 256, 0, st>>>(S.x_in[i], BN, Tin, Tout, dil, W.xcat);
        STEP_LAUNCH_CHECK("im2col");
        {   // pre = xcat @ Wcat^T + b
            StepGemm g = gemm_desc((int)npos, 64, 64, W.xcat, 64, 1, W.wcat + i * 4096, 1, 64, W.pre, 64);
            g.bias = W.bcat + i * 64;
            g.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g, st));
        }
        gate_act_kernel<<<g1(npos * C), 256, 0, st>>>(W.pre, npos, S.tf[i], S.sg[i], S.cat[i]);
        STEP_LAUNCH_CHECK("gate_act");
        {   // skip[bn][:] (+)= Wskip z[bn][Tout-1]    (biases are summed once in the head)
            StepGemm g = gemm_desc((int)BN, CS, C, S.cat[i] + (long)(Tout - 1) * CAT, (long)Tout * CAT, 1, p->skip_w[i], 1, C, S.skip, CS);
            g.accumulate = i == 0 ? 0 : 1;
            g.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g, st));
        }
        if (i == NL - 1) break;
        STEP_TRY(nconv_fwd3(S.Pstk, S.PT16, S.cat[i], 0, 0, 1, B, N, Tout, p->gemm_bf16, st));      // slots 1,3,5 = P_s z
        STEP_TRY(nconv_fwd3(S.Pstk, S.PT16, S.cat[i], 1, 2, 2, B, N, Tout, p->gemm_bf16, st));      // slots 2,4,6 = P_s (P_s z)
        {   // h = cat @ Wmix^T + b
            StepGemm g = gemm_desc((int)npos, C, CAT, S.cat[i], CAT, 1, p->gconv_w[i], 1, CAT, W.h, C);
            g.bias = p->gconv_b[i];
            g.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g, st));
        }
        int nblk = (int)((npos + 7) / 8);
        if (nblk > BN_BLOCKS) nblk = BN_BLOCKS;
        mix_post_kernel<<<nblk, 256, 0, st>>>(W.h, S.x_in[i], BN, Tin, Tout, dil, use_drop ? dropout_p : 0.f, (uint32_t)seed,
                                              (uint32_t)(seed >> 32), (uint32_t)i, S.mask[i], S.y[i], W.partial);
        STEP_LAUNCH_CHECK("mix_post");
        bn_cl_finalize_kernel<<<1, 1024, 0, st>>>(W.partial, nblk, (double)npos, p->bn_w[i], p->bn_b[i], p->bn_rm[i], p->bn_rv[i], training,
                                                momentum, S.bnstat[i]);
        bn_cl_apply_kernel<<<g1(npos * C), 256, 0, st>>>(S.y[i], npos * C, S.bnstat[i], S.x_in[i + 1]);
        STEP_LAUNCH_CHECK("bn_apply");
    }
    // head (model.py:215-220)
    {
        StepGemm g = gemm_desc((int)BN, CE, HID, hidden_last, HID, 1, p->fc_his0_w, 1, HID, S.h1, CE);
        g.bias = p->fc_his0_b; g.relu = 1;
        g.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g, st));
        StepGemm g2 = gemm_desc((int)BN, CS, CE, S.h1, CE, 1, p->fc_his2_w, 1, CE, S.h2, CS);
        g2.bias = p->fc_his2_b; g2.relu = 1;
        g2.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g2, st));
        // sum of the 8 skip biases
        Ptr8 sb;
        for (int i = 0; i < NL; ++i) sb.p[i] = p->skip_b[i];
        sum8_kernel<<<1, 256, 0, st>>>(W.bsum, sb, CS);
        head_combine_kernel<<<g1(BN * CS), 256, 0, st>>>(S.skip, W.bsum, S.h2, BN * CS, S.xh);
        STEP_LAUNCH_CHECK("head_combine");
        StepGemm g3 = gemm_desc((int)BN, CE, CS, S.xh, CS, 1, p->end1_w, 1, CS, S.e1, CE);
        g3.bias = p->end1_b; g3.relu = 1;
        g3.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g3, st));
        // pred[b][o][n] = e1[b,n,:] . W2[o,:] + b2[o]      (written directly as [B,12,N], step.py:65)
        StepGemm g4 = gemm_desc(N, OUT, CE, S.e1, CE, 1, p->end2_w, 1, CE, pred, 1);
        g4.batch = B; g4.sab = (long)N * CE; g4.scb = (long)OUT * N; g4.scn = N;
        g4.bias = p->end2_b;
        g4.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g4, st));
    }
    return STEP_OK;
}


extern "C" int step_gwnet_backward(const float* hist, int B, int N, int Cin, const float* hidden_last, const StepGwnetParams* p,
                                   const float* saved, float* work, const float* dpred, const StepGwnetParams* grads,
                                   float* dadj, int dropout, void* stream) {
    STEP_REQUIRE(hist && hidden_last && p && saved && work && dpred && grads && dadj, "gwnet_backward: null argument");
    STEP_REQUIRE(B > 0 && N > 0 && Cin >= 2, "gwnet_backward: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    Saved S = carve_saved((float*)saved, B, N, dropout != 0);
    Work W = carve_work(work, B, N, true);
    const long BN = (long)B * N;
    const int allbf16 = p->gemm_bf16;
    auto split_for = [](long) { return -1; };      // -1: step_gemm picks a split that fills the chip

    {
        GatePtrs gp;
        for (int i = 0; i < NL; ++i) { gp.wf[i] = p->filter_w[i]; gp.bf[i] = p->filter_b[i]; gp.wg[i] = p->gate_w[i]; gp.bg[i] = p->gate_b[i]; }
        pack_gate_kernel<<<dim3(16, NL), 256, 0, st>>>(gp, W.wcat, W.bcat);
        STEP_LAUNCH_CHECK("pack_gate");
    }
    STEP_TRY(zero(W.dwcat, NL * 4096, st));
    STEP_TRY(zero(W.dbcat, NL * 64, st));
    const long NN = (long)N * N;
    STEP_TRY(zero(W.dPstk, 3L * B * NN, st));

    // ---------------------------------------------------------------- head (model.py:215-220)
    {
        // d_e1[b,n,:] = sum_o dpred[b][o][n] W2[o,:], masked by relu(e1)
        StepGemm g = gemm_desc(N, CE, OUT, dpred, 1, N, p->end2_w, CE, 1, W.d_e1, CE);
        g.batch = B; g.sab = (long)OUT * N; g.scb = (long)N * CE;
        STEP_TRY(step_gemm_launch(g, st));
        // dW2[o,:] += sum_{b,n} dpred[b][o][n] e1[b,n,:]   (batches accumulate atomically)
        StepGemm gw = gemm_desc(OUT, CE, N, dpred, N, 1, S.e1, CE, 1, grads->end2_w, CE);
        gw.batch = B; gw.sab = (long)OUT * N; gw.sbb = (long)N * CE; gw.scb = 0; gw.accumulate = 2;
        STEP_TRY(step_gemm_launch(gw, st));
        rowsum_mod_kernel<<<B * OUT, 256, 0, st>>>(dpred, N, OUT, grads->end2_b);
        STEP_LAUNCH_CHECK("end2_bias_grad");
        relu_bwd_kernel<<<g1(BN * CE), 256, 0, st>>>(W.d_e1, S.e1, BN * CE);
        StepGemm gw1 = gemm_desc(CE, CS, (int)BN, W.d_e1, 1, CE, S.xh, CS, 1, grads->end1_w, CS);
        gw1.accumulate = 2; gw1.splitk = split_for(BN);
        gw1.a_rowsum = grads->end1_b;                         // bias gradient = row sums of the same A
        gw1.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw1, st));
        StepGemm gx = gemm_desc((int)BN, CS, CE, W.d_e1, CE, 1, p->end1_w, CS, 1, W.d_xh, CS);
        gx.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gx, st));
        relu_bwd_kernel<<<g1(BN * CS), 256, 0, st>>>(W.d_xh, S.xh, BN * CS);       // = d skip = d h2 (pre-mask)
        {   // the 8 skip biases all receive colsum(d skip)
            STEP_TRY(zero(W.bsum, CS, st));
            STEP_TRY(step_colsum_launch(W.d_xh, BN, CS, CS, W.bsum, st));
            MPtr8 gb;
            for (int i = 0; i < NL; ++i) gb.p[i] = grads->skip_b[i];
            add_to8_kernel<<<1, 256, 0, st>>>(gb, W.bsum, CS);
        }
        // fc_his
        if (hipMemcpyAsync(W.d_h2, W.d_xh, (size_t)BN * CS * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
            step_set_error("gwnet_backward: copy failed");
            return STEP_ERR_HIP;
        }
        relu_bwd_kernel<<<g1(BN * CS), 256, 0, st>>>(W.d_h2, S.h2, BN * CS);
        StepGemm gw2 = gemm_desc(CS, CE, (int)BN, W.d_h2, 1, CS, S.h1, CE, 1, grads->fc_his2_w, CE);
        gw2.accumulate = 2; gw2.splitk = split_for(BN);
        gw2.a_rowsum = grads->fc_his2_b;
        gw2.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw2, st));
        StepGemm gh1 = gemm_desc((int)BN, CE, CS, W.d_h2, CS, 1, p->fc_his2_w, CE, 1, W.d_h1, CE);
        gh1.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gh1, st));
        relu_bwd_kernel<<<g1(BN * CE), 256, 0, st>>>(W.d_h1, S.h1, BN * CE);
        StepGemm gw0 = gemm_desc(CE, HID, (int)BN, W.d_h1, 1, CE, hidden_last, HID, 1, grads->fc_his0_w, HID);
        gw0.accumulate = 2; gw0.splitk = split_for(BN);
        gw0.a_rowsum = grads->fc_his0_b;
        gw0.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw0, st));
    }

    // ---------------------------------------------------------------- WaveNet layers, reversed
    float* dx_next = nullptr;
    float* dxbuf[2] = {W.dxa, W.dxb};
    for (int i = NL - 1; i >= 0; --i) {
        const int Tin = TIN[i], Tout = TOUT[i], dil = DIL[i];
        const long npos = BN * Tout;
        const float* cat = S.cat[i];
        if (i < NL - 1) {
            int nblk = (int)((npos + 7) / 8);
            if (nblk > BN_BLOCKS) nblk = BN_BLOCKS;
            bn_cl_bwd_reduce_kernel<<<nblk, 256, 0, st>>>(dx_next, S.y[i], npos, S.bnstat[i], W.partial);
            bn_cl_bwd_finalize_kernel<<<1, 1024, 0, st>>>(W.partial, nblk, (double)npos, p->bn_w[i], S.bnstat[i], grads->bn_w[i], grads->bn_b[i], W.coef);
            bn_cl_bwd_apply_kernel<<<g1(npos * C), 256, 0, st>>>(dx_next, S.y[i], npos * C, S.bnstat[i], W.coef, S.mask[i], W.dres, W.dh);
            STEP_LAUNCH_CHECK("bn_bwd");
            // mix (gconv.i.mlp) gradients
            StepGemm gw = gemm_desc(C, CAT, (int)npos, W.dh, 1, C, cat, CAT, 1, grads->gconv_w[i], CAT);
            gw.accumulate = 2; gw.splitk = split_for(npos);
            gw.a_rowsum = grads->gconv_b[i];
            gw.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw, st));
            StepGemm gd = gemm_desc((int)npos, CAT, C, W.dh, C, 1, p->gconv_w[i], CAT, 1, W.dcat, CAT);
            gd.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gd, st));
            // diffusion hops, the three supports per launch: slots (1,2) <- P_f, (3,4) <- P_b, (5,6) <- P_a
            STEP_TRY(nconv_bwd_data3(S.PTstk, S.P16, W.dcat, 2, 1, 2, B, N, Tout, p->gemm_bf16, st));          // d_x1 += P (d_x2)
            STEP_TRY(nconv_bwd_adj3(cat, 1, 2, W.dcat, 2, W.dPstk, B, N, Tout, p->gemm_bf16, st));      // dP += x1 (x) d_x2
            STEP_TRY(nconv_bwd_adj3(cat, 0, 0, W.dcat, 1, W.dPstk, B, N, Tout, p->gemm_bf16, st));      // dP += z  (x) d_x1
            STEP_TRY(nconv_bwd_data3(S.PTstk, S.P16, W.dcat, 1, 0, 0, B, N, Tout, p->gemm_bf16, st));          // d_z  += sum_s P_s (d_x1_s)
        } else {
            STEP_TRY(zero(W.dcat, npos * CAT, st));
        }
        // skip connection: gradient enters z at the last time index only
        {
            StepGemm g = gemm_desc((int)BN, C, CS, W.d_xh, CS, 1, p->skip_w[i], C, 1, W.dcat + (long)(Tout - 1) * CAT, (long)Tout * CAT);
            g.accumulate = 1;
            g.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g, st));
            StepGemm gw = gemm_desc(CS, C, (int)BN, W.d_xh, 1, CS, cat + (long)(Tout - 1) * CAT, (long)Tout * CAT, 1, grads->skip_w[i], C);
            gw.accumulate = 2; gw.splitk = split_for(BN);
            gw.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw, st));
        }
        // gated TCN
        gate_bwd_kernel<<<g1(npos * C), 256, 0, st>>>(W.dcat, S.tf[i], S.sg[i], npos, W.dpre);
        im2col_kernel<<<g1(npos * 64), 256, 0, st>>>(S.x_in[i], BN, Tin, Tout, dil, W.xcat);
        STEP_LAUNCH_CHECK("gate_bwd");
        {
            StepGemm gw = gemm_desc(64, 64, (int)npos, W.dpre, 1, 64, W.xcat, 64, 1, W.dwcat + i * 4096, 64);
            gw.accumulate = 2; gw.splitk = split_for(npos);
            gw.a_rowsum = W.dbcat + i * 64;
            gw.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw, st));
            StepGemm gx = gemm_desc((int)npos, 64, 64, W.dpre, 64, 1, W.wcat + i * 4096, 64, 1, W.dxcat, 64);
            gx.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gx, st));
        }
        float* dx = dxbuf[i & 1];
        col2im_kernel<<<g1(BN * Tin * C), 256, 0, st>>>(W.dxcat, i < NL - 1 ? W.dres : nullptr, BN, Tin, Tout, dil, dx);
        STEP_LAUNCH_CHECK("col2im");
        dx_next = dx;
    }
    {
        GatePtrs gg;
        for (int i = 0; i < NL; ++i) { gg.wf[i] = grads->filter_w[i]; gg.bf[i] = grads->filter_b[i]; gg.wg[i] = grads->gate_w[i]; gg.bg[i] = grads->gate_b[i]; }
        unpack_gate_grad_kernel<<<dim3(16, NL), 256, 0, st>>>(W.dwcat, W.dbcat, gg);
    }
    start_conv_bwd_kernel<<<128, 256, 0, st>>>(hist, B, N, Cin, dx_next, grads->start_w, grads->start_b);
    STEP_LAUNCH_CHECK("start_conv_bwd");

    // ---------------------------------------------------------------- supports
    {   // adaptive adjacency softmax(relu(E1 E2), dim=1); its gradient is the sum over samples of stack slot 2
        sum_batches_kernel<<<g1(NN), 256, 0, st>>>(W.dPstk + 2 * B * NN, NN, B, W.dPa);
        row_dot_kernel<<<N, 256, 0, st>>>(W.dPa, S.Pa, N, W.rf);
        softmax_relu_bwd_kernel<<<g1(NN), 256, 0, st>>>(S.Madp, S.Pa, W.dPa, W.rf, N, W.dM);
        STEP_LAUNCH_CHECK("adp_bwd");
        StepGemm g1_ = gemm_desc(N, 10, N, W.dM, N, 1, p->nodevec2, 1, N, grads->nodevec1, 10);
        g1_.accumulate = 1;
        STEP_TRY(step_gemm_launch(g1_, st));
        StepGemm g2_ = gemm_desc(10, N, N, p->nodevec1, 1, 10, W.dM, N, 1, grads->nodevec2, N);
        g2_.accumulate = 1;
        STEP_TRY(step_gemm_launch(g2_, st));
    }
    row_dot_kernel<<<(unsigned)BN, 256, 0, st>>>(W.dPstk, S.Pstk, N, W.rf);
    row_dot_kernel<<<(unsigned)BN, 256, 0, st>>>(W.dPstk + B * NN, S.Pstk + B * NN, N, W.rb);
    rw_bwd_kernel<<<dim3(cdiv(N, 32), cdiv(N, 32), B), 256, 0, st>>>(W.dPstk, W.dPstk + B * NN, N, S.rs, S.cs, W.rf, W.rb, dadj);
    STEP_LAUNCH_CHECK("rw_bwd");
    return STEP_OK;
}
