// TSFormer pre-training path (reference: step/step_arch/tsformer/tsformer.py:71-160,180-188): the kernels that,
// together with step_gemm, make up forward AND backward of the masked-autoencoder stage in exact f32:
// token gather/scatter by the mask index lists, positional-embedding add, decoder-input assembly,
// LayerNorm, multi-head self-attention (scores/softmax/PV recomputed in the backward from the saved row
// statistics -- no [T,T] matrix is ever stored), dropout with a counter-based (replayable) mask, ReLU.
// This first version is position-wise VALU code plus f32-MFMA GEMMs (not fused like the forecasting-mode
// encoder); it exists so that config C3 runs natively with parity against the oracle.
#include "common.h"
#include "step_internal.h"

namespace {

constexpr int D = 96, H = 4, DH = 24;

__device__ __forceinline__ float keep_scale(uint32_t lo, uint32_t hi, uint32_t site, long elem, float p) {
    uint32_t r[4];
    const long blk = elem >> 2;
    philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), site, 0xD20Fu, lo, hi, r);
    return u32_to_unit(r[elem & 3]) >= p ? 1.f / (1.f - p) : 0.f;
}

// y = x * mask (in place or out of place); the same call with the same (seed, site) replays the mask (backward)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p, uint32_t lo, uint32_t hi,
                               uint32_t site) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] * keep_scale(lo, hi, site, i, p);
}
// out = a + dropout(b)
__global__ void add_dropout_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n, float p,
                                   uint32_t lo, uint32_t hi, uint32_t site) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + (p > 0.f ? b[i] * keep_scale(lo, hi, site, i, p) : b[i]);
}
// x[s][p][:] += vec[idx ? idx[p] : p][:]
__global__ void add_rows_kernel(float* __restrict__ x, long S, int P, const float* __restrict__ vec, const int* __restrict__ idx) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= S * P * D) return;
    int f = i % D, p = (i / D) % P;
    x[i] += vec[(long)(idx ? idx[p] : p) * D + f];
}
// dvec[idx ? idx[p] : p][f] += sum_s dx[s][p][f]        one block per token position p
__global__ __launch_bounds__(384) void sum_over_seq_kernel(const float* __restrict__ dx, long S, int P, int p_off, int p_cnt, int ldp,
                                                           const int* __restrict__ idx, float* __restrict__ dvec) {
    __shared__ float red[4][D];
    const int pj = blockIdx.x;                 // 0..p_cnt
    const int f = threadIdx.x % D, part = threadIdx.x / D;      // 4 partial sums per feature
    float s = 0.f;
    for (long q = part; q < S; q += 4) s += dx[(q * ldp + p_off + pj) * D + f];
    red[part][f] = s;
    __syncthreads();
    if (part == 0) atomicAdd(&dvec[(long)(idx ? idx[pj] : pj) * D + f], red[0][f] + red[1][f] + red[2][f] + red[3][f]);
}
// dst[s][t][:] = src[s][idx[t]][:] * scale
__global__ void token_gather_kernel(const float* __restrict__ src, long S, int P, const int* __restrict__ idx, int T, float scale,
                                    float* __restrict__ dst) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= S * T * D) return;
    int f = i % D, t = (i / D) % T;
    long s = i / ((long)D * T);
    dst[i] = src[(s * P + idx[t]) * D + f] * scale;
}
// dsrc[s][idx[t]][:] = ddst[s][t][:] * scale   (dsrc pre-zeroed; idx entries are distinct)
__global__ void token_scatter_kernel(const float* __restrict__ ddst, long S, int P, const int* __restrict__ idx, int T, float scale,
                                     float* __restrict__ dsrc) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= S * T * D) return;
    int f = i % D, t = (i / D) % T;
    long s = i / ((long)D * T);
    dsrc[(s * P + idx[t]) * D + f] = ddst[i] * scale;
}
// decoder input (tsformer.py:120-127 + transformer_layers.py:15): out[s][t] = sqrt(96) * (t < Pu ? z[s][t]
//                                                                   : dropout(mask_token + pos[midx[t-Pu]]))
__global__ void dec_input_kernel(const float* __restrict__ z, const float* __restrict__ mask_token, const float* __restrict__ pos,
                                 const int* __restrict__ midx, long S, int P, int Pu, float scale, float p, uint32_t lo, uint32_t hi,
                                 uint32_t site, float* __restrict__ out) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= S * P * D) return;
    int f = i % D, t = (i / D) % P;
    long s = i / ((long)D * P);
    float v;
    if (t < Pu) {
        v = z[(s * Pu + t) * D + f];
    } else {
        v = mask_token[f] + pos[(long)midx[t - Pu] * D + f];
        if (p > 0.f) v *= keep_scale(lo, hi, site, i, p);
    }
    out[i] = v * scale;
}
// backward: dz[s][t] = scale * dout[s][t] (t < Pu);  dm[s][j] = scale * mask * dout[s][Pu + j]   (dm: [S][Pm][96] scratch)
__global__ void dec_input_bwd_kernel(const float* __restrict__ dout, long S, int P, int Pu, float scale, float p, uint32_t lo, uint32_t hi,
                                     uint32_t site, float* __restrict__ dz, float* __restrict__ dm) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= S * P * D) return;
    int f = i % D, t = (i / D) % P;
    long s = i / ((long)D * P);
    float g = dout[i] * scale;
    if (t < Pu) {
        dz[(s * Pu + t) * D + f] = g;
    } else {
        if (p > 0.f) g *= keep_scale(lo, hi, site, i, p);
        dm[(s * (P - Pu) + (t - Pu)) * D + f] = g;
    }
}

// ------------------------------------------------------------------------------------ LayerNorm(96), one wave per row
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long R, const float* __restrict__ g,
                                                     const float* __restrict__ b, float* __restrict__ y, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + row * D;
    float v0 = xr[lane], v1 = lane < D - 64 ? xr[64 + lane] : 0.f;
    float mean = wave_sum(v0 + v1) * (1.f / D);
    float d0 = v0 - mean, d1 = lane < D - 64 ? v1 - mean : 0.f;
    float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / D) + 1e-5f);
    y[row * D + lane] = d0 * rstd * g[lane] + b[lane];
    if (lane < D - 64) y[row * D + 64 + lane] = d1 * rstd * g[64 + lane] + b[64 + lane];
    if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}
// dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)),  dyg = dy * gamma; per-block partial dgamma/dbeta -> atomics
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, long R,
                                                     const float* __restrict__ g, const float* __restrict__ stats,
                                                     float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sg[4][D], sb[4][D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float ag0 = 0.f, ag1 = 0.f, ab0 = 0.f, ab1 = 0.f;
    for (long row = (long)blockIdx.x * 4 + w; row < R; row += (long)gridDim.x * 4) {
        const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
        const bool hi = lane < D - 64;
        float xh0 = (x[row * D + lane] - mean) * rstd, xh1 = hi ? (x[row * D + 64 + lane] - mean) * rstd : 0.f;
        float y0 = dy[row * D + lane], y1 = hi ? dy[row * D + 64 + lane] : 0.f;
        float q0 = y0 * g[lane], q1 = hi ? y1 * g[64 + lane] : 0.f;
        float m1 = wave_sum(q0 + q1) * (1.f / D);
        float m2 = wave_sum(q0 * xh0 + q1 * xh1) * (1.f / D);
        dx[row * D + lane] = rstd * (q0 - m1 - xh0 * m2);
        if (hi) dx[row * D + 64 + lane] = rstd * (q1 - m1 - xh1 * m2);
        ag0 += y0 * xh0; ab0 += y0;
        ag1 += y1 * xh1; ab1 += y1;
    }
    sg[w][lane] = ag0; sb[w][lane] = ab0;
    if (lane < D - 64) { sg[w][64 + lane] = ag1; sb[w][64 + lane] = ab1; }
    __syncthreads();
    if (threadIdx.x < D) {
        const int f = threadIdx.x;
        atomicAdd(&dgamma[f], sg[0][f] + sg[1][f] + sg[2][f] + sg[3][f]);
        atomicAdd(&dbeta[f], sb[0][f] + sb[1][f] + sb[2][f] + sb[3][f]);
    }
}

// ------------------------------------------------------------------------------------ self-attention, one block per (sequence, head)
// qkv [S][T][288] (q | k | v, head h = columns 24h..24h+23 of each third), out [S][T][96], stats [S][H][T][2] = (row max, row sum)
// attention-probability dropout uses element index ((s*H + h)*T + i)*T + j
template <bool BWD>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ qkv, long S, int T, float p, uint32_t lo, uint32_t hi,
                                                   uint32_t site, float* __restrict__ out, float* __restrict__ stats,
                                                   const float* __restrict__ dout, float* __restrict__ dqkv) {
    extern __shared__ float sm[];
    float* sq = sm;                       // [T][DH]
    float* sk = sq + T * DH;
    float* sv = sk + T * DH;
    float* sdo = sv + T * DH;             // BWD only: dO  [T][DH]
    float* sdl = sdo + T * DH;            // BWD only: delta_i = sum_d dO_i O_i  [T]
    float* smx = sdl + T;                 // [T] row max, [T] 1/row sum
    float* sinv = smx + T;
    const long s = blockIdx.x / H;
    const int h = blockIdx.x % H;
    const float scale = 0.20412414523193154f;      // 1/sqrt(24)
    const float* base = qkv + s * (long)T * 288;
    for (int e = threadIdx.x; e < T * DH; e += 256) {
        int t = e / DH, d = e % DH;
        sq[e] = base[(long)t * 288 + h * DH + d];
        sk[e] = base[(long)t * 288 + 96 + h * DH + d];
        sv[e] = base[(long)t * 288 + 192 + h * DH + d];
        if (BWD) sdo[e] = dout[(s * T + t) * D + h * DH + d];
    }
    __syncthreads();
    const long srow = (s * H + h) * (long)T;
    if (!BWD) {
        for (int i = threadIdx.x; i < T; i += 256) {
            float q[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) q[d] = sq[i * DH + d] * scale;
            float mx = -INFINITY;
            for (int j = 0; j < T; ++j) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) a += q[d] * sk[j * DH + d];
                mx = fmaxf(mx, a);
            }
            float l = 0.f, o[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] = 0.f;
            for (int j = 0; j < T; ++j) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) a += q[d] * sk[j * DH + d];
                float e = __expf(a - mx);
                l += e;
                if (p > 0.f) e *= keep_scale(lo, hi, site, (srow + i) * T + j, p);
#pragma unroll
                for (int d = 0; d < DH; ++d) o[d] += e * sv[j * DH + d];
            }
            const float inv = 1.f / l;
#pragma unroll
            for (int d = 0; d < DH; ++d) out[(s * T + i) * D + h * DH + d] = o[d] * inv;
            stats[(srow + i) * 2] = mx;
            stats[(srow + i) * 2 + 1] = l;
        }
    } else {
        // delta_i and cached row statistics
        for (int i = threadIdx.x; i < T; i += 256) {
            float dl = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) dl += sdo[i * DH + d] * out[(s * T + i) * D + h * DH + d];
            sdl[i] = dl;
            smx[i] = stats[(srow + i) * 2];
            sinv[i] = 1.f / stats[(srow + i) * 2 + 1];
        }
        __syncthreads();
        float* db = dqkv + s * (long)T * 288;
        // pass A (thread = query i): dQ_i = scale * sum_j dS_ij K_j,  dS_ij = P_ij (mask_ij dP_ij - delta_i)
        for (int i = threadIdx.x; i < T; i += 256) {
            float q[DH], dq[DH], dout_i[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) { q[d] = sq[i * DH + d] * scale; dq[d] = 0.f; dout_i[d] = sdo[i * DH + d]; }
            for (int j = 0; j < T; ++j) {
                float a = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) { a += q[d] * sk[j * DH + d]; dp += dout_i[d] * sv[j * DH + d]; }
                const float pij = __expf(a - smx[i]) * sinv[i];
                if (p > 0.f) dp *= keep_scale(lo, hi, site, (srow + i) * T + j, p);
                const float ds = pij * (dp - sdl[i]);
#pragma unroll
                for (int d = 0; d < DH; ++d) dq[d] += ds * sk[j * DH + d];
            }
#pragma unroll
            for (int d = 0; d < DH; ++d) db[(long)i * 288 + h * DH + d] = dq[d] * scale;
        }
        // pass B (thread = key j): dK_j = scale * sum_i dS_ij Q_i,  dV_j = sum_i mask_ij P_ij dO_i
        for (int j = threadIdx.x; j < T; j += 256) {
            float kj[DH], vj[DH], dk[DH], dv[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) { kj[d] = sk[j * DH + d]; vj[d] = sv[j * DH + d]; dk[d] = 0.f; dv[d] = 0.f; }
            for (int i = 0; i < T; ++i) {
                float a = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) { a += sq[i * DH + d] * kj[d]; dp += sdo[i * DH + d] * vj[d]; }
                const float pij = __expf(a * scale - smx[i]) * sinv[i];
                float m = 1.f;
                if (p > 0.f) m = keep_scale(lo, hi, site, (srow + i) * T + j, p);
                const float ds = pij * (dp * m - sdl[i]);
                const float pm = pij * m;
#pragma unroll
                for (int d = 0; d < DH; ++d) { dk[d] += ds * sq[i * DH + d]; dv[d] += pm * sdo[i * DH + d]; }
            }
#pragma unroll
            for (int d = 0; d < DH; ++d) {
                db[(long)j * 288 + 96 + h * DH + d] = dk[d] * scale;
                db[(long)j * 288 + 192 + h * DH + d] = dv[d];
            }
        }
    }
}

__global__ void relu_mask_kernel(float* __restrict__ d, const float* __restrict__ y, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && !(y[i] > 0.f)) d[i] = 0.f;
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

#define SEED_LO(s) ((uint32_t)(s))
#define SEED_HI(s) ((uint32_t)((s) >> 32))

extern "C" int step_pt_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint32_t site, void* stream) {
    STEP_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "pt_dropout: bad arguments");
    dropout_kernel<<<g1(n), 256, 0, (hipStream_t)stream>>>(x, y, n, p, SEED_LO(seed), SEED_HI(seed), site);
    STEP_LAUNCH_CHECK("pt_dropout");
    return STEP_OK;
}
extern "C" int step_pt_add_dropout(const float* a, const float* b, float* out, long n, float p, uint64_t seed, uint32_t site, void* stream) {
    STEP_REQUIRE(a && b && out && n > 0 && p >= 0.f && p < 1.f, "pt_add_dropout: bad arguments");
    add_dropout_kernel<<<g1(n), 256, 0, (hipStream_t)stream>>>(a, b, out, n, p, SEED_LO(seed), SEED_HI(seed), site);
    STEP_LAUNCH_CHECK("pt_add_dropout");
    return STEP_OK;
}
extern "C" int step_pt_add_rows(float* x, long S, int P, const float* vec, const int* idx, void* stream) {
    STEP_REQUIRE(x && vec && S > 0 && P > 0, "pt_add_rows: bad arguments");
    add_rows_kernel<<<g1(S * P * D), 256, 0, (hipStream_t)stream>>>(x, S, P, vec, idx);
    STEP_LAUNCH_CHECK("pt_add_rows");
    return STEP_OK;
}
extern "C" int step_pt_sum_over_seq(const float* dx, long S, int ldp, int p_off, int p_cnt, const int* idx, float* dvec, void* stream) {
    STEP_REQUIRE(dx && dvec && S > 0 && p_cnt > 0, "pt_sum_over_seq: bad arguments");
    sum_over_seq_kernel<<<p_cnt, 384, 0, (hipStream_t)stream>>>(dx, S, 0, p_off, p_cnt, ldp, idx, dvec);
    STEP_LAUNCH_CHECK("pt_sum_over_seq");
    return STEP_OK;
}
extern "C" int step_pt_token_gather(const float* src, long S, int P, const int* idx, int T, float scale, float* dst, void* stream) {
    STEP_REQUIRE(src && idx && dst && S > 0 && T > 0, "pt_token_gather: bad arguments");
    token_gather_kernel<<<g1(S * T * D), 256, 0, (hipStream_t)stream>>>(src, S, P, idx, T, scale, dst);
    STEP_LAUNCH_CHECK("pt_token_gather");
    return STEP_OK;
}
extern "C" int step_pt_token_scatter(const float* ddst, long S, int P, const int* idx, int T, float scale, float* dsrc, void* stream) {
    STEP_REQUIRE(ddst && idx && dsrc && S > 0 && T > 0, "pt_token_scatter: bad arguments");
    token_scatter_kernel<<<g1(S * T * D), 256, 0, (hipStream_t)stream>>>(ddst, S, P, idx, T, scale, dsrc);
    STEP_LAUNCH_CHECK("pt_token_scatter");
    return STEP_OK;
}
extern "C" int step_pt_dec_input(const float* z, const float* mask_token, const float* pos, const int* midx, long S, int P, int Pu,
                                 float p, uint64_t seed, uint32_t site, float* out, void* stream) {
    STEP_REQUIRE(z && mask_token && pos && midx && out && S > 0 && P > Pu && Pu > 0, "pt_dec_input: bad arguments");
    dec_input_kernel<<<g1(S * P * D), 256, 0, (hipStream_t)stream>>>(z, mask_token, pos, midx, S, P, Pu, 9.797958971132712f, p,
                                                                     SEED_LO(seed), SEED_HI(seed), site, out);
    STEP_LAUNCH_CHECK("pt_dec_input");
    return STEP_OK;
}
extern "C" int step_pt_dec_input_bwd(const float* dout, long S, int P, int Pu, float p, uint64_t seed, uint32_t site, float* dz,
                                     float* dm, void* stream) {
    STEP_REQUIRE(dout && dz && dm && S > 0 && P > Pu && Pu > 0, "pt_dec_input_bwd: bad arguments");
    dec_input_bwd_kernel<<<g1(S * P * D), 256, 0, (hipStream_t)stream>>>(dout, S, P, Pu, 9.797958971132712f, p, SEED_LO(seed),
                                                                         SEED_HI(seed), site, dz, dm);
    STEP_LAUNCH_CHECK("pt_dec_input_bwd");
    return STEP_OK;
}
extern "C" int step_pt_layernorm_fwd(const float* x, long R, const float* g, const float* b, float* y, float* stats, void* stream) {
    STEP_REQUIRE(x && g && b && y && stats && R > 0, "pt_layernorm_fwd: bad arguments");
    ln_fwd_kernel<<<(unsigned)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, R, g, b, y, stats);
    STEP_LAUNCH_CHECK("pt_layernorm_fwd");
    return STEP_OK;
}
extern "C" int step_pt_layernorm_bwd(const float* dy, const float* x, long R, const float* g, const float* stats, float* dx, float* dgamma,
                                     float* dbeta, void* stream) {
    STEP_REQUIRE(dy && x && g && stats && dx && dgamma && dbeta && R > 0, "pt_layernorm_bwd: bad arguments");
    long blocks = (R + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    ln_bwd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(dy, x, R, g, stats, dx, dgamma, dbeta);
    STEP_LAUNCH_CHECK("pt_layernorm_bwd");
    return STEP_OK;
}
static void attn_attrs() {
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute((const void*)attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
}
extern "C" int step_pt_attention_fwd(const float* qkv, long S, int T, float p, uint64_t seed, uint32_t site, float* out, float* stats,
                                     void* stream) {
    STEP_REQUIRE(qkv && out && stats && S > 0 && T > 0 && T <= 336, "pt_attention_fwd: bad arguments (T=%d; at most 336 tokens, the limit of the backward)", T);
    attn_attrs();
    size_t lds = (size_t)(3 * T * DH) * sizeof(float);
    attn_kernel<false><<<(unsigned)(S * H), 256, lds, (hipStream_t)stream>>>(qkv, S, T, p, SEED_LO(seed), SEED_HI(seed), site, out, stats,
                                                                            nullptr, nullptr);
    STEP_LAUNCH_CHECK("pt_attention_fwd");
    return STEP_OK;
}
extern "C" int step_pt_attention_bwd(const float* qkv, const float* out, const float* dout, const float* stats, long S, int T, float p,
                                     uint64_t seed, uint32_t site, float* dqkv, void* stream) {
    STEP_REQUIRE(qkv && out && dout && stats && dqkv && S > 0 && T > 0 && T <= 336, "pt_attention_bwd: bad arguments (T=%d)", T);
    size_t lds = (size_t)(4 * T * DH + 3 * T) * sizeof(float);
    attn_attrs();
    attn_kernel<true><<<(unsigned)(S * H), 256, lds, (hipStream_t)stream>>>(qkv, S, T, p, SEED_LO(seed), SEED_HI(seed), site,
                                                                           const_cast<float*>(out), const_cast<float*>(stats), dout, dqkv);
    STEP_LAUNCH_CHECK("pt_attention_bwd");
    return STEP_OK;
}
extern "C" int step_pt_relu_mask(float* d, const float* y, long n, void* stream) {
    STEP_REQUIRE(d && y && n > 0, "pt_relu_mask: bad arguments");
    relu_mask_kernel<<<g1(n), 256, 0, (hipStream_t)stream>>>(d, y, n);
    STEP_LAUNCH_CHECK("pt_relu_mask");
    return STEP_OK;
}
extern "C" int step_colsum(const float* x, long rows, int cols, long ld, float* out, void* stream) {
    STEP_REQUIRE(x && out && rows > 0 && cols > 0, "colsum: bad arguments");
    return step_colsum_launch(x, rows, cols, ld, out, (hipStream_t)stream);
}
