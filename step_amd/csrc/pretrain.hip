// TSFormer pre-training path (reference: step/step_arch/tsformer/tsformer.py:71-160,180-188): the kernels that,
// together with step_gemm, make up forward AND backward of the masked-autoencoder stage in exact f32:
// token gather/scatter by the mask index lists, positional-embedding add, decoder-input assembly,
// LayerNorm, multi-head self-attention (scores/softmax/PV recomputed in the backward from the saved row
// statistics -- no [T,T] matrix is ever stored), dropout with a counter-based (replayable) mask, ReLU.
// This first version is position-wise VALU code plus f32-MFMA GEMMs (not fused like the forecasting-mode
// encoder); it exists so that config C3 runs natively with parity against the oracle.
#include "common.h"
#include <stdlib.h>
#include "step_internal.h"

namespace {

constexpr int D = 96, H = 4, DH = 24;

__device__ __forceinline__ float keep_scale(uint32_t lo, uint32_t hi, uint32_t site, long elem, float p) {
    uint32_t r[4];
    const long blk = elem >> 2;
    philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), site, 0xD20Fu, lo, hi, r);
    return u32_to_unit(r[elem & 3]) >= p ? 1.f / (1.f - p) : 0.f;
}

// keep multipliers of the 4 consecutive elements 4*blk .. 4*blk+3 (one Philox call)
__device__ __forceinline__ void keep_scale4(uint32_t lo, uint32_t hi, uint32_t site, long blk, float p, float* m) {
    uint32_t r[4];
    philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), site, 0xD20Fu, lo, hi, r);
    const float ks = 1.f / (1.f - p);
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = u32_to_unit(r[j]) >= p ? ks : 0.f;
}
// y = x * mask (in place or out of place); the same call with the same (seed, site) replays the mask (backward).
// One thread = 4 consecutive elements = one Philox call (n is a multiple of 4 on this path: rows of 96 / 384; the tail is scalar).
// relu_of != nullptr: additionally y = 0 where relu_of <= 0 (the ReLU mask of the feed-forward backward in the same pass).
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p, uint32_t lo, uint32_t hi,
                               uint32_t site, const float* __restrict__ relu_of) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b * 4 + 3 < n) {
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (p > 0.f) keep_scale4(lo, hi, site, b, p, m);
        const float4 v = ((const float4*)x)[b];
        float4 o = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
        if (relu_of) {
            const float4 r = ((const float4*)relu_of)[b];
            o.x = r.x > 0.f ? o.x : 0.f; o.y = r.y > 0.f ? o.y : 0.f; o.z = r.z > 0.f ? o.z : 0.f; o.w = r.w > 0.f ? o.w : 0.f;
        }
        ((float4*)y)[b] = o;
    } else {
        for (long i = b * 4; i < n; ++i) {
            float v = x[i] * (p > 0.f ? keep_scale(lo, hi, site, i, p) : 1.f);
            if (relu_of && !(relu_of[i] > 0.f)) v = 0.f;
            y[i] = v;
        }
    }
}
// out = a + dropout(b)
__global__ void add_dropout_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n, float p,
                                   uint32_t lo, uint32_t hi, uint32_t site) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k * 4 + 3 < n) {
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (p > 0.f) keep_scale4(lo, hi, site, k, p, m);
        const float4 va = ((const float4*)a)[k], vb = ((const float4*)b)[k];
        ((float4*)out)[k] = make_float4(va.x + vb.x * m[0], va.y + vb.y * m[1], va.z + vb.z * m[2], va.w + vb.w * m[3]);
    } else {
        for (long i = k * 4; i < n; ++i) out[i] = a[i] + (p > 0.f ? b[i] * keep_scale(lo, hi, site, i, p) : b[i]);
    }
}
// x[s][p][:] += vec[idx ? idx[p] : p][:]        (one thread = 4 consecutive features; n4 = S * P * 24 < 2^32)
__global__ void add_rows_kernel(float* __restrict__ x, uint32_t n4, int P, const float* __restrict__ vec, const int* __restrict__ idx) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    const uint32_t f4 = i % 24u, p = (i / 24u) % (uint32_t)P;
    const float4 v = ((const float4*)vec)[(uint32_t)(idx ? idx[p] : (int)p) * 24u + f4];
    float4 a = ((float4*)x)[i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    ((float4*)x)[i] = a;
}
// dvec[idx ? idx[p] : p][f] += sum_s dx[s][p][f]        one block per token position p
__global__ __launch_bounds__(384) void sum_over_seq_kernel(const float* __restrict__ dx, long S, int P, int p_off, int p_cnt, int ldp,
                                                           const int* __restrict__ idx, float* __restrict__ dvec) {
    __shared__ float red[4][D];
    const int pj = blockIdx.x;                 // 0..p_cnt
    const int f = threadIdx.x % D, part = threadIdx.x / D;      // 4 partial sums per feature
    // grid.y slices of the sequences (already an atomic accumulation): one block per token walked all 5200 sequences of C3 alone
    const long per = (S + gridDim.y - 1) / gridDim.y, qbeg = blockIdx.y * per, qend = qbeg + per < S ? qbeg + per : S;
    float s = 0.f;
    for (long q = qbeg + part; q < qend; q += 4) s += dx[(q * ldp + p_off + pj) * D + f];
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
// One thread = 4 consecutive features = one Philox call of the step_pt_dropout stream (element index of `out`); n4 = S * P * 24 < 2^32.
__global__ void dec_input_kernel(const float* __restrict__ z, const float* __restrict__ mask_token, const float* __restrict__ pos,
                                 const int* __restrict__ midx, uint32_t n4, int P, int Pu, float scale, float p, uint32_t lo, uint32_t hi,
                                 uint32_t site, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    const uint32_t f4 = i % 24u, st = i / 24u, t = st % (uint32_t)P, s = st / (uint32_t)P;
    float4 v;
    if (t < (uint32_t)Pu) {
        v = ((const float4*)z)[(s * (uint32_t)Pu + t) * 24u + f4];
    } else {
        const float4 a = ((const float4*)mask_token)[f4], b = ((const float4*)pos)[(uint32_t)midx[t - Pu] * 24u + f4];
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (p > 0.f) keep_scale4(lo, hi, site, (long)i, p, m);
        v = make_float4((a.x + b.x) * m[0], (a.y + b.y) * m[1], (a.z + b.z) * m[2], (a.w + b.w) * m[3]);
    }
    ((float4*)out)[i] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
}
// backward: dz[s][t] = scale * dout[s][t] (t < Pu);  dm[s][j] = scale * mask * dout[s][Pu + j]   (dm: [S][Pm][96] scratch)
__global__ void dec_input_bwd_kernel(const float* __restrict__ dout, uint32_t n4, int P, int Pu, float scale, float p, uint32_t lo, uint32_t hi,
                                     uint32_t site, float* __restrict__ dz, float* __restrict__ dm) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n4) return;
    const uint32_t f4 = i % 24u, st = i / 24u, t = st % (uint32_t)P, s = st / (uint32_t)P;
    float4 g = ((const float4*)dout)[i];
    g.x *= scale; g.y *= scale; g.z *= scale; g.w *= scale;
    if (t < (uint32_t)Pu) {
        ((float4*)dz)[(s * (uint32_t)Pu + t) * 24u + f4] = g;
    } else {
        if (p > 0.f) {
            float m[4];
            keep_scale4(lo, hi, site, (long)i, p, m);
            g.x *= m[0]; g.y *= m[1]; g.z *= m[2]; g.w *= m[3];
        }
        ((float4*)dm)[(s * (uint32_t)(P - Pu) + (t - (uint32_t)Pu)) * 24u + f4] = g;
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

// ------------------------------------------------------------------------------------ fused residual + LayerNorm, 16-byte accesses
// One row (96 floats = 24 float4) per 32 lanes, 8 rows per block.  Forward:  pre = a + dropout(b)  (b == nullptr: pre = a, nothing
// stored for it),  y = LayerNorm(pre), stats = (mean, rstd): the add_dropout pass, its store -> load round trip through HBM and the
// 4-byte accesses of ln_fwd_kernel in one kernel.  The dropout stream is step_pt_add_dropout's (one Philox call per float4).
__device__ __forceinline__ float half_sum(float v) {          // sum over the 32 lanes of a half wave
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 32);
    return v;
}
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long R, float p, uint32_t lo,
                                                         uint32_t hi, uint32_t site, const float* __restrict__ g, const float* __restrict__ beta,
                                                         float* __restrict__ pre, float* __restrict__ y, float* __restrict__ stats) {
    const int q = threadIdx.x & 31;
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= R) return;
    const bool on = q < D / 4;
    const long k = row * (D / 4) + (on ? q : 0);
    float4 v = ((const float4*)a)[k];
    if (b) {
        const float4 w = ((const float4*)b)[k];
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (p > 0.f) keep_scale4(lo, hi, site, k, p, m);
        v.x += w.x * m[0]; v.y += w.y * m[1]; v.z += w.z * m[2]; v.w += w.w * m[3];
        if (on) ((float4*)pre)[k] = v;
    }
    if (!on) v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float mean = half_sum((v.x + v.y) + (v.z + v.w)) * (1.f / D);
    const float4 d = on ? make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float rstd = rsqrtf(half_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.f / D) + 1e-5f);
    if (on) {
        const float4 g4 = ((const float4*)g)[q], b4 = ((const float4*)beta)[q];
        ((float4*)y)[k] = make_float4(d.x * rstd * g4.x + b4.x, d.y * rstd * g4.y + b4.y, d.z * rstd * g4.z + b4.z, d.w * rstd * g4.w + b4.w);
    }
    if (q == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}
// Backward: dx = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)), dyg = dy * gamma; optionally dxd = dropout(dx) with the stream of
// step_pt_dropout at (seed, site) in the same pass (the gradient that continues into the dropped branch); dgamma / dbeta by atomics.
__global__ __launch_bounds__(256) void ln_bwd_drop_kernel(const float* __restrict__ dy, const float* __restrict__ x, long R,
                                                          const float* __restrict__ g, const float* __restrict__ stats, float* __restrict__ dx,
                                                          float* __restrict__ dxd, float p, uint32_t lo, uint32_t hi, uint32_t site,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ ocol) {
    __shared__ float sg[8][D], sb[8][D], sc[8][D];
    const int q = threadIdx.x & 31, w = threadIdx.x >> 5;
    const bool on = q < D / 4;
    const float4 g4 = on ? ((const float4*)g)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f}, ac[4] = {0.f, 0.f, 0.f, 0.f};
    // the NEXT row's operands are requested before this row is worked on (round 6): a half wave had one row = 2 x 16 bytes per lane in flight
    // and then two dependent 5-step reductions; with ~19 waves per unit that is ~40 KB in flight per unit = 4.9 TB/s (Little), now twice that
    const long stride = (long)gridDim.x * 8;
    long row = (long)blockIdx.x * 8 + w;
    float4 nxv = make_float4(0.f, 0.f, 0.f, 0.f), nyv = nxv;
    float2 nst = make_float2(0.f, 1.f);
    if (row < R) {
        const long k0 = row * (D / 4) + (on ? q : 0);
        nxv = ((const float4*)x)[k0]; nyv = ((const float4*)dy)[k0]; nst = *(const float2*)(stats + row * 2);
    }
    for (; row < R; row += stride) {
        const float mean = nst.x, rstd = nst.y;
        const long k = row * (D / 4) + (on ? q : 0);
        float4 xv = nxv, yv = nyv;
        if (row + stride < R) {
            const long kn = (row + stride) * (D / 4) + (on ? q : 0);
            nxv = ((const float4*)x)[kn]; nyv = ((const float4*)dy)[kn]; nst = *(const float2*)(stats + (row + stride) * 2);
        }
        if (!on) { xv = make_float4(mean, mean, mean, mean); yv = make_float4(0.f, 0.f, 0.f, 0.f); }
        const float xh[4] = {(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
        const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
        const float qq[4] = {yv.x * g4.x, yv.y * g4.y, yv.z * g4.z, yv.w * g4.w};
        const float m1 = half_sum((qq[0] + qq[1]) + (qq[2] + qq[3])) * (1.f / D);
        const float m2 = half_sum((qq[0] * xh[0] + qq[1] * xh[1]) + (qq[2] * xh[2] + qq[3] * xh[3])) * (1.f / D);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = rstd * (qq[i] - m1 - xh[i] * m2); ag[i] += yy[i] * xh[i]; ab[i] += yy[i]; }
        if (on) {
            ((float4*)dx)[k] = make_float4(o[0], o[1], o[2], o[3]);
            float m[4] = {1.f, 1.f, 1.f, 1.f};
            if (dxd) {
                if (p > 0.f) keep_scale4(lo, hi, site, k, p, m);
                ((float4*)dxd)[k] = make_float4(o[0] * m[0], o[1] * m[1], o[2] * m[2], o[3] * m[3]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) ac[i] += o[i] * m[i];          // column sums of the gradient that continues (dxd, or dx without it)
        }
    }
    if (on) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { sg[w][4 * q + i] = ag[i]; sb[w][4 * q + i] = ab[i]; sc[w][4 * q + i] = ac[i]; }
    }
    __syncthreads();
    if (threadIdx.x < D) {
        const int f = threadIdx.x;
        float tg = 0.f, tb = 0.f, tc = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) { tg += sg[r][f]; tb += sb[r][f]; tc += sc[r][f]; }
        atomicAdd(&dgamma[f], tg);
        atomicAdd(&dbeta[f], tb);
        if (ocol) atomicAdd(&ocol[f], tc);
    }
}

// ------------------------------------------------------------------------------------ dropout stream of the attention probabilities
// One Philox4x32 call = eight 16-bit fields = eight keep decisions (field >= p * 2^16; 16 bits resolve p to 1.5e-5, and the fields of
// a Philox output are as independent as its words).  Decision of (row, key k), row = (s H + head) T + query: field number
//     row * T32 + 32 (k >> 5) + sigma(k & 31),   sigma(8 g + 4 h + j) = 16 h + 4 g + j,   T32 = T rounded up to 32
// -- sigma makes the 16 keys one lane of a matrix-core score tile holds (keys 8 g + 4 h + j, g = 0..3, j = 0..3) 16 CONSECUTIVE
// fields, i.e. two Philox calls per lane and tile (the element-per-word stream of step_pt_dropout needed four, and they were half
// of the forward kernel's time: profiles/r03_l_pretrain_attention_ablations.log).  The f32 kernels evaluate the same stream.
__device__ __forceinline__ uint32_t attn_keep_thr(float p) { return (uint32_t)(p * 65536.f); }
__device__ __forceinline__ long attn_keep_index(long row, int T32, int k) {
    const int k32 = k & 31;
    return row * T32 + (k & ~31) + 16 * ((k32 >> 2) & 1) + 4 * (k32 >> 3) + (k32 & 3);
}
__device__ __forceinline__ float attn_keep(uint32_t lo, uint32_t hi, uint32_t site, long row, int T32, int k, float p) {
    const long f = attn_keep_index(row, T32, k), blk = f >> 3;
    const int sub = (int)(f & 7);
    uint32_t r[4];
    philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), site, 0xD20Fu, lo, hi, r);
    return ((r[sub >> 1] >> (16 * (sub & 1))) & 0xffffu) >= attn_keep_thr(p) ? 1.f / (1.f - p) : 0.f;
}

// ------------------------------------------------------------------------------------ self-attention, one block per (sequence, head)
// qkv [S][T][288] (q | k | v, head h = columns 24h..24h+23 of each third), out [S][T][96], stats [S][H][T][2] = (row max, row sum)
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
                if (p > 0.f) e *= attn_keep(lo, hi, site, srow + i, (T + 31) & ~31, j, p);
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
                if (p > 0.f) dp *= attn_keep(lo, hi, site, srow + i, (T + 31) & ~31, j, p);
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
                if (p > 0.f) m = attn_keep(lo, hi, site, srow + i, (T + 31) & ~31, j, p);
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

// ------------------------------------------------------------------------------------ self-attention on the matrix cores (bf16 mode)
// Same operation as attn_kernel with bf16 operands and f32 accumulation on v_mfma_f32_32x32x16_bf16, one workgroup per (sequence,
// head), one wave per 32-token tile.  Everything is computed TRANSPOSED, S^T = K Q^T: an accumulator then holds 16 keys of ONE query
// per lane (row statistics are per-lane scalars) and becomes the B operand of the next product by a plain pack -- a 16-deep k step
// takes accumulator registers [8s, 8s+8), i.e. keys {16s+4h+0..3, 16s+8+4h+0..3} of lane half h, and the A operand (rows of a
// transposed LDS copy) reads exactly those keys (any pairing of k is a valid contraction order).
// Dropout of the attention probabilities: the same Philox stream and element index as attn_kernel (one call covers 4 keys).
#ifndef MA_ABLATE
#define MA_ABLATE 0       // timing experiments only (WRONG results): 1 no Philox in the forward, 2 no global loads in the fills, 4 no output stores
#endif
constexpr int MA_RP = 40;                 // row pitch of the row-major LDS arrays (bf16 elements): 80 B, conflict-free 16-byte reads
__device__ __forceinline__ int ma_tpitch(int Tp) { return Tp + 8; }      // pitch of the transposed arrays
__device__ __forceinline__ bf16x8 ma_row8(const uint16_t* row) { return __builtin_bit_cast(bf16x8, *(const uint4*)row); }
__device__ __forceinline__ bf16x8 ma_tr8(const uint16_t* trow, int k0, int h) {      // keys k0 + {4h..4h+3, 8+4h..8+4h+3} of one d row
    const uint2 a = *(const uint2*)(trow + k0 + 4 * h), b = *(const uint2*)(trow + k0 + 8 + 4 * h);
    return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}
__device__ __forceinline__ bf16x8 ma_pack(const float* v) { return pack8(v); }
// the 16 keep decisions (as multipliers 1/(1-p) or 0) of one lane of a score tile -- keys 8 g + 4 h + j of the tile, m[4 g + j] --
// from the attention-dropout stream (attn_keep_index): fields 16 h .. 16 h + 15 of the tile = two Philox calls
__device__ __forceinline__ void ma_keep16(uint32_t lo, uint32_t hi, uint32_t site, long row, int T32, int kt, int h, float p, float ks, float* m) {
    const long blk = (row * T32 + kt * 32 + 16 * h) >> 3;
    const uint32_t thr = attn_keep_thr(p);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        uint32_t r[4];
        philox4x32((uint32_t)(blk + b), (uint32_t)((blk + b) >> 32), site, 0xD20Fu, lo, hi, r);
#pragma unroll
        for (int f = 0; f < 8; ++f) m[8 * b + f] = ((r[f >> 1] >> (16 * (f & 1))) & 0xffffu) >= thr ? ks : 0.f;
    }
}
__device__ __forceinline__ uint32_t ma_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ int ma_key(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }     // accumulator register -> row of the tile

// Activations in HBM are bf16 (qkv [S][T][288], out / d out [S][T][96], d qkv [S][T][288]: the GEMM epilogues write them and the
// GEMM loaders read them in that type -- the rounding is the one the matrix-core operands get anyway).  Global <-> LDS traffic moves
// 16 bytes per lane: a fill item is (token t, 8 head dimensions), 4 items per token (3 real + 1 padding) = three uint4 of one 48-byte
// run, copied to LDS as they are; results leave through an f32 staging tile [Tp][MA_SP] and are packed to uint4 pieces of those runs.  (With one 4-byte access per lane the loads were 30 % and the stores 30 %
// (forward) / 65 % (backward) of the kernel time: profiles/r03_l_pretrain_attention_ablations.log.)
// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2); this hands XCD x a CONTIGUOUS range of (sequence, head)
// units, so the four heads of a sequence -- which share every 128-byte line of the sequence's qkv / out / dqkv rows -- run
// back to back on one L2.
__device__ __forceinline__ unsigned ma_unit(unsigned b, unsigned n) {
    const unsigned per = n >> 3, rem = n & 7u, x = b & 7u;
    return x * per + (x < rem ? x : rem) + (b >> 3);
}
constexpr int MA_SP = 28;                 // row pitch (floats) of the staging tile: 112 B, 16-byte aligned, 2-way conflicts at most
__device__ __forceinline__ void ma_stage_rows(float* stage, int q, int h, const f32x16& v, float scale) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int d = ma_key(e, h);
        if (d < DH) stage[q * MA_SP + d] = v[e] * scale;
    }
}
// rows [0, T) of the staging tile -> dst[t * ld + (0..23)] as bf16, uint4 pieces (8 values)
__device__ __forceinline__ uint4 ma_pack8(const float* v) {
    const float4 a = *(const float4*)v, b = *(const float4*)(v + 4);
    return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}
__device__ __forceinline__ void ma_store_rows(const float* stage, int T, uint16_t* __restrict__ dst, long ld, int tid, int nthreads) {
    if (MA_ABLATE & 4) return;
    for (int e = tid; e < T * 3; e += nthreads) {
        const int t = e / 3, d8 = e - 3 * t;
        *(uint4*)(dst + (long)t * ld + d8 * 8) = ma_pack8(stage + t * MA_SP + d8 * 8);
    }
}
__device__ __forceinline__ uint16_t ma_half(const uint4& v, int c) {      // element c (0..7) of 8 packed bf16
    const uint32_t w = c < 2 ? v.x : (c < 4 ? v.y : (c < 6 ? v.z : v.w));
    return (uint16_t)(w >> (16 * (c & 1)));
}

// forward: out [S][T][96], stats [S][H][T][2] = (row max, row sum) like attn_kernel
__global__ __launch_bounds__(704) void attn_mfma_fwd_kernel(const uint16_t* __restrict__ qkv, int T, int Tp, float p, uint32_t lo, uint32_t hi,
                                                            uint32_t site, uint16_t* __restrict__ out, float* __restrict__ stats,
                                                            uint32_t* __restrict__ keepbits, const uint32_t* __restrict__ pool32, uint32_t pmask32) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ml[];
    const int TPt = ma_tpitch(Tp);
    uint16_t* Qs = ml;                       // [Tp][MA_RP]  q (the 1/sqrt(24) of the scores is applied to the f32 products)
    uint16_t* Ks = Qs + Tp * MA_RP;          // [Tp][MA_RP]
    uint16_t* VT = Ks + Tp * MA_RP;          // [32][TPt]
    const unsigned unit = ma_unit(blockIdx.x, gridDim.x);
    const long s = unit / H;
    const int hd = unit % H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
    const float scale = 0.20412414523193154f;
    const uint16_t* base = qkv + s * (long)T * 288 + hd * DH;
    for (int e = tid; e < Tp * 4; e += blockDim.x) {             // (token, 8 head dimensions): q | k rows copied as they are, v transposed
        const int t = e >> 2, d8 = e & 3;
        uint4 q8 = make_uint4(0u, 0u, 0u, 0u), k8 = q8, v8 = q8;
        if (t < T && d8 < 3 && !(MA_ABLATE & 2)) {
            const uint16_t* r = base + (long)t * 288 + d8 * 8;
            q8 = *(const uint4*)r; k8 = *(const uint4*)(r + 96); v8 = *(const uint4*)(r + 192);
        }
        *(uint4*)(Qs + t * MA_RP + d8 * 8) = q8;
        *(uint4*)(Ks + t * MA_RP + d8 * 8) = k8;
#pragma unroll
        for (int c = 0; c < 8; ++c) VT[(d8 * 8 + c) * TPt + t] = ma_half(v8, c);
    }
    __syncthreads();
    const int q = wave * 32 + col, nkt = Tp / 32;
    const long srow = (s * H + hd) * (long)T;
    // pool32 != NULL: the keep decisions of (query, key tile) are one 32-bit word of the step's Bernoulli pool (step_dropout_pool_fill) --
    // this (sequence, head) reads T * nkt consecutive words at a hashed offset -- instead of two Philox calls per lane and tile
    const uint32_t pbase = ma_mix32(lo + unit * 0x9E3779B1u + (site + 1u) * 0x632BE5ABu);
    bf16x8 bq[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) bq[st] = ma_row8(Qs + q * MA_RP + 16 * st + 8 * h);
    auto scores = [&](int kt, f32x16& acc) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int st = 0; st < 2; ++st)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma_row8(Ks + (kt * 32 + col) * MA_RP + 16 * st + 8 * h), bq[st], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) if (kt * 32 + ma_key(e, h) >= T) acc[e] = -1e30f;
    };
    float mx = -1e30f;
    for (int kt = 0; kt < nkt; ++kt) {
        f32x16 acc;
        scores(kt, acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, acc[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float l = 0.f;
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    for (int kt = 0; kt < nkt; ++kt) {
        uint32_t pword = 0u;                 // requested before the score products: its latency hides behind them and the exponentials
        if (p > 0.f && pool32 && q < T) pword = pool32[(pbase + (uint32_t)(q * nkt + kt)) & pmask32];
        f32x16 acc;
        scores(kt, acc);
        float pv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { pv[e] = __expf((acc[e] - mx) * scale); l += pv[e]; }
        if (p > 0.f) {
            uint32_t word = 0u;
            if (q < T && pool32) {
                word = pword;
                const uint32_t wh = word >> (4 * h);
#pragma unroll
                for (int e = 0; e < 16; ++e) pv[e] = (wh & (1u << ((e & 3) + 8 * (e >> 2)))) ? pv[e] * ks : 0.f;
            } else if (q < T) {
                float m[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) m[e] = ks;
                if (!(MA_ABLATE & 1)) ma_keep16(lo, hi, site, srow + q, Tp, kt, h, p, ks, m);
#pragma unroll
                for (int e = 0; e < 16; ++e) { pv[e] *= m[e]; if (m[e] > 0.f) word |= 1u << ma_key(e, h); }
            }
            // the keep decisions of (query, key tile) as one word: the backward reads them instead of re-running Philox
            if (!pool32) word |= __shfl_xor(word, 32, 64);
            if (keepbits && h == 0 && q < T) keepbits[(srow + q) * nkt + kt] = word;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma_tr8(VT + col * TPt, kt * 32 + 16 * st, h), ma_pack(pv + 8 * st), o, 0, 0, 0);
    }
    l += __shfl_xor(l, 32, 64);
    if (q < T && h == 0) { stats[(srow + q) * 2] = mx * scale; stats[(srow + q) * 2 + 1] = l; }
    // the output tile leaves through LDS (the operand arrays are dead) as float4 pieces of each token's 96-byte run
    __syncthreads();
    float* stage = (float*)ml;
    ma_stage_rows(stage, q, h, o, 1.f / l);
    __syncthreads();
    ma_store_rows(stage, T, out + s * (long)T * D + hd * DH, D, tid, blockDim.x);
}

// backward: phase A (wave = query tile) -> dQ and the keep bits, phase B (wave = key tile) -> dK, dV
// LDS: only the 24 real head dimensions are stored -- row-major rows of 24 (48 B: three 16-byte pieces, the fourth operand piece
// is a zero constant) and 24 transposed rows plus ONE shared zero row for operand rows 24..31 -- 71 KB at T = 168 (two
// workgroups per compute unit; 107 KB and one before) and 137 KB at T = 336 (the f32 kernel had to take over above 256 tokens).
constexpr int MB_RP = 24;                 // row pitch (bf16 elements) of the backward's row-major arrays
__device__ __forceinline__ bf16x8 mb_row8(const uint16_t* arr, int row, int st, int h) {      // dims 16 st + 8 h .. + 7 of `row`
    const int off = 16 * st + 8 * h;
    const uint4 v = *(const uint4*)(arr + row * MB_RP + (off < DH ? off : 0));
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    return __builtin_bit_cast(bf16x8, off < DH ? v : z);
}
__global__ __launch_bounds__(704) void attn_mfma_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                            const uint16_t* __restrict__ dout, const float* __restrict__ stats, int T, int Tp,
                                                            float p, uint32_t lo, uint32_t hi, uint32_t site, uint16_t* __restrict__ dqkv,
                                                            const uint32_t* __restrict__ keepbits) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ml[];
    const int TPt = ma_tpitch(Tp), nt = Tp / 32;
    uint16_t* Qs = ml;                        // row-major [Tp][MB_RP]: q, k, v, dO
    uint16_t* Ks = Qs + Tp * MB_RP;
    uint16_t* Vs = Ks + Tp * MB_RP;
    uint16_t* Os = Vs + Tp * MB_RP;
    uint16_t* QT = Os + Tp * MB_RP;           // transposed [DH][TPt]: q, k, dO
    uint16_t* KT = QT + DH * TPt;
    uint16_t* OT = KT + DH * TPt;
    uint16_t* ZT = OT + DH * TPt;             // [TPt] zeros: the padded head dimensions 24 .. 31 of all three
    float* smx = (float*)(ZT + TPt);          // [Tp] exponent offset of a query row: P = exp2(score * scale * log2(e) + smx) (rows >= T: -1e30)
    float* sinv = smx + Tp;                   // (unused slot kept for the layout of ma_bwd_lds)
    float* sdl = sinv + Tp;                   // [Tp] delta
    uint32_t* bits = (uint32_t*)(sdl + Tp);   // [nt][Tp] keep bits of (key tile, query): four consecutive queries are one 16-byte read
    const unsigned unit = ma_unit(blockIdx.x, gridDim.x);
    const long s = unit / H;
    const int hd = unit % H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
    const float scale = 0.20412414523193154f;
    const uint16_t* base = qkv + s * (long)T * 288 + hd * DH;
    const long srow = (s * H + hd) * (long)T;
    // fill: item = (token, 8 head dimensions), 4 lanes per token, uint4 loads; blockDim = 2 Tp, so every thread owns exactly two
    // items -- all ten loads are issued before the first is consumed
    uint4 q8[2], k8[2], v8[2], g8[2], o8[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = tid + it * blockDim.x, t = e >> 2, d8 = e & 3;
        q8[it] = make_uint4(0u, 0u, 0u, 0u); k8[it] = q8[it]; v8[it] = q8[it]; g8[it] = q8[it]; o8[it] = q8[it];
        if (t < T && d8 < 3 && !(MA_ABLATE & 2)) {
            const uint16_t* r = base + (long)t * 288 + d8 * 8;
            const long oi = (s * T + t) * D + hd * DH + d8 * 8;
            q8[it] = *(const uint4*)r; k8[it] = *(const uint4*)(r + 96); v8[it] = *(const uint4*)(r + 192);
            g8[it] = *(const uint4*)(dout + oi); o8[it] = *(const uint4*)(out + oi);
        }
    }
    if (keepbits && p > 0.f)                 // the forward's keep decisions of this (sequence, head): one contiguous run of T nt words
        for (int i = tid; i < Tp * nt; i += blockDim.x) bits[(i % nt) * Tp + i / nt] = i < T * nt ? keepbits[srow * nt + i] : 0u;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = tid + it * blockDim.x, t = e >> 2, d8 = e & 3;
        const uint32_t gw[4] = {g8[it].x, g8[it].y, g8[it].z, g8[it].w}, ow[4] = {o8[it].x, o8[it].y, o8[it].z, o8[it].w};
        float dl = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dl += __uint_as_float(gw[j] << 16) * __uint_as_float(ow[j] << 16) + __uint_as_float(gw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
        dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64);                    // delta_t = sum_d dO O (4 adjacent lanes)
        if (d8 == 0) sdl[t] = dl;
        if (d8 < 3) {
            const int ro = t * MB_RP + d8 * 8;
            *(uint4*)(Qs + ro) = q8[it]; *(uint4*)(Ks + ro) = k8[it]; *(uint4*)(Vs + ro) = v8[it]; *(uint4*)(Os + ro) = g8[it];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int d = d8 * 8 + c;
                QT[d * TPt + t] = ma_half(q8[it], c); KT[d * TPt + t] = ma_half(k8[it], c); OT[d * TPt + t] = ma_half(g8[it], c);
            }
        }
    }
    for (int i = tid; i < TPt; i += blockDim.x) ZT[i] = 0;
    for (int i = tid; i < Tp; i += blockDim.x) {
        float2 st2 = make_float2(0.f, 1.f);
        if (i < T) st2 = *(const float2*)(stats + (srow + i) * 2);
        // stats = (row max of the scaled scores, row sum): P = exp(s * scale - max) / sum = exp2(s * scale * log2(e) + smx)
        smx[i] = i < T ? -st2.x * 1.4426950408889634f - __log2f(st2.y) : -1e30f;
    }
    __syncthreads();
    const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
    f32x16 dq_keep;
    {   // ---- phase A: this wave's 32 queries against every key tile (keys in registers, query = lane)
        const int q = wave * 32 + col;
        const float mq = smx[q], dq_ = sdl[q];
        const float sl2 = scale * 1.4426950408889634f;
        bf16x8 bq[2], bo[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) { bq[st] = mb_row8(Qs, q, st, h); bo[st] = mb_row8(Os, q, st, h); }
        f32x16 dqa;
#pragma unroll
        for (int e = 0; e < 16; ++e) dqa[e] = 0.f;
        for (int kt = 0; kt < nt; ++kt) {
            f32x16 sc, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sc[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb_row8(Ks, kt * 32 + col, st, h), bq[st], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb_row8(Vs, kt * 32 + col, st, h), bo[st], dp, 0, 0, 0);
            }
            float mk[16];
            uint32_t word = 0u;
            if (keepbits && p > 0.f) {               // the forward's keep decisions
                word = q < T ? bits[kt * Tp + q] : 0u;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mk[4 * g + j] = (word >> (8 * g + 4 * h + j)) & 1u ? ks : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) mk[e] = p > 0.f && q >= T ? 0.f : ks;
                if (p > 0.f && q < T) ma_keep16(lo, hi, site, srow + q, Tp, kt, h, p, ks, mk);
#pragma unroll
                for (int e = 0; e < 16; ++e) if (mk[e] > 0.f) word |= 1u << ma_key(e, h);
                word |= __shfl_xor(word, 32, 64);
            }
            if (h == 0 && !(keepbits && p > 0.f)) bits[kt * Tp + q] = word;
            float ds[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool kok = kt * 32 + ma_key(e, h) < T;
                const float pij = kok ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], sl2, mq)) : 0.f;
                ds[e] = pij * (dp[e] * mk[e] - dq_);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
                dqa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma_tr8(col < DH ? KT + col * TPt : ZT, kt * 32 + 16 * st, h), ma_pack(ds + 8 * st), dqa, 0, 0, 0);
        }
        dq_keep = dqa;                         // leaves with dK and dV through the staging tile, after phase B is done with the operands
    }
    __syncthreads();
    {   // ---- phase B: this wave's 32 keys against every query tile (queries in registers, key = lane)
        const int kj = wave * 32 + col;
        bf16x8 bk[2], bv[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) { bk[st] = mb_row8(Ks, kj, st, h); bv[st] = mb_row8(Vs, kj, st, h); }
        f32x16 dka, dva;
#pragma unroll
        for (int e = 0; e < 16; ++e) { dka[e] = 0.f; dva[e] = 0.f; }
        for (int qt = 0; qt < nt; ++qt) {
            f32x16 sc, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { sc[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb_row8(Qs, qt * 32 + col, st, h), bk[st], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(mb_row8(Os, qt * 32 + col, st, h), bv[st], dp, 0, 0, 0);
            }
            float pm[16], ds[16];
            const float sl2 = scale * 1.4426950408889634f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // queries qt * 32 + 8 g + 4 h + (0..3) are accumulator registers 4 g .. 4 g + 3: their statistics and keep words are 16-byte reads
                const int q0 = qt * 32 + 8 * g + 4 * h;
                const float4 eq = *(const float4*)(smx + q0), dl4 = *(const float4*)(sdl + q0);
                const uint4 bw = *(const uint4*)(bits + wave * Tp + q0);
                const float eqv[4] = {eq.x, eq.y, eq.z, eq.w}, dlv[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
                const uint32_t bwv[4] = {bw.x, bw.y, bw.z, bw.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * g + j;
                    const float mkv = (bwv[j] >> col) & 1u ? ks : 0.f;
                    const float pij = kj < T ? __builtin_amdgcn_exp2f(__builtin_fmaf(sc[e], sl2, eqv[j])) : 0.f;      // rows qi >= T: exp2(-1e30) = 0
                    pm[e] = pij * mkv;
                    ds[e] = pij * (dp[e] * mkv - dlv[j]);
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                dva = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma_tr8(col < DH ? OT + col * TPt : ZT, qt * 32 + 16 * st, h), ma_pack(pm + 8 * st), dva, 0, 0, 0);
                dka = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma_tr8(col < DH ? QT + col * TPt : ZT, qt * 32 + 16 * st, h), ma_pack(ds + 8 * st), dka, 0, 0, 0);
            }
        }
        // dQ | dK | dV of this wave's 32 tokens -> staging tile [Tp][3 MA_SP] over the dead operands -> float4 pieces of the three
        // 96-byte runs each token owns in dqkv
        __syncthreads();
        float* stage = (float*)ml;
        ma_stage_rows(stage, 3 * kj, h, dq_keep, scale);
        ma_stage_rows(stage + MA_SP, 3 * kj, h, dka, scale);
        ma_stage_rows(stage + 2 * MA_SP, 3 * kj, h, dva, 1.f);
        __syncthreads();
        if (!(MA_ABLATE & 4)) {
            uint16_t* db = dqkv + s * (long)T * 288 + hd * DH;
            for (int e = tid; e < T * 9; e += blockDim.x) {
                const int t = e / 9, r = e - 9 * t, sec = r / 3, d8 = r - 3 * sec;
                *(uint4*)(db + (long)t * 288 + sec * 96 + d8 * 8) = ma_pack8(stage + (3 * t + sec) * MA_SP + d8 * 8);
            }
        }
    }
}

__global__ void relu_mask_kernel(float* __restrict__ d, const float* __restrict__ y, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && !(y[i] > 0.f)) d[i] = 0.f;
}

// out[c] += sum over rows of x[row][c], x bf16 [rows][cols], cols % 8 == 0, cols <= 2048 (bias gradient of the bf16 hidden-layer
// gradient).  A thread owns 8 consecutive columns (one 16-byte load per row) of every (256 / (cols / 8))-th row of the block's slab;
// the row lanes are combined through LDS, one atomic per column and block.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const uint16_t* __restrict__ x, long rows, int cols, int slab, float* __restrict__ out) {
    __shared__ float part[256 * 8];
    const int vc = cols >> 3, lanes = 256 / vc;                  // vector columns, row lanes
    const int cx = threadIdx.x % vc, ry = threadIdx.x / vc;
    const long r0 = (long)blockIdx.x * slab, r1 = r0 + slab < rows ? r0 + slab : rows;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ry < lanes)
        for (long r = r0 + ry; r < r1; r += lanes) {
            const uint4 v = *(const uint4*)(x + r * cols + cx * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[2 * j] += __uint_as_float(w[j] << 16); a[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u); }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[threadIdx.x * 8 + j] = a[j];
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += 256) {
        float t = 0.f;
        for (int l = 0; l < lanes; ++l) t += part[(l * vc + (c >> 3)) * 8 + (c & 7)];
        atomicAdd(out + c, t);
    }
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

#define SEED_LO(s) ((uint32_t)(s))
#define SEED_HI(s) ((uint32_t)((s) >> 32))

extern "C" int step_pt_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint32_t site, void* stream) {
    STEP_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "pt_dropout: bad arguments");
    STEP_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pt_dropout: 16-byte aligned buffers");
    dropout_kernel<<<g1((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, y, n, p, SEED_LO(seed), SEED_HI(seed), site, nullptr);
    STEP_LAUNCH_CHECK("pt_dropout");
    return STEP_OK;
}
// d = dropout(d) masked by relu_of > 0 in one pass (backward of relu -> dropout: the same mask stream as step_pt_dropout at `site`)
extern "C" int step_pt_dropout_relu_mask(float* d, const float* relu_of, long n, float p, uint64_t seed, uint32_t site, void* stream) {
    STEP_REQUIRE(d && relu_of && n > 0 && p >= 0.f && p < 1.f, "pt_dropout_relu_mask: bad arguments");
    STEP_REQUIRE((((uintptr_t)d | (uintptr_t)relu_of) & 15) == 0, "pt_dropout_relu_mask: 16-byte aligned buffers");
    dropout_kernel<<<g1((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(d, d, n, p, SEED_LO(seed), SEED_HI(seed), site, relu_of);
    STEP_LAUNCH_CHECK("pt_dropout_relu_mask");
    return STEP_OK;
}
extern "C" int step_pt_add_dropout(const float* a, const float* b, float* out, long n, float p, uint64_t seed, uint32_t site, void* stream) {
    STEP_REQUIRE(a && b && out && n > 0 && p >= 0.f && p < 1.f, "pt_add_dropout: bad arguments");
    STEP_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "pt_add_dropout: 16-byte aligned buffers");
    add_dropout_kernel<<<g1((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(a, b, out, n, p, SEED_LO(seed), SEED_HI(seed), site);
    STEP_LAUNCH_CHECK("pt_add_dropout");
    return STEP_OK;
}
extern "C" int step_pt_add_rows(float* x, long S, int P, const float* vec, const int* idx, void* stream) {
    STEP_REQUIRE(x && vec && S > 0 && P > 0, "pt_add_rows: bad arguments");
    STEP_REQUIRE(S * P * 24 < (1L << 32), "pt_add_rows: %ld x %d tokens exceed the 32-bit index range", S, P);
    add_rows_kernel<<<g1(S * P * 24), 256, 0, (hipStream_t)stream>>>(x, (uint32_t)(S * P * 24), P, vec, idx);
    STEP_LAUNCH_CHECK("pt_add_rows");
    return STEP_OK;
}
extern "C" int step_pt_sum_over_seq(const float* dx, long S, int ldp, int p_off, int p_cnt, const int* idx, float* dvec, void* stream) {
    STEP_REQUIRE(dx && dvec && S > 0 && p_cnt > 0, "pt_sum_over_seq: bad arguments");
    int slices = (int)(2048 / p_cnt);
    if (slices > S / 16) slices = (int)(S / 16);
    if (slices < 1) slices = 1;
    sum_over_seq_kernel<<<dim3(p_cnt, slices), 384, 0, (hipStream_t)stream>>>(dx, S, 0, p_off, p_cnt, ldp, idx, dvec);
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
    STEP_REQUIRE(S * P * 24 < (1L << 32), "pt_dec_input: %ld x %d tokens exceed the 32-bit index range", S, P);
    dec_input_kernel<<<g1(S * P * 24), 256, 0, (hipStream_t)stream>>>(z, mask_token, pos, midx, (uint32_t)(S * P * 24), P, Pu, 9.797958971132712f, p,
                                                                     SEED_LO(seed), SEED_HI(seed), site, out);
    STEP_LAUNCH_CHECK("pt_dec_input");
    return STEP_OK;
}
extern "C" int step_pt_dec_input_bwd(const float* dout, long S, int P, int Pu, float p, uint64_t seed, uint32_t site, float* dz,
                                     float* dm, void* stream) {
    STEP_REQUIRE(dout && dz && dm && S > 0 && P > Pu && Pu > 0, "pt_dec_input_bwd: bad arguments");
    STEP_REQUIRE(S * P * 24 < (1L << 32), "pt_dec_input_bwd: %ld x %d tokens exceed the 32-bit index range", S, P);
    dec_input_bwd_kernel<<<g1(S * P * 24), 256, 0, (hipStream_t)stream>>>(dout, (uint32_t)(S * P * 24), P, Pu, 9.797958971132712f, p, SEED_LO(seed),
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
// pre = a + dropout(b) (b may be NULL: pre is not written), y = LayerNorm(pre), stats = (mean, rstd) per row, in one pass
extern "C" int step_pt_add_layernorm_fwd(const float* a, const float* b, long R, float p, uint64_t seed, uint32_t site, const float* g,
                                         const float* beta, float* pre, float* y, float* stats, void* stream) {
    STEP_REQUIRE(a && g && beta && y && stats && R > 0 && p >= 0.f && p < 1.f && (b == nullptr || pre != nullptr), "pt_add_layernorm_fwd: bad arguments");
    STEP_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)pre | (uintptr_t)y | (uintptr_t)g | (uintptr_t)beta) & 15) == 0,
                 "pt_add_layernorm_fwd: 16-byte aligned buffers");
    add_ln_fwd_kernel<<<(unsigned)((R + 7) / 8), 256, 0, (hipStream_t)stream>>>(a, b, R, p, SEED_LO(seed), SEED_HI(seed), site, g, beta, pre, y, stats);
    STEP_LAUNCH_CHECK("pt_add_layernorm_fwd");
    return STEP_OK;
}
// LayerNorm backward; dx_dropped (may be NULL) = dropout(dx) with the stream of step_pt_dropout(seed, site), written in the same pass
extern "C" int step_pt_layernorm_bwd_dropout(const float* dy, const float* x, long R, const float* g, const float* stats, float* dx,
                                             float* dx_dropped, float p, uint64_t seed, uint32_t site, float* dgamma, float* dbeta,
                                             float* out_colsum, void* stream) {
    STEP_REQUIRE(dy && x && g && stats && dx && dgamma && dbeta && R > 0 && p >= 0.f && p < 1.f, "pt_layernorm_bwd_dropout: bad arguments");
    STEP_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dx_dropped | (uintptr_t)g) & 15) == 0,
                 "pt_layernorm_bwd_dropout: 16-byte aligned buffers");
    // every block ends with 3 x 96 atomic adds onto the same 288 addresses: the cap trades that serial tail against rows in flight
    // (STEP_LN_BWD_BLOCKS: A/B measurements, profiles/r04_u_layernorm_backward_blocks.log)
    // measured at config C3's sizes: 873 600 rows 293 / 272 / 289 us with 1024 / 2048 / 4096 blocks, 218 400 rows 85 / 102 / 125 us
    static long cap = -1;
    if (cap < 0) {
        const char* e = getenv("STEP_LN_BWD_BLOCKS");
        cap = e && atol(e) > 0 ? atol(e) : 0;
    }
    long blocks = (R + 7) / 8;
    const long want = cap > 0 ? cap : (R / 256 < 1024 ? 1024 : (R / 256 > 2048 ? 2048 : R / 256));
    if (blocks > want) blocks = want;
    ln_bwd_drop_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(dy, x, R, g, stats, dx, dx_dropped, p, SEED_LO(seed), SEED_HI(seed), site,
                                                                         dgamma, dbeta, out_colsum);
    STEP_LAUNCH_CHECK("pt_layernorm_bwd_dropout");
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
static int attn_attrs() {
    STEP_TRY(step_raise_lds_once((const void*)attn_kernel<true>, 160 * 1024, "pt_attention"));
    STEP_TRY(step_raise_lds_once((const void*)attn_kernel<false>, 160 * 1024, "pt_attention"));
    return STEP_OK;
}
extern "C" int step_pt_attention_fwd(const float* qkv, long S, int T, float p, uint64_t seed, uint32_t site, float* out, float* stats,
                                     void* stream) {
    STEP_REQUIRE(qkv && out && stats && S > 0 && T > 0 && T <= 336, "pt_attention_fwd: bad arguments (T=%d; at most 336 tokens, the limit of the backward)", T);
    STEP_TRY(attn_attrs());
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
    STEP_TRY(attn_attrs());
    attn_kernel<true><<<(unsigned)(S * H), 256, lds, (hipStream_t)stream>>>(qkv, S, T, p, SEED_LO(seed), SEED_HI(seed), site,
                                                                           const_cast<float*>(out), const_cast<float*>(stats), dout, dqkv);
    STEP_LAUNCH_CHECK("pt_attention_bwd");
    return STEP_OK;
}
// the same attention on the matrix cores (bf16 operands, f32 accumulation) -- what TSFormer(mode="pre-train") uses with matmul_precision = "bf16"
// second version of the two kernels (pretrain_attn2.hip): keep decisions from the step's pool (or no dropout); STEP_PT_ATTN_V1=1 keeps the first one (A/B runs)
bool step_attn2_fits(int T);
int step_attn2_fwd(const uint16_t* qkv, long S, int T, float p, uint64_t seed, uint32_t site, uint16_t* out, float* stats, uint32_t* keepbits,
                   const uint64_t* pool, long pool_words, hipStream_t st);
int step_attn2_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* stats, long S, int T, float p, uint16_t* dqkv,
                   const uint32_t* keepbits, hipStream_t st);
static bool attn_v1_forced() {
    static const bool v1 = [] { const char* e = getenv("STEP_PT_ATTN_V1"); return e && e[0] == '1'; }();
    return v1;
}
static size_t ma_fwd_lds(int Tp) {      // operands, re-used by the f32 staging tile of the output
    const size_t ops = (size_t)(2 * Tp * MA_RP + 32 * (Tp + 8)) * 2, stage = (size_t)Tp * MA_SP * 4;
    return ops > stage ? ops : stage;
}
static size_t ma_bwd_lds(int Tp) { return (size_t)(4 * Tp * MB_RP + (3 * DH + 1) * (Tp + 8)) * 2 + (size_t)(3 * Tp + Tp * (Tp / 32)) * 4; }
extern "C" int step_pt_attention_fwd_bf16(const uint16_t* qkv, long S, int T, float p, uint64_t seed, uint32_t site, uint16_t* out, float* stats,
                                          uint32_t* keepbits, const uint64_t* pool, long pool_words, void* stream) {
    STEP_REQUIRE(qkv && out && stats && S > 0 && T > 0 && T <= 352 && p >= 0.f && p < 1.f && ((((uintptr_t)qkv) | ((uintptr_t)out)) & 15) == 0,
                 "pt_attention_fwd_bf16: bad arguments (T=%d; at most 352 tokens, 16-byte aligned bf16 tensors)", T);
    if (pool && p > 0.f) {
        STEP_REQUIRE(keepbits, "pt_attention_fwd_bf16: pool-drawn keep decisions reach the backward through `keepbits` only");
        STEP_REQUIRE(pool_words >= 4096 && (pool_words & (pool_words - 1)) == 0 && pool_words <= (1L << 30),
                     "pt_attention_fwd_bf16: pool of %ld words is not a power of two in [4096, 2^30]", pool_words);
    }
    const int Tp = (T + 31) & ~31;
    if (!attn_v1_forced() && step_attn2_fits(T) && (p == 0.f || pool))
        return step_attn2_fwd(qkv, S, T, p, SEED_LO(seed), site, out, stats, keepbits, pool, pool_words, (hipStream_t)stream);
    STEP_TRY(step_raise_lds_once((const void*)attn_mfma_fwd_kernel, 160 * 1024, "pt_attention_fwd_bf16"));
    attn_mfma_fwd_kernel<<<(unsigned)(S * H), 64 * (Tp / 32), ma_fwd_lds(Tp), (hipStream_t)stream>>>(qkv, T, Tp, p, SEED_LO(seed), SEED_HI(seed),
                                                                                                    site, out, stats, keepbits,
                                                                                                    pool && p > 0.f ? (const uint32_t*)pool : nullptr,
                                                                                                    pool && p > 0.f ? (uint32_t)(2 * pool_words - 1) : 0u);
    STEP_LAUNCH_CHECK("pt_attention_fwd_bf16");
    return STEP_OK;
}
extern "C" int step_pt_attention_bwd_bf16(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* stats, long S, int T, float p,
                                          uint64_t seed, uint32_t site, uint16_t* dqkv, const uint32_t* keepbits, void* stream) {
    STEP_REQUIRE(qkv && out && dout && stats && dqkv && S > 0 && T > 0 && T <= 352 && p >= 0.f && p < 1.f &&
                 ((((uintptr_t)qkv) | ((uintptr_t)out) | ((uintptr_t)dout) | ((uintptr_t)dqkv)) & 15) == 0,
                 "pt_attention_bwd_bf16: bad arguments (T=%d; at most 352 tokens, 16-byte aligned bf16 tensors)", T);
    const int Tp = (T + 31) & ~31;
    if (!attn_v1_forced() && step_attn2_fits(T) && (p == 0.f || keepbits))
        return step_attn2_bwd(qkv, out, dout, stats, S, T, p, dqkv, keepbits, (hipStream_t)stream);
    STEP_REQUIRE(ma_bwd_lds(Tp) <= 160 * 1024, "pt_attention_bwd_bf16: %d tokens do not fit the LDS", T);
    STEP_TRY(step_raise_lds_once((const void*)attn_mfma_bwd_kernel, 160 * 1024, "pt_attention_bwd_bf16"));
    attn_mfma_bwd_kernel<<<(unsigned)(S * H), 64 * (Tp / 32), ma_bwd_lds(Tp), (hipStream_t)stream>>>(qkv, out, dout, stats, T, Tp, p, SEED_LO(seed),
                                                                                                    SEED_HI(seed), site, dqkv, keepbits);
    STEP_LAUNCH_CHECK("pt_attention_bwd_bf16");
    return STEP_OK;
}
// Linear layer whose result is stored as bf16 (GEMM_FUSED_BF16OUT epilogue of the staged GEMM): out = x w + bias
extern "C" int step_pt_linear_bf16out(const float* x, const float* w, long swk, long swn, const float* bias, long R, int N, int K, uint16_t* out,
                                      void* stream) {
    STEP_REQUIRE(x && w && out && R > 0 && R < (1L << 31) && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0 && (swk == 1 || swn == 1),
                 "pt_linear_bf16out: bad arguments");
    StepGemm g = gemm_desc((int)R, N, K, x, K, 1, w, swk, swn, (float*)out, N);
    g.compute_bf16 = 1;
    GemmFused fu;
    memset(&fu, 0, sizeof(fu));
    fu.channels = 1; fu.period = N; fu.flags = GEMM_FUSED_BF16OUT;
    fu.ffn_bias = bias;
    return step_gemm_launch_fused(g, fu, (hipStream_t)stream);
}
// Feed-forward hidden layer with its ReLU and dropout in the GEMM epilogue, stored as bf16 (bf16 contraction mode):
//   hidden[r][j] = dropout(relu(x[r,:] . w1[j,:] + b1[j]))       x [R,96], w1 [384,96], hidden bf16 [R,384]
// The keep decisions are those of step_pt_dropout(seed, site) on an [R,384] tensor, so the unfused path replays the same masks.
extern "C" int step_pt_ffn_hidden_fwd(const float* x, const float* w1, const float* b1, long R, float p, uint64_t seed, uint32_t site,
                                      uint16_t* hidden, void* stream) {
    STEP_REQUIRE(x && w1 && b1 && hidden && R > 0 && R < (1L << 31) && p >= 0.f && p < 1.f, "pt_ffn_hidden_fwd: bad arguments");
    StepGemm g = gemm_desc((int)R, 4 * D, D, x, D, 1, w1, 1, D, (float*)hidden, 4 * D);
    g.compute_bf16 = 1;
    GemmFused fu;
    memset(&fu, 0, sizeof(fu));
    fu.channels = 1; fu.period = 4 * D; fu.flags = GEMM_FUSED_FFN_FWD;
    fu.ffn_bias = b1; fu.p = p; fu.seed_lo = SEED_LO(seed); fu.seed_hi = SEED_HI(seed); fu.site = site;
    return step_gemm_launch_fused(g, fu, (hipStream_t)stream);
}
// Its backward through dropout and ReLU: dhidden[r][j] = hidden[r][j] != 0 ? (dy[r,:] . w2[:,j]) / (1 - p) : 0, bf16 [R,384]
// (dy [R,96] = gradient of the second linear layer's output, w2 [96,384]).
extern "C" int step_pt_ffn_hidden_bwd(const float* dy, const float* w2, const uint16_t* hidden, long R, float p, uint16_t* dhidden,
                                      void* stream) {
    STEP_REQUIRE(dy && w2 && hidden && dhidden && R > 0 && R < (1L << 31) && p >= 0.f && p < 1.f, "pt_ffn_hidden_bwd: bad arguments");
    StepGemm g = gemm_desc((int)R, 4 * D, D, dy, D, 1, w2, 4 * D, 1, (float*)dhidden, 4 * D);
    g.compute_bf16 = 1;
    GemmFused fu;
    memset(&fu, 0, sizeof(fu));
    fu.channels = 1; fu.period = 4 * D; fu.flags = GEMM_FUSED_MASKNZ;
    fu.maskx = hidden; fu.p = p;
    return step_gemm_launch_fused(g, fu, (hipStream_t)stream);
}
// out[c] += sum_r x[r][c] for a bf16 matrix
extern "C" int step_pt_colsum_bf16(const uint16_t* x, long rows, int cols, float* out, void* stream) {
    STEP_REQUIRE(x && out && rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 2048 && (((uintptr_t)x) & 15) == 0,
                 "pt_colsum_bf16: bad arguments (cols must be a multiple of 8, at most 2048, x 16-byte aligned)");
    const int slab = 1024;
    colsum_bf16_kernel<<<(unsigned)((rows + slab - 1) / slab), 256, 0, (hipStream_t)stream>>>(x, rows, cols, slab, out);
    STEP_LAUNCH_CHECK("pt_colsum_bf16");
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
