// Self-attention of the TSFormer pre-training step on the matrix cores, second version (round 6).  Reference: the attention inside
// nn.TransformerEncoderLayer (step/step_arch/tsformer/transformer_layers.py:7-21: softmax(q k^T / sqrt(24)) -> dropout -> . v, four heads
// of 24), forward and backward, bf16 activations [S][T][288] / [S][T][96] as in pretrain.hip (whose kernels remain the path for keep
// decisions drawn from Philox and the reference implementation of this file's tests).
//
// Same decomposition as attn_mfma_{fwd,bwd}_kernel -- one workgroup per (sequence, head), one wave per 32-token tile, scores computed
// TRANSPOSED (S^T = K Q^T) so that an accumulator tile is the next product's operand by a plain pack, backward in two orientations
// (phase A: wave = query tile -> dQ; phase B: wave = key tile -> dK, dV) -- rebuilt around what one gfx950 SIMD charges per vector
// instruction (tools/valu_rate_probe.cpp, profiles/r06_s_valu_rate_probe.log: v_fma / v_and / v_sub 2.9 cycles, v_cndmask / v_bfe 4.5,
// packs 5.0, v_exp 8.5, and a 32x32x16 product ~16 cycles of the other waves' vector issue): the kernels are bound by vector ISSUE, so
// every score element is touched by as few instructions as the arithmetic allows:
//   * P = exp2(fma(score, c, e_q)) with c = log2(e) / sqrt(24) and ONE per-query offset e_q that already holds the row maximum, the
//     1 / row sum and (training) the survivor scale 1 / (1 - p): no subtract, no scale, no multiply by the keep factor;
//   * keep decisions: one v_bfe_i32 (bit -> 0 / -1) and one v_and per use (P >= 0 and d P are masked as bit patterns) instead of
//     shift + and + compare + select + multiply;
//   * d S = P' (dP masked - delta'): the survivor scale lives in P' = P / (1 - p) and delta' = delta (1 - p);
//   * padded keys are masked in the LAST key tile only (a wave-uniform branch), padded queries through e_q = -1e30, and phase B does not
//     mask padded keys at all (their columns are never stored);
//   * the operands of head dimensions 24 .. 31 come from a zero area through a lane-dependent base address and stride (no selects in the loops);
//   * fills: rows are copied as 16-byte pieces; the transposed arrays are built from the row-major ones with 16-byte LDS reads,
//     v_perm_b32 and 8-byte writes (the first version wrote 24 two-byte pieces per token and array); results leave as bf16 through an
//     8-byte-per-lane staging tile.
#include "common.h"
#include <stdlib.h>
#include "step_internal.h"

namespace {

#ifndef A2_ABLATE
#define A2_ABLATE 0       // timing experiments only (WRONG results): 1 no phase A, 2 no phase B, 4 no transposes, 8 no keep-word fill, 16 no global loads, 32 no global stores, 64 no forward key loop
#endif
#ifndef A2_BWD_WAVES
#define A2_BWD_WAVES 4    // waves per SIMD the register allocation has to admit (see the occupancy note at the launchers)
#endif
#ifndef A2_FWD_WAVES
#define A2_FWD_WAVES 5
#endif
constexpr int D = 96, H = 4, DH = 24;
constexpr int RP = 24;                    // row pitch (bf16 elements) of the row-major LDS arrays: 48 B, conflict-free 16-byte reads per 16 lanes
__device__ __host__ __forceinline__ int tpitch(int Tp) { return Tp + 8; }          // pitch of the transposed arrays
constexpr float SCALE = 0.20412414523193154f;              // 1 / sqrt(24)
constexpr float C2 = 0.20412414523193154f * 1.4426950408889634f;

// workgroups are dealt round-robin to the 8 XCDs: XCD x gets a CONTIGUOUS range of (sequence, head) units (the four heads of a sequence share
// every 128-byte line of its rows)
__device__ __forceinline__ unsigned unit_of(unsigned b, unsigned n) {
    const unsigned per = n >> 3, rem = n & 7u, x = b & 7u;
    return x * per + (x < rem ? x : rem) + (b >> 3);
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bf16x8 as_op(const uint4& v) { return __builtin_bit_cast(bf16x8, v); }
// keys k0 + {4h .. 4h+3, 8+4h .. 8+4h+3} of one row of a transposed array (the k-slot pairing of a packed accumulator half)
__device__ __forceinline__ bf16x8 tr8(const uint16_t* trow, int k0, int h) {
    const uint2 a = *(const uint2*)(trow + k0 + 4 * h), b = *(const uint2*)(trow + k0 + 8 + 4 * h);
    return as_op(make_uint4(a.x, a.y, b.x, b.y));
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}
__device__ __forceinline__ int reg_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }      // accumulator register -> row of the tile
__device__ __forceinline__ float and_mask(float v, int m) { return __int_as_float(__float_as_int(v) & m); }
// bit `off` of w as 0 / -1 in ONE v_bfe_i32 (left to itself the compiler turns a constant-position test into and + compare + select)
__device__ __forceinline__ int bit_mask(int w, int off) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "n"(off));
    return m;
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ float swap_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swap_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Row operand of a row-major array: the A (or B) fragment of k-step st for row `r0 + lane & 31` is dims 16 st + 8 h .. + 7.  Lanes of half 1
// have nothing at st = 1 (dims 24 .. 31): their pointer goes to the zero area and does not move with the tile.
struct RowPtr {
    const uint16_t* p0;      // st = 0
    const uint16_t* p1;      // st = 1
    int step1;               // elements per 32-row tile at st = 1 (0 for the zero area)
};
__device__ __forceinline__ RowPtr row_ptr(const uint16_t* arr, const uint16_t* zero, int col, int h) {
    RowPtr r;
    r.p0 = arr + col * RP + 8 * h;
    r.p1 = h ? zero : arr + col * RP + 16;
    r.step1 = h ? 0 : 32 * RP;
    return r;
}
// the same with per-row values in slots 24 .. 31 (an array of 8 elements per row): the backward's query operand, whose slots 24 / 25 carry the
// exponent offset e_q through the contraction (the key side holds ones there)
__device__ __forceinline__ RowPtr row_ptr_x(const uint16_t* arr, const uint16_t* extra8, int col, int h) {
    RowPtr r;
    r.p0 = arr + col * RP + 8 * h;
    r.p1 = h ? extra8 + col * 8 : arr + col * RP + 16;
    r.step1 = h ? 32 * 8 : 32 * RP;
    return r;
}
__device__ __forceinline__ bf16x8 row_op0(const RowPtr& r, int tile) { return as_op(*(const uint4*)(r.p0 + tile * (32 * RP))); }
__device__ __forceinline__ bf16x8 row_op1(const RowPtr& r, int tile) { return as_op(*(const uint4*)(r.p1 + tile * r.step1)); }

// 4 rows x 8 columns (one 16-byte piece of four consecutive rows) -> 8 columns x 4 rows as 8-byte pieces dst[c * pitch .. + 3]
__device__ __forceinline__ void transpose_4x8(const uint4 (&r)[4], uint16_t* dst, int pitch) {
    const uint32_t w[4][4] = {{r[0].x, r[0].y, r[0].z, r[0].w}, {r[1].x, r[1].y, r[1].z, r[1].w}, {r[2].x, r[2].y, r[2].z, r[2].w}, {r[3].x, r[3].y, r[3].z, r[3].w}};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t sel = (c & 1) ? 0x07060302u : 0x05040100u;
        const uint32_t lo = __builtin_amdgcn_perm(w[1][c >> 1], w[0][c >> 1], sel), hi = __builtin_amdgcn_perm(w[3][c >> 1], w[2][c >> 1], sel);
        *(uint2*)(dst + c * pitch) = make_uint2(lo, hi);
    }
}
// three of an accumulator tile's four register groups (rows 8 g + 4 h + 0..3, g = 0..2: the 24 head dimensions) -> bf16, 8 bytes each
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__device__ __forceinline__ void stage24(uint16_t* dst, int h, const f32x16& v, float scale) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const f32x4 t = {v[4 * g] * scale, v[4 * g + 1] * scale, v[4 * g + 2] * scale, v[4 * g + 3] * scale};
        *(uint2*)(dst + 8 * g + 4 * h) = __builtin_bit_cast(uint2, __builtin_convertvector(t, bf16x4));      // 2 x v_cvt_pk_bf16_f32
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward
// out [S][T][96] bf16, stats [S][H][T][2] = (row maximum of the scaled scores, row sum) as attn_kernel writes them, keepbits [S][H][T][nt]:
// the keep word of (query, key tile) -- here the pool's 32-bit word at the (sequence, head)'s hashed offset, handed on to the backward
template <bool DROP>
__global__ __launch_bounds__(704, A2_FWD_WAVES) void attn2_fwd_kernel(const uint16_t* __restrict__ qkv, int T, int Tp, float p, uint32_t lo, uint32_t site,
                                                        uint16_t* __restrict__ out, float* __restrict__ stats, uint32_t* __restrict__ keepbits,
                                                        const uint32_t* __restrict__ pool32, uint32_t pmask32) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ml[];
    const int TPt = tpitch(Tp), nt = Tp >> 5;
    uint16_t* Ks = ml;                        // [Tp][RP]
    uint16_t* VT = Ks + Tp * RP;              // [DH][TPt]
    uint16_t* ZT = VT + DH * TPt;             // [TPt] zeros
    uint32_t* bits = (uint32_t*)(ZT + TPt);   // [nt][Tp] keep word of (key tile, query)
    const unsigned unit = unit_of(blockIdx.x, gridDim.x);
    const long s = unit / H;
    const int hd = unit % H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
    const uint16_t* base = qkv + s * (long)T * 288 + hd * DH;
    const int q = wave * 32 + col;
    const long srow = (s * H + hd) * (long)T;
    if (DROP) {
        // the keep words of this (sequence, head): T * nt consecutive 32-bit words of the pool at a hashed offset, word q * nt + kt for (query q,
        // key tile kt) -- copied to `keepbits` for the backward as one contiguous run and kept in LDS for the loop below (read per lane and
        // tile inside the loop they were scattered 4-byte loads and stores: 0.2 of the forward's 0.5 ms at 168 tokens)
        const uint32_t pbase = mix32(lo + unit * 0x9E3779B1u + (site + 1u) * 0x632BE5ABu);
        for (int i = tid; i < T * nt; i += blockDim.x) {
            const uint32_t w = pool32[(pbase + (uint32_t)i) & pmask32];
            keepbits[srow * nt + i] = w;
            const int qi = i / nt;
            bits[(i - qi * nt) * Tp + qi] = w;
        }
    }
    // this lane's query operand straight from memory (dims 8 h .. and 16 + 8 h ..; the latter exists for half 0 only)
    uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0;
    if (q < T && !(A2_ABLATE & 16)) {
        q0 = *(const uint4*)(base + (long)q * 288 + 8 * h);
        if (h == 0) q1 = *(const uint4*)(base + (long)q * 288 + 16);
    }
    for (int e = tid; e < Tp * 3; e += blockDim.x) {                  // K rows as they are
        const int t = e / 3, d8 = e - 3 * t;
        uint4 k8 = make_uint4(0u, 0u, 0u, 0u);
        if (t < T && !(A2_ABLATE & 16)) k8 = *(const uint4*)(base + (long)t * 288 + 96 + d8 * 8);
        *(uint4*)(Ks + t * RP + d8 * 8) = k8;
    }
    for (int e = tid; e < (Tp >> 2) * 3; e += blockDim.x) {           // V transposed: four tokens x eight dimensions per item
        const int tq = e / 3, d8 = e - 3 * tq;
        uint4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = 4 * tq + i;
            r[i] = make_uint4(0u, 0u, 0u, 0u);
            if (t < T && !(A2_ABLATE & 16)) r[i] = *(const uint4*)(base + (long)t * 288 + 192 + d8 * 8);
        }
        transpose_4x8(r, VT + (d8 * 8) * TPt + 4 * tq, TPt);
    }
    for (int i = tid; i < TPt; i += blockDim.x) ZT[i] = 0;
    __syncthreads();
    const bf16x8 bq0 = as_op(q0), bq1 = as_op(q1);
    const RowPtr kp = row_ptr(Ks, ZT, col, h);
    const bool ragged = (T & 31) != 0;
    const f32x16 zero = zero16();
    // pass 1: row maximum (the keys of the last tile that do not exist are taken out)
    float m0 = -1e30f, m1 = -1e30f;
    // (every loop below requests the NEXT tile's row operands right behind the products that consumed this tile's, and the transposed operands
    //  of a tile in front of its vector work: left to the compiler, each read sits directly in front of its product with a full wait)
    for (int kt = 0; kt < nt; ++kt) {
        f32x16 sc = mfma(row_op0(kp, kt), bq0, zero);
        sc = mfma(row_op1(kp, kt), bq1, sc);
        if (ragged && kt == nt - 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) if (kt * 32 + reg_row(e, h) >= T) sc[e] = -1e30f;
        }
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            m0 = fmaxf(fmaxf(m0, sc[e]), sc[e + 1]);
            m1 = fmaxf(fmaxf(m1, sc[e + 2]), sc[e + 3]);
        }
    }
    const float mx = swap_max(fmaxf(m0, m1));
    const float eq = -mx * C2;
    const uint16_t* vrow = col < DH ? VT + col * TPt : ZT;
    float l = 0.f;
    f32x16 o = zero;
    // (what speeds these kernels up is WAVES per SIMD, not fewer instructions or earlier reads -- profiles/r06_w_attn2_pmc.txt: at 166 registers the
    //  backward ran one workgroup per compute unit, 1.4 waves per SIMD, and a lone wave issues a vector instruction every ~7.5 cycles -- so the
    //  loops keep few values alive: the probabilities overwrite the scores, one k-step half at a time)
    for (int kt = 0; kt < ((A2_ABLATE & 64) ? 0 : nt); ++kt) {
        int wh = 0;
        if (DROP) wh = (int)(bits[kt * Tp + q] >> (4 * h));          // (queries >= T: whatever is there -- their columns are never stored)
        f32x16 sc = mfma(row_op0(kp, kt), bq0, zero);
        sc = mfma(row_op1(kp, kt), bq1, sc);
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = __builtin_fmaf(sc[e], C2, eq);
        if (ragged && kt == nt - 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) if (kt * 32 + reg_row(e, h) >= T) sc[e] = -1e30f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 vt = tr8(vrow, kt * 32 + 16 * s2, h);
            float pv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 8 * s2 + i;
                pv[i] = __builtin_amdgcn_exp2f(sc[e]);
                l += pv[i];
                if (DROP) pv[i] = and_mask(pv[i], bit_mask(wh, (e & 3) + 8 * (e >> 2)));
            }
            o = mfma(vt, pack8(pv), o);
            FENCE();
        }
    }
    l = swap_sum(l);
    if (q < T && h == 0) { stats[(srow + q) * 2] = mx * SCALE; stats[(srow + q) * 2 + 1] = l; }
    // the output tile leaves through LDS (the operand arrays are dead) as bf16 rows of 24, then 16-byte pieces of each token's 48-byte run
    __syncthreads();
    uint16_t* stage = ml;                     // [Tp][24]
    stage24(stage + q * DH, h, o, (DROP ? 1.f / (1.f - p) : 1.f) / l);
    __syncthreads();
    uint16_t* dst = out + s * (long)T * D + hd * DH;
    for (int e = tid; e < T * 3 && !(A2_ABLATE & 32); e += blockDim.x) {
        const int t = e / 3, d8 = e - 3 * t;
        *(uint4*)(dst + (long)t * D + d8 * 8) = *(const uint4*)(stage + t * DH + d8 * 8);
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward
template <bool DROP>
__global__ __launch_bounds__(704, A2_BWD_WAVES) void attn2_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ out,
                                                        const uint16_t* __restrict__ dout, const float* __restrict__ stats, int T, int Tp, float p,
                                                        uint16_t* __restrict__ dqkv, const uint32_t* __restrict__ keepbits) {
    extern __shared__ __attribute__((aligned(16))) uint16_t ml[];
    const int TPt = tpitch(Tp), nt = Tp >> 5;
    uint16_t* Qs = ml;                        // row-major [Tp][RP]: q, k, v, dO
    uint16_t* Ks = Qs + Tp * RP;
    uint16_t* Vs = Ks + Tp * RP;
    uint16_t* Gs = Vs + Tp * RP;
    uint16_t* QT = Gs + Tp * RP;              // transposed [DH][TPt]: q, k, dO
    uint16_t* KT = QT + DH * TPt;
    uint16_t* GT = KT + DH * TPt;
    uint16_t* ZT = GT + DH * TPt;             // [TPt] zeros
    uint16_t* ONE8 = ZT + TPt;                // (1, 1, 0 ...): slots 24 .. 31 of every key
    uint16_t* E8 = ONE8 + 8;                  // [Tp][8]: slots 24 .. 31 of the queries: (hi, lo, 0 ...) with hi + lo = e_q / C2, P' = exp2(C2 (q . k) + e_q) = P / (1 - p)
    float* sdl = (float*)(E8 + Tp * 8);       // [Tp] delta' = (1 - p) sum_d dO O
    uint32_t* bits = (uint32_t*)(sdl + Tp);   // [nt][Tp] keep word of (key tile, query)
    const unsigned unit = unit_of(blockIdx.x, gridDim.x);
    const long s = unit / H;
    const int hd = unit % H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, h = lane >> 5;
    const uint16_t* base = qkv + s * (long)T * 288 + hd * DH;
    const long srow = (s * H + hd) * (long)T;
    const float keep = DROP ? 1.f - p : 1.f;
    // fill, stage 1: item = (token, 8 head dimensions), 4 lanes per token (the fourth idles); blockDim = 2 Tp: two items per thread, all ten
    // loads in flight before the first is consumed
    uint4 q8[2], k8[2], v8[2], g8[2], o8[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = tid + it * blockDim.x, t = e >> 2, d8 = e & 3;
        q8[it] = make_uint4(0u, 0u, 0u, 0u); k8[it] = q8[it]; v8[it] = q8[it]; g8[it] = q8[it]; o8[it] = q8[it];
        if (t < T && d8 < 3 && !(A2_ABLATE & 16)) {
            const uint16_t* r = base + (long)t * 288 + d8 * 8;
            const long oi = (s * T + t) * D + hd * DH + d8 * 8;
            q8[it] = *(const uint4*)r; k8[it] = *(const uint4*)(r + 96); v8[it] = *(const uint4*)(r + 192);
            g8[it] = *(const uint4*)(dout + oi); o8[it] = *(const uint4*)(out + oi);
        }
    }
    if (DROP && !(A2_ABLATE & 8)) {
        for (int i = tid; i < Tp * nt; i += blockDim.x) {          // the forward's keep words: one contiguous run of T nt words
            const int qi = i / nt;
            bits[(i - qi * nt) * Tp + qi] = i < T * nt ? keepbits[srow * nt + i] : 0u;
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = tid + it * blockDim.x, t = e >> 2, d8 = e & 3;
        const uint32_t gw[4] = {g8[it].x, g8[it].y, g8[it].z, g8[it].w}, ow[4] = {o8[it].x, o8[it].y, o8[it].z, o8[it].w};
        float dl = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dl += __uint_as_float(gw[j] << 16) * __uint_as_float(ow[j] << 16) + __uint_as_float(gw[j] & 0xffff0000u) * __uint_as_float(ow[j] & 0xffff0000u);
        dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64);                    // delta_t = sum_d dO O (4 adjacent lanes)
        if (d8 == 0) sdl[t] = dl * keep;
        if (d8 < 3) {
            const int ro = t * RP + d8 * 8;
            *(uint4*)(Qs + ro) = q8[it]; *(uint4*)(Ks + ro) = k8[it]; *(uint4*)(Vs + ro) = v8[it]; *(uint4*)(Gs + ro) = g8[it];
        }
    }
    for (int i = tid; i < TPt; i += blockDim.x) ZT[i] = 0;
    if (tid < 8) ONE8[tid] = tid < 2 ? 0x3F80 : 0;
    for (int i = tid; i < Tp; i += blockDim.x) {
        float2 st2 = make_float2(0.f, 1.f);
        if (i < T) st2 = *(const float2*)(stats + (srow + i) * 2);
        // stats = (row max of the scaled scores, row sum): P / (1 - p) = exp2(s * C2 - max * log2(e) - log2(sum (1 - p))); queries >= T: -1e30.
        // e_q / C2 as the sum of two bfloat16 (relative error 2^-17: 2e-4 of a probability at |e_q| = 100)
        const float eqc = (i < T ? -st2.x * 1.4426950408889634f - __log2f(st2.y * keep) : -1e30f) * (1.0f / C2);
        const uint32_t hi = f32_to_bf16_bits(eqc), lo2 = f32_to_bf16_bits(eqc - bf16_bits_to_f32(hi));
        *(uint4*)(E8 + i * 8) = make_uint4(hi | (lo2 << 16), 0u, 0u, 0u);
    }
    __syncthreads();
    // fill, stage 2: the transposed arrays from the row-major ones: item = (array, four tokens, 8 dimensions)
    for (int e = tid; e < (Tp >> 2) * 9 && !(A2_ABLATE & 4); e += blockDim.x) {
        const int a = e / ((Tp >> 2) * 3), r2 = e - a * ((Tp >> 2) * 3), tq = r2 / 3, d8 = r2 - 3 * tq;
        const uint16_t* src = (a == 0 ? Qs : a == 1 ? Ks : Gs) + (4 * tq) * RP + d8 * 8;
        uint16_t* dstT = (a == 0 ? QT : a == 1 ? KT : GT) + (d8 * 8) * TPt + 4 * tq;
        uint4 r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *(const uint4*)(src + i * RP);
        transpose_4x8(r, dstT, TPt);
    }
    __syncthreads();
    const bool ragged = (T & 31) != 0;
    const f32x16 zero = zero16();
    f32x16 dq_keep;
    {   // ---- phase A: this wave's 32 queries against every key tile (keys in registers, query = lane)
        const int q = wave * 32 + col;
        const float dlq = sdl[q];
        const RowPtr qp = row_ptr_x(Qs, E8, col, h), gp = row_ptr(Gs, ZT, col, h), kp = row_ptr(Ks, ONE8, col, h), vp = row_ptr(Vs, ZT, col, h);
        const bf16x8 bq0 = row_op0(qp, wave), bq1 = row_op1(qp, wave), bo0 = row_op0(gp, wave), bo1 = row_op1(gp, wave);
        const uint16_t* ktrow = col < DH ? KT + col * TPt : ZT;
        f32x16 dqa = zero;
        for (int kt = 0; kt < ((A2_ABLATE & 1) ? 0 : nt); ++kt) {
            int wh = 0;
            if (DROP) wh = (int)(bits[kt * Tp + q] >> (4 * h));
            f32x16 sc = mfma(row_op0(kp, kt), bq0, zero);
            f32x16 dp = mfma(row_op0(vp, kt), bo0, zero);
            sc = mfma(row_op1(kp, kt), bq1, sc);
            dp = mfma(row_op1(vp, kt), bo1, dp);
            const bf16x8 ta = tr8(ktrow, kt * 32, h), tb = tr8(ktrow, kt * 32 + 16, h);
            float ds[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) ds[e] = sc[e] * C2;
            if (ragged && kt == nt - 1) {
#pragma unroll
                for (int e = 0; e < 16; ++e) if (kt * 32 + reg_row(e, h) >= T) ds[e] = -1e30f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pp = __builtin_amdgcn_exp2f(ds[e]);
                const float dpm = DROP ? and_mask(dp[e], bit_mask(wh, (e & 3) + 8 * (e >> 2))) : dp[e];
                ds[e] = pp * (dpm - dlq);
            }
            dqa = mfma(ta, pack8(ds), dqa);
            dqa = mfma(tb, pack8(ds + 8), dqa);
        }
        dq_keep = dqa;                         // leaves with dK and dV through the staging tile, after phase B is done with the operands
    }
    f32x16 dka = zero, dva = zero;
    {   // ---- phase B: this wave's 32 keys against every query tile (queries in registers, key = lane); padded keys are not masked: their
        //      columns are never stored
        const RowPtr qp = row_ptr_x(Qs, E8, col, h), gp = row_ptr(Gs, ZT, col, h), kp = row_ptr(Ks, ONE8, col, h), vp = row_ptr(Vs, ZT, col, h);
        const bf16x8 bk0 = row_op0(kp, wave), bk1 = row_op1(kp, wave), bv0 = row_op0(vp, wave), bv1 = row_op1(vp, wave);
        const uint16_t* gtrow = col < DH ? GT + col * TPt : ZT;
        const uint16_t* qtrow = col < DH ? QT + col * TPt : ZT;
        for (int qt = 0; qt < ((A2_ABLATE & 2) ? 0 : nt); ++qt) {
            f32x16 sc = mfma(row_op0(qp, qt), bk0, zero);
            f32x16 dp = mfma(row_op0(gp, qt), bv0, zero);
            sc = mfma(row_op1(qp, qt), bk1, sc);
            dp = mfma(row_op1(gp, qt), bv1, dp);
            // one half of the tile (accumulator registers 8 s2 .. 8 s2 + 7 = one k-step of the two products that follow) at a time, P' m and d S
            // written over the scores / d P they come from: the kernel has to stay within 128 registers (two workgroups per compute unit)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 gt = tr8(gtrow, qt * 32 + 16 * s2, h), qtt = tr8(qtrow, qt * 32 + 16 * s2, h);
#pragma unroll
                for (int g = 2 * s2; g < 2 * s2 + 2; ++g) {
                    // queries qt * 32 + 8 g + 4 h + (0..3) are accumulator registers 4 g .. 4 g + 3: their deltas / keep words are 16-byte reads
                    const int q0 = qt * 32 + 8 * g + 4 * h;
                    const float4 dl4 = *(const float4*)(sdl + q0);
                    uint4 bw = make_uint4(0u, 0u, 0u, 0u);
                    if (DROP) bw = *(const uint4*)(bits + wave * Tp + q0);
                    const float dlv[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
                    const uint32_t bwv[4] = {bw.x, bw.y, bw.z, bw.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g + j;
                        const float pp = __builtin_amdgcn_exp2f(sc[e] * C2);      // (e_q rode through the contraction; queries >= T: exp2(-1e30 C2) = 0)
                        if (DROP) {
                            const int m = __builtin_amdgcn_sbfe((int)bwv[j], col, 1);
                            sc[e] = and_mask(pp, m);
                            dp[e] = pp * (and_mask(dp[e], m) - dlv[j]);
                        } else {
                            sc[e] = pp;
                            dp[e] = pp * (dp[e] - dlv[j]);
                        }
                    }
                }
                float pm8[8], ds8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { pm8[i] = sc[8 * s2 + i]; ds8[i] = dp[8 * s2 + i]; }
                dva = mfma(gt, pack8(pm8), dva);
                dka = mfma(qtt, pack8(ds8), dka);
                FENCE();
            }
        }
    }
    // dQ | dK | dV of this wave's 32 tokens -> bf16 staging tile [Tp][3][24] over the dead operands -> 16-byte pieces of the three 48-byte
    // runs each token owns in dqkv
    __syncthreads();
    uint16_t* stage = ml;
    const int tok = wave * 32 + col;
    stage24(stage + (tok * 3) * DH, h, dq_keep, SCALE);
    stage24(stage + (tok * 3 + 1) * DH, h, dka, SCALE);
    stage24(stage + (tok * 3 + 2) * DH, h, dva, 1.f);
    __syncthreads();
    uint16_t* db = dqkv + s * (long)T * 288 + hd * DH;
    for (int e = tid; e < T * 9 && !(A2_ABLATE & 32); e += blockDim.x) {
        const int t = e / 9, r = e - 9 * t, sec = r / 3, d8 = r - 3 * sec;
        *(uint4*)(db + (long)t * 288 + sec * 96 + d8 * 8) = *(const uint4*)(stage + (t * 3 + sec) * DH + d8 * 8);
    }
}

size_t fwd_lds(int Tp) {
    const size_t ops = (size_t)(Tp * RP + (DH + 1) * tpitch(Tp)) * 2 + (size_t)Tp * (Tp / 32) * 4, stage = (size_t)Tp * DH * 2;
    return ops > stage ? ops : stage;
}
size_t bwd_lds(int Tp) {
    const size_t ops = (size_t)(4 * Tp * RP + (3 * DH + 1) * tpitch(Tp) + 8 + 8 * Tp) * 2 + (size_t)(Tp + Tp * (Tp / 32)) * 4, stage = (size_t)Tp * 3 * DH * 2;
    return ops > stage ? ops : stage;
}

}  // namespace

// launchers used by step_pt_attention_{fwd,bwd}_bf16 (pretrain.hip): keep decisions from the step's pool (or none)
bool step_attn2_fits(int T) { return bwd_lds((T + 31) & ~31) <= 160 * 1024 && T <= 352; }
int step_attn2_fwd(const uint16_t* qkv, long S, int T, float p, uint64_t seed, uint32_t site, uint16_t* out, float* stats, uint32_t* keepbits,
                   const uint64_t* pool, long pool_words, hipStream_t st) {
    const int Tp = (T + 31) & ~31;
    const bool drop = p > 0.f;
    const size_t lds = fwd_lds(Tp);
    if (drop) {
        STEP_TRY(step_raise_lds_once((const void*)attn2_fwd_kernel<true>, 160 * 1024, "pt_attention_fwd_bf16"));
        attn2_fwd_kernel<true><<<(unsigned)(S * H), 64 * (Tp / 32), lds, st>>>(qkv, T, Tp, p, (uint32_t)seed, site, out, stats, keepbits,
                                                                              (const uint32_t*)pool, (uint32_t)(2 * pool_words - 1));
    } else {
        STEP_TRY(step_raise_lds_once((const void*)attn2_fwd_kernel<false>, 160 * 1024, "pt_attention_fwd_bf16"));
        attn2_fwd_kernel<false><<<(unsigned)(S * H), 64 * (Tp / 32), lds, st>>>(qkv, T, Tp, p, (uint32_t)seed, site, out, stats, keepbits, nullptr, 0u);
    }
    STEP_LAUNCH_CHECK("pt_attention_fwd_bf16");
    return STEP_OK;
}
int step_attn2_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* stats, long S, int T, float p, uint16_t* dqkv,
                   const uint32_t* keepbits, hipStream_t st) {
    const int Tp = (T + 31) & ~31;
    const size_t lds = bwd_lds(Tp);
    if (p > 0.f) {
        STEP_TRY(step_raise_lds_once((const void*)attn2_bwd_kernel<true>, 160 * 1024, "pt_attention_bwd_bf16"));
        attn2_bwd_kernel<true><<<(unsigned)(S * H), 64 * (Tp / 32), lds, st>>>(qkv, out, dout, stats, T, Tp, p, dqkv, keepbits);
    } else {
        STEP_TRY(step_raise_lds_once((const void*)attn2_bwd_kernel<false>, 160 * 1024, "pt_attention_bwd_bf16"));
        attn2_bwd_kernel<false><<<(unsigned)(S * H), 64 * (Tp / 32), lds, st>>>(qkv, out, dout, stats, T, Tp, p, dqkv, keepbits);
    }
    STEP_LAUNCH_CHECK("pt_attention_bwd_bf16");
    return STEP_OK;
}
