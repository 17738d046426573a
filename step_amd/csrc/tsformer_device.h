// Device helpers of the fused TSFormer encoder kernel (tsformer_encoder.hip).  Everything here is internal linkage.
#pragma once
#include "common.h"
#include "step_internal.h"
#include "tsformer_layout.h"

namespace {

struct EncArgs {
    const float* series;
    int S, L, P, depth, nkt;
    const char* wpack;
    uint16_t* hid_bf16;
    float* hid_f32;
    float* last_f32;
    float* sqn;
    float keep;                              // 1 - dropout probability (1 when dropout is off)
    float inv_keep, inv_keep2;               // 1 / keep and its square, computed on the host: as kernel arguments they are scalar operands
                                             // of the packed fmas, not loop-invariant vector register pairs that end up in scratch
    uint32_t seed;
    const unsigned long long* pool;          // Bernoulli(keep) lane-mask words (step_dropout_pool_fill); NULL when dropout is off
    uint32_t pool_mask;                      // number of pool words - 1 (a power of two); 16 more words (a copy of the first 16) follow
    unsigned int* fallback;                  // optional 64 counters (summed by the caller): (wave, head) pairs that ran the re-shifting softmax loop; NULL = not counted
    bool f16;          // operand fragments of wpack are float16 (else bfloat16)
    bool always_rescale;                     // test hook: take the softmax re-shift path on every key tile
    unsigned int* range_word;                // STEP_ENC_RANGE_FLAG: fallback_count + 64, OR-ed with 1 when a wave's output is not finite (float16 operand overflow); else NULL
    int nseq;                                // sequences per workgroup: 1, or 2 (each half of the waves owns one; P <= 192)
    int grid_limit;                          // > 0: persistent launch of at most this many workgroups (each loops over sequences)
};

// The 16-bit operand type of the kernel is a template parameter: bfloat16 (v_mfma_f32_32x32x16_bf16) or float16
// (v_mfma_f32_32x32x16_f16: same rate, same 2-byte fragment layout, 3 more mantissa bits; operand magnitudes on this path
// stay below 200, tools/encoder_precision_study.py).  The packed weight buffer says which one it holds (header word 3).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <bool F16> struct Opnd { typedef bf16x8 v8; typedef __bf16 elem; };
template <> struct Opnd<true> { typedef f16x8 v8; typedef _Float16 elem; };

template <bool F16>
__device__ __forceinline__ f32x16 mfma16(typename Opnd<F16>::v8 a, typename Opnd<F16>::v8 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ typename Opnd<F16>::v8 lfrag(const char* lds, int frag, int lane) {
    return *(const typename Opnd<F16>::v8*)(lds + frag * TSF_FRAG + lane * 16);
}
template <bool F16>
__device__ __forceinline__ typename Opnd<F16>::v8 pack_half(const f32x16& v, int s) {
    if constexpr (F16) {
        f32x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[8 * s + j];
        return __builtin_convertvector(t, f16x8);              // 4 x v_cvt_pk_f16_f32 (round to nearest even)
    } else {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[8 * s + j];
        return pack8(t);
    }
}

// ---- dropout keep-masks ---------------------------------------------------------------------------------------------
// A keep-mask is a 64-bit LANE mask per accumulator register: bit l of word i says whether lane l keeps register i.  The words
// are not generated in this kernel: they are read with scalar loads (s_load_dwordx8/x16, no vector-ALU work at all) from a pool
// of Bernoulli(keep) bits that step_dropout_pool_fill() regenerates from the step's seed (Philox4x32-10) before every launch,
// and applied with ONE v_cndmask_b32 per element (the mask is the instruction's SGPR-pair operand).  Every (sequence, layer)
// owns a contiguous window of the pool ("chunk") at a hashed offset of WORD granularity (scalar loads only need dword alignment),
// so two chunks that overlap are shifted against each other by a whole number of registers unless their bases coincide exactly
// (probability 1 / pool words per pair; tests/test_encoder_dropout_pool.py counts them).  The pool is followed by a copy of its
// first 16 words, so a 16-word group that starts in the last 15 words reads on without wrapping.  Inside the chunk every site
// has its own words, so no two elements of one (sequence, layer) ever share a bit (tests/enc_dropout_host.py mirrors this layout):
//     attention probabilities  (head hd, query tile = wave, key tile kt)   ATT + ((hd*nkt + wave)*nkt + kt)*16 + reg
//     FFN hidden units         (wave, chunk ch of 32 units)                FFN + (wave*12 + ch)*16 + reg
//     dropout1 / dropout2      (wave, 32-feature block t)                  D1 / D2 + (wave*3 + t)*16 + reg
// and chunk `depth` (one past the last layer) carries the positional-encoding dropout in its D1 words.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
struct DropLayout {
    uint32_t ffn, d1, d2, words;            // word offsets inside a chunk, chunk length
    __device__ __host__ explicit DropLayout(int nkt) {
        ffn = 4u * nkt * nkt * 16u;
        d1 = ffn + nkt * 12u * 16u;
        d2 = d1 + nkt * 48u;
        words = d2 + nkt * 48u;
    }
};
__device__ __forceinline__ uint32_t drop_chunk_base(uint32_t seed, uint32_t seq, uint32_t layer, uint32_t pool_mask) {
    return mix32(seed + seq * 0x9E3779B1u + (layer + 1u) * 0x632BE5ABu) & pool_mask;
}
// The pool is read through the constant address space: that is what lets hipcc use scalar loads for it (a plain global
// pointer inside the by-value argument struct is not provably unclobbered, and would be fetched with vector loads +
// v_readfirstlane).  Nothing in this kernel writes the pool.
typedef const __attribute__((address_space(4))) unsigned long long* mask_ptr;
// v[i] = keep ? v[i] : 0 for the 16 accumulator registers of this lane; w is wave-uniform (scalar loads)
__device__ __forceinline__ void keep16(f32x16& v, mask_ptr w) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_inverse_ballot_w64(w[i]) ? v[i] : 0.f;
}

// Experimental (A/B builds only, default 0 = not emitted): raise the wave's issue priority while it feeds the matrix pipe.
//   1: around the score / PV MFMAs of the attention loop;  2: also around the QKV, out-projection and FFN MFMA chains.
#ifndef TSF_SETPRIO
#define TSF_SETPRIO 0
#endif
#define TSF_PRIO_ATTN(x) do { if (TSF_SETPRIO >= 1) __builtin_amdgcn_s_setprio(x); } while (0)
#define TSF_PRIO_CHAIN(x) do { if (TSF_SETPRIO >= 2) __builtin_amdgcn_s_setprio(x); } while (0)

// training-mode tail of a sub-layer: acc = keep-mask(acc) * scale + x, with x taken from the 16-bit operand copy of the
// residual stream (the f32 copy is not kept live across the sub-layer).  w[t]: the 16 mask words of 32-feature block t (each
// 16-word group is contiguous in the pool; consecutive groups may wrap around its end).
template <bool F16>
__device__ __forceinline__ void add_residual_op(f32x16 (&acc)[3], const typename Opnd<F16>::v8 (&xb)[6], const mask_ptr (&w)[3], float scale) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float x[16];
        if constexpr (F16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { x[j] = (float)xb[2 * t][j]; x[8 + j] = (float)xb[2 * t + 1][j]; }
        } else {
            u32x4 lo = __builtin_bit_cast(u32x4, xb[2 * t]), hi = __builtin_bit_cast(u32x4, xb[2 * t + 1]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t wl = lo[j >> 1], wh = hi[j >> 1];
                x[j] = bf16_bits_to_f32((j & 1) ? (wl >> 16) : (wl & 0xffffu));
                x[8 + j] = bf16_bits_to_f32((j & 1) ? (wh >> 16) : (wh & 0xffffu));
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[t][i] = __builtin_amdgcn_inverse_ballot_w64(w[t][i]) ? __builtin_fmaf(acc[t][i], scale, x[i]) : x[i];
    }
}

// the lane id, recomputed (two VALU ops) instead of kept alive: `volatile` keeps the compiler from merging it with earlier copies
__device__ __forceinline__ int fresh_lane_id() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// a wave-uniform value (kernel argument, pointer base) made opaque at its point of use: inside the persistent sequence loop the compiler
// otherwise hoists what it derives from it -- a VGPR copy of a table base, a scale product -- out of the loop and keeps it alive in
// SCRATCH memory for the whole sequence (12 B/lane = 20 MB of HBM writes per launch at PEMS04, profiles/r05_zz_encoder_pmc.json)
template <typename T>
__device__ __forceinline__ T* fresh_uniform(T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ float fresh_uniform(float x) {
    uint32_t b = __float_as_uint(x);
    asm volatile("" : "+s"(b));
    return __uint_as_float(b);
}

// value of the 16-bit operand type nearest to x (what the MFMA will see when x is stored into an operand slot)
template <bool F16>
__device__ __forceinline__ float round_to_operand(float x) {
    if constexpr (F16) return (float)(_Float16)fminf(fmaxf(x, -60000.f), 60000.f);
    else return bf16_bits_to_f32(f32_to_bf16_bits(x));
}
// {x of lane & 31, x of (lane & 31) + 32} in every lane: one v_permlane32_swap, no LDS round trip
__device__ __forceinline__ void both_halves(float x, float& lo, float& hi) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}

// sum over the 64 lanes: xor butterflies inside each half on the LDS crossbar (ds_swizzle, bit mode), then the two halves
__device__ __forceinline__ float wave_sum_swz(float v) {
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x041F));      // and 0x1f, xor 1
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x081F));      // xor 2
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F));      // xor 4
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x201F));      // xor 8
    v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));      // xor 16
    float lo, hi;
    both_halves(v, lo, hi);
    return lo + hi;
}

// LayerNorm over the 96 features of token (lane&31): each lane holds 48, its partner lane^32 the rest.
// g / b point at this lane-half's 48 values (accumulator-register order).
#ifndef TSF_ABLATE
#define TSF_ABLATE 0
#endif
#ifndef TSF_LN_TWO_FMA
#define TSF_LN_TWO_FMA 0     // 1: LayerNorm's last step as two fmas per element (A/B builds: encoder alone 2.00-2.02 -> 1.96-1.98 ms, C2 3.377 -> 3.364 ms,
                             // profiles/r06_zl_encoder_ln_two_fma.log).  Not the default: the unparked eight-tile variants without dropout then need 20 bytes of
                             // scratch per lane, and leaving only those on the old form breaks the bit-identity of one vs two sequences per workgroup
#endif
#ifndef TSF_LN_PACKED
#define TSF_LN_PACKED 1      // 0: the scalar reduction chains of the first version (A/B builds)
#endif
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
template <bool TWO_FMA = (TSF_LN_TWO_FMA != 0)>
__device__ __forceinline__ void layer_norm96(f32x16 (&a)[3], const float* gg, const float* bb) {
    if (TSF_ABLATE & 128) return;
    float s, q;
    if (TSF_LN_PACKED) {
        // both reductions as two chains of packed adds / fmas (v_pk_add_f32, v_pk_fma_f32): half the instructions of a
        // scalar chain and a quarter of its dependency depth
        f32x2_t s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                s0 += f32x2_t{a[t][i], a[t][i + 1]};
                s1 += f32x2_t{a[t][i + 2], a[t][i + 3]};
            }
        s = (s0[0] + s0[1]) + (s1[0] + s1[1]);
    } else {
        s = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += a[t][i];
    }
    {   // the partner half's sum by v_permlane32_swap: no lane-address register to keep alive (a ds_bpermute needs one)
        float lo, hi;
        both_halves(s, lo, hi);
        s = lo + hi;
    }
    const float mean = s * (1.0f / 96.0f);
    if (TSF_LN_PACKED) {
        const f32x2_t m2 = {mean, mean};
        f32x2_t q0 = {0.f, 0.f}, q1 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const f32x2_t d0 = f32x2_t{a[t][i], a[t][i + 1]} - m2, d1 = f32x2_t{a[t][i + 2], a[t][i + 3]} - m2;
                q0 += d0 * d0;
                q1 += d1 * d1;
            }
        q = (q0[0] + q0[1]) + (q1[0] + q1[1]);
    } else {
        q = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { float d = a[t][i] - mean; q += d * d; }
    }
    {
        float lo, hi;
        both_halves(q, lo, hi);
        q = lo + hi;
    }
    const float rstd = rsqrtf(q * (1.0f / 96.0f) + 1e-5f);
    if constexpr (TWO_FMA) {
    // two fused multiply-adds per element ((a rstd - mean rstd) g + b: packed f32 fmas, 4.6 cycles per pair each) instead of subtract, multiply
    // and one fma (2.9 + 2.3 + 2.3 per element)
    const float nm = -mean * rstd;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[t][i] = __builtin_fmaf(__builtin_fmaf(a[t][i], rstd, nm), gg[t * 16 + i], bb[t * 16 + i]);
    } else {
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[t][i] = (a[t][i] - mean) * rstd * gg[t * 16 + i] + bb[t * 16 + i];
    }
}

// One 1 KB piece global -> LDS by the DMA path (no VGPR round trip, invisible to hipcc's waitcnt
// bookkeeping: the kernel waits with an explicit vmcnt(0) at the next stage boundary).
// lds_dst is wave-uniform; lane i's 16 bytes land at lds_dst + 16 i.
__device__ __forceinline__ void dma_1k(const char* gsrc_lane, uint32_t lds_dst) {
    uint32_t keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);      // provably scalar for the "s" operand (the compiler has been seen to hand over a vector register)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_dst)
                 : "memory");
}
#define LDS_ADDR(p) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(p))


}  // namespace
