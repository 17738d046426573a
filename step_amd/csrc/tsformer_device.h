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
    float drop_p;
    uint32_t seed;
    bool f16;          // operand fragments of wpack are float16 (else bfloat16)
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

// dropout keep-mask bits: a per-lane xorshift32 stream (6 full-rate VALU ops per 32 bits, no integer
// multiplies in the hot loops) yields four 8-bit Bernoulli draws per step (p_eff = thresh/256; the
// survivor scale uses p_eff, so the estimator stays unbiased).  The stream is re-seeded per
// (sequence, layer, site, token) with a multiplicative hash, so results are launch-deterministic.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// Experimental (A/B builds only, default 0 = not emitted): raise the wave's issue priority while it feeds the matrix pipe.
//   1: around the score / PV MFMAs of the two attention passes;  2: also around the QKV, out-projection and FFN MFMA chains.
#ifndef TSF_SETPRIO
#define TSF_SETPRIO 0
#endif
#define TSF_PRIO_ATTN(x) do { if (TSF_SETPRIO >= 1) __builtin_amdgcn_s_setprio(x); } while (0)
#define TSF_PRIO_CHAIN(x) do { if (TSF_SETPRIO >= 2) __builtin_amdgcn_s_setprio(x); } while (0)

#ifndef TSF_DROPOUT_LCG
#define TSF_DROPOUT_LCG 0          // 1: experimental 24-bit LCG keep-mask generator (A/B builds only, see below)
#endif

#if !TSF_DROPOUT_LCG
struct Dropper {
    uint32_t base;      // seed ^ per-(seq, layer, site) salt
    uint32_t thresh;    // keep iff draw8 >= thresh
    float scale;        // 1/(1-p_eff)
    uint32_t st;        // xorshift state
    __device__ __forceinline__ void seed(uint32_t elem_salt) { st = mix32(base + elem_salt * 0x9E3779B1u) | 1u; }
    __device__ __forceinline__ uint32_t next() {
        st ^= st << 13; st ^= st >> 17; st ^= st << 5;
        return st;
    }
    // 16 accumulator registers of this lane; the stream must have been seeded by the caller
    __device__ __forceinline__ void apply16(f32x16& v) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const uint32_t r = next();
            v[i] = ((r & 0xffu) >= thresh) ? v[i] * scale : 0.f;
            v[i + 1] = (((r >> 8) & 0xffu) >= thresh) ? v[i + 1] * scale : 0.f;
            v[i + 2] = (((r >> 16) & 0xffu) >= thresh) ? v[i + 2] * scale : 0.f;
            v[i + 3] = ((r >> 24) >= thresh) ? v[i + 3] * scale : 0.f;
        }
    }
    // unscaled variant (the caller folds the survivor scale into a later multiply)
    __device__ __forceinline__ void mask16(f32x16& v) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const uint32_t r = next();
            v[i] = ((r & 0xffu) >= thresh) ? v[i] : 0.f;
            v[i + 1] = (((r >> 8) & 0xffu) >= thresh) ? v[i + 1] : 0.f;
            v[i + 2] = (((r >> 16) & 0xffu) >= thresh) ? v[i + 2] : 0.f;
            v[i + 3] = ((r >> 24) >= thresh) ? v[i + 3] : 0.f;
        }
    }
};
#else
// Experimental generator (tools/dropout_generator_study.py): x <- x * 0x43FD45 + 0xC39EC3 mod 2^24 is ONE full-rate
// v_mad_u32_u24 (which reads only bits 0..23 of x) and yields two Bernoulli bytes (bits 16..23, then 8..15): 0.5 VALU op
// per draw instead of 1.5.  Inline asm because hipcc lowers __umul24 of an unmasked value to the quarter-rate v_mul_lo_u32.
struct Dropper {
    uint32_t base, thresh;
    float scale;
    uint32_t st;
    __device__ __forceinline__ void seed(uint32_t elem_salt) { st = mix32(base + elem_salt * 0x9E3779B1u); }
    __device__ __forceinline__ uint32_t next() {
        uint32_t r;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(st), "s"(0x43FD45u), "v"(0xC39EC3u));
        st = r;
        return r;
    }
    __device__ __forceinline__ void apply16(f32x16& v) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const uint32_t r = next();
            v[i] = (((r >> 16) & 0xffu) >= thresh) ? v[i] * scale : 0.f;
            v[i + 1] = (((r >> 8) & 0xffu) >= thresh) ? v[i + 1] * scale : 0.f;
        }
    }
    __device__ __forceinline__ void mask16(f32x16& v) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const uint32_t r = next();
            v[i] = (((r >> 16) & 0xffu) >= thresh) ? v[i] : 0.f;
            v[i + 1] = (((r >> 8) & 0xffu) >= thresh) ? v[i + 1] : 0.f;
        }
    }
};
#endif

// training-mode tail of a sub-layer: acc = dropout(acc) + x, with x taken from the 16-bit operand
// copy of the residual stream (the f32 copy is not kept live across the sub-layer)
template <bool F16>
__device__ __forceinline__ void add_residual_op(f32x16 (&acc)[3], const typename Opnd<F16>::v8 (&xb)[6], Dropper& dr,
                                                uint32_t salt) {
    dr.seed(salt);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        dr.apply16(acc[t]);
        if constexpr (F16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[t][j] += (float)xb[2 * t][j];
                acc[t][8 + j] += (float)xb[2 * t + 1][j];
            }
        } else {
            u32x4 lo = __builtin_bit_cast(u32x4, xb[2 * t]), hi = __builtin_bit_cast(u32x4, xb[2 * t + 1]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t wl = lo[j >> 1], wh = hi[j >> 1];
                acc[t][j] += bf16_bits_to_f32((j & 1) ? (wl >> 16) : (wl & 0xffffu));
                acc[t][8 + j] += bf16_bits_to_f32((j & 1) ? (wh >> 16) : (wh & 0xffffu));
            }
        }
    }
}

// LayerNorm over the 96 features of token (lane&31): each lane holds 48, its partner lane^32 the rest.
// g / b point at this lane-half's 48 values (accumulator-register order).
__device__ __forceinline__ void layer_norm96(f32x16 (&a)[3], const float* gg, const float* bb) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += a[t][i];
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / 96.0f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { float d = a[t][i] - mean; q += d * d; }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / 96.0f) + 1e-5f);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[t][i] = (a[t][i] - mean) * rstd * gg[t * 16 + i] + bb[t * 16 + i];
}

// One 1 KB piece global -> LDS by the DMA path (no VGPR round trip, invisible to hipcc's waitcnt
// bookkeeping: the kernel waits with an explicit vmcnt(0) at the next stage boundary).
// lds_dst is wave-uniform; lane i's 16 bytes land at lds_dst + 16 i.
__device__ __forceinline__ void dma_1k(const char* gsrc_lane, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_dst)
                 : "memory");
}
#define LDS_ADDR(p) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(p))


}  // namespace
