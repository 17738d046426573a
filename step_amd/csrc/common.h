// Shared host/device helpers for libstep_hip (gfx950 / CDNA4 only).
#pragma once
#include "../../include/step_hip.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define STEP_OK 0
#define STEP_ERR_ARG 1
#define STEP_ERR_HIP 2

void step_set_error(const char* fmt, ...);
int step_raise_lds_once(const void* kernel, int bytes, const char* what);      // errors.cpp: once per (kernel, device), thread-safe

#define STEP_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            step_set_error(__VA_ARGS__);        \
            return STEP_ERR_ARG;                \
        }                                       \
    } while (0)

#define STEP_LAUNCH_CHECK(name)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            step_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return STEP_ERR_HIP;                                                  \
        }                                                                         \
    } while (0)

#define STEP_TRY(expr)                       \
    do {                                     \
        int rc_ = (expr);                    \
        if (rc_ != STEP_OK) return rc_;      \
    } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 (round to nearest even; NaN not expected on these paths) -------------
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// 8 consecutive f32 -> one MFMA bf16 operand (4 VGPRs); lowers to 4 x v_cvt_pk_bf16_f32 (RNE)
typedef __attribute__((ext_vector_type(8))) float f32x8;
__device__ __forceinline__ bf16x8 pack8(const float* v) {
    f32x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = v[j];
    return __builtin_convertvector(t, bf16x8);
}

// ---- wave64 helpers --------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

#ifndef STEP_PHILOX_MUL64
#define STEP_PHILOX_MUL64 1     // 0: __umulhi + low product (two quarter-rate multiplies per round half; identical results)
#endif
// ---- counter-based RNG (Philox4x32-10) --------------------------------------------------
// Used for on-device Gumbel noise and dropout masks; deterministic in (seed, counter).
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#if STEP_PHILOX_MUL64
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;      // one v_mad_u64_u32 each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#endif
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniform in [0,1) with 24 bits, like torch.rand's float path
__device__ __forceinline__ float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// Epilogue of one 32x32 MFMA accumulator tile held by a lane: column pointer `col` (= C + remapped n), rows
// mb + (e&3) + 8*(e>>2).  The accumulate mode is uniform, so it is branched on once per tile: the plain-store
// and atomic paths issue their 16 memory operations back to back (no s_waitcnt in between), the read-modify-write
// path issues its 16 loads first.
// column-block affine of a result column (StepGemm.c_nscale / c_nshift / c_mvec): value <- cs * value + csh * mvec[row]
struct GemmColAffine { float cs, csh; const float* mvec; };
__device__ __forceinline__ GemmColAffine gemm_col_affine(const StepGemm& g, int gn) {
    GemmColAffine a = {1.f, 0.f, nullptr};
    if (g.c_nscale) { const int c = gn / g.c_nperiod; a.cs = g.c_nscale[c]; a.csh = g.c_nshift[c]; a.mvec = g.c_mvec; }
    return a;
}

__device__ __forceinline__ void gemm_store_tile(const f32x16& acc, float* col, int mb, int M, long ldc, float alpha, int accumulate,
                                                float bv, int relu, GemmColAffine ca = GemmColAffine{1.f, 0.f, nullptr}) {
    float val[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int gm = mb + (e & 3) + 8 * (e >> 2);
        val[e] = alpha * acc[e] * ca.cs;
        if (ca.mvec) val[e] += ca.csh * (gm < M ? ca.mvec[gm] : 0.f);
    }
    if (accumulate == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int gm = mb + (e & 3) + 8 * (e >> 2);
            if (gm < M) atomicAdd(col + (long)gm * ldc, val[e]);
        }
    } else if (accumulate == 1) {
        float old[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int gm = mb + (e & 3) + 8 * (e >> 2);
            old[e] = gm < M ? col[(long)gm * ldc] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int gm = mb + (e & 3) + 8 * (e >> 2);
            float v = val[e] + old[e] + bv;
            if (relu) v = fmaxf(v, 0.f);
            if (gm < M) col[(long)gm * ldc] = v;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int gm = mb + (e & 3) + 8 * (e >> 2);
            float v = val[e] + bv;
            if (relu) v = fmaxf(v, 0.f);
            if (gm < M) col[(long)gm * ldc] = v;
        }
    }
}
