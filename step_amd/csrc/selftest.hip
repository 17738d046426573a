// Device self-test of the MFMA lane maps libstep_hip relies on (run once by the test-suite
// and by __graft_entry__.smoke()).  Small-integer operands: every product and sum is exact.
#include "common.h"
#include "step_internal.h"

namespace {

__device__ __forceinline__ float fa(int r, int k) { return (float)(((r * 7 + k * 3) % 11) - 5); }
__device__ __forceinline__ float fb(int k, int c) { return (float)(((k * 5 + c * 2) % 13) - 6); }

__global__ void selftest_kernel(int32_t* out) {
    const int l = threadIdx.x;
    int bad0 = 0, bad1 = 0;
    {   // v_mfma_f32_32x32x2_f32: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]
        f32x16 d;
        for (int e = 0; e < 16; ++e) d[e] = 0.f;
        d = __builtin_amdgcn_mfma_f32_32x32x2f32(fa(l & 31, l >> 5), fb(l >> 5, l & 31), d, 0, 0, 0);
        for (int e = 0; e < 16; ++e) {
            int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31;
            float want = fa(row, 0) * fb(0, col) + fa(row, 1) * fb(1, col);
            bad0 += (d[e] != want);
        }
    }
    {   // v_mfma_f32_32x32x16_bf16 in "slot space": A lane (r,h) slot j  x  B lane (c,h) slot j
        float av[8], bv[8];
        const int r = l & 31, h = l >> 5;
        for (int j = 0; j < 8; ++j) { av[j] = fa(r, h * 8 + j); bv[j] = fb(h * 8 + j, r); }
        f32x16 d;
        for (int e = 0; e < 16; ++e) d[e] = 0.f;
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(av), pack8(bv), d, 0, 0, 0);
        for (int e = 0; e < 16; ++e) {
            int row = (e & 3) + 8 * (e >> 2) + 4 * h, col = r;
            float want = 0.f;
            for (int k = 0; k < 16; ++k) want += fa(row, k) * fb(k, col);
            bad1 += (d[e] != want);
        }
    }
    atomicAdd(&out[0], bad0);
    atomicAdd(&out[1], bad1);
    if (l == 0) out[7] = 0x600DC0DE;
}

// keep-mask streams of the encoder's dropout generator: stream (block, thread) is seeded like Dropper::seed and emits
// `words` 32-bit draws = 4*words Bernoulli bytes.  gen 0: xorshift32 (6 VALU ops per word), gen 1: v_prng_b32 (1 op),
// gen 2: the 24-bit LCG of the TSF_DROPOUT_LCG build (draw bytes packed in draw order).
__device__ __forceinline__ uint32_t st_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__global__ void dropout_stream_kernel(uint32_t base, int gen, int words, uint32_t* __restrict__ out) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t st = st_mix32(base + id * 0x9E3779B1u) | 1u;
    for (int w = 0; w < words; ++w) {
        if (gen == 0) { st ^= st << 13; st ^= st >> 17; st ^= st << 5; }
        else if (gen == 1) st = __builtin_amdgcn_prng_b32(st);
        else {       // gen 2: 24-bit LCG, one v_mad_u32_u24 per step; two draws (bits 16..23, 8..15) per step, two steps per word
            uint32_t a, b;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a) : "v"(st), "s"(0x43FD45u), "v"(0xC39EC3u));
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(b) : "v"(a), "s"(0x43FD45u), "v"(0xC39EC3u));
            st = b;
            out[(long)id * words + w] = ((a >> 16) & 0xffu) | (((a >> 8) & 0xffu) << 8) | (((b >> 16) & 0xffu) << 16) | (((b >> 8) & 0xffu) << 24);
            continue;
        }
        out[(long)id * words + w] = st;
    }
}

}  // namespace

extern "C" int step_selftest_dropout_stream(uint32_t seed, int gen, int streams, int words, uint32_t* out, void* stream) {
    STEP_REQUIRE(out && streams > 0 && streams % 64 == 0 && words > 0 && (gen >= 0 && gen <= 2), "selftest_dropout_stream: bad arguments");
    dropout_stream_kernel<<<streams / 64, 64, 0, (hipStream_t)stream>>>(seed, gen, words, out);
    STEP_LAUNCH_CHECK("step_selftest_dropout_stream");
    return STEP_OK;
}

extern "C" int step_selftest_mfma(int32_t* out, void* stream) {
    STEP_REQUIRE(out != nullptr, "selftest: null output");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, 8 * sizeof(int32_t), st) != hipSuccess) {
        step_set_error("selftest: memset failed");
        return STEP_ERR_HIP;
    }
    selftest_kernel<<<1, 64, 0, st>>>(out);
    STEP_LAUNCH_CHECK("step_selftest_mfma");
    return STEP_OK;
}
