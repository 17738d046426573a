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

}  // namespace

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
