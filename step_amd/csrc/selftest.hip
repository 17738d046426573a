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

// ~us of wall time on one wave (s_memrealtime ticks at 100 MHz)
__global__ void spin_kernel(long ticks, int* sink) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && ticks < 0) *sink = 1;
}

}  // namespace

// Do two streams run CONCURRENTLY on this device?  The HIP runtime multiplexes a process's streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES, four by default, handed out by least use): two streams that share a queue serialise, silently.  With a process group
// in the process (RCCL and torch.distributed create streams of their own) the step's second stream has been seen to land on the main
// stream's queue -- every kernel of the graph learner then waits for the encoder, +0.5 ms per step (profiles/r05_h_*, r05_i_*).  The
// probe: a 200 us spin on `a`, an empty kernel + event on `b`; `b` finishing within 100 us of the spin's start means they overlap.
// Host-blocking (synchronises both streams): for set-up code only.
extern "C" int step_streams_concurrent(void* stream_a, void* stream_b, int* concurrent) {
    STEP_REQUIRE(concurrent != nullptr, "streams_concurrent: null output");
    hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipStreamSynchronize(a);
    if (e == hipSuccess) e = hipStreamSynchronize(b);
    float best = 1e30f;
    for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {          // (the first round also pays the kernels' load)
        e = hipEventRecord(e0, a);
        if (e != hipSuccess) break;
        spin_kernel<<<1, 64, 0, a>>>(20000, nullptr);
        spin_kernel<<<1, 64, 0, b>>>(0, nullptr);
        e = hipEventRecord(e1, b);
        if (e == hipSuccess) e = hipStreamSynchronize(a);
        if (e == hipSuccess) e = hipStreamSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess && rep == 1) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) {
        step_set_error("streams_concurrent: %s", hipGetErrorString(e));
        return STEP_ERR_HIP;
    }
    *concurrent = best < 0.1f ? 1 : 0;
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
