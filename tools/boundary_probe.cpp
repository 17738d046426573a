// What one dependent kernel boundary costs on this box (round 5): chains of N small kernels in ONE stream, eager launches, timed with
// events around the whole chain (host launches run ahead: the chain is enqueued before the first kernel ends where possible).
//   (a) empty kernels of 256 workgroups; (b) kernels that each rewrite 4 MB (dirty lines to write back at the boundary);
//   (c) the same while a long-running kernel occupies 160 compute units on another stream (the persistent encoder's situation).
// hipcc --offload-arch=gfx950 -O3 -o boundary_probe tools/boundary_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1000) *p = 1; }
__global__ void write_kernel(float4* p, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(704) void hog_kernel(long ticks) {          // 704 threads + 150 KB of LDS: one workgroup per compute unit
    extern __shared__ char smem[];
    smem[threadIdx.x] = 1;
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

static float chain(hipStream_t st, int n, int mode, float4* buf, long n4) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < n; ++i) {
        if (mode == 0) empty_kernel<<<256, 256, 0, st>>>(nullptr);
        else write_kernel<<<512, 256, 0, st>>>(buf, n4, (float)i);
    }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main() {
    hipStream_t a, b; hipStreamCreate(&a); hipStreamCreate(&b);
    float4* buf; const long n4 = (4L << 20) / 16; hipMalloc(&buf, n4 * 16);
    hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int N = 200;
    for (int hog = 0; hog < 2; ++hog)
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                if (hog) hog_kernel<<<160, 704, 150 * 1024, b>>>(100000 * 30);      // ~30 ms on 160 units
                float single = 0.f;
                if (rep == 0) { chain(a, 3, mode, buf, n4); }
                float ms = chain(a, N, mode, buf, n4);
                hipDeviceSynchronize();
                if (ms < best) best = ms;
            }
            float one = 1e9f;
            for (int rep = 0; rep < 5; ++rep) { float ms = chain(a, 1, mode, buf, n4); if (ms < one) one = ms; }
            printf("%s, %s: %d dependent launches %.1f us each (a single launch between two events: %.1f us)\n",
                   hog ? "next to a 160-unit persistent kernel" : "alone", mode ? "4 MB rewritten per kernel" : "empty kernels", N, best * 1e3f / N, one * 1e3f);
        }
    return 0;
}
