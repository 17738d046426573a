// Fused gradient-norm clip + Adam on flat buffers.
//
// Replaces, for the native module, what easytorch's Runner.backward does with torch calls for the reference
// (cfg: step/STEP_PEMS04.py:90-106): torch.nn.utils.clip_grad_norm_(params, max_norm=3.0) followed by
// torch.optim.Adam(lr, weight_decay, eps).step().  Same arithmetic as torch (L2 weight decay added to the
// gradient, bias-corrected moments, eps added after the sqrt), two launches instead of ~25, one pass over
// parameters / moments / gradients (16 B read + 12 B written per element: HBM-bound).
#include "common.h"
#include "step_internal.h"

namespace {

constexpr int NB = 1024;     // partial-sum blocks

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n / 4;
    const float4* g4 = (const float4*)g;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                                                        float wd, float bc1, float bc2_sqrt, float max_norm,
                                                        const float* __restrict__ partial, int npartial, float* __restrict__ out_norm) {
    __shared__ float red[4];
    __shared__ float s_coef;
    {   // total gradient norm from the partials (every block recomputes it: 1024 floats, L2-resident)
        float s = 0.f;
        for (int i = threadIdx.x; i < npartial; i += 256) s += partial[i];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
            float c = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;      // clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
            s_coef = c < 1.f ? c : 1.f;
            if (out_norm && blockIdx.x == 0) *out_norm = norm;
        }
        __syncthreads();
    }
    const float coef = s_coef;
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] * coef + wd * pi;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

}  // namespace

extern "C" long step_adam_work_floats(void) { return NB + 8; }

extern "C" int step_adam_clip(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float max_norm, float* work, float* out_norm,
                              void* stream) {
    STEP_REQUIRE(params && grads && exp_avg && exp_avg_sq && work && n > 0 && step >= 1, "adam_clip: bad arguments");
    STEP_REQUIRE((((uintptr_t)grads) & 15) == 0, "adam_clip: gradient buffer must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    sumsq_partial_kernel<<<NB, 256, 0, st>>>(grads, n, work);
    STEP_LAUNCH_CHECK("sumsq_partial");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    adam_clip_kernel<<<blocks, 256, 0, st>>>(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2),
                                             max_norm, work, NB, out_norm);
    STEP_LAUNCH_CHECK("adam_clip");
    return STEP_OK;
}
