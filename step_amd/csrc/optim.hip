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
                                                        const float* __restrict__ partial, int npartial, const float* __restrict__ extra_sumsq,
                                                        float* __restrict__ out_norm, const StepDynState* __restrict__ dyn) {
    __shared__ float red[4];
    __shared__ float s_coef;
    if (dyn) {      // replayed steps (step_hip.h): learning rate and step count live on the device
        const float t = (float)dyn->adam_step;
        lr = dyn->lr;
        bc1 = 1.f - powf(beta1, t);
        bc2_sqrt = sqrtf(1.f - powf(beta2, t));
    }
    {   // total gradient norm from the partials (every block recomputes it: 1024 floats, L2-resident)
        float s = 0.f;
        for (int i = threadIdx.x; i < npartial; i += 256) s += partial[i];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            // max_norm < 0: the caller supplies the WHOLE squared norm in extra_sumsq (every rank of a sharded model passes the same
            // number, so every rank forms bit-identically the same clip factor and the replicated parameters stay identical), clip at -max_norm
            const bool given = max_norm < 0.f;
            const float mx = given ? -max_norm : max_norm;
            const float norm = sqrtf(given ? *extra_sumsq : red[0] + red[1] + red[2] + red[3] + (extra_sumsq ? *extra_sumsq : 0.f));
            float c = mx > 0.f ? mx / (norm + 1e-6f) : 1.f;      // clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
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
    return step_adam_clip_sharded(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, max_norm, nullptr, work,
                                  out_norm, stream);
}

// extra_sumsq (device scalar, may be NULL): sum of squares of gradient elements that live on OTHER ranks (parameter shards), added
// to this buffer's own sum before the clip coefficient is formed, so that every rank clips with the norm of the whole model
static int adam_clip_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, float max_norm, const float* extra_sumsq,
                            float* work, float* out_norm, const StepDynState* dyn, void* stream);
extern "C" int step_adam_clip_sharded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                                      float beta2, float eps, float weight_decay, int step, float max_norm, const float* extra_sumsq,
                                      float* work, float* out_norm, void* stream) {
    return adam_clip_launch(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, max_norm, extra_sumsq, work, out_norm,
                            nullptr, stream);
}
extern "C" int step_adam_clip_dyn(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float beta1, float beta2,
                                  float eps, float weight_decay, float max_norm, const float* extra_sumsq, float* work, float* out_norm,
                                  const StepDynState* dyn, void* stream) {
    STEP_REQUIRE(dyn, "adam_clip_dyn: null state");
    return adam_clip_launch(params, grads, exp_avg, exp_avg_sq, n, 0.f, beta1, beta2, eps, weight_decay, 1, max_norm, extra_sumsq, work, out_norm, dyn,
                            stream);
}
// hash step of the replay seed: splitmix64
__global__ void dyn_advance_kernel(StepDynState* s) {
    uint64_t z = s->seed_xor + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    s->seed_xor = z ^ (z >> 31);
    s->adam_step += 1;
}
extern "C" int step_dyn_advance(StepDynState* dyn, void* stream) {
    STEP_REQUIRE(dyn, "dyn_advance: null state");
    dyn_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(dyn);
    STEP_LAUNCH_CHECK("dyn_advance");
    return STEP_OK;
}
static int adam_clip_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, float max_norm, const float* extra_sumsq,
                            float* work, float* out_norm, const StepDynState* dyn, void* stream) {
    STEP_REQUIRE(params && grads && exp_avg && exp_avg_sq && work && n > 0 && step >= 1, "adam_clip: bad arguments");
    STEP_REQUIRE((((uintptr_t)grads) & 15) == 0, "adam_clip: gradient buffer must be 16-byte aligned");
    STEP_REQUIRE(max_norm >= 0.f || extra_sumsq, "adam_clip: max_norm < 0 (squared norm supplied by the caller) needs extra_sumsq");
    hipStream_t st = (hipStream_t)stream;
    sumsq_partial_kernel<<<NB, 256, 0, st>>>(grads, n, work);
    STEP_LAUNCH_CHECK("sumsq_partial");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    adam_clip_kernel<<<blocks, 256, 0, st>>>(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2),
                                             max_norm, work, NB, extra_sumsq, out_norm, dyn);
    STEP_LAUNCH_CHECK("adam_clip");
    return STEP_OK;
}

// ---------------------------------------------------------------------------------------------------------
// step_loss (reference step/step_loss/step_loss.py:5-16 + basicts/metrics/mae.py:5-28), forward value and both
// gradients in two launches:  loss = sum(|p - y| m) / sum(m) + coef * mean(BCE(theta, prior)),  m = |y - null| > 5e-5.
namespace {
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ pred, const float* __restrict__ real, long n1, long rs,
                                                          float scale, float shift,
                                                          const float* __restrict__ theta, const float* __restrict__ prior, long n2,
                                                          float null_val, double* __restrict__ acc /*[3]*/) {
    __shared__ double red[4][3];
    double s_abs = 0.0, s_cnt = 0.0, s_bce = 0.0;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n1; i += stride) {
        const float y = real[i * rs] * scale + shift;
        if (fabsf(y - null_val) > 5e-5f) { s_abs += fabsf((pred[i] * scale + shift) - y); s_cnt += 1.0; }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
        const float t = theta[i], y = prior[i];
        const float l1 = fmaxf(logf(t), -100.f), l0 = fmaxf(logf(1.f - t), -100.f);      // torch BCELoss clamps the logs at -100
        s_bce -= (double)(y * l1 + (1.f - y) * l0);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s_abs += __shfl_xor(s_abs, o, 64); s_cnt += __shfl_xor(s_cnt, o, 64); s_bce += __shfl_xor(s_bce, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s_abs; red[threadIdx.x >> 6][1] = s_cnt; red[threadIdx.x >> 6][2] = s_bce; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&acc[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ pred, const float* __restrict__ real, long n1, long rs,
                                                          float scale, float shift,
                                                          const float* __restrict__ theta, const float* __restrict__ prior, long n2,
                                                          float null_val, float coef, const double* __restrict__ acc,
                                                          float* __restrict__ loss, float* __restrict__ dpred, float* __restrict__ dtheta,
                                                          const StepDynState* __restrict__ dyn) {
    if (dyn) coef = dyn->gsl_coef;      // replayed steps (step_hip.h)
    const double cnt = acc[1];
    const float inv_cnt = cnt > 0.0 ? (float)(1.0 / cnt) : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *loss = (cnt > 0.0 ? (float)(acc[0] / cnt) : 0.f) + coef * (float)(acc[2] / (double)n2);
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n1; i += stride) {
        const float y = real[i * rs] * scale + shift, d = (pred[i] * scale + shift) - y;
        const float m = fabsf(y - null_val) > 5e-5f ? inv_cnt * scale : 0.f;      // d/d pred of the loss on pred * scale + shift
        dpred[i] = d > 0.f ? m : (d < 0.f ? -m : 0.f);
    }
    const float sc = coef / (float)n2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
        const float t = theta[i], y = prior[i];
        // d/dt of -(y log t + (1-y) log(1-t)), with torch's denominator floor
        dtheta[i] = sc * (t - y) / fmaxf(t * (1.f - t), 1e-12f);
    }
}
__global__ __launch_bounds__(256) void scale2_kernel(const float* __restrict__ a, long na, const float* __restrict__ b, long nb,
                                                     const float* __restrict__ g, float* __restrict__ oa, float* __restrict__ ob) {
    const float s = *g;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < na; i += stride) oa[i] = a[i] * s;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) ob[i] = b[i] * s;
}
}  // namespace

static int loss_launch(const float* pred, const float* real, long n_pred, long real_stride, float scale, float shift, const float* theta,
                       const float* prior, long n_adj, float null_val, float coef, double* work, float* loss, float* dpred, float* dtheta,
                       const StepDynState* dyn, void* stream);
extern "C" int step_loss_scaled_fwd_bwd(const float* pred, const float* real, long n_pred, long real_stride, float scale, float shift,
                                        const float* theta, const float* prior, long n_adj, float null_val, float coef,
                                        double* work /*3 doubles*/, float* loss, float* dpred, float* dtheta, void* stream) {
    return loss_launch(pred, real, n_pred, real_stride, scale, shift, theta, prior, n_adj, null_val, coef, work, loss, dpred, dtheta, nullptr, stream);
}
extern "C" int step_loss_scaled_fwd_bwd_dyn(const float* pred, const float* real, long n_pred, long real_stride, float scale, float shift,
                                            const float* theta, const float* prior, long n_adj, float null_val, double* work, float* loss,
                                            float* dpred, float* dtheta, const StepDynState* dyn, void* stream) {
    STEP_REQUIRE(dyn, "step_loss_dyn: null state");
    return loss_launch(pred, real, n_pred, real_stride, scale, shift, theta, prior, n_adj, null_val, 0.f, work, loss, dpred, dtheta, dyn, stream);
}
static int loss_launch(const float* pred, const float* real, long n_pred, long real_stride, float scale, float shift, const float* theta,
                       const float* prior, long n_adj, float null_val, float coef, double* work, float* loss, float* dpred, float* dtheta,
                       const StepDynState* dyn, void* stream) {
    STEP_REQUIRE(pred && real && theta && prior && work && loss && dpred && dtheta && n_pred > 0 && n_adj > 0 && real_stride >= 1,
                 "step_loss: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(work, 0, 3 * sizeof(double), st) != hipSuccess) { step_set_error("step_loss: memset failed"); return STEP_ERR_HIP; }
    long nmax = n_pred > n_adj ? n_pred : n_adj;
    int blocks = (int)((nmax + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    // three f64 atomics per block onto three addresses: at least 16 elements per thread (191 blocks at PEMS04, 1024 at N = 4096)
    int rblocks = (int)(nmax / 4096);
    rblocks = rblocks < 1 ? 1 : (rblocks > 1024 ? 1024 : rblocks);
    loss_reduce_kernel<<<rblocks, 256, 0, st>>>(pred, real, n_pred, real_stride, scale, shift, theta, prior, n_adj, null_val, work);
    STEP_LAUNCH_CHECK("loss_reduce");
    loss_finish_kernel<<<blocks, 256, 0, st>>>(pred, real, n_pred, real_stride, scale, shift, theta, prior, n_adj, null_val, coef, work, loss,
                                               dpred, dtheta, dyn);
    STEP_LAUNCH_CHECK("loss_finish");
    return STEP_OK;
}
extern "C" int step_loss_fwd_bwd(const float* pred, const float* real, long n_pred, const float* theta, const float* prior, long n_adj,
                                 float null_val, float coef, double* work /*3 doubles*/, float* loss, float* dpred, float* dtheta,
                                 void* stream) {
    return step_loss_scaled_fwd_bwd(pred, real, n_pred, 1, 1.f, 0.f, theta, prior, n_adj, null_val, coef, work, loss, dpred, dtheta, stream);
}
// The three training metrics the reference's runner evaluates every iteration (base_tsf_runner.py:252-254: masked_mae, masked_rmse,
// masked_mape of basicts/metrics -- ~30 element-wise torch launches) in ONE launch:
//   m  = |y - null| > 5e-5                       (mae.py:18-21, rmse.py:17-20: ~isclose(y, null, atol 5e-5, rtol 0))
//   MAE = sum |p - y| m / sum m,  RMSE = sqrt(sum (p - y)^2 m / sum m)
//   y0 = |y| < 1e-4 ? 0 : y,  m0 = |y0| > 5e-5,  MAPE = sum |(|p - y0|) / y0| m0 / sum m0          (mape.py:21-35, null value fixed at 0)
// (an all-masked batch gives 0, like the reference's nan -> 0 replacement).  work: 6 doubles, ZERO before the first call; the last block to
// finish turns the sums into out[0..2] and clears them again, so no memset is queued per call.
namespace {
__global__ __launch_bounds__(256) void masked_metrics_kernel(const float* __restrict__ pred, long ps, const float* __restrict__ real, long rs, long n,
                                                             float null_val, double* __restrict__ work, float* __restrict__ out) {
    __shared__ double red[4][5];
    __shared__ bool last;
    double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float p = pred[i * ps], y = real[i * rs];
        if (fabsf(y - null_val) > 5e-5f) { const float d = p - y; a[0] += fabsf(d); a[1] += (double)d * d; a[2] += 1.0; }
        const float y0 = fabsf(y) < 1e-4f ? 0.f : y;
        if (fabsf(y0) > 5e-5f) { a[3] += fabsf(fabsf(p - y0) / y0); a[4] += 1.0; }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o, 64);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = a[k];
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&work[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd((unsigned int*)(work + 5), 1u) == gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        volatile double* w = work;
        const double s_abs = w[0], s_sq = w[1], cnt = w[2], s_ape = w[3], cnt0 = w[4];
        out[0] = cnt > 0.0 ? (float)(s_abs / cnt) : 0.f;
        out[1] = cnt > 0.0 ? sqrtf((float)(s_sq / cnt)) : 0.f;
        out[2] = cnt0 > 0.0 ? (float)(s_ape / cnt0) : 0.f;
        for (int k = 0; k < 6; ++k) w[k] = 0.0;
    }
}
}  // namespace
extern "C" int step_masked_metrics(const float* pred, long pred_stride, const float* real, long real_stride, long n, float null_val,
                                   double* work /*6 doubles, zero before the first call*/, float* out /*[3]: MAE, RMSE, MAPE*/, void* stream) {
    STEP_REQUIRE(pred && real && work && out && n > 0 && pred_stride >= 1 && real_stride >= 1, "step_masked_metrics: bad arguments");
    int blocks = (int)(n / 4096);
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    masked_metrics_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(pred, pred_stride, real, real_stride, n, null_val, work, out);
    STEP_LAUNCH_CHECK("masked_metrics");
    return STEP_OK;
}
// out_a = a * *g, out_b = b * *g (g a device scalar): both gradients of step_loss times the incoming gradient of the loss, one launch
extern "C" int step_scale2(const float* a, long na, const float* b, long nb, const float* g, float* out_a, float* out_b, void* stream) {
    STEP_REQUIRE(a && b && g && out_a && out_b && na > 0 && nb > 0, "step_scale2: bad arguments");
    long nmax = na > nb ? na : nb;
    int blocks = (int)((nmax + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    scale2_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(a, na, b, nb, g, out_a, out_b);
    STEP_LAUNCH_CHECK("scale2");
    return STEP_OK;
}
