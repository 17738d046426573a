// Internal C++ declarations shared by the translation units of libstep_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "../../include/step_hip.h"

// Fused epilogues of the DiscreteGraphLearning fc backward, internal to the library (wide-store epilogue of the staged GEMM).
// The result C [M, N] has `channels` column blocks of `period` columns (N = channels * period).
//   dotw / dots: dots[2c] += sum_{m, n in block c} W[m][n] * (alpha A.B)[m][n],  dots[2c+1] += sum W[m][n] * c_mvec[m]
//                (W laid out like C).  With A.B = dgpre^T a2 and W = fc.weight these are the two sums the BatchNorm2 backward needs
//                (sum dy, sum dy*xhat over nodes and time) without ever materialising dy = dgpre fc.weight -- see dgl.hip.
//   bnx ...:     the stored value becomes the BatchNorm-backward of the result, masked by the ReLU in front of the BatchNorm:
//                C = x > 0 ? coef[2C+c] * (v - coef[c] - (x - stat[2C+c]) * stat[3C+c] * coef[C+c]) : 0,  x = bnx[m][n] (laid out like C)
//                coef = [m1 | m2 | gamma*rstd], stat = [scale | shift | mean | rstd], C = channels.
// flags: GEMM_FUSED_INTERLEAVED -- channels-last bf16 activations (bf16 mode): column n belongs to channel n % channels, C and bnx are bf16
//        (ldc in bf16 elements), B already carries the BatchNorm scale:  C = x > 0 ? v - kc (m1 + (x - mean) rstd m2) : 0,  kc = coef[2C+c].
//        GEMM_FUSED_FFN_FWD -- feed-forward hidden layer of the TSFormer pre-training step (pretrain.hip): C (bf16, ldc in bf16 elements)
//        = dropout(relu(v + ffn_bias[n])), the keep decisions from the Philox stream of step_pt_dropout at (seed, site) and element
//        m * N + n -- the ReLU output and its dropped copy are never written in f32.
//        GEMM_FUSED_MASKNZ -- its backward: C (bf16) = maskx[m][n] != 0 ? v / (1 - p) : 0 with maskx (bf16, laid out like C) the
//        stored forward value: non-zero exactly where the ReLU was open and the element was kept.
//        GEMM_FUSED_BF16OUT -- C (bf16, ldc in bf16 elements) = v + ffn_bias[n] (ffn_bias nullable): a linear layer whose output is stored as
//        the matrix-core operand type of its consumers (qkv and the attention-output gradient of the pre-training step).
enum { GEMM_FUSED_INTERLEAVED = 1, GEMM_FUSED_FFN_FWD = 2, GEMM_FUSED_MASKNZ = 4, GEMM_FUSED_BF16OUT = 8 };
struct GemmFused {
    const float* dotw; float* dots;
    const float* bnx; const float* bncoef; const float* bnstat;
    int channels, period;
    int flags;
    const float* ffn_bias; const void* maskx; float p; unsigned seed_lo, seed_hi, site;      // GEMM_FUSED_FFN_FWD / GEMM_FUSED_MASKNZ
};

// one 32-wide block of a segmented contraction (step_gemm_segmented_launch): element offsets from the A / B base pointers
struct GemmKSeg { long a_off, b_off, a_bs, b_bs; int a_rs, b_rs; };
constexpr int GEMM_KSEG_MAX = 128;
int step_gemm_segmented_launch(StepGemm g, const GemmKSeg* ktab, int ktab_per, hipStream_t st);

int step_gemm_launch(StepGemm g, hipStream_t st);
int step_gemm_auto_splitk(int M, int N, int K, int batch);      // splits chosen for splitk = -1 on the staged bf16 path
int step_gemm_launch_fused(StepGemm g, const GemmFused& fused, hipStream_t st);      // STEP_ERR_ARG (message set) when the operands do not qualify
int step_gemm_bf16_launch(StepGemm g, hipStream_t st, const GemmFused* fused = nullptr);
int step_gemm_f32_fast_launch(StepGemm g, hipStream_t st, const GemmFused* fused = nullptr);      // -1: operands do not qualify

// convenience builder for the common dense cases (f32 operands, batch 1)
static inline StepGemm gemm_desc(int M, int N, int K, const float* A, long sam, long sak, const float* B, long sbk,
                                 long sbn, float* C, long ldc) {
    StepGemm g;
    memset(&g, 0, sizeof(g));
    g.M = M; g.N = N; g.K = K; g.batch = 1;
    g.A = A; g.sam = sam; g.sak = sak;
    g.B = B; g.sbk = sbk; g.sbn = sbn;
    g.C = C; g.ldc = ldc; g.scn = 1;
    g.alpha = 1.f; g.splitk = 1;
    return g;
}

// StepGemm.a_rowsum on the general (unaligned-operand) kernels: a separate column-sum launch (needs A(m,k) = X[k*sak + m]); clears g->a_rowsum
int step_gemm_rowsum_separate(StepGemm* g, hipStream_t st);
// bf16 matrix-core conv2 stage of the DGL (dgl_conv_mfma.hip)
int dgl_conv2_fwd_mfma(const float* a1, const float* w, const float* b, const float* sc, const float* sh, float* a2, float* partial,
                       int N, int T1, int* nblk, hipStream_t st);
int dgl_conv2_dgrad_mfma(const float* dz, const float* w, float* din, int N, int T1, const float* bnx, const float* coef, const float* stat,
                         int own, hipStream_t st);
int dgl_conv2_wgrad_xhat_mfma(const float* dz, const float* a1, const float* stat1, float* scratch, float* graw, int N, int T1, hipStream_t st);
int dgl_conv2_wgrad_finish(const float* graw, const float* w, const float* stat1, const float* gamma1, const float* beta1, double count,
                           float* dw, float* db, float* dgamma1, float* dbeta1, float* coef1, int raw, hipStream_t st);
// channels-last bf16 storage of the conv activations (bf16 contraction mode), dgl_conv_mfma.hip
int dgl_conv1_fwd_cl(const float* x, const float* w, const float* b, void* a1h, float* partial, int N, int T, int stat_limit, int* nblk,
                     hipStream_t st);
long dgl_conv2_pack_floats();
int dgl_conv2_fwd_cl(const void* a1h, const float* w, const float* b, const float* sc, const float* sh, void* a2h, float* partial, int N,
                     int T1, int* nblk, float* pack, hipStream_t st);
int dgl_conv2_dgrad_cl(const void* dz2h, const float* w, const float* sc, void* dz1h, int N, int T1, const void* a1h, const float* coef,
                       const float* stat, int own, float* pack, hipStream_t st);
int dgl_conv2_wgrad_cl(const void* dz2h, const void* a1h, float* scratch, float* graw, int N, int T1, hipStream_t st);
int dgl_conv1_wgrad_cl(const void* dz1h, const float* x, float* scratch, float* dw, float* db, int N, int T, hipStream_t st);
long dgl_conv2_wgrad_scratch_floats(int N, int T1);
int dgl_conv2_wgrad_mfma(const float* dz, const float* a1, const float* sc, const float* sh, float* scratch, float* dw, float* db, int N,
                         int T1, hipStream_t st);
int dgl_conv1_wgrad_mfma(const float* dz, const float* x, float* scratch, float* dw, float* db, int N, int T, hipStream_t st);
int step_colsum_launch(const float* x, long rows, int cols, long ld, float* out, hipStream_t st);

// Fork / join of leaf work onto a second stream (events from a per-thread, per-device pool): the data-gradient chain of a backward
// stays on `main`, launches whose results nothing in the chain reads go to fork().
struct AuxLane {
    hipStream_t main, aux;
    bool on;
    int next = 0, base = 0;            // base: first event of this lane in the pool (two lanes of one call use disjoint events)
    AuxLane(hipStream_t m, hipStream_t a, int base_ = 0) : main(m), aux(a), on(a != nullptr && a != m), base(base_) {}
    static hipEvent_t event(int k) {
        thread_local std::vector<hipEvent_t> pool[16];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::vector<hipEvent_t>& v = pool[dev & 15];
        while ((int)v.size() <= k) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            v.push_back(e);
        }
        return v[k];
    }
    // the stream leaf work goes to, ordered after everything queued on the main stream so far
    hipStream_t fork() {
        if (!on) return main;
        hipEvent_t e = event(base + (next++ & 63));
        if (!e || hipEventRecord(e, main) != hipSuccess || hipStreamWaitEvent(aux, e, 0) != hipSuccess) { on = false; return main; }
        return aux;
    }
    // One record on the main stream for several waiters (an event record costs the main stream's dependent chain ~6 us of packet
    // processing between two kernels): mark() records, after(e) makes this lane's stream wait for it.
    hipEvent_t mark() {
        if (!on) return nullptr;
        hipEvent_t e = event(base + (next++ & 63));
        if (!e || hipEventRecord(e, main) != hipSuccess) { on = false; return nullptr; }
        return e;
    }
    hipStream_t after(hipEvent_t e) {
        if (!on || !e) return main;
        if (hipStreamWaitEvent(aux, e, 0) != hipSuccess) { on = false; return main; }
        return aux;
    }
    // done(): an event on the AUXILIARY stream after what is queued on it so far; wait_done(e): the main stream waits for that point
    // only -- later leaves on the same stream do not delay it (join() = wait for everything queued so far).
    hipEvent_t done() {
        if (!on) return nullptr;
        hipEvent_t e = event(base + (next++ & 63));
        if (!e || hipEventRecord(e, aux) != hipSuccess) return nullptr;
        return e;
    }
    int wait_done(hipEvent_t e) {
        if (!on) return STEP_OK;
        if (!e) return join();
        if (hipStreamWaitEvent(main, e, 0) != hipSuccess) { step_set_error("backward: stream join failed"); return STEP_ERR_HIP; }
        return STEP_OK;
    }
    // the main stream waits for everything queued on the auxiliary stream so far
    int join() {
        if (!on) return STEP_OK;
        hipEvent_t e = event(base + (next++ & 63));
        if (!e || hipEventRecord(e, aux) != hipSuccess || hipStreamWaitEvent(main, e, 0) != hipSuccess) {
            step_set_error("backward: stream join failed");
            return STEP_ERR_HIP;
        }
        return STEP_OK;
    }
};
