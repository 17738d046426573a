// Internal C++ declarations shared by the translation units of libstep_hip.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/step_hip.h"

int step_gemm_launch(StepGemm g, hipStream_t st);
int step_gemm_bf16_launch(StepGemm g, hipStream_t st);
int step_gemm_f32_fast_launch(StepGemm g, hipStream_t st);      // -1: operands do not qualify

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
int dgl_conv2_dgrad_mfma(const float* dz, const float* w, float* din, int N, int T1, hipStream_t st);
long dgl_conv2_wgrad_scratch_floats(int N, int T1);
int dgl_conv2_wgrad_mfma(const float* dz, const float* a1, const float* sc, const float* sh, float* scratch, float* dw, float* db, int N,
                         int T1, hipStream_t st);
int dgl_conv1_wgrad_mfma(const float* dz, const float* x, float* scratch, float* dw, float* db, int N, int T, hipStream_t st);
int step_colsum_launch(const float* x, long rows, int cols, long ld, float* out, hipStream_t st);
