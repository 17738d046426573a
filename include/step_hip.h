/* libstep_hip -- C ABI of the MI355X-native STEP training-step hot path.
 *
 * The reference (GestaltCogTeam/STEP) has no FFI: its "operator API" is the python class wired
 * in by the config (CFG.MODEL.ARCH = STEP, step/STEP_PEMS04.py:41; instantiated at
 * basicts/runners/base_runner.py:49-51; called at step/step_runner/step_runner.py:66).  This
 * header is the C-ABI boundary a binding for that path would target: plain device pointers,
 * sizes and a hipStream_t (passed as void*).  The library never allocates, never retains a
 * pointer and launches only on the caller's stream (SURVEY.md section 8b "Ownership",
 * "Threading").  Every function returns 0 on success; on failure a thread-local message is
 * available from step_last_error() (python binding raises RuntimeError, mirroring the
 * reference's exception-only error convention).
 *
 * All tensors are dense row-major device memory.  "f32" = float, "bf16" = uint16_t bits.
 * Each entry point cites the reference code it replaces (paths relative to the reference).
 */
#ifndef STEP_HIP_H
#define STEP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* step_last_error(void);
int step_abi_version(void);

/* ---------------------------------------------------------------- generic contraction ---
 * C[b](m,n) (op)= alpha * sum_k A[b](m,k) B[b](k,n) (+bias[n]) (relu), element strides.
 * Replaces torch.einsum/matmul/bmm/1x1-Conv2d call sites on the path:
 * graphwavenet/model.py:13-15 (nconv), :23 (1x1 convs), discrete_graph_learning.py:134
 * (fc), similarity.py:12 (Gram), and all of their autograd backward contractions. */
typedef struct StepGemm {
    int M, N, K, batch;
    const void* A; long sam, sak, sab; int a_bf16;
    const void* B; long sbk, sbn, sbb; int b_bf16;
    float* C; long ldc, scn, scb;          /* C(m,n) at C + b*scb + m*ldc + n*scn (scn 0 -> 1) */
    float alpha;
    int accumulate;                        /* 0: C = ..., 1: C += ..., 2: atomicAdd(C, ...) */
    const float* bias; int relu;           /* epilogue, only with accumulate 0/1 */
    int splitk;                            /* >1 requires accumulate == 2 */
} StepGemm;
int step_gemm(const StepGemm* g, void* stream);

/* ---------------------------------------------------------------- TSFormer encoder -------
 * Forecasting-mode TSFormer forward, fused in one launch: patch embedding + positional
 * encoding + 4 post-norm encoder layers + encoder_norm.
 * Replaces TSFormer.forward(mode="forecasting") = tsformer/tsformer.py:71-105,179,190,
 * patch.py:20-42, positional_encoding.py:13-35, transformer_layers.py:13-20.
 *
 *  series      f32 [S, L]            one contiguous row per sequence s=(b,n) (see step_pack_long_history)
 *  wpack       packed weights (step_amd/tsformer_pack.py documents the layout; bf16 MFMA
 *              operand fragments + f32 vectors), built once per checkpoint
 *  hidden_bf16 bf16 [S, P, 96] or NULL
 *  hidden_f32  f32  [S, P, 96] or NULL   (parity tests)
 *  last_f32    f32  [S, 96]    or NULL   (state of the last patch = step.py:64)
 *  sqnorm_part f32  [S, 16]    or NULL   per-wave partial sums of hidden_bf16^2 (cosine norms)
 *  dropout_p   0 disables; otherwise inverted dropout at the reference's 1+4*depth sites
 */
int step_tsformer_encode(const float* series, int S, int L, const void* wpack, long wpack_bytes,
                         int depth, uint16_t* hidden_bf16, float* hidden_f32, float* last_f32,
                         float* sqnorm_part, float dropout_p, uint64_t seed, void* stream);

/* [B, L, N, C] f32 (the layout the reference DataLoader delivers, forecasting_dataset.py:62-71)
 * channel `ch` -> [B*N, L] f32.  Replaces the permute at tsformer.py:179 + `[..., [0]]` at
 * discrete_graph_learning.py:139. */
int step_pack_long_history(const float* x, int B, int L, int N, int C, int ch, float* out, void* stream);

/* ---------------------------------------------------------------- kNN prior graph --------
 * Replaces batch_cosine_similarity (similarity.py:6-16) + get_k_nn_neighbor
 * (discrete_graph_learning.py:91-111) + the diagonal clear (:165-166).
 *  hidden  bf16 [B, N, F];  sqnorm_part f32 [B*N, 16] from the encoder (or NULL: recomputed)
 *  sim     f32 [B, N, N] scratch/out (cosine similarities)
 *  adj     f32 [B, N, N] out in {0,1}
 *  work    scratch, at least step_knn_workspace_bytes(B, N, F) bytes
 */
long step_knn_workspace_bytes(int B, int N, int F);
int step_knn_graph(const uint16_t* hidden, const float* sqnorm_part, int B, int N, int F, int k_total,
                   float* sim, float* adj, void* work, long work_bytes, void* stream);
/* same selection on a caller-provided f32 similarity matrix (tests, N-scaling config) */
int step_topk_mask(const float* sim, int B, int N, int k_total, float* adj, void* work, long work_bytes,
                   void* stream);

/* ---------------------------------------------------------------- self test --------------
 * Verifies on the device the MFMA operand/accumulator lane maps this library is built on
 * (cdna_hip_programming.md section 3).  out: int32[8] failure counters, all zero when ok. */
int step_selftest_mfma(int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
