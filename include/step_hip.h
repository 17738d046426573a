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
int step_abi_version(void);        /* 10: step_tsformer_encode(STEP_ENC_RANGE_FLAG); 9: step_comm_* / step_grad_allreduce* (data-parallel collectives on RCCL's C API); 8: step_pt_ffn_pack / step_pt_ffn_fused_{fwd,bwd_data,bwd_weights} (the pre-training feed-forward block without a stored hidden layer), step_pt_attention_fwd_bf16(pool, pool_words), step_pt_rows_linear, step_pt_proj_wgrad, step_pt_embed_unmasked_*; 7: StepDynState + step_dyn_advance and the *_dyn entry points (captured / replayed training steps); 6: unjoined leaves on aux_stream / leaf_stream (step_gwnet_backward, step_dgl_edges_backward), StepGemm.splitk_ws, bf16 I/O of step_pt_attention_*_bf16, step_pt_linear_bf16out, step_pt_layernorm_bwd_dropout(out_colsum), step_loss_scaled_fwd_bwd, step_scale2, step_dgl_edges_theta_offset; 5: step_tsformer_encode(fallback_count), keep-mask chunks at word granularity + 16-word wrap copy, step_gwnet_backward(aux_stream), step_pt_ffn_hidden_{fwd,bwd}, step_pt_colsum_bf16, step_pt_add_layernorm_fwd, step_pt_layernorm_bwd_dropout; 4: step_tsformer_encode(flags, drop_pool), step_dropout_pool_fill; 3: step_tsformer_encode(operand_f16);
                                      2: StepGemm.compute_bf16, Step{Dgl,Gwnet}Params.gemm_bf16 */

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
    /* optional index remaps idx -> (idx / blk) * stride + idx % blk (blk == 0: off); they let the
       contraction read / write one 32-channel slot of the [.., T, 224] concatenated gcn buffer
       (graphwavenet/model.py:45) in place */
    int a_kblk; long a_kstride;
    int b_kblk; long b_kstride;
    int b_nblk; long b_nstride;
    int c_nblk; long c_nstride;
    /* optional affine on A along k: A'(m,k) = A(m,k) * a_kscale[k / a_kperiod] + a_kshift[k / a_kperiod]
       (folds BatchNorm1d(16) into the DGL fc, discrete_graph_learning.py:132-134) */
    const float* a_kscale; const float* a_kshift; int a_kperiod;
    /* optional second batch level: batch index z = i1 * batch0 + i0 (batch0 == 0: off), operand offsets
       i0 * s?b + i1 * s?b1.  Lets one launch run the same contraction for the three diffusion supports
       (different adjacency stack entry and gcn slot, same sample index) */
    int batch0; long sab1, sbb1, scb1;
    /* 0: exact f32 matrix cores (v_mfma_f32_32x32x2_f32); 1: operands rounded to bf16 in LDS, v_mfma_f32_32x32x16_bf16,
       f32 accumulate and output (16x the matrix-pipe rate) */
    int compute_bf16;
    /* optional: a_rowsum[m] += sum_k A(m,k)  (batch 1, alpha 1).  The bias gradient that accompanies a weight-gradient
       GEMM; the staged kernels get it from the matrix cores as one extra all-ones column of B. */
    float* a_rowsum;
    /* optional column-block affine of the result: C(m,n) <- c_nscale[n / c_nperiod] * (alpha * A.B)(m,n)
       + c_nshift[n / c_nperiod] * c_mvec[m]  (then accumulate / bias as usual).  Folds a BatchNorm affine of the B operand's
       columns into a weight-gradient GEMM (the DGL fc weight gradient w.r.t. the normalised conv2 output). */
    const float *c_nscale, *c_nshift, *c_mvec;
    int c_nperiod;
    /* optional scratch for split-K (accumulate == 2, compute_bf16, staged path): with at least splitk * batch * M * N floats here the
       splits store their partial tiles (plain 16-byte stores) and a second launch adds their sum to C, instead of every split adding
       every element of its tile with an atomic -- 256 splits of a 96 x 384 weight gradient are 9.4 M atomics, which took longer than
       streaming the two operands (profiles/r03_w_split_target_ab_C3.log).  NULL / too small: atomics as before. */
    float* splitk_ws; long splitk_ws_floats;
} StepGemm;
int step_gemm(const StepGemm* g, void* stream);

/* ---------------------------------------------------------------- TSFormer encoder -------
 * Forecasting-mode TSFormer forward, fused in one launch: patch embedding + positional
 * encoding + 4 post-norm encoder layers + encoder_norm.
 * Replaces TSFormer.forward(mode="forecasting") = tsformer/tsformer.py:71-105,179,190,
 * patch.py:20-42, positional_encoding.py:13-35, transformer_layers.py:13-20.
 *
 *  series      f32 [S, L]            one contiguous row per sequence s=(b,n) (see step_pack_long_history)
 *  wpack       packed weights (step_amd/tsformer_pack.py documents the layout; 16-bit MFMA
 *              operand fragments + f32 vectors), built once per checkpoint
 *  flags       STEP_ENC_F16: the fragments of wpack are float16 (v_mfma_f32_32x32x16_f16; same rate as bfloat16, 3 more
 *              mantissa bits; P and V of the attention stay bfloat16 for their exponent range) -- must match how wpack was
 *              packed; 0: bfloat16 fragments.  STEP_ENC_ALWAYS_RESHIFT (tests): skip the fixed-shift softmax schedule and run
 *              the re-shifting loop, re-shifting on every new running maximum instead of only when its head room is used up.
 *  hidden_bf16 bf16 [S, P, 96] or NULL
 *  hidden_f32  f32  [S, P, 96] or NULL   (parity tests)
 *  last_f32    f32  [S, 96]    or NULL   (state of the last patch = step.py:64)
 *  sqnorm_part f32  [S, 16]    or NULL   per-wave partial sums of hidden_bf16^2 (cosine norms)
 *  dropout_p   0 disables; otherwise inverted dropout at the reference's 1+4*depth sites (positional_encoding.py:32 and
 *              torch.nn.TransformerEncoderLayer's attention-probability, dropout1, FFN and dropout2 sites), keep-masks taken
 *              from drop_pool: pool_words (a power of two, >= 2 * step_tsformer_dropout_words(L, depth)) 64-bit words of
 *              Bernoulli(1 - dropout_p) bits written by step_dropout_pool_fill(dropout_p) -- bit l of a word is lane l's
 *              keep flag for one accumulator register.  Sequence s, layer l reads the step_tsformer_dropout_words() words
 *              that start at word (mix32(seed32 + s*0x9E3779B1 + (l+1)*0x632BE5AB) mod pool_words) (wrapping; the buffer holds
 *              pool_words + 16 words, the last 16 a copy of the first 16, which step_dropout_pool_fill writes); the
 *              layout inside that window is documented in csrc/tsformer_device.h and mirrored by tests/enc_dropout_host.py.
 *              The pool is only read.  Refill it (new seed) before every training step.
 *  fallback_count  optional device counters, uint32 [64] (NULL: off): their SUM grows by the number of (32-token tile, head, layer,
 *              sequence) units whose softmax left the fixed-shift fast schedule and ran the re-shifting loop (the kernel's
 *              data-dependent slow path); of S * depth * 4 * ceil(P / 32) units per launch.
 */
#define STEP_ENC_F16 1
#define STEP_ENC_ALWAYS_RESHIFT 2
#define STEP_ENC_RANGE_FLAG 4       /* fallback_count holds 65 words; word [64] is OR-ed with 1 when a hidden state written by this launch is
                                       not finite -- what a float16 operand beyond 65 504 ends as (the reference computes in fp32,
                                       transformer_layers.py:13-20, and has no such limit).  The host mirror re-runs the launch on bfloat16
                                       fragments (step_arch/tsformer.py, "range guard") */
#define STEP_ENC_WORKGROUPS(n) (((n) & 0xffff) << 8)   /* persistent launch: at most n workgroups (= compute units: a workgroup fills one), each
                                                          looping over sequences; 0 = one workgroup per sequence */
int step_tsformer_encode(const float* series, int S, int L, const void* wpack, long wpack_bytes,
                         int depth, int flags, uint16_t* hidden_bf16, float* hidden_f32, float* last_f32,
                         float* sqnorm_part, float dropout_p, const uint64_t* drop_pool, long pool_words, uint64_t seed,
                         unsigned int* fallback_count, void* stream);
/* Keep-mask pool for the encoder's dropout: word w, bit l = (Philox4x32-10(counter = (w, w >> 32, l / 4, 0x5EEDD80F),
 * key = (seed, seed >> 32))[l % 4] >= dropout_p * 2^32).  `words` must be a power of two >= 16 and the buffer must hold
 * words + 16 words: the first 16 are repeated behind the pool.  Replaces the device generator draws of
 * torch.nn.functional.dropout at the sites listed above. */
int step_dropout_pool_fill(uint64_t* pool, long words, float dropout_p, uint64_t seed, void* stream);
long step_tsformer_dropout_words(int L, int depth);       /* mask words one (sequence, layer) reads; 0 on bad arguments */

/* [B, L, N, C] f32 (the layout the reference DataLoader delivers, forecasting_dataset.py:62-71)
 * channel `ch` -> [B*N, L] f32.  Replaces the permute at tsformer.py:179 + `[..., [0]]` at
 * discrete_graph_learning.py:139. */
int step_pack_long_history(const float* x, int B, int L, int N, int C, int ch, float* out, void* stream);
/* Device-resident dataset, index-only loader (replaces ForecastingDataset.__getitem__ + collate + H2D,
 * step/step_data/forecasting_dataset.py:52-71): data f32 [T, N, C] stays in HBM, t0 int64 [B] are the forecast origins.
 *  long_series f32 [B*N, L]   channel ch of rows t0-L .. t0-1, in the encoder's layout (zero-filled when t0 < L, :66-67), or NULL
 *  hist        f32 [B, H, N, C] rows t0-H .. t0-1, or NULL;  fut f32 [B, H, N, C] rows t0 .. t0+H-1, or NULL */
int step_gather_windows(const float* data, int T, int N, int C, int ch, const long* t0, int B, int L, int H,
                        float* long_series, float* hist, float* fut, void* stream);

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

/* ---------------------------------------------------------------- DiscreteGraphLearning ---
 * Device pointers named after the reference state_dict keys of `discrete_graph_learning.*`
 * (discrete_graph_learning.py:63-78).  The same struct carries gradients (running stats unused);
 * backward entry points ACCUMULATE (+=) into it, the caller zeroes.  fc_mean is never used by
 * the reference forward (:142 is commented out) and is therefore absent. */
typedef struct StepDglParams {
    float *conv1_w, *conv1_b;                  /* [8,1,10], [8] */
    float *conv2_w, *conv2_b;                  /* [16,8,10], [16] */
    float *fc_w, *fc_b;                        /* [100, 16*(T-18)], [100] */
    float *bn1_w, *bn1_b, *bn1_rm, *bn1_rv;    /* BatchNorm1d(8): weight, bias, running_mean, running_var */
    float *bn2_w, *bn2_b, *bn2_rm, *bn2_rv;    /* BatchNorm1d(16) */
    float *bn3_w, *bn3_b, *bn3_rm, *bn3_rv;    /* BatchNorm1d(100) */
    float *fc_out_w, *fc_out_b;                /* [100,200], [100] */
    float *fc_cat_w, *fc_cat_b;                /* [2,100], [2] */
    int gemm_bf16;                             /* 1: the fc forward/backward contractions run on the bf16 matrix cores */
} StepDglParams;

/* Global node feature g[N,100] (discrete_graph_learning.py:131-136) from the constant train
 * series series_nt[N,T] (node-major copy of node_feats).  training!=0: batch statistics +
 * running-stat update (momentum 0.1, unbiased variance) exactly like torch.nn.BatchNorm1d.
 * saved/work: caller-owned scratch of step_dgl_global_{saved,work}_floats floats. */
long step_dgl_global_saved_floats(int N, int T);
long step_dgl_global_work_floats(int N, int T, int backward);
int step_dgl_global_forward(const float* series_nt, int N, int T, const StepDglParams* p, int training,
                            float momentum, float* saved, float* work, float* g, void* stream);
int step_dgl_global_backward(const float* series_nt, int N, int T, const StepDglParams* p, const float* saved,
                             const float* dg, float* work, const StepDglParams* grads, void* stream);
/* The same backward in two calls: phase 1 ends with the finished fc weight gradient (87 MB at PEMS04, the bulk of the
   data-parallel all-reduce, which the caller starts while phase 2 -- conv / BatchNorm backward, same `work` -- runs);
   phase 0 = both.  phase | STEP_DGL_FRESH_FC_GRAD (also for step_dgl_global_backward_shard): grads->fc_w holds no previous value --
   the fc weight gradient (98 % of the gradient bytes) is STORED instead of accumulated, so the caller need not zero that buffer and
   the kernel does not read it. */
#define STEP_DGL_FRESH_FC_GRAD 16
int step_dgl_global_backward_phase(const float* series_nt, int N, int T, const StepDglParams* p, const float* saved,
                             const float* dg, float* work, const StepDglParams* grads, int phase, void* stream);

/* Time slice of the global feature for data-parallel ranks (the fc weight is 98 % of the gradient bytes
 * and the conv / fc work is identical on every rank).  Rank r owns conv2-output columns [a, b) of the T-18: it gets the series
 * columns [a, b+18) as series_slice [N, Ts = b-a+18] and the weight columns fc.weight.view(100,16,T-18)[:, :, a:b] as p->fc_w
 * [100, 16*(b-a)], and runs the same kernels on that smaller problem.  What couples the slices is reduced by the CALLER between
 * the phases (sum over the ranks): the BatchNorm1/2 sums `sums` (f64: [0,16) after phase 1, [16,48) after phase 2), the partial
 * fc product gpre (step_dgl_global_offset item 2, after phase 3); backward: the 32 "dots" (item 10, after phase 1) and the 1296
 * "graw" (item 11, after phase 3).  own1 = conv1-output columns of the slice this rank owns (b-a, or b-a+9 on the last rank: the
 * other 9 are halo shared with the next rank); count1 / count2 = N*(T-9) / N*(T-18) of the WHOLE series.  The results equal the
 * unsliced forward / backward up to summation order.  bf16 contraction mode only (the fused BatchNorm backward). */
typedef struct StepDglShard { int own1; double count1, count2; } StepDglShard;
int step_dgl_global_forward_shard(const float* series_slice, int N, int Ts, const StepDglParams* p, int training, float momentum,
                                  float* saved, float* work, double* sums, float* g, const StepDglShard* shard, int phase,
                                  void* stream);
int step_dgl_global_backward_shard(const float* series_slice, int N, int Ts, const StepDglParams* p, const float* saved,
                                   const float* dg, float* work, const StepDglParams* grads, const StepDglShard* shard, int phase,
                                   void* stream);
long step_dgl_global_offset(int N, int T, int item);

/* Edge logits + Gumbel-softmax hard sample (discrete_graph_learning.py:148-161, :11-45).
 *  g [N,100]; u f32 [B, N*N, 2] uniform noise as drawn by torch.rand (:12) or NULL for the
 *  on-device Philox stream keyed by seed.  Outputs theta[B,N,N] = softmax(logits)[...,0]
 *  (step.py:72) and the sampled adjacency [B,N,N] with the diagonal cleared. */
long step_dgl_edges_saved_floats(int B, int N);
long step_dgl_edges_work_floats(int N);
long step_dgl_edges_theta_offset(int N);      /* theta_out == saved + this offset (floats): theta is written once, in place (no copy) */
int step_dgl_edges_forward(const float* g, int N, int B, const StepDglParams* p, const float* u, uint64_t seed,
                           float temperature, float* saved, float* theta_out, float* adj_out, void* stream);
/* aux_stream (nullable): the fc_out weight / bias gradients are queued there, ordered after the kernels that produce their inputs and
 * NOT joined (same contract as step_gwnet_backward: order the first reader of `grads` and the release of `work` after aux_stream);
 * dg is ordered on `stream`. */
int step_dgl_edges_backward(const float* g, int N, int B, const StepDglParams* p, const float* saved,
                            const float* dtheta, const float* dadj, float temperature, float* work,
                            const StepDglParams* grads, float* dg, void* aux_stream, void* stream);

/* ---------------------------------------------------------------- GraphWaveNet backbone ----
 * Device pointers named after the reference state_dict keys of `backend.*`
 * (graphwavenet/model.py:57-117).  residual_convs.* are never executed when gcn_bool is set
 * (model.py:202-208) and therefore absent; gconv[7] / bn[7] exist in the struct but the last
 * layer's gcn output is dead code in the reference, so they are neither read nor given a
 * gradient.  The same struct carries gradients: backward ACCUMULATES (+=), the caller zeroes. */
typedef struct StepGwnetParams {
    float *nodevec1, *nodevec2;                       /* [N,10], [10,N] */
    float *start_w, *start_b;                         /* [32,2,1,1], [32] */
    float *filter_w[8], *filter_b[8];                 /* [32,32,1,2], [32] */
    float *gate_w[8], *gate_b[8];
    float *skip_w[8], *skip_b[8];                     /* [256,32,1,1], [256] */
    float *bn_w[8], *bn_b[8], *bn_rm[8], *bn_rv[8];   /* BatchNorm2d(32) */
    float *gconv_w[8], *gconv_b[8];                   /* [32,224,1,1], [32] */
    float *fc_his0_w, *fc_his0_b, *fc_his2_w, *fc_his2_b;   /* [512,96], [256,512] */
    float *end1_w, *end1_b, *end2_w, *end2_b;         /* [512,256,1,1], [12,512,1,1] */
    int gemm_bf16;                                    /* 1: diffusion hops and their adjoints on the bf16 matrix cores */
} StepGwnetParams;

/* GraphWaveNet.forward (model.py:132-224) fused with the transposes of step.py:65,72:
 *  hist [B,12,N,Cin] f32 (first two channels used), hidden_last [B*N,96] f32, adj [B,N,N] f32
 *  -> pred [B,12,N] f32.  training bit 0: batch-stat BatchNorm (+running-stat update); bit 1 (value 2, with bit 0): also evaluate the
 *  last layer's gcn, whose output the reference computes and drops (model.py:202-213), for bn.7's running statistics; and, if
 *  dropout_p>0, inverted dropout after each gcn (model.py:47) from the Philox stream `seed`.
 *  saved/work: caller-owned scratch of step_gwnet_{saved,work}_floats floats; `saved` must be
 *  kept unchanged until step_gwnet_backward. */
long step_gwnet_saved_floats(int B, int N, int dropout);
long step_gwnet_work_floats(int B, int N, int backward);
long step_gwnet_saved_offset(int B, int N, int dropout, int item, int layer);
int step_gwnet_forward(const float* hist, int B, int N, int Cin, const float* hidden_last, const float* adj,
                       const StepGwnetParams* p, int training, float dropout_p, uint64_t seed, float momentum,
                       float* saved, float* work, float* pred, void* stream);
/* The same in two calls, so that the caller can run the part that does not depend on the TSFormer on another stream while
 * the encoder is busy: phase 1 = supports + the 8 WaveNet layers (hidden_last / pred unused), phase 2 = head (hist / adj
 * unused; same saved / work buffers, after phase 1 in stream order), phase 0 = both.  The head in two: phase 3 = the fc_his branch
 * (needs hidden_last only: it can be queued behind the encoder before phase 1 has finished elsewhere), phase 4 = the rest of the
 * head (after phases 1 and 3; needs pred).  Phase 1 in two (ABI 10): phase 5 = its part that does not read `adj` (start convolution,
 * adaptive support, weight packing, layer 0's gated TCN: model.py:143-155,165,183-189; adj may be NULL) -- it can run next to the graph
 * learner that produces adj -- and phase 6 = the rest of phase 1 (after phase 5 in stream / event order; hist unused). */
int step_gwnet_forward_phase(const float* hist, int B, int N, int Cin, const float* hidden_last, const float* adj,
                             const StepGwnetParams* p, int training, float dropout_p, uint64_t seed, float momentum,
                             float* saved, float* work, float* pred, int phase, void* stream);
/* dpred [B,12,N] -> parameter gradients (+=) and dadj [B,N,N] (gradient w.r.t. the sampled adjacency,
 * through both random-walk normalisations, model.py:121-130,160).
 * aux_stream, leaf_stream (each may be NULL or equal to stream: that work stays on `stream`; leaf_stream NULL: aux_stream takes its
 * work too): two more streams of the same device.  aux_stream runs the adjacency-gradient contractions of the layer ranges next to
 * the data-gradient chain and is joined inside the call (dadj needs them).  leaf_stream runs the LEAVES of the backward -- the
 * weight / bias gradients of every layer, the fc_his branch, unpacking the gate / skip gradients, the adaptive adjacency's backward
 * with the two node-embedding gradients, the start convolution's gradients -- forked after the kernels that produce their inputs
 * (event on `stream`, wait on leaf_stream) and NOT joined: on return dadj is ordered on `stream`, the leaves may still be running, and
 * the caller orders the first reader of `grads` -- and the release of `work` -- after leaf_stream and aux_stream (one stream-wait
 * before the gradient all-reduce / the optimizer; step_arch/step.py does it after the graph learner's backward, by when the leaves
 * have long finished). */
int step_gwnet_backward(const float* hist, int B, int N, int Cin, const float* hidden_last, const StepGwnetParams* p,
                        const float* saved, float* work, const float* dpred, const StepGwnetParams* grads,
                        float* dadj, int dropout, void* aux_stream, void* leaf_stream, void* stream);

/* ---------------------------------------------------------------- TSFormer pre-training -----
 * Building blocks (exact f32) of the masked-autoencoder stage, forward and backward; the contractions
 * go through step_gemm.  Replaces TSFormer.forward(mode="pre-train") = tsformer/tsformer.py:71-160,180-188
 * (+ mask.py index lists, positional_encoding.py:28-32 with index, transformer_layers.py:13-20 /
 * torch.nn.TransformerEncoderLayer) and its autograd.  All activations are [rows, 96] f32 with
 * rows = sequence-major tokens.  Dropout masks are a pure function of (seed, site, element index), so a
 * backward call with the same arguments replays the forward mask. */
int step_pt_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint32_t site, void* stream);
/* bf16 mode: the feed-forward hidden layer with ReLU and dropout in the GEMM epilogue, stored as bf16 (nothing [R,384] is ever written
 * in f32):  hidden[r][j] = dropout(relu(x[r,:] . w1[j,:] + b1[j])), the keep decisions of step_pt_dropout(seed, site) on [R,384];
 * and its backward through dropout and ReLU:  dhidden[r][j] = hidden[r][j] != 0 ? (dy[r,:] . w2[:,j]) / (1 - p) : 0.
 * Replace linear1 + ReLU + dropout of torch.nn.TransformerEncoderLayer (transformer_layers.py:10) and their autograd. */
int step_pt_ffn_hidden_fwd(const float* x, const float* w1, const float* b1, long R, float p, uint64_t seed, uint32_t site,
                           uint16_t* hidden, void* stream);
int step_pt_ffn_hidden_bwd(const float* dy, const float* w2, const uint16_t* hidden, long R, float p, uint16_t* dhidden, void* stream);
int step_pt_colsum_bf16(const uint16_t* x, long rows, int cols, float* out, void* stream);      /* out[c] += sum_r x[r][c] */
/* Fused feed-forward block of a pre-training layer (bf16 operands, f32 accumulate; csrc/pretrain_fused.hip): replaces, for
 * nn.TransformerEncoderLayer's linear1 -> ReLU -> dropout -> linear2 (reference transformer_layers.py:7-21 under tsformer.py:71-160),
 * step_pt_ffn_hidden_fwd + the linear2 step_gemm in the forward and two weight-gradient GEMMs, step_pt_ffn_hidden_bwd,
 * step_pt_colsum_bf16 and the input-gradient GEMM in the backward.  The [R, 384] hidden layer is never stored: the backward recomputes it.
 *   step_pt_ffn_pack          w1 [384, 96], b1 [384], w2 [96, 384], b2 [96] (f32) -> operand fragments (step_pt_ffn_pack_bytes() bytes, 16-byte
 *                             aligned); call it whenever the weights changed (once per training step)
 *   step_pt_ffn_fused_fwd     f2 [R, 96] = w2 . dropout(relu(w1 . h1 + b1)) + b2
 *   step_pt_ffn_fused_bwd_data     dh1 [R, 96] += ((df2 . w2) * relu' * keep / (1 - p)) . w1
 *   step_pt_ffn_fused_bwd_weights  dw1 [384, 96] += dhid^T h1, db1 [384] += column sums of dhid, dw2 [96, 384] += df2^T hid;
 *                             ws: step_pt_ffn_wgrad_ws_floats(R) floats of scratch
 * Dropout (p > 0): keep decisions are bits of the per-step pool written by step_dropout_pool_fill (pool_words a power of two >= 512, followed by
 * the 16-word wrap copy); rows 32 k .. 32 k + 31 of call site `site` own 192 consecutive words at a hashed offset, the same in all three calls. */
/* The four thin products around the attention of a pre-training layer as row kernels with LDS-resident weights (csrc/pretrain_fused.hip):
 * y[R, 96 nog] (accumulate ? += : =) x[R, 96 nkc] . M^T + bias, nkc * nog <= 3, M(out, in) = w[out * swo + in * swi] packed once per step
 * by step_pt_rows_linear_pack (bias nullable; step_pt_rows_linear_pack_bytes(nkc, nog) bytes, 16-byte aligned).  Forms:
 *   (nkc 1, nog 3 | 1, x f32, y bf16)  qkv = x . Wi^T + bi,  da = do . Wo          (what step_pt_linear_bf16out computes)
 *   (1, 1, x bf16, y f32)              o = a . Wo^T + bo
 *   (3, 1, x bf16, y f32, accumulate)  dx += dqkv . Wi */
long step_pt_rows_linear_pack_bytes(int nkc, int nog);
int step_pt_rows_linear_pack(const float* w, long swo, long swi, int nkc, int nog, const float* bias, void* pack, void* stream);
int step_pt_rows_linear(const void* x, int x_bf16, long R, const void* pack, int nkc, int nog, void* y, int y_bf16, int accumulate, void* stream);
/* every fragment buffer of one layer in one launch: what step_pt_ffn_pack and the four step_pt_rows_linear_pack calls of a layer write
 * (qkv: (1, 3) of Wi with bi; o: (1, 1) of Wo with bo; da: (1, 1) of Wo transposed; dx: (3, 1) of Wi transposed) */
int step_pt_layer_pack(const float* wi, const float* bi, const float* wo, const float* bo, const float* w1, const float* b1, const float* w2,
                       const float* b2, void* ffn, void* qkv, void* o, void* da, void* dx, void* stream);
/* the weight gradients of a layer's two projections and the qkv bias gradient in one pass over the four row tensors:
 * dwi [288, 96] += dqkv^T x, dbi [288] += column sums of dqkv, dwo [96, 96] += dov^T a   (x, dov f32 [R, 96]; dqkv bf16 [R, 288]; a bf16 [R, 96];
 * ws: step_pt_proj_wgrad_ws_floats(R) floats of scratch) */
long step_pt_proj_wgrad_ws_floats(long R);
int step_pt_proj_wgrad(const float* x, const uint16_t* dqkv, const float* dov, const uint16_t* a, long R, float* ws, float* dwi, float* dbi, float* dwo,
                       void* stream);
/* The forward row kernels with step_pt_add_layernorm_fwd as their output stage (same Philox stream, same results to f32 summation order): the
 * branch's [R, 96] tile never reaches HBM.
 *   step_pt_ffn_fused_fwd_ln   pre (nullable) = h1 + dropout(feed-forward(h1)) at site_out, y = LayerNorm(pre), stats [R, 2] = (mean, rstd)
 *   step_pt_rows_linear_ln     pre = res + dropout(x . M^T + bias), x bf16 [R, 96], pack of form (1, 1): the attention's out-projection */
int step_pt_ffn_fused_fwd_ln(const float* h1, long R, const void* pack, float p, const uint64_t* pool, long pool_words, uint64_t seed, uint32_t site_hidden,
                             uint32_t site_out, const float* gamma, const float* beta, float* pre, float* y, float* stats, void* stream);
int step_pt_rows_linear_ln(const uint16_t* x, long R, const void* pack, const float* res, float p, uint64_t seed, uint32_t site, const float* gamma,
                           const float* beta, float* pre, float* y, float* stats, void* stream);
/* Encoder input of the pre-training step for the unmasked tokens only (tsformer.py:88-104; the masked tokens' embeddings are dead):
 * x [S, Pu, 96] = sqrt(96) * dropout(w . patch(s, um[t]) + b + pos[um[t]]), w [96, 12] the patch embedding, um [Pu] int32 on the device;
 * the backward turns d x into the three parameter gradients (accumulated): dpos [*, 96] rows um[t], dw [96, 12], db [96]. */
int step_pt_embed_unmasked_fwd(const float* series, const int* um, const float* w, const float* b, const float* pos, long S, int L, int Pu, float p,
                               uint64_t seed, uint32_t site, float* x, void* stream);
int step_pt_embed_unmasked_bwd(const float* dx, const float* series, const int* um, long S, int L, int Pu, float p, uint64_t seed, uint32_t site,
                               float* dpos, float* dw, float* db, void* stream);
/* step_pt_dec_input_bwd + step_pt_sum_over_seq(midx) + step_colsum in one pass over dout [S, P, 96] (no [S, P - Pu, 96] scratch):
 * dz [S, Pu, 96] = sqrt(96) * dout rows t < Pu; dpos [*, 96] rows midx[j] += and dmask [96] += the dropped, scaled rows Pu + j summed over s. */
int step_pt_dec_input_bwd_sums(const float* dout, long S, int P, int Pu, float p, uint64_t seed, uint32_t site, const int* midx, float* dz, float* dpos,
                               float* dmask, void* stream);
long step_pt_ffn_pack_bytes(void);
long step_pt_ffn_wgrad_workgroups(long R);
long step_pt_ffn_wgrad_ws_floats(long R);
int step_pt_ffn_pack(const float* w1, const float* b1, const float* w2, const float* b2, void* pack, void* stream);
int step_pt_ffn_fused_fwd(const float* h1, long R, const void* pack, float p, const uint64_t* pool, long pool_words, uint64_t seed, uint32_t site,
                          float* f2, void* stream);
int step_pt_ffn_fused_bwd_data(const float* df2, const float* h1, long R, const void* pack, float p, const uint64_t* pool, long pool_words,
                               uint64_t seed, uint32_t site, float* dh1, void* stream);
int step_pt_ffn_fused_bwd_weights(const float* df2, const float* h1, long R, const void* pack, const float* b1, float p, const uint64_t* pool,
                                  long pool_words, uint64_t seed, uint32_t site, float* ws, float* dw1, float* db1, float* dw2, void* stream);
/* d = dropout(d) * [relu_of > 0] in one pass: the backward of relu -> dropout (same mask stream as step_pt_dropout at `site`) */
int step_pt_dropout_relu_mask(float* d, const float* relu_of, long n, float p, uint64_t seed, uint32_t site, void* stream);
int step_pt_add_dropout(const float* a, const float* b, float* out, long n, float p, uint64_t seed, uint32_t site, void* stream);
/* x[s][p][:] += vec[idx ? idx[p] : p][:]   (positional embedding, positional_encoding.py:28-31) */
int step_pt_add_rows(float* x, long S, int P, const float* vec, const int* idx, void* stream);
/* dvec[idx ? idx[j] : j][:] += sum_s dx[s][p_off + j][:], j < p_cnt, dx row pitch ldp tokens per sequence */
int step_pt_sum_over_seq(const float* dx, long S, int ldp, int p_off, int p_cnt, const int* idx, float* dvec, void* stream);
/* dst[s][t] = scale * src[s][idx[t]]  and its adjoint (tsformer.py:94-96) */
int step_pt_token_gather(const float* src, long S, int P, const int* idx, int T, float scale, float* dst, void* stream);
int step_pt_token_scatter(const float* ddst, long S, int P, const int* idx, int T, float scale, float* dsrc, void* stream);
/* decoder input = sqrt(96) * [ z | dropout(mask_token + pos[midx]) ]  (tsformer.py:120-127) and its adjoint */
int step_pt_dec_input(const float* z, const float* mask_token, const float* pos, const int* midx, long S, int P, int Pu,
                      float p, uint64_t seed, uint32_t site, float* out, void* stream);
int step_pt_dec_input_bwd(const float* dout, long S, int P, int Pu, float p, uint64_t seed, uint32_t site, float* dz,
                          float* dm, void* stream);
/* LayerNorm(96, eps 1e-5) rows; stats [R][2] = mean, rstd; backward accumulates dgamma/dbeta (+=) */
int step_pt_layernorm_fwd(const float* x, long R, const float* g, const float* b, float* y, float* stats, void* stream);
int step_pt_layernorm_bwd(const float* dy, const float* x, long R, const float* g, const float* stats, float* dx, float* dgamma,
                          float* dbeta, void* stream);
/* The same two with their neighbours fused and 16-byte accesses (one row per 32 lanes): forward  pre = a + dropout(b) (the residual
 * add of transformer_layers.py:10's encoder layer; b may be NULL, then pre is not written), y = LayerNorm(pre), stats = (mean, rstd);
 * backward  dx as step_pt_layernorm_bwd and, when dx_dropped is given, dropout(dx) with the stream of step_pt_dropout(seed, site);
 * out_colsum (nullable, [96]) += column sums of dx_dropped (of dx without it): the bias gradient of the linear layer that gradient
 * enters next. */
int step_pt_add_layernorm_fwd(const float* a, const float* b, long R, float p, uint64_t seed, uint32_t site, const float* g,
                              const float* beta, float* pre, float* y, float* stats, void* stream);
int step_pt_layernorm_bwd_dropout(const float* dy, const float* x, long R, const float* g, const float* stats, float* dx,
                                  float* dx_dropped, float p, uint64_t seed, uint32_t site, float* dgamma, float* dbeta,
                                  float* out_colsum, void* stream);
/* 4-head self-attention on qkv [S][T][288] -> out [S][T][96]; stats [S][4][T][2] = row max, row sum (for the backward) */
int step_pt_attention_fwd(const float* qkv, long S, int T, float p, uint64_t seed, uint32_t site, float* out, float* stats,
                          void* stream);
int step_pt_attention_bwd(const float* qkv, const float* out, const float* dout, const float* stats, long S, int T, float p,
                          uint64_t seed, uint32_t site, float* dqkv, void* stream);
/* The same attention on the matrix cores with the activations stored as bfloat16 (f32 accumulation and statistics, same statistics
 * layout; the probability dropout draws the 16-bit-field stream of csrc/pretrain.hip `attn_keep_index`, which the f32 kernels above
 * evaluate too): qkv bf16 [S][T][288], out / dout bf16 [S][T][96], dqkv bf16 [S][T][288], all 16-byte aligned; what the pre-training
 * module uses when matmul_precision == "bf16" (the producing / consuming GEMMs write / read bf16: step_pt_linear_bf16out, step_gemm
 * with a_bf16 / b_bf16).  T <= 352 (the backward holds seven operand copies and the keep bits in LDS: 139 KB at 352 tokens).
 * keepbits (nullable): [S][4][T][ceil(T/32)] words -- the forward stores the keep decisions of every (query, key tile), a backward
 * given the same buffer reads them instead of regenerating the Philox stream.
 * pool (nullable; needs keepbits): the forward takes the keep word of every (query, key tile) from the step's Bernoulli pool
 * (step_dropout_pool_fill; pool_words a power of two >= 4096; a (sequence, head) reads T * ceil(T/32) consecutive 32-bit words at a
 * hashed offset) instead of running Philox -- 0.73 -> 0.4 ms at config C3's decoder layer.
 * Round 6: with the pool (or p == 0) both entry points run the second-version kernels of csrc/pretrain_attn2.hip -- same tensors, same
 * statistics, same keep words, rebuilt for occupancy (<= 128 / 88 registers: two / three workgroups per compute unit): forward 620 -> 434 us,
 * backward 1497 -> 883 us at 168 tokens; the kernels of csrc/pretrain.hip remain the path of Philox-drawn keep decisions
 * (STEP_PT_ATTN_V1=1 in the environment forces them). */
int step_pt_attention_fwd_bf16(const uint16_t* qkv, long S, int T, float p, uint64_t seed, uint32_t site, uint16_t* out, float* stats,
                               uint32_t* keepbits, const uint64_t* pool, long pool_words, void* stream);
int step_pt_attention_bwd_bf16(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* stats, long S, int T, float p,
                               uint64_t seed, uint32_t site, uint16_t* dqkv, const uint32_t* keepbits, void* stream);
/* out[r][n] = bf16(sum_k x[r][k] w(k, n) + bias[n]), x f32 [R][K] rows (16-byte aligned, K % 4 == 0), w(k, n) at w + k * swk + n * swn
 * with swk == 1 (a torch Linear weight [N][K]) or swn == 1 (its transpose: the data gradient of that layer), bias nullable, N % 4 == 0,
 * out bf16 [R][N]: a linear layer whose result is only ever read as a matrix-core operand, stored in that type (bf16 contraction). */
int step_pt_linear_bf16out(const float* x, const float* w, long swk, long swn, const float* bias, long R, int N, int K, uint16_t* out,
                           void* stream);
int step_pt_relu_mask(float* d, const float* y, long n, void* stream);      /* d *= (y > 0) */
int step_colsum(const float* x, long rows, int cols, long ld, float* out, void* stream);   /* out[c] += sum_r x[r*ld + c] */

/* ---------------------------------------------------------------- optimizer side ---------
 * clip_grad_norm_(max_norm) + Adam (L2 weight decay, bias correction, eps after sqrt) on flat f32 buffers
 * of n elements; replaces easytorch's torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step for the
 * native module (cfg step/STEP_PEMS04.py:90-106).  step >= 1 is the Adam step count after this update.
 * work: step_adam_work_floats() floats of scratch; out_norm (nullable) receives the pre-clip gradient norm. */
long step_adam_work_floats(void);
int step_adam_clip(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float max_norm, float* work, float* out_norm,
                   void* stream);
/* The same with extra_sumsq (device scalar, NULL = 0): the sum of squares of gradient elements held by OTHER ranks (the fc weight
 * slices of a sharded graph learner), so that every rank clips with the norm of the whole model.  max_norm < 0: *extra_sumsq IS the
 * whole squared norm (this buffer's own sum is not added) and the clip threshold is -max_norm -- data-parallel ranks that hold different
 * shards pass the same number (sum over the replicated part + all-reduced sum over the shards) and so form bit-identically the same clip
 * factor: own + others, added in a different order on every rank, differs in the last bit and lets the replicated parameters drift. */
int step_adam_clip_sharded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, float max_norm, const float* extra_sumsq,
                           float* work, float* out_norm, void* stream);
/* step_loss = masked_mae(pred, real, null_val) + coef * BCELoss(theta, prior)  (step/step_loss/step_loss.py:5-16,
 * basicts/metrics/mae.py:5-28; pred/real are the RESCALED tensors the runner passes, base_tsf_runner.py:240-250).
 * Writes the scalar loss and d loss/d pred, d loss/d theta.  work: 3 doubles of scratch. */
int step_loss_fwd_bwd(const float* pred, const float* real, long n_pred, const float* theta, const float* prior, long n_adj,
                      float null_val, float coef, double* work, float* loss, float* dpred, float* dtheta, void* stream);
/* The same on NORMALISED tensors: the loss is taken on x * scale + shift (the runner's inverse scaling, base_tsf_runner.py:240-250,
 * step_runner.py:86-92, folded into the two kernels); dpred is the gradient w.r.t. the normalised prediction.  real is read with an
 * element stride (real_stride = C reads feature 0 of a contiguous [..., C] batch tensor in place). */
int step_loss_scaled_fwd_bwd(const float* pred, const float* real, long n_pred, long real_stride, float scale, float shift,
                             const float* theta, const float* prior, long n_adj, float null_val, float coef, double* work, float* loss,
                             float* dpred, float* dtheta, void* stream);
/* The runner's three training metrics in one launch (ABI 10): out = [masked MAE, masked RMSE, masked MAPE] of `pred` against `real`
 * with the semantics of basicts/metrics/{mae,rmse,mape}.py (mask = |real - null_val| > 5e-5; MAPE zeroes |real| < 1e-4 and masks 0),
 * evaluated every iteration by base_tsf_runner.py:252-254 (~30 element-wise torch launches).  Element i of pred / real sits at
 * i * stride floats.  work: 6 doubles, zero before the FIRST call; the kernel leaves them zero again. */
int step_masked_metrics(const float* pred, long pred_stride, const float* real, long real_stride, long n, float null_val, double* work,
                        float* out, void* stream);
/* out_a = a * *g, out_b = b * *g with g a device scalar (autograd's incoming gradient of the loss applied to both gradients). */
int step_scale2(const float* a, long na, const float* b, long nb, const float* g, float* out_a, float* out_b, void* stream);

/* ---------------------------------------------------------------- replayable steps --------
 * A training step that is captured once into a hipGraph and replayed (step_amd.GraphedTrainStep) cannot take per-step scalars as
 * launch arguments: they would be frozen into the graph.  What changes from step to step lives in ONE small device struct that
 * the *_dyn variants of five entry points read at run time (dyn == NULL: exactly the plain entry point):
 *   seed_xor  -- XORed into the seed argument of the keep-mask pool (step_dropout_pool_fill_dyn), of the Gumbel noise
 *                (step_dgl_edges_forward_dyn) and of the gcn dropout (step_gwnet_forward_phase_dyn);
 *   adam_step -- the optimizer step count t of step_adam_clip_dyn (bias corrections 1 - beta^t), lr its learning rate;
 *   gsl_coef  -- step_loss's coefficient of the graph term (step.py:68-69; it changes with the epoch).
 * step_dyn_advance (one launch at the head of every replay): seed_xor <- hash(seed_xor), adam_step += 1.  The host writes lr /
 * gsl_coef with an ordinary copy when the scheduler / the epoch changes them. */
typedef struct StepDynState {
    uint64_t seed_xor;
    int32_t adam_step;
    float lr;
    float gsl_coef;
    float reserved;
} StepDynState;
int step_dyn_advance(StepDynState* dyn, void* stream);
int step_dropout_pool_fill_dyn(uint64_t* pool, long words, float dropout_p, uint64_t seed, const StepDynState* dyn, void* stream);
int step_dgl_edges_forward_dyn(const float* g, int N, int B, const StepDglParams* p, const float* u, uint64_t seed,
                               float temperature, float* saved, float* theta_out, float* adj_out, const StepDynState* dyn, void* stream);
int step_gwnet_forward_phase_dyn(const float* hist, int B, int N, int Cin, const float* hidden_last, const float* adj,
                                 const StepGwnetParams* p, int training, float dropout_p, uint64_t seed, float momentum,
                                 float* saved, float* work, float* pred, int phase, const StepDynState* dyn, void* stream);
/* lr and the step count come from dyn (non-NULL) */
int step_adam_clip_dyn(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float beta1, float beta2, float eps,
                       float weight_decay, float max_norm, const float* extra_sumsq, float* work, float* out_norm,
                       const StepDynState* dyn, void* stream);
/* coef comes from dyn->gsl_coef (non-NULL) */
int step_loss_scaled_fwd_bwd_dyn(const float* pred, const float* real, long n_pred, long real_stride, float scale, float shift,
                                 const float* theta, const float* prior, long n_adj, float null_val, double* work, float* loss,
                                 float* dpred, float* dtheta, const StepDynState* dyn, void* stream);

/* ---------------------------------------------------------------- data-parallel collectives ---
 * The exchange step of the path (SURVEY.md 8b "grad_allreduce(handle, stream, flat_grad*, n)", 8e): ONE gradient all-reduce (mean) per
 * training step over the flat gradient buffer, plus -- with the graph learner cut into time slices (8f row 2) -- six small sums.  Replaces
 * what torch.nn.parallel.DistributedDataParallel does for the reference when GPU_NUM > 1 (step/STEP_PEMS04.py:30, STEP_PEMS07.py:29,83,117,
 * wrapped by easytorch's launcher): bucketed ncclAllReduce calls behind autograd hooks.  Here the calls are RCCL's C API in stream order:
 * no work objects, no host synchronisation.  librccl is resolved with dlopen at the first call (the copy already in the process first).
 * A communicator is a handle: created from a 128-byte unique id that rank 0 makes and the caller distributes (any side channel:
 * torch.distributed's store, MPI, a file), destroyed explicitly.  Reductions are in place on the caller's buffers.
 *   step_comm_available      1 when librccl could be loaded
 *   step_comm_allreduce      buf <- sum / mean over the ranks, queued on `stream`
 *   step_grad_allreduce      = step_comm_allreduce(f32, mean) of the flat gradient buffer on `stream`
 *   step_grad_allreduce_begin / _join   the same OVERLAPPED with the rest of the backward: begin orders the reduction behind what is queued
 *                            on `stream` and runs it on the communicator's own stream; join makes `stream` wait for every reduction begun
 *                            since the last join (the fc weight gradient, 98 % of the bytes, is finished early in the backward) */
#define STEP_COMM_ID_BYTES 128
#define STEP_COMM_F32 0
#define STEP_COMM_F64 1
#define STEP_COMM_U8 2
int step_comm_available(void);
int step_comm_version(void);          /* RCCL's version code (0 when unavailable) */
int step_comm_unique_id(void* id128);
int step_comm_init_rank(const void* id128, int nranks, int rank, void** comm_out);
int step_comm_destroy(void* comm);
int step_comm_set_side_stream(void* comm, void* stream);      /* run the overlapped all-reduce on the caller's stream (kept by reference) */
int step_comm_allreduce(void* comm, void* buf, long n, int dtype, int average, void* stream);
int step_comm_broadcast(void* comm, void* buf, long n, int dtype, int root, void* stream);
int step_grad_allreduce(void* comm, float* flat_grad, long n, void* stream);
int step_grad_allreduce_begin(void* comm, float* flat_grad, long n, void* stream);
int step_grad_allreduce_join(void* comm, void* stream);

/* ---------------------------------------------------------------- self test --------------
 * Verifies on the device the MFMA operand/accumulator lane maps this library is built on
 * (cdna_hip_programming.md section 3).  out: int32[8] failure counters, all zero when ok. */
int step_selftest_mfma(int32_t* out, void* stream);
/* Set-up helper of the host side: *concurrent = 1 when work queued on the two streams overlaps on this device (the runtime maps streams
 * onto a few hardware queues; two streams on one queue serialise).  Host-blocking: a 200 us probe. */
int step_streams_concurrent(void* stream_a, void* stream_b, int* concurrent);

#ifdef __cplusplus
}
#endif
#endif
