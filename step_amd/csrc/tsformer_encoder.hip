// Fused TSFormer encoder forward for gfx950 (forecasting mode, frozen weights).
//
// One workgroup per sequence s = (sample, node); one wave per 32-token tile (P <= 512 tokens,
// 336 for PEMS04).  The residual stream of a wave's 32 tokens lives in f32 accumulator
// registers for the whole kernel: activations are kept TRANSPOSED ([feature][token], token =
// lane & 31), weights are the MFMA A operand, and thanks to the k-slot map in
// tsformer_layout.h every accumulator tile becomes the next MFMA's B operand by a plain
// f32->bf16 pack.  Inter-wave traffic: the per-head K and V operand fragments (2 KB per key tile
// each) through LDS, and the weights, which every workgroup streams ONCE from L2 into a 2-slot
// LDS ring (global_load_lds DMA, 25 KB stage blocks, prefetched one stage ahead) shared by all waves.
//
//   patch embed + pos-emb (f32 VALU)  ->  4 x { per head: Q,K,V (MFMA, K=96) -> S^T = K Q^T
//   -> exact two-pass softmax (exp2, scale folded into Wq) -> O^T = V^T P^T (row 24 of V is
//   all-ones, so the softmax denominator falls out of the same MFMA) -> out-proj accumulates
//   onto (x + b_o) ; LN1 ; FFN in 12 chunks of 32 hidden units, never leaving registers ;
//   LN2 }  ->  encoder_norm  ->  hidden (bf16 and/or f32), last-patch state, squared norms.
//
// MFMA work per token-layer: 221 184*1.33(q/k/v/o head padding 24->32 only) ... see DESIGN.md
// for the flop accounting used by bench.py's roofline.
#include "tsformer_device.h"

namespace {

template <int MAXW, bool DROP, bool PARK, bool F16>
__global__ __launch_bounds__(MAXW * 64) void tsformer_encoder_kernel(EncArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool drop = DROP;
    typedef typename Opnd<F16>::v8 op8;            // one MFMA operand: 8 x bfloat16 or 8 x float16
    typedef typename Opnd<F16>::elem ope;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int seq = blockIdx.x;
    const int P = A.P, nkt = A.nkt;
    const int tok = wave * 32 + c;
    const bool tok_ok = tok < P;
    const int tokc = tok_ok ? tok : 0;
    const char* W = A.wpack;
    Dropper dr;
    dr.thresh = (uint32_t)(A.drop_p * 256.0f + 0.5f);
    dr.scale = drop ? 256.0f / (256.0f - (float)dr.thresh) : 1.0f;
    const uint32_t seq_salt = A.seed ^ ((uint32_t)seq * 0x7FEB352Du);

    // LDS: [K frags nkt*2 KB][V frags nkt*2 KB][weight ring: 2 stage blocks of 25 KB]
    char* kbuf = smem;
    char* vbuf = smem + nkt * 2 * TSF_FRAG;
    char* ring = smem + nkt * 4 * TSF_FRAG;
    // PARK: the bf16 operand copy of the residual stream (6 fragments per wave) lives in a wave-private LDS
    // area while the attention loops run, which frees 24 VGPRs for the software-pipelined score tiles
    char* xpark = ring + 2 * TSF_BLOCK + wave * 6 * TSF_FRAG;
    const uint32_t ring_addr = __builtin_amdgcn_readfirstlane(LDS_ADDR(ring));
    const int nstage = A.depth * TSF_STAGES;

    auto issue_fill = [&](int g) {            // stage block g -> ring slot g & 1, pieces spread over the waves
        const char* src = W + TSF_LAYER0 + (long)g * TSF_BLOCK + lane * 16;
        const uint32_t dst = ring_addr + (uint32_t)(g & 1) * TSF_BLOCK;
        for (int pc = wave; pc < 25; pc += nkt) dma_1k(src + pc * TSF_FRAG, dst + (uint32_t)pc * TSF_FRAG);
    };
    // Stage boundary: my DMA pieces of block g have landed (vmcnt), everybody's have (barrier), and
    // everybody is done reading the other slot, which is then refilled with block g+1.
    auto stage_begin = [&](int g) -> const char* {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (g + 1 < nstage) issue_fill(g + 1);
        return ring + (g & 1) * TSF_BLOCK;
    };

    issue_fill(0);

    // ------------------------------------------------------------------ patch embedding + pos
    f32x16 xT[3];
    {
        float xin[12];
        const float4* src = (const float4*)(A.series + (long)seq * A.L + (long)tokc * TSF_PATCH);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            float4 t4 = tok_ok ? src[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            xin[4 * v] = t4.x; xin[4 * v + 1] = t4.y; xin[4 * v + 2] = t4.z; xin[4 * v + 3] = t4.w;
        }
        const float4* wpe = (const float4*)(W + TSF_G_WPE) + h * 48 * 3;
        const float* bpe = (const float*)(W + TSF_G_BPE) + h * 48;
        const float* pos = (const float*)(W + TSF_POS_OFF(A.depth)) + ((long)tokc * 2 + h) * 48;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int qd = t * 16 + i;
                float4 w0 = wpe[qd * 3], w1 = wpe[qd * 3 + 1], w2 = wpe[qd * 3 + 2];
                float acc = bpe[qd];
                acc += w0.x * xin[0] + w0.y * xin[1] + w0.z * xin[2] + w0.w * xin[3];
                acc += w1.x * xin[4] + w1.y * xin[5] + w1.z * xin[6] + w1.w * xin[7];
                acc += w2.x * xin[8] + w2.y * xin[9] + w2.z * xin[10] + w2.w * xin[11];
                xT[t][i] = acc + pos[qd];
            }
        if constexpr (drop) {
            dr.base = seq_salt ^ 0xA511E9B3u;
            dr.seed((uint32_t)(tok * 2 + h));
#pragma unroll
            for (int t = 0; t < 3; ++t) dr.apply16(xT[t]);
        }
        const float sc = 9.797958971132712f;   // sqrt(96), transformer_layers.py:15
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) xT[t][i] *= sc;
    }

    // ------------------------------------------------------------------ encoder layers
    int g = 0;                                  // global stage index (10 per layer)
#pragma unroll 1
    for (int layer = 0; layer < A.depth; ++layer) {
        const uint32_t lsalt = seq_salt + (uint32_t)(layer + 1) * 0x632BE5ABu;
        op8 xb[6];
        f32x16 acc[3];

        // the first head's stage is opened outside the loop so that the f32 residual stream xT is
        // dead (folded into acc / xb) while the head loop runs
        const char* blk = stage_begin(g);
        const float* tail = (const float*)(blk + TSF_TAIL);
        {
#pragma unroll
            for (int t = 0; t < 3; ++t) { xb[2 * t] = pack_half<F16>(xT[t], 0); xb[2 * t + 1] = pack_half<F16>(xT[t], 1); }
            if constexpr (PARK) {
#pragma unroll
                for (int f = 0; f < 6; ++f) *(op8*)(xpark + f * TSF_FRAG + lane * 16) = xb[f];
            }
            const float* bo = tail + 64 + h * 48;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = (drop ? 0.f : xT[t][i]) + bo[t * 16 + i];
        }
#pragma unroll 1
        for (int hd = 0; hd < TSF_HEADS; ++hd, ++g) {
            if (hd > 0) {
                blk = stage_begin(g);
                tail = (const float*)(blk + TSF_TAIL);
            }
            if constexpr (PARK) {
#pragma unroll
                for (int f = 0; f < 6; ++f) xb[f] = lfrag<F16>(xpark, f, lane);
            }
            // ---- Q^T (kept in registers as the B operand of S^T = K Q^T)
            TSF_PRIO_CHAIN(1);
            op8 qb[2];
            {
                f32x16 q;
                const float* bq = tail + h * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = bq[i];           // bias rides in the accumulator
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) q = mfma16<F16>(lfrag<F16>(blk, ks, lane), xb[ks], q);
                qb[0] = pack_half<F16>(q, 0);
                qb[1] = pack_half<F16>(q, 1);
                // head-dim slots 25 / 26 (the head dim is 24 of 32) carry the softmax shift and the key-padding mask
                // through the contraction: the key side holds (1, is_padding), the query side (-rowmax, -30000)
                if (h == 0) qb[1][6] = (ope)(-30000.0f);
            }
            // ---- K^T -> this tile's A-operand fragments (bias dropped: it cancels in softmax)
            {
                f32x16 kk;
#pragma unroll
                for (int i = 0; i < 16; ++i) kk[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) kk = mfma16<F16>(lfrag<F16>(blk, 6 + ks, lane), xb[ks], kk);
                if (h == 0) { kk[13] = 1.0f; kk[14] = tok_ok ? 0.0f : 1.0f; }     // slots 25 / 26 of this key (rows 25, 26)
                *(op8*)(kbuf + (wave * 2 + 0) * TSF_FRAG + lane * 16) = pack_half<F16>(kk, 0);
                *(op8*)(kbuf + (wave * 2 + 1) * TSF_FRAG + lane * 16) = pack_half<F16>(kk, 1);
            }
            // ---- V (tokens as rows) -> this tile's A-operand fragments of V^T
            {
                f32x16 vv;
                const float bv = tail[32 + c];
#pragma unroll
                for (int i = 0; i < 16; ++i) vv[i] = bv;
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) vv = mfma16<F16>(xb[ks], lfrag<F16>(blk, 12 + ks, lane), vv);
                *(op8*)(vbuf + (wave * 2 + 0) * TSF_FRAG + lane * 16) = pack_half<F16>(vv, 0);
                *(op8*)(vbuf + (wave * 2 + 1) * TSF_FRAG + lane * 16) = pack_half<F16>(vv, 1);
            }
            TSF_PRIO_CHAIN(0);
            // K/V fragments visible to every wave; the in-flight weight DMA is NOT drained here
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

            // ---- pass 1: row maxima of S^T (keys on accumulator rows, queries on lanes).  Software pipeline: the MFMAs
            // of key tile kt+1 are issued before the VALU work on tile kt; padded keys score -30000 through slot 26 and
            // never win.  (Unrolling by two to avoid the tile copy was measured 20 % SLOWER: profiles/r01_m_encoder_ab.md.)
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;
            auto score_tile = [&](int kt) -> f32x16 {
                TSF_PRIO_ATTN(1);
                f32x16 s = mfma16<F16>(lfrag<F16>(kbuf, kt * 2, lane), qb[0], zero);
                s = mfma16<F16>(lfrag<F16>(kbuf, kt * 2 + 1, lane), qb[1], s);
                TSF_PRIO_ATTN(0);
                return s;
            };
            float mx = -INFINITY;
            {
                auto fold = [&](const f32x16& sc) {
#pragma unroll
                    for (int i = 0; i < 16; i += 2) mx = fmaxf(fmaxf(mx, sc[i]), sc[i + 1]);     // v_max3_f32
                };
                f32x16 sa = score_tile(0);
#pragma unroll 1
                for (int kt = 0; kt < nkt; ++kt) {
                    const f32x16 sb = score_tile(kt + 1 < nkt ? kt + 1 : kt);
                    fold(sa);
                    sa = sb;
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (h == 0) qb[1][5] = (ope)(-mx);        // slot 25: S - max comes out of the MFMA (bf16 rounding of max cancels)

            // ---- pass 2: P = exp2(S - max), O^T += V^T P^T (row 24 of V^T is all ones: the denominator), same pipeline
            f32x16 o;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = 0.f;
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            f32x2 lsum2 = {0.f, 0.f};
            if constexpr (drop) { dr.base = lsalt ^ (0x1000193u * (uint32_t)(hd + 1)); dr.seed((uint32_t)(tok * 2 + h)); }
            {
                auto consume = [&](f32x16& sc, int kt) {
                    const op8 v0 = lfrag<F16>(vbuf, kt * 2, lane), v1 = lfrag<F16>(vbuf, kt * 2 + 1, lane);
#pragma unroll
                    for (int i = 0; i < 16; ++i) sc[i] = __builtin_amdgcn_exp2f(sc[i]);
                    if constexpr (drop) {
                        // attention-prob dropout acts on the normalised probabilities: keep the denominator
                        // dropout-free (packed VALU sum), mask the numerator only; the survivor scale is folded into 1/den
#pragma unroll
                        for (int i = 0; i < 16; i += 2) lsum2 += f32x2{sc[i], sc[i + 1]};         // v_pk_add_f32
                        dr.mask16(sc);
                    }
                    const op8 p0 = pack_half<F16>(sc, 0), p1 = pack_half<F16>(sc, 1);
                    TSF_PRIO_ATTN(1);
                    o = mfma16<F16>(v0, p0, o);
                    o = mfma16<F16>(v1, p1, o);
                    TSF_PRIO_ATTN(0);
                };
                f32x16 sa = score_tile(0);
#pragma unroll 1
                for (int kt = 0; kt < nkt; ++kt) {
                    f32x16 sb = score_tile(kt + 1 < nkt ? kt + 1 : kt);
                    consume(sa, kt);
                    sa = sb;
                }
            }
            const float lsum = lsum2[0] + lsum2[1];
            float den;
            if constexpr (drop) {
                den = (lsum + __shfl_xor(lsum, 32, 64)) * (1.0f / dr.scale);
            } else {
                den = __shfl(o[12], c, 64);      // V^T row 24 == ones: lane-half 0, register 12
            }
            const float inv = 1.0f / den;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] *= inv;
            op8 ob0 = pack_half<F16>(o, 0), ob1 = pack_half<F16>(o, 1);
            // ---- out-projection of this head accumulates onto the residual
            TSF_PRIO_CHAIN(1);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                acc[t] = mfma16<F16>(lfrag<F16>(blk, 18 + t * 2, lane), ob0, acc[t]);
                acc[t] = mfma16<F16>(lfrag<F16>(blk, 19 + t * 2, lane), ob1, acc[t]);
            }
            TSF_PRIO_CHAIN(0);
        }  // heads
        if constexpr (drop) {
            // dropout1 on (attention output + b_o); residual re-read from its bf16 operand copy
            dr.base = lsalt ^ 0x51ED27u;
            if constexpr (PARK) {
#pragma unroll
                for (int f = 0; f < 6; ++f) xb[f] = lfrag<F16>(xpark, f, lane);
            }
            add_residual_op<F16>(acc, xb, dr, (uint32_t)(tok * 2 + h));
        }
        layer_norm96(acc, tail + 64 + h * 48, tail + 160 + h * 48);      // LN1 params ride in head 3's block

        // ---- FFN 96 -> 384 -> 96: 6 stages of two 32-unit chunks, hidden units never leave registers
        if constexpr (drop) dr.base = lsalt ^ 0x2545F491u;
        blk = stage_begin(g);
        tail = (const float*)(blk + TSF_TAIL);
        {
#pragma unroll
            for (int t = 0; t < 3; ++t) { xb[2 * t] = pack_half<F16>(acc[t], 0); xb[2 * t + 1] = pack_half<F16>(acc[t], 1); }
            const float* b2 = tail + 64 + h * 48;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = (drop ? 0.f : acc[t][i]) + b2[t * 16 + i];
        }
#pragma unroll 1
        for (int j = 0; j < 6; ++j, ++g) {
            if (j > 0) {
                blk = stage_begin(g);
                tail = (const float*)(blk + TSF_TAIL);
            }
            TSF_PRIO_CHAIN(1);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                f32x16 hh;
                const float* b1 = tail + (cc * 2 + h) * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) hh[i] = b1[i];
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) hh = mfma16<F16>(lfrag<F16>(blk, cc * 12 + ks, lane), xb[ks], hh);
#pragma unroll
                for (int i = 0; i < 16; ++i) hh[i] = __builtin_amdgcn_fmed3f(hh[i], 0.f, 3.0e38f);      // relu, one VALU op
                if constexpr (drop) dr.apply16(hh);
                op8 hb0 = pack_half<F16>(hh, 0), hb1 = pack_half<F16>(hh, 1);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t] = mfma16<F16>(lfrag<F16>(blk, cc * 12 + 6 + t * 2, lane), hb0, acc[t]);
                    acc[t] = mfma16<F16>(lfrag<F16>(blk, cc * 12 + 7 + t * 2, lane), hb1, acc[t]);
                }
            }
            TSF_PRIO_CHAIN(0);
        }
        if constexpr (drop) {
            dr.base = lsalt ^ 0x9E3779B9u;
            add_residual_op<F16>(acc, xb, dr, (uint32_t)(tok * 2 + h));
        }
        layer_norm96(acc, tail + 64 + h * 48, tail + 160 + h * 48);      // LN2 params ride in the last ffn block
#pragma unroll
        for (int t = 0; t < 3; ++t) xT[t] = acc[t];
    }  // layers

    // ------------------------------------------------------------------ encoder_norm + outputs
    layer_norm96(xT, (const float*)(W + TSF_G_NORM_G) + h * 48, (const float*)(W + TSF_G_NORM_B) + h * 48);
    float sq = 0.f;
    if (tok_ok) {
        const long row = ((long)seq * P + tok) * TSF_D;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int f0 = t * 32 + 8 * g4 + 4 * h;
                float v0 = xT[t][4 * g4], v1 = xT[t][4 * g4 + 1], v2 = xT[t][4 * g4 + 2], v3 = xT[t][4 * g4 + 3];
                u32x2 pk;
                pk[0] = pack_bf16x2(v0, v1);
                pk[1] = pack_bf16x2(v2, v3);
                if (A.hid_bf16) *(u32x2*)(A.hid_bf16 + row + f0) = pk;
                if (A.hid_f32) *(float4*)(A.hid_f32 + row + f0) = make_float4(v0, v1, v2, v3);
                if (A.last_f32 && tok == P - 1) *(float4*)(A.last_f32 + (long)seq * TSF_D + f0) = make_float4(v0, v1, v2, v3);
                float r0 = bf16_bits_to_f32(pk[0] & 0xffffu), r1 = bf16_bits_to_f32(pk[0] >> 16);
                float r2 = bf16_bits_to_f32(pk[1] & 0xffffu), r3 = bf16_bits_to_f32(pk[1] >> 16);
                sq += r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
            }
    }
    if (A.sqn) {
        sq = wave_sum(sq);
        if (lane == 0) A.sqn[(long)seq * 16 + wave] = sq;
        if (wave == 0 && lane >= nkt && lane < 16) A.sqn[(long)seq * 16 + lane] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------
// [B, L, N, C] -> [B*N, L] (channel ch): LDS-tiled transpose, reads coalesced along n, writes
// coalesced along t.
__global__ __launch_bounds__(256) void pack_long_history_kernel(const float* __restrict__ x, int B, int L, int N,
                                                                int C, int ch, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int t = t0 + r, n = n0 + tx;
        tile[r][tx] = (t < L && n < N) ? x[(((long)b * L + t) * N + n) * C + ch] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int n = n0 + r, t = t0 + tx;
        if (n < N && t < L) out[((long)b * N + n) * L + t] = tile[tx][r];
    }
}

// Same output as pack_long_history_kernel, read straight from the device-resident series [T][N][C]: sequence (b, n) is
// channel ch of rows t0[b] - L .. t0[b] - 1 (windows that start before L rows exist are zero-filled, like the reference's
// dataset does for them, forecasting_dataset.py:66-67).  No [B, L, N, C] tensor is ever materialised.
__global__ __launch_bounds__(256) void gather_long_history_kernel(const float* __restrict__ data, int T, int N, int C, int ch,
                                                                  const long* __restrict__ t0, int L, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const long start = t0[b] - L;
    const int l0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int l = l0 + r, n = n0 + tx;
        const long t = start + l;
        tile[r][tx] = (l < L && n < N && start >= 0 && t < T) ? data[(t * N + n) * C + ch] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, l = l0 + tx;
        if (n < N && l < L) out[((long)b * N + n) * L + l] = tile[tx][r];
    }
}
// hist[b][j][n][c] = data[t0[b] - H + j][n][c], fut[b][j][n][c] = data[t0[b] + j][n][c]   (H = horizon = 12)
__global__ void gather_short_windows_kernel(const float* __restrict__ data, int T, int N, int C, const long* __restrict__ t0, int H,
                                            float* __restrict__ hist, float* __restrict__ fut) {
    const int b = blockIdx.y;
    const long per = (long)H * N * C;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= per) return;
    const long th = t0[b] - H, tf = t0[b];
    hist[b * per + idx] = (th >= 0 && th + H <= T) ? data[th * N * C + idx] : 0.f;
    if (fut) fut[b * per + idx] = (tf >= 0 && tf + H <= T) ? data[tf * N * C + idx] : 0.f;
}

template <int MAXW, bool DROP, bool PARK, bool F16>
int launch_enc_t(const EncArgs& a, hipStream_t st) {
    size_t lds = (size_t)a.nkt * 4 * TSF_FRAG + 2 * TSF_BLOCK + (PARK ? (size_t)a.nkt * 6 * TSF_FRAG : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)tsformer_encoder_kernel<MAXW, DROP, PARK, F16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            step_set_error("tsformer_encode: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
            return STEP_ERR_HIP;
        }
        attr_set = true;
    }
    tsformer_encoder_kernel<MAXW, DROP, PARK, F16><<<a.S, a.nkt * 64, lds, st>>>(a);
    STEP_LAUNCH_CHECK("step_tsformer_encode");
    return STEP_OK;
}
template <int MAXW, bool DROP, bool PARK>
int launch_enc(const EncArgs& a, hipStream_t st) {
    return a.f16 ? launch_enc_t<MAXW, DROP, PARK, true>(a, st) : launch_enc_t<MAXW, DROP, PARK, false>(a, st);
}

}  // namespace

extern "C" int step_tsformer_encode(const float* series, int S, int L, const void* wpack, long wpack_bytes,
                                    int depth, int operand_f16, uint16_t* hidden_bf16, float* hidden_f32, float* last_f32,
                                    float* sqnorm_part, float dropout_p, uint64_t seed, void* stream) {
    STEP_REQUIRE(series && wpack, "tsformer_encode: null input");
    STEP_REQUIRE(S > 0 && L > 0 && L % TSF_PATCH == 0, "tsformer_encode: L=%d must be a positive multiple of %d", L, TSF_PATCH);
    const int P = L / TSF_PATCH;
    STEP_REQUIRE(P <= 512, "tsformer_encode: %d tokens exceed the 512-token workgroup limit", P);
    STEP_REQUIRE(depth >= 1 && depth <= 16, "tsformer_encode: bad depth %d", depth);
    STEP_REQUIRE(wpack_bytes >= TSF_TOTAL_BYTES(depth, P), "tsformer_encode: packed weights too small (%ld < %ld)",
                 wpack_bytes, (long)TSF_TOTAL_BYTES(depth, P));
    STEP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "tsformer_encode: bad dropout %f", dropout_p);
    EncArgs a;
    a.series = series; a.S = S; a.L = L; a.P = P; a.depth = depth; a.nkt = (P + 31) / 32;
    a.wpack = (const char*)wpack; a.hid_bf16 = hidden_bf16; a.hid_f32 = hidden_f32; a.last_f32 = last_f32;
    a.sqn = sqnorm_part; a.drop_p = dropout_p; a.seed = (uint32_t)(seed ^ (seed >> 32));
    a.f16 = operand_f16 != 0;
    hipStream_t st = (hipStream_t)stream;
    const bool dr = dropout_p > 0.f;
    // parking the operand copy needs nkt * 10 KB + 50 KB of LDS (<= 160 KB up to 11 token tiles = 352 tokens)
    if (a.nkt <= 4) return dr ? launch_enc<4, true, true>(a, st) : launch_enc<4, false, true>(a, st);
    if (a.nkt <= 8) return dr ? launch_enc<8, true, true>(a, st) : launch_enc<8, false, true>(a, st);
    if (a.nkt <= 11) return dr ? launch_enc<12, true, true>(a, st) : launch_enc<12, false, true>(a, st);
    if (a.nkt <= 12) return dr ? launch_enc<12, true, false>(a, st) : launch_enc<12, false, false>(a, st);
    return dr ? launch_enc<16, true, false>(a, st) : launch_enc<16, false, false>(a, st);
}

extern "C" int step_gather_windows(const float* data, int T, int N, int C, int ch, const long* t0, int B, int L, int H,
                                   float* long_series, float* hist, float* fut, void* stream) {
    STEP_REQUIRE(data && t0 && T > 0 && N > 0 && C > 0 && ch >= 0 && ch < C && B > 0 && L >= 0 && H > 0, "gather_windows: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (long_series) {
        STEP_REQUIRE(L > 0, "gather_windows: long history length must be positive");
        gather_long_history_kernel<<<dim3(cdiv(L, 32), cdiv(N, 32), B), 256, 0, st>>>(data, T, N, C, ch, t0, L, long_series);
        STEP_LAUNCH_CHECK("gather_long_history");
    }
    if (hist) {
        gather_short_windows_kernel<<<dim3(cdiv((long)H * N * C, 256), B), 256, 0, st>>>(data, T, N, C, t0, H, hist, fut);
        STEP_LAUNCH_CHECK("gather_short_windows");
    }
    return STEP_OK;
}

extern "C" int step_pack_long_history(const float* x, int B, int L, int N, int C, int ch, float* out, void* stream) {
    STEP_REQUIRE(x && out && B > 0 && L > 0 && N > 0 && C > 0 && ch >= 0 && ch < C, "pack_long_history: bad arguments");
    dim3 grid(cdiv(L, 32), cdiv(N, 32), B);
    pack_long_history_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, B, L, N, C, ch, out);
    STEP_LAUNCH_CHECK("step_pack_long_history");
    return STEP_OK;
}
