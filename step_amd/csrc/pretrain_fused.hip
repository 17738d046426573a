// Row kernels of the TSFormer pre-training step (reference: nn.TransformerEncoderLayer inside step/step_arch/tsformer/transformer_layers.py:7-21,
// trained by tsformer.py:71-160), bf16 mode: everything of a layer that acts on one token at a time, as kernels that stream [R, 96] row
// tensors once.  In this file, in order:
//   the fused feed-forward block        ffn_pack / ffn_rows (forward, backward-data) / ffn_wgrad / ffn_reduce        (described below)
//   tile staging helpers                row tiles <-> transposed register layout through wave-private swizzled LDS, and
//                                       residual add + dropout + LayerNorm as an output stage (tile_out_ln)
//   the projections of the attention    rows_linear (qkv, out-projection, d a, d x +=; weights resident in LDS), proj_wgrad (d Wi, d bi, d Wo in one pass)
//   the ends of the network             embed_unmasked_fwd / _bwd (patch + positional embedding of the unmasked tokens), dec_input_bwd_sums
//   layer_pack                          one launch that writes all operand-fragment buffers of a layer
// The attention itself, the LayerNorm backward and the f32 path are in pretrain.hip.
//
// Fused feed-forward block.  The layer-by-layer path (pretrain.hip + step_gemm) stores the [R, 384] hidden layer and its gradient as bf16 tensors
// (R = sequences x tokens = 873 600 rows in the decoder layer of config C3: 671 MB each) and walks them eight times per layer.
// Here the hidden layer never leaves registers, in the forward AND in the backward (which recomputes it):
//
//   ffn_rows_kernel<BWD = false>   f2   = W2 . drop(relu(W1 . h1 + b1)) + b2 (+ residual, dropout, LayerNorm 2)    reads h1, writes f2 (or pre / y / stats)
//   ffn_rows_kernel<BWD = true>    dh1 += W1^T . [ (W2^T . df2) * relu' * keep ]                               reads h1, df2, read-modify-writes dh1
//   ffn_wgrad_kernel<W1K = false>  dW2[o,j] = sum over rows of df2[row,o] * hid[row,j]                          reads h1, df2
//   ffn_wgrad_kernel<W1K = true>   dW1[j,i] = sum over rows of dhid[row,j] * h1[row,i], db1 = column sums of dhid   reads h1, df2
//
// The first two keep activations TRANSPOSED like the forecasting-mode encoder (tsformer_encoder.hip): token = lane & 31, the 96
// features of a token in three accumulator tiles, weights as the A operand streamed through an LDS ring of stage blocks
// (global_load_lds, two chunks of 32 hidden units per block), and every accumulator tile is the next product's B operand after a
// plain f32 -> bf16 pack (k-slot map F(s, h, j) of tsformer_layout.h).
// The two weight-gradient kernels contract over ROWS, so there a wave owns a chunk of 32 hidden units for the whole launch (its
// slices of W1 / W2 as B operands in registers, its 96 x 32 gradient tile in accumulators) and all twelve waves of the workgroup
// walk the same row tiles, whose operand fragments -- rows as the non-contracted index ("X") and rows as the contracted index
// ("Y") -- are staged in LDS once per tile.  hid / dhid come out of the matrix cores with the hidden unit as the lane, which is
// already the operand layout of the products that follow.  Per-workgroup partial gradients go to a workspace; ffn_reduce_kernel
// adds them into the gradient buffers.
//
// Dropout of the hidden units: keep decisions are bits of the per-step Bernoulli pool (step_dropout_pool_fill, the forecasting
// encoder's scheme, tsformer_device.h): the 32-row tile `tile32` of call site `site` owns 12 x 16 consecutive 64-bit words at a
// hashed offset; word c * 16 + i, bit 32 h + r  <->  row 32 tile32 + r, hidden unit 32 c + (i & 3) + 8 (i >> 2) + 4 h.
// Forward and backward-data read them as lane masks (scalar loads); the weight-gradient kernels, where the lane is the hidden
// unit, load the one word that holds their unit and test the row's bit.
#include "common.h"
#include <mutex>
#include <stdlib.h>
#include "step_internal.h"
#include "tsformer_device.h"

#ifndef FF_ABLATE
#define FF_ABLATE 0       // timing experiments only (WRONG results): 1 no matrix-core chains in the row kernels, 2 no global loads of the row tiles, 4 no stores,
                          // 8 no global loads in the weight-gradient kernels' staging, 16 no matrix-core work there,
                          // 32 no weight-fragment DMA in the row kernels (the ring keeps whatever it holds)
#endif

namespace {

typedef bf16x8 op8;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int FF_FRAG = 1024;
constexpr int FF_BLOCK_F = 25 * FF_FRAG;            // forward stage block: 2 x (6 W1 + 6 W2 fragments) + 1 KB of f32 vectors
constexpr int FF_BLOCK_B = 37 * FF_FRAG;            // backward-data stage block: 2 x (6 W1 + 6 W2^T + 6 W1^T fragments) + 1 KB
constexpr int FF_OFF_B = 6 * FF_BLOCK_F;
constexpr int FF_PACK_BYTES = 6 * FF_BLOCK_F + 6 * FF_BLOCK_B;

__device__ __forceinline__ int chain_f(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }
__device__ __forceinline__ int row16(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

__device__ __forceinline__ op8 relu_bf16(op8 v) {
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(op8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
}
__device__ __forceinline__ op8 mfrag(const char* base, int frag, int lane) { return *(const op8*)(base + frag * FF_FRAG + lane * 16); }
__device__ __forceinline__ f32x16 mma(op8 a, op8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ op8 pack_lo_hi(const f32x16& v, int s) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = v[8 * s + j];
    return pack8(t);
}

// ---------------------------------------------------------------------------------------------------------------- weight fragments
// A-operand fragment: lane (h, r), slot j.  Forward block jb, chunk c = 2 jb + cc, fragments 12 cc + ...:
//     0..5   W1 rows 32 c + r, k-step ks: W1[32 c + r][32 (ks >> 1) + F(ks & 1, h, j)]
//     6..11  W2 tile t, k-step s:         W2[32 t + r][32 c + F(s, h, j)]
// backward block jb, fragments 18 cc + ...:
//     0..5   the same W1 fragments
//     6..11  W2^T rows = hidden units:    W2[32 (ks >> 1) + F(ks & 1, h, j)][32 c + r]
//     12..17 W1^T tile t, k-step s:       W1[32 c + F(s, h, j)][32 t + r]
// f32 tail of every block: [0..63] b1 of both chunks in accumulator-register order [cc][h][16]; forward blocks also [64..159] b2 [h][48].
constexpr int FFN_PACK_THREADS = (6 * 24 + 6 * 36) * 64 + 12 * 256;
__device__ __forceinline__ void ffn_pack_body(int gid, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                              const float* __restrict__ b2, char* __restrict__ pack) {
    constexpr int NF_F = 6 * 24, NF_B = 6 * 36;
    if (gid < (NF_F + NF_B) * 64) {
        const int frag = gid >> 6, lane = gid & 63, r = lane & 31, h = lane >> 5;
        float v[8];
        char* dst;
        int q, c;
        bool bwd = frag >= NF_F;
        if (!bwd) {
            const int jb = frag / 24, f = frag % 24;
            c = 2 * jb + f / 12;
            q = f % 12;
            dst = pack + (long)jb * FF_BLOCK_F + f * FF_FRAG + lane * 16;
        } else {
            const int fb = frag - NF_F, jb = fb / 36, f = fb % 36;
            c = 2 * jb + f / 18;
            q = f % 18;
            dst = pack + FF_OFF_B + (long)jb * FF_BLOCK_B + f * FF_FRAG + lane * 16;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (q < 6) v[j] = w1[(32 * c + r) * 96 + 32 * (q >> 1) + chain_f(q & 1, h, j)];
            else if (!bwd) v[j] = w2[(32 * ((q - 6) >> 1) + r) * 384 + 32 * c + chain_f((q - 6) & 1, h, j)];
            else if (q < 12) v[j] = w2[(32 * ((q - 6) >> 1) + chain_f((q - 6) & 1, h, j)) * 384 + 32 * c + r];
            else v[j] = w1[(32 * c + chain_f((q - 12) & 1, h, j)) * 96 + 32 * ((q - 12) >> 1) + r];
        }
        *(op8*)dst = pack8(v);
        return;
    }
    const int t = gid - (NF_F + NF_B) * 64;          // tails: 12 blocks x 256 floats
    if (t < 12 * 256) {
        const int blk = t >> 8, i = t & 255;
        const bool bwd = blk >= 6;
        const int jb = bwd ? blk - 6 : blk;
        float* dst = (float*)(pack + (bwd ? FF_OFF_B + (long)jb * FF_BLOCK_B + 36 * FF_FRAG : (long)jb * FF_BLOCK_F + 24 * FF_FRAG));
        float v = 0.f;
        if (i < 64) {
            const int cc = i >> 5, h = (i >> 4) & 1, e = i & 15;
            v = b1[32 * (2 * jb + cc) + row16(e, h)];
        } else if (i < 160 && !bwd) {
            const int k = i - 64, h = k / 48, tt = (k % 48) >> 4, e = k & 15;
            v = b2[32 * tt + row16(e, h)];
        }
        dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void ffn_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, char* __restrict__ pack) {
    ffn_pack_body(blockIdx.x * 256 + threadIdx.x, w1, b1, w2, b2, pack);
}

// ---------------------------------------------------------------------------------------------------------------- shared pieces
struct LnEpi {
    const float* res; float* pre; float* y; float* stats; const float* gamma; const float* beta;
    float p; uint32_t lo, hi, site;
};
struct FfnArgs {
    const float* h1;                 // [R, 96] input of the block (LayerNorm1 output)
    const float* df2;                // [R, 96] gradient of the block's output (backward kernels)
    float* out;                      // forward: f2 [R, 96]; backward-data: dh1 [R, 96], accumulated in place
    long R;
    const char* pack;
    const float* b1;                 // [384] (weight-gradient kernels: bias by lane)
    float inv_keep;                  // 1 / (1 - p), 1 when dropout is off
    const unsigned long long* pool;  // keep-mask pool (NULL: no dropout)
    uint32_t pool_mask;
    uint32_t seed, site;
    float* ws;                       // weight-gradient kernels: per-workgroup partial results
    LnEpi ln;                        // forward with LN = true: the output stage (residual add + dropout + LayerNorm 2)
};

__device__ __forceinline__ uint32_t ffn_mask_base(uint32_t seed, uint32_t site, long tile32, uint32_t pool_mask) {
    return mix32(seed + (uint32_t)tile32 * 0x9E3779B1u + (site + 1u) * 0x632BE5ABu) & pool_mask;
}

// Row tiles move between HBM and the transposed register layout through wave-private LDS, so that every global access is a full
// 128-byte line (a lane reading its own row's 16 bytes touches 32 lines per instruction: measured 2.4x the time of the kernel's
// matrix work, profiles/r04_p_ffn_ablations.log):
//   in : the tile's 32 x 96 floats are one contiguous 12 KB run: 12 coalesced float4 loads -> bf16 -> [32 rows][24 chunks of 8 bytes],
//        chunk c of row r at position c ^ ((r >> 1) & 7) (conflict-free for the 16 lanes of a ds_read_b64 pass); an operand fragment
//        (k-step f = 2 b + s, lane (r, h)) is chunks 4 f + h and 4 f + h + 2 of row r
//   out: one 32-feature block at a time, [32 rows][8 chunks of 16 bytes] f32, chunk position c ^ (r & 7); read back as 8 lanes per
//        128-byte line of a row (plain store, or read-modify-write for the accumulated input gradient)
constexpr int STG_IN = 32 * 192, STG_OUT = 32 * 128, STG_WAVE = STG_IN + STG_OUT;
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }      // LDS executes a wave's accesses in order; this pins the program order

// tile_load only issues the 12 loads (the next pass's tile is requested before the current pass's matrix work and converted after it)
__device__ __forceinline__ void tile_load(const float* __restrict__ x, long row0, long R, int lane, float4 (&v)[12]) {
    const float* src = x + row0 * 96;
    const long left = R - row0;
    const int nvalid = (int)(left >= 32 ? 32 : (left > 0 ? left : 0)) * 96;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int flat = k * 256 + lane * 4;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (flat < nvalid && !(FF_ABLATE & 2)) v[k] = *(const float4*)(src + flat);
    }
}
__device__ __forceinline__ void tile_put(const float4 (&v)[12], char* stg, int lane) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int flat = k * 256 + lane * 4, r = flat / 96, c = (flat - r * 96) >> 2;
        u32x2 pk;
        pk[0] = pack_bf16x2(v[k].x, v[k].y);
        pk[1] = pack_bf16x2(v[k].z, v[k].w);
        *(u32x2*)(stg + r * 192 + 8 * (c ^ ((r >> 1) & 7))) = pk;
    }
    lds_order();
}
__device__ __forceinline__ void tile_frags(const char* stg, int lane, op8 (&b)[6]) {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
    const char* row = stg + r * 192;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const u32x2 lo = *(const u32x2*)(row + 8 * ((4 * f + h) ^ sw)), hi = *(const u32x2*)(row + 8 * ((4 * f + h + 2) ^ sw));
        u32x4 w;
        w[0] = lo[0]; w[1] = lo[1]; w[2] = hi[0]; w[3] = hi[1];
        b[f] = __builtin_bit_cast(op8, w);
    }
    lds_order();
}
template <bool RMW>
__device__ __forceinline__ void tile_out(float* __restrict__ y, long row0, long R, char* ost, int lane, const f32x16 (&acc)[3]) {
    if (FF_ABLATE & 4) return;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4*)(ost + r * 128 + 16 * ((2 * q + h) ^ (r & 7))) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        lds_order();
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int id = n * 64 + lane, rr = id >> 3, cq = id & 7;
            float4 v = *(const float4*)(ost + rr * 128 + 16 * (cq ^ (rr & 7)));
            if (row0 + rr < R) {
                float* g = y + (row0 + rr) * 96 + 32 * t + 4 * cq;
                if constexpr (RMW) {
                    const float4 o = *(const float4*)g;
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *(float4*)g = v;
            }
        }
        lds_order();
    }
}

// Residual add + dropout + LayerNorm as the output stage of a row kernel (what step_pt_add_layernorm_fwd does in a pass of its own, with
// the same Philox stream: one call per 4 consecutive elements of the [R, 96] tensor): pre = res + dropout(branch), y = LayerNorm(pre),
// stats = (mean, rstd).  It runs in the row-major domain of tile_out: a lane holds 4 consecutive features of 4 rows per 32-feature block,
// the 8 lanes of a row are adjacent, so the row statistics are three xor-shuffles; the branch's 32 x 96 tile never reaches HBM.
__device__ __forceinline__ float sum8(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ __forceinline__ void tile_out_ln(const LnEpi& E, long row0, long R, char* ost, int lane, const f32x16 (&acc)[3]) {
    const int r = lane & 31, h = lane >> 5, cq = lane & 7;
    const float ks = E.p > 0.f ? 1.f / (1.f - E.p) : 1.f;
    float4 val[4][3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4*)(ost + r * 128 + 16 * ((2 * q + h) ^ (r & 7))) = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        lds_order();
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int rr = n * 8 + (lane >> 3);
            const float4 v = *(const float4*)(ost + rr * 128 + 16 * (cq ^ (rr & 7)));
            const long row = row0 + rr, k = row * 24 + 8 * t + cq;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            float m[4] = {1.f, 1.f, 1.f, 1.f};
            if (row < R) {
                a = ((const float4*)E.res)[k];
                if (E.p > 0.f) {
                    uint32_t rnd[4];
                    philox4x32((uint32_t)k, (uint32_t)(k >> 32), E.site, 0xD20Fu, E.lo, E.hi, rnd);
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[j] = u32_to_unit(rnd[j]) >= E.p ? ks : 0.f;
                }
            }
            val[n][t] = make_float4(a.x + v.x * m[0], a.y + v.y * m[1], a.z + v.z * m[2], a.w + v.w * m[3]);
        }
        lds_order();
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const long row = row0 + n * 8 + (lane >> 3);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) s += (val[n][t].x + val[n][t].y) + (val[n][t].z + val[n][t].w);
        const float mean = sum8(s) * (1.f / 96.f);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float dx = val[n][t].x - mean, dy = val[n][t].y - mean, dz = val[n][t].z - mean, dw = val[n][t].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        const float rstd = rsqrtf(sum8(q) * (1.f / 96.f) + 1e-5f);
        if (row < R) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const long k = row * 24 + 8 * t + cq;
                const float4 g4 = ((const float4*)E.gamma)[8 * t + cq], b4 = ((const float4*)E.beta)[8 * t + cq], v = val[n][t];
                if (E.pre) ((float4*)E.pre)[k] = v;
                ((float4*)E.y)[k] = make_float4((v.x - mean) * rstd * g4.x + b4.x, (v.y - mean) * rstd * g4.y + b4.y, (v.z - mean) * rstd * g4.z + b4.z,
                                               (v.w - mean) * rstd * g4.w + b4.w);
            }
            if (cq == 0) { E.stats[row * 2] = mean; E.stats[row * 2 + 1] = rstd; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward / backward-data
// One workgroup = NW waves = NW tiles of 32 rows per pass; persistent over passes.  BWD = false: forward, BWD = true: backward-data.
// FLAGS (experiment, STEP_FFN_RING_FLAGS=1): the weight ring without workgroup barriers.  Every wave is producer (of its DMA pieces) and
// consumer of every stage block; instead of s_barrier it (a) adds its piece count to the slot's `arrived` counter once its pieces have
// landed and waits until the counter shows all 25 / 37 pieces of the block it is about to read, (b) adds 1 to the slot's `done` counter when
// it has read a block, and refills a slot only when all NW waves are done with the block that was in it.  With NSLOT = 3 the waves may
// drift up to two stages apart, so one wave's tile loads and stores run under the others' matrix work.
template <int NW, bool BWD, bool DROP, bool LN = false, int NSLOT = 2, bool FLAGS = false>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 2 : 1)) void ffn_rows_kernel(FfnArgs A) {      // (four-wave variant: two waves per SIMD = two workgroups per unit)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int BLOCK = BWD ? FF_BLOCK_B : FF_BLOCK_F;
    constexpr int NPIECE = BLOCK / FF_FRAG;
    constexpr int TAILOFF = (NPIECE - 1) * FF_FRAG;
    constexpr int CF = BWD ? 18 : 12;                   // fragments per chunk
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const long npass = (A.R + 32 * NW - 1) / (32 * NW);
    if ((long)blockIdx.x >= npass) return;
    const int mine = (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1;
    const int nstage = mine * 6;
    const char* W = A.pack + (BWD ? FF_OFF_B : 0);
    const uint32_t ring_addr = __builtin_amdgcn_readfirstlane(LDS_ADDR(smem));
    const mask_ptr pool = (mask_ptr)(uintptr_t)A.pool;
    uint32_t* flag = (uint32_t*)(smem + NSLOT * BLOCK);                      // FLAGS: arrived[NSLOT], done[NSLOT]
    // this wave's staging area (tile_in / tile_out).  Four-wave workgroups (STEP_FFN_FWD_WAVES=4: two workgroups per compute unit) stage their
    // output over the input tile, which is dead once its fragments are in registers: 74 KB of LDS per workgroup instead of 90
    constexpr int WSTRIDE = NW == 4 ? STG_IN : STG_WAVE, OUT_OFF = NW == 4 ? 0 : STG_IN;
    char* stg = smem + NSLOT * BLOCK + (FLAGS ? 64 : 0) + wave * WSTRIDE;
    if constexpr (FLAGS) {
        if (threadIdx.x < 2 * NSLOT) flag[threadIdx.x] = 0u;
        __syncthreads();
    }
    const uint32_t my_pieces = (uint32_t)((NPIECE - wave + NW - 1) / NW);

    auto issue_fill = [&](int g) {                      // stage g reads block g mod 6 into ring slot g mod NSLOT
        if constexpr (FLAGS) {                          // ... once every wave has read the block that was there
            const uint32_t need = (uint32_t)NW * (uint32_t)(g / NSLOT);
            while (__atomic_load_n(&flag[NSLOT + g % NSLOT], __ATOMIC_RELAXED) < need) __builtin_amdgcn_s_sleep(1);
        }
        const char* src = W + (long)(g % 6) * BLOCK + lane * 16;
        const uint32_t dst = ring_addr + (uint32_t)(g % NSLOT) * BLOCK;
        if (!(FF_ABLATE & 32))
            for (int pc = wave; pc < NPIECE; pc += NW) dma_1k(src + pc * FF_FRAG, dst + (uint32_t)pc * FF_FRAG);
    };
    int issued = 1, signaled = 0;
    auto stage_begin = [&](int g) -> const char* {      // see tsformer_encoder.hip: my pieces landed, everybody's did, slot of g - 1 is free
        if constexpr (FLAGS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // my pieces of every block requested so far have landed
            for (; signaled < issued; ++signaled)
                if (lane == 0) atomicAdd(&flag[signaled % NSLOT], my_pieces);
            const uint32_t want = (uint32_t)NPIECE * (uint32_t)(g / NSLOT + 1);
            while (__atomic_load_n(&flag[g % NSLOT], __ATOMIC_RELAXED) < want) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const int upto = g + NSLOT - 1 < nstage ? g + NSLOT - 1 : nstage - 1;
        for (; issued <= upto; ++issued) issue_fill(issued);
        return smem + (g % NSLOT) * BLOCK;
    };
    auto stage_end = [&](int g) {                       // FLAGS: this wave has read block g
        if constexpr (FLAGS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(&flag[NSLOT + g % NSLOT], 1u);
        }
    };
    issue_fill(0);

    int g = 0;
    float4 nx[12];                                      // the next pass's h1 tile, in flight while the current pass computes
    tile_load(A.h1, ((long)blockIdx.x * NW + wave) * 32, A.R, lane, nx);
#pragma unroll 1
    for (long pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        const long tile32 = pass * NW + wave;
        op8 xb[6], db[6];
        f32x16 acc[3];
        tile_put(nx, stg, lane);
        tile_frags(stg, lane, xb);
        if constexpr (BWD) {
            float4 dv[12];
            tile_load(A.df2, tile32 * 32, A.R, lane, dv);
            tile_put(dv, stg, lane);
            tile_frags(stg, lane, db);
        }
        if (pass + gridDim.x < npass) tile_load(A.h1, ((pass + gridDim.x) * NW + wave) * 32, A.R, lane, nx);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const uint32_t mbase = DROP ? ffn_mask_base(A.seed, A.site, tile32, A.pool_mask) : 0u;
        // keep-mask words of a stage's two chunks (scalar loads).  They are requested at the END of the previous stage: LDS reads and scalar
        // loads share one wait counter, so a request next to its use is waited for by the first fragment read behind it (0.07 ms of a
        // 0.28 ms forward); behind the last matrix instructions of a stage it completes under the barrier.
        unsigned long long mw[2][16];
        auto load_masks = [&](int jb_) {
            if constexpr (DROP) {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const mask_ptr mp = pool + ((mbase + (uint32_t)(jb_ * 2 + cc) * 16u) & A.pool_mask);
#pragma unroll
                    for (int i = 0; i < 16; ++i) mw[cc][i] = mp[i];
                }
            }
        };
        load_masks(0);
        const char* blk = nullptr;
#pragma unroll 1
        for (int jb = 0; jb < ((FF_ABLATE & 1) ? 1 : 6); ++jb, g += ((FF_ABLATE & 1) ? 6 : 1)) {
            blk = stage_begin(g);
            const float* tail = (const float*)(blk + TAILOFF);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                op8 wu[6];
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) wu[ks] = mfrag(blk, cc * CF + ks, lane);
                f32x16 hh;
                const float* b1 = tail + (cc * 2 + h) * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) hh[i] = b1[i];
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) hh = mma(wu[ks], xb[ks], hh);
                if constexpr (!BWD) {
                    op8 wd[6];
#pragma unroll
                    for (int f = 0; f < 6; ++f) wd[f] = mfrag(blk, cc * CF + 6 + f, lane);
                    if constexpr (DROP) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) hh[i] = __builtin_amdgcn_inverse_ballot_w64(mw[cc][i]) ? hh[i] : 0.f;
                    }
                    const op8 hb0 = relu_bf16(pack_lo_hi(hh, 0)), hb1 = relu_bf16(pack_lo_hi(hh, 1));
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc[t] = mma(wd[2 * t], hb0, acc[t]);
                        acc[t] = mma(wd[2 * t + 1], hb1, acc[t]);
                    }
                } else {
                    op8 wt[6];
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) wt[ks] = mfrag(blk, cc * CF + 6 + ks, lane);
                    f32x16 dd;
#pragma unroll
                    for (int i = 0; i < 16; ++i) dd[i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) dd = mma(wt[ks], db[ks], dd);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        bool open = hh[i] > 0.f;
                        if constexpr (DROP) open = open && __builtin_amdgcn_inverse_ballot_w64(mw[cc][i]);
                        dd[i] = open ? dd[i] * A.inv_keep : 0.f;
                    }
                    const op8 d0 = pack_lo_hi(dd, 0), d1 = pack_lo_hi(dd, 1);
                    op8 wc[6];
#pragma unroll
                    for (int f = 0; f < 6; ++f) wc[f] = mfrag(blk, cc * CF + 12 + f, lane);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        acc[t] = mma(wc[2 * t], d0, acc[t]);
                        acc[t] = mma(wc[2 * t + 1], d1, acc[t]);
                    }
                }
            }
            if (jb < 5) { load_masks(jb + 1); stage_end(g); }
        }
        if constexpr (!BWD) {
            const float* b2 = (const float*)(blk + TAILOFF) + 64 + h * 48;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = __builtin_fmaf(acc[t][i], A.inv_keep, b2[t * 16 + i]);
        }
        stage_end(g - 1);                               // (the last block of the pass: b2 rides in its tail)
        if constexpr (LN) tile_out_ln(A.ln, tile32 * 32, A.R, stg + OUT_OFF, lane, acc);
        else tile_out<BWD>(A.out, tile32 * 32, A.R, stg + OUT_OFF, lane, acc);     // backward-data: added onto the residual branch's gradient already in dh1
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight gradients
// Twelve waves, wave = chunk of 32 hidden units.  Per 32-row tile the workgroup stages operand fragments in LDS (bf16):
//   X(h1), X(df2): 6 fragments each, lane = row, slots = features in chain order (A operand of hid / dhid)
//   Y(v), 6 fragments [t][s]: lane = feature 32 t + r, slots = rows F(s, h, j) (rows as the contracted index)
// W2 kernel: X(h1), Y(df2);   W1 kernel: X(h1), X(df2), Y(h1).
constexpr int FW_WAVES = 12;
constexpr int FW_TILES = 2;                       // 32-row tiles per stage

template <bool W1K>
struct FwLayout {
    static constexpr int FRAGS = W1K ? 18 : 12;   // per 32-row tile
    static constexpr int STAGE = FW_TILES * FRAGS * FF_FRAG;
    static constexpr int JOBS = W1K ? 5 : 4;      // staging jobs per tile: X jobs (one tensor each), Y jobs (two fragments... see below)
};

// stage one 32-row tile's fragments into `dst`; job ids: W2 kernel {0: X(h1) -> frags 0..5, 1..3: Y(df2) tile t = job - 1 -> frags 6 + 2 t ..};
// W1 kernel {0: X(h1) -> 0..5, 1: X(df2) -> 6..11, 2..4: Y(h1) tile t = job - 2 -> frags 12 + 2 t ..}
template <bool W1K>
__device__ __forceinline__ void fw_stage_job(const FfnArgs& A, int job, long row0, char* dst, int lane) {
    const int r = lane & 31, h = lane >> 5;
    const int nx = W1K ? 2 : 1;
    if (job < nx) {
        const float* src = job == 0 ? A.h1 : A.df2;
        const long row = row0 + r;
        const bool ok = row < A.R;
        // one 32-feature block at a time (not unrolled): the gradient accumulators and the weight slices stay in registers meanwhile
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            f32x16 a;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok && !(FF_ABLATE & 8)) v = *(const float4*)(src + row * 96 + 32 * t + 8 * q + 4 * h);
                a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
            }
            *(op8*)(dst + (job * 6 + 2 * t) * FF_FRAG + lane * 16) = pack_lo_hi(a, 0);
            *(op8*)(dst + (job * 6 + 2 * t + 1) * FF_FRAG + lane * 16) = pack_lo_hi(a, 1);
        }
    } else {
        const int t = job - nx;
        const float* src = W1K ? A.h1 : A.df2;
        const int fbase = (W1K ? 12 : 6) + 2 * t;
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long row = row0 + chain_f(s, h, j);
                v[j] = row < A.R && !(FF_ABLATE & 8) ? src[row * 96 + 32 * t + r] : 0.f;
            }
            *(op8*)(dst + (fbase + s) * FF_FRAG + lane * 16) = pack8(v);
        }
    }
}

template <bool W1K, bool DROP>
__global__ __launch_bounds__(FW_WAVES * 64) void ffn_wgrad_kernel(FfnArgs A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef FwLayout<W1K> LY;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // = chunk of 32 hidden units
    const int u = lane & 31, h = lane >> 5;
    const long ntile = (A.R + 31) / 32;
    const long nstage_all = (ntile + FW_TILES - 1) / FW_TILES;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float dbias = 0.f;
    // this wave's slices of the weights as B operands (lane = hidden unit), from the backward stage blocks of the pack
    op8 w1a[6], w2t[6];
    {
        const char* blk = A.pack + FF_OFF_B + (long)(wave >> 1) * FF_BLOCK_B + (wave & 1) * 18 * FF_FRAG;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            w1a[ks] = mfrag(blk, ks, lane);
            if constexpr (W1K) w2t[ks] = mfrag(blk, 6 + ks, lane);
        }
    }
    const float bias = A.b1[32 * wave + u];
    // the pool word of this lane's hidden unit inside a chunk's 16 words, and which half of it holds the row bits
    const uint32_t widx = (uint32_t)((u & 3) + 4 * (u >> 3));
    const int hp = (u >> 2) & 1;

    auto stage_fill = [&](long st, char* buf) {
        for (int jb = wave; jb < FW_TILES * LY::JOBS; jb += FW_WAVES) {
            const int tl = jb / LY::JOBS, job = jb % LY::JOBS;
            fw_stage_job<W1K>(A, job, (st * FW_TILES + tl) * 32, buf + tl * LY::FRAGS * FF_FRAG, lane);
        }
    };
    if ((long)blockIdx.x < nstage_all) stage_fill(blockIdx.x, smem);
    __syncthreads();
    int par = 0;
#pragma unroll 1
    for (long st = blockIdx.x; st < nstage_all; st += gridDim.x, par ^= 1) {
        const long nxt = st + gridDim.x;
        if (nxt < nstage_all) stage_fill(nxt, smem + (par ^ 1) * LY::STAGE);
        const char* buf = smem + par * LY::STAGE;
#pragma unroll 1
        for (int tl = 0; tl < FW_TILES; ++tl) {
            const long tile32 = st * FW_TILES + tl;
            if (tile32 >= ntile || ((FF_ABLATE & 16) && tile32 > 0)) break;
            const char* fr = buf + tl * LY::FRAGS * FF_FRAG;
            f32x16 hh;
#pragma unroll
            for (int i = 0; i < 16; ++i) hh[i] = bias;
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) hh = mma(mfrag(fr, ks, lane), w1a[ks], hh);          // hid[row, unit]: lane = unit, registers = rows
            unsigned long long word = ~0ull;
            if constexpr (DROP) {
                const uint32_t mbase = ffn_mask_base(A.seed, A.site, tile32, A.pool_mask);
                word = A.pool[(mbase + (uint32_t)wave * 16u + widx) & A.pool_mask] >> (32 * hp);
            }
            const uint32_t bits = (uint32_t)word >> (4 * h);                                  // bit (i & 3) + 8 (i >> 2): the row of register i
            if constexpr (!W1K) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool open = hh[i] > 0.f && (bits & (1u << ((i & 3) + 8 * (i >> 2))));
                    hh[i] = open ? hh[i] : 0.f;
                }
                const op8 hb0 = pack_lo_hi(hh, 0), hb1 = pack_lo_hi(hh, 1);                    // B operand: k = rows (chain order), n = unit
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t] = mma(mfrag(fr, 6 + 2 * t, lane), hb0, acc[t]);                      // dW2[o, unit] += df2[rows, o]^T hid[rows, unit]
                    acc[t] = mma(mfrag(fr, 7 + 2 * t, lane), hb1, acc[t]);
                }
            } else {
                f32x16 dd;
#pragma unroll
                for (int i = 0; i < 16; ++i) dd[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) dd = mma(mfrag(fr, 6 + ks, lane), w2t[ks], dd);   // (df2 . W2)[row, unit]
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const bool open = hh[i] > 0.f && (bits & (1u << ((i & 3) + 8 * (i >> 2))));
                    dd[i] = open ? dd[i] : 0.f;
                    dbias += dd[i];
                }
                const op8 da0 = pack_lo_hi(dd, 0), da1 = pack_lo_hi(dd, 1);                    // A operand: m = unit, k = rows (chain order)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t] = mma(da0, mfrag(fr, 12 + 2 * t, lane), acc[t]);                     // dW1[unit, i] += dhid[rows, unit]^T h1[rows, i]
                    acc[t] = mma(da1, mfrag(fr, 13 + 2 * t, lane), acc[t]);
                }
            }
        }
        __syncthreads();
    }
    // partial results of this workgroup (the dropout survivor scale is applied by the reduction)
    float* ws = A.ws + (long)blockIdx.x * (W1K ? 384 * 96 + 384 : 96 * 384);
    if constexpr (!W1K) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) ws[(32 * t + row16(i, h)) * 384 + 32 * wave + u] = acc[t][i];      // dW2[o][unit]
    } else {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) ws[(32 * wave + row16(i, h)) * 96 + 32 * t + u] = acc[t][i];       // dW1[unit][i]
        const float other = __shfl_xor(dbias, 32, 64);
        if (h == 0) ws[384 * 96 + 32 * wave + u] = dbias + other;
    }
}

// out[i] += scale * sum over the workgroups' partial results; blockIdx.y takes every gridDim.y-th partial (f32 atomics: at most 8 per element)
__global__ __launch_bounds__(256) void ffn_reduce_kernel(const float* __restrict__ ws, int parts, long n, long stride, float scale, float* __restrict__ out0,
                                                          long n0, float* __restrict__ out1) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f;
    int p = blockIdx.y;
    for (; p + (int)gridDim.y < parts; p += 2 * gridDim.y) { s0 += ws[(long)p * stride + i]; s1 += ws[(long)(p + gridDim.y) * stride + i]; }
    if (p < parts) s0 += ws[(long)p * stride + i];
    atomicAdd(i < n0 ? out0 + i : out1 + (i - n0), scale * (s0 + s1));
}

// ---------------------------------------------------------------------------------------------------------------- projections of the attention block
// y[R, 96 NOG] (= | +=) x[R, 96 NKC] . M^T + bias with NKC * NOG <= 3: the four thin products around the attention of a layer --
//   qkv  = x . Wi^T + bi        (f32 in, bf16 out, NOG = 3)        o    = a . Wo^T + bo   (bf16 in, f32 out)
//   da   = do . Wo              (f32 in, bf16 out)                 dx  += dqkv . Wi       (bf16 in, NKC = 3, f32 read-modify-write)
// Same transposed layout and tile staging as the feed-forward row kernels, but the whole operand-fragment set of M (at most 54 KB)
// stays in LDS for the launch: after the first barrier a wave never waits for another one, so the eight waves of a workgroup drift
// apart and one wave's loads and stores run under the others' matrix work.  The step_gemm path these replace moves the same bytes at
// 1.5-2.3 TB/s (K = 96: one k step per tile, nothing to pipeline).
struct LinArgs {
    const void* x; void* y; long R; int ldx, ldy;        // leading dimensions in elements
    const char* pack;                                     // NOG * 3 * NKC * 6 fragments, then bias [NOG][2][48] f32 (zeros when there is none)
    LnEpi ln;                                             // LN = true: residual add + dropout + LayerNorm as the output stage
};
// M(out, in) = w[out * swo + in * swi]; fragment ((og * 3 + t) * NKC + kc) * 6 + ks: lane (h, r), slot j -> M(96 og + 32 t + r, 96 kc + 32 (ks >> 1) + F(ks & 1, h, j))
__device__ __forceinline__ void lin_pack_body(int gid, const float* __restrict__ w, long swo, long swi, int NKC, int NOG, const float* __restrict__ bias,
                                              char* __restrict__ pack) {
    const int nfrag = NOG * 3 * NKC * 6;
    if (gid < nfrag * 64) {
        const int frag = gid >> 6, lane = gid & 63, r = lane & 31, h = lane >> 5;
        const int ks = frag % 6, kc = (frag / 6) % NKC, ot = frag / (6 * NKC);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w[(long)(32 * ot + r) * swo + (long)(96 * kc + 32 * (ks >> 1) + chain_f(ks & 1, h, j)) * swi];
        *(op8*)(pack + (long)frag * FF_FRAG + lane * 16) = pack8(v);
        return;
    }
    const int i = gid - nfrag * 64;
    if (i < NOG * 96) {
        const int og = i / 96, k = i % 96, h = k / 48, t = (k % 48) >> 4, e = k & 15;
        ((float*)(pack + (long)nfrag * FF_FRAG))[i] = bias ? bias[96 * og + 32 * t + row16(e, h)] : 0.f;
    }
}
__global__ __launch_bounds__(256) void lin_pack_kernel(const float* __restrict__ w, long swo, long swi, int NKC, int NOG, const float* __restrict__ bias,
                                                        char* __restrict__ pack) {
    lin_pack_body(blockIdx.x * 256 + threadIdx.x, w, swo, swi, NKC, NOG, bias, pack);
}
// every fragment buffer of one layer in one launch: the feed-forward pack and the four projections (qkv, out-projection, d a, d x)
struct LayerPackArgs {
    const float *wi, *bi, *wo, *bo, *w1, *b1, *w2, *b2;
    char *ffn, *qkv, *o, *da, *dx;
};
constexpr int lin_pack_threads(int nkc, int nog) { return nog * 3 * nkc * 6 * 64 + nog * 96; }
constexpr int LP_B0 = (FFN_PACK_THREADS + 255) / 256, LP_B1 = LP_B0 + (lin_pack_threads(1, 3) + 255) / 256, LP_B2 = LP_B1 + (lin_pack_threads(1, 1) + 255) / 256,
              LP_B3 = LP_B2 + (lin_pack_threads(1, 1) + 255) / 256, LP_B4 = LP_B3 + (lin_pack_threads(3, 1) + 255) / 256;
__global__ __launch_bounds__(256) void layer_pack_kernel(LayerPackArgs A) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b < LP_B0) ffn_pack_body(b * 256 + t, A.w1, A.b1, A.w2, A.b2, A.ffn);
    else if (b < LP_B1) lin_pack_body((b - LP_B0) * 256 + t, A.wi, 96, 1, 1, 3, A.bi, A.qkv);
    else if (b < LP_B2) lin_pack_body((b - LP_B1) * 256 + t, A.wo, 96, 1, 1, 1, A.bo, A.o);
    else if (b < LP_B3) lin_pack_body((b - LP_B2) * 256 + t, A.wo, 1, 96, 1, 1, nullptr, A.da);
    else lin_pack_body((b - LP_B3) * 256 + t, A.wi, 1, 96, 3, 1, nullptr, A.dx);
}

// a 32 x 96 block of a bf16 tensor (leading dimension ld elements, first column col0) -> the swizzled bf16 tile of tile_put
__device__ __forceinline__ void tile_load_bf16(const uint16_t* __restrict__ x, long row0, long R, int ld, int col0, int lane, uint4 (&v)[6]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int idx = k * 64 + lane, r = idx / 12, c16 = idx - r * 12;
        v[k] = make_uint4(0u, 0u, 0u, 0u);
        if (row0 + r < R) v[k] = *(const uint4*)(x + (row0 + r) * ld + col0 + c16 * 8);
    }
}
__device__ __forceinline__ void tile_put_bf16(const uint4 (&v)[6], char* stg, int lane) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int idx = k * 64 + lane, r = idx / 12, c16 = idx - r * 12, sw = (r >> 1) & 7;
        // 8-byte chunks 2 c16 and 2 c16 + 1 land at positions p and p ^ 1 of one aligned 16-byte slot: one write, halves swapped when sw is odd
        const uint4 o = (sw & 1) ? make_uint4(v[k].z, v[k].w, v[k].x, v[k].y) : v[k];
        *(uint4*)(stg + r * 192 + 8 * (((2 * c16) ^ sw) & ~1)) = o;
    }
    lds_order();
}
// three accumulator tiles (96 features of 32 rows) -> bf16 rows y[row][col0 ..] through the swizzled tile
__device__ __forceinline__ void tile_out_bf16(uint16_t* __restrict__ y, long row0, long R, int ld, int col0, char* stg, int lane, const f32x16 (&acc)[3]) {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x2 pk;
            pk[0] = pack_bf16x2(acc[t][4 * q], acc[t][4 * q + 1]);
            pk[1] = pack_bf16x2(acc[t][4 * q + 2], acc[t][4 * q + 3]);
            *(u32x2*)(stg + r * 192 + 8 * ((8 * t + 2 * q + h) ^ sw)) = pk;
        }
    lds_order();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int idx = k * 64 + lane, rr = idx / 12, c16 = idx - rr * 12, s2 = (rr >> 1) & 7;
        const uint4 v = *(const uint4*)(stg + rr * 192 + 8 * (((2 * c16) ^ s2) & ~1));
        const uint4 o = (s2 & 1) ? make_uint4(v.z, v.w, v.x, v.y) : v;
        if (row0 + rr < R) *(uint4*)(y + (row0 + rr) * ld + col0 + c16 * 8) = o;
    }
    lds_order();
}

// NW waves per workgroup (round 6: 12 = three per SIMD, what 150-170 registers admit; 8 before, and still for the LayerNorm output stage's 217).  A wave's staging area is ONE 6 KB tile: the operand fragments of the input
// tile are in registers before the first product, so the f32 output blocks go through the same bytes (the first version kept a second 4 KB
// area and, with 134 KB per eight waves, one workgroup = two waves per SIMD per compute unit; a wave of these kernels waits for its own
// loads, then for its own stores: what hides that is other waves).
template <int NKC, int NOG, bool IN_BF16, bool OUT_BF16, bool ACCUM, bool LN = false, int NW = 12>
__global__ __launch_bounds__(NW * 64) void rows_linear_kernel(LinArgs A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NFRAG = NOG * 3 * NKC * 6, WBYTES = NFRAG * FF_FRAG + NOG * 96 * 4;
    static_assert(NKC * NOG <= 3 && !(IN_BF16 && NKC == 1 && false), "fragment set must fit the LDS");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    for (int i = threadIdx.x; i < WBYTES / 16; i += NW * 64) ((uint4*)smem)[i] = ((const uint4*)A.pack)[i];
    __syncthreads();
    const float* bias = (const float*)(smem + NFRAG * FF_FRAG);
    char* stg = smem + ((WBYTES + 1023) & ~1023) + wave * STG_IN;
    const long ntile = (A.R + 31) / 32;
#pragma unroll 1
    for (long tile = (long)blockIdx.x * NW + wave; tile < ntile; tile += (long)gridDim.x * NW) {
        const long row0 = tile * 32;
        if constexpr (NKC == 1) {
            op8 xb[6];
            if constexpr (IN_BF16) {
                uint4 v[6];
                tile_load_bf16((const uint16_t*)A.x, row0, A.R, A.ldx, 0, lane, v);
                tile_put_bf16(v, stg, lane);
            } else {
                float4 v[12];
                tile_load((const float*)A.x, row0, A.R, lane, v);
                tile_put(v, stg, lane);
            }
            tile_frags(stg, lane, xb);
#pragma unroll 1
            for (int og = 0; og < NOG; ++og) {
                f32x16 acc[3];
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] = bias[og * 96 + h * 48 + t * 16 + i];
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) acc[t] = mma(mfrag(smem, (og * 3 + t) * 6 + ks, lane), xb[ks], acc[t]);
                if constexpr (LN) tile_out_ln(A.ln, row0, A.R, stg, lane, acc);
                else if constexpr (OUT_BF16) tile_out_bf16((uint16_t*)A.y, row0, A.R, A.ldy, og * 96, stg, lane, acc);
                else tile_out<ACCUM>((float*)A.y + og * 96, row0, A.R, stg, lane, acc);
            }
        } else {
            static_assert(NKC == 1 || (IN_BF16 && NOG == 1 && !OUT_BF16), "the K = 288 form reads bf16 and writes f32");
            f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = bias[h * 48 + t * 16 + i];
#pragma unroll 1
            for (int kc = 0; kc < NKC; ++kc) {
                op8 xb[6];
                uint4 v[6];
                tile_load_bf16((const uint16_t*)A.x, row0, A.R, A.ldx, kc * 96, lane, v);
                tile_put_bf16(v, stg, lane);
                tile_frags(stg, lane, xb);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int ks = 0; ks < 6; ++ks) acc[t] = mma(mfrag(smem, (t * NKC + kc) * 6 + ks, lane), xb[ks], acc[t]);
            }
            tile_out<ACCUM>((float*)A.y, row0, A.R, stg, lane, acc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- weight gradients of the projections
// d Wi[j, i] = sum_r d qkv[r, j] x[r, i] (288 x 96), d bi[j] = sum_r d qkv[r, j], d Wo[j, i] = sum_r d o[r, j] a[r, i] (96 x 96) in ONE pass over
// the four row tensors (the split-K GEMMs they replace read them at 2.9 TB/s, and the bias gradient was a fifth pass over d qkv).
// Twelve waves: waves 0..8 own a block of 32 rows j of d Wi (three accumulator tiles + the bias sum), waves 9..11 a block of d Wo.  A 32-row
// tile of all four tensors (48 KB: each a contiguous run) is copied to LDS as it is -- 16-byte coalesced loads, requested one tile
// ahead -- and every wave gathers its operand fragments from the raw rows: both operands of these products contract over ROWS, i.e. lane
// = column, slots = rows F(s, h, j), which in a row-major tile is 32 consecutive elements per read (conflict-free).
constexpr int PW_DQ = 32 * 288 * 2, PW_X = 32 * 96 * 4, PW_DO = 32 * 96 * 4, PW_A = 32 * 96 * 2, PW_TILE = PW_DQ + PW_X + PW_DO + PW_A;     // 49152
constexpr int PW_WS = 288 * 96 + 288 + 96 * 96;
struct ProjWgradArgs { const float* x; const uint16_t* dqkv; const float* dov; const uint16_t* a; long R; float* ws; };

__global__ __launch_bounds__(FW_WAVES * 64) void proj_wgrad_kernel(ProjWgradArgs A) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 31, h = lane >> 5;
    const long ntile = (A.R + 31) / 32;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;
    // piece p (1 KB) of a tile's 48: dqkv 0..17, x 18..29, d o 30..41, a 42..47; this wave moves pieces wave, wave + 12, wave + 24, wave + 36, i.e. one
    // of each of the ranges [0, 12) dqkv, [12, 24) dqkv | x, [24, 36) x | d o, [36, 48) d o | a
    const int pc0 = wave, pc1 = wave + 12, pc2 = wave + 24, pc3 = wave + 36;
    const char *ps0 = (const char*)A.dqkv, *ps1 = pc1 < 18 ? (const char*)A.dqkv : (const char*)A.x, *ps2 = pc2 < 30 ? (const char*)A.x : (const char*)A.dov,
               *ps3 = pc3 < 42 ? (const char*)A.dov : (const char*)A.a;
    const long po0 = (long)pc0 * 1024 + lane * 16, po1 = (long)(pc1 < 18 ? pc1 : pc1 - 18) * 1024 + lane * 16,
               po2 = (long)(pc2 < 30 ? pc2 - 18 : pc2 - 30) * 1024 + lane * 16, po3 = (long)(pc3 < 42 ? pc3 - 30 : pc3 - 42) * 1024 + lane * 16;
    const long pl0 = A.R * 576, pl1 = pc1 < 18 ? A.R * 576 : A.R * 384, pl2 = A.R * 384, pl3 = pc3 < 42 ? A.R * 384 : A.R * 192;
    const long ts0 = PW_DQ, ts1 = pc1 < 18 ? PW_DQ : PW_X, ts2 = pc2 < 30 ? PW_X : PW_DO, ts3 = pc3 < 42 ? PW_DO : PW_A;
    uint4 nx0, nx1, nx2, nx3;
#define PW_REQUEST(tile_)                                                                                                              \
    do {                                                                                                                               \
        const long o0 = (tile_) * ts0 + po0, o1 = (tile_) * ts1 + po1, o2 = (tile_) * ts2 + po2, o3 = (tile_) * ts3 + po3;             \
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);          /* (row sizes are multiples of 16 bytes: a piece never straddles the end) */ \
        /* (out-of-range lanes read the tensor's first 16 bytes and are zeroed afterwards: a select between the address and a zero */  \
        /*  constant becomes a flat load from a scratch copy of the constant) */                                                        \
        nx0 = *(const uint4*)(ps0 + (o0 < pl0 ? o0 : 0)); nx1 = *(const uint4*)(ps1 + (o1 < pl1 ? o1 : 0));                           \
        nx2 = *(const uint4*)(ps2 + (o2 < pl2 ? o2 : 0)); nx3 = *(const uint4*)(ps3 + (o3 < pl3 ? o3 : 0));                           \
        if (!(o0 < pl0)) nx0 = z4;                                                                                                     \
        if (!(o1 < pl1)) nx1 = z4;                                                                                                     \
        if (!(o2 < pl2)) nx2 = z4;                                                                                                     \
        if (!(o3 < pl3)) nx3 = z4;                                                                                                     \
    } while (0)
#define PW_COMMIT(buf_)                                                                                                                \
    do {                                                                                                                               \
        *(uint4*)((buf_) + pc0 * 1024 + lane * 16) = nx0;                                                                              \
        *(uint4*)((buf_) + pc1 * 1024 + lane * 16) = nx1;                                                                              \
        *(uint4*)((buf_) + pc2 * 1024 + lane * 16) = nx2;                                                                              \
        *(uint4*)((buf_) + pc3 * 1024 + lane * 16) = nx3;                                                                              \
    } while (0)
    if ((long)blockIdx.x < ntile) { PW_REQUEST((long)blockIdx.x); PW_COMMIT(smem); }
    __syncthreads();
    int par = 0;
#pragma unroll 1
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x, par ^= 1) {
        const long nxt = tile + gridDim.x;
        if (nxt < ntile) PW_REQUEST(nxt);
        const char* buf = smem + par * PW_TILE;
        const uint16_t* dq = (const uint16_t*)buf;
        const float* xx = (const float*)(buf + PW_DQ);
        const float* dd = (const float*)(buf + PW_DQ + PW_X);
        const uint16_t* aa = (const uint16_t*)(buf + PW_DQ + PW_X + PW_DO);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            op8 af;
            if (wave < 9) {
                u32x4 v;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const uint32_t lo = dq[chain_f(s, h, j) * 288 + 32 * wave + c], hi = dq[chain_f(s, h, j + 1) * 288 + 32 * wave + c];
                    v[j >> 1] = lo | (hi << 16);
                    bsum += __uint_as_float(lo << 16) + __uint_as_float(hi << 16);
                }
                af = __builtin_bit_cast(op8, v);
            } else {
                f32x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = dd[chain_f(s, h, j) * 96 + 32 * (wave - 9) + c];
                af = __builtin_convertvector(v, op8);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                op8 bf;
                if (wave < 9) {
                    f32x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = xx[chain_f(s, h, j) * 96 + 32 * t + c];
                    bf = __builtin_convertvector(v, op8);
                } else {
                    u32x4 v;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) v[j >> 1] = (uint32_t)aa[chain_f(s, h, j) * 96 + 32 * t + c] | ((uint32_t)aa[chain_f(s, h, j + 1) * 96 + 32 * t + c] << 16);
                    bf = __builtin_bit_cast(op8, v);
                }
                acc[t] = mma(af, bf, acc[t]);
            }
        }
        if (nxt < ntile) PW_COMMIT(smem + (par ^ 1) * PW_TILE);
        __syncthreads();
    }
    float* ws = A.ws + (long)blockIdx.x * PW_WS;
    if (wave < 9) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) ws[(32 * wave + row16(i, h)) * 96 + 32 * t + c] = acc[t][i];
        const float other = __shfl_xor(bsum, 32, 64);
        if (h == 0) ws[288 * 96 + 32 * wave + c] = bsum + other;
    } else {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) ws[288 * 96 + 288 + (32 * (wave - 9) + row16(i, h)) * 96 + 32 * t + c] = acc[t][i];
    }
}

// ---------------------------------------------------------------------------------------------------------------- patch embedding of the unmasked tokens
// Encoder input of the pre-training step (tsformer.py:88-104: patch embedding -> positional encoding (+ dropout) -> keep the unmasked
// tokens -> * sqrt(d), transformer_layers.py:15).  The masked tokens' embeddings are dead -- the decoder puts mask_token + position there --
// so only the Pu = P / 4 unmasked tokens are computed: x[s][t] = sqrt(96) * dropout(W_pe . patch(s, um[t]) + b_pe + pos[um[t]]), and the backward
// produces the three parameter gradients from d x alone.  (The layer-by-layer path embeds all P tokens, adds / drops / gathers in four passes over
// [S, P, 96] and scatters back into a zeroed [S, P, 96] in the backward: 0.9 ms per step at config C3.)
// One thread = 4 consecutive features of one token = one Philox call of the step_pt_dropout stream over the [S, Pu, 96] tensor.
__device__ __forceinline__ void embed_keep4(uint32_t lo, uint32_t hi, uint32_t site, long i4, float p, float* m) {
    uint32_t r[4];
    philox4x32((uint32_t)i4, (uint32_t)(i4 >> 32), site, 0xD20Fu, lo, hi, r);
    const float ks = 1.f / (1.f - p);
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = u32_to_unit(r[j]) >= p ? ks : 0.f;
}
__global__ __launch_bounds__(256) void embed_unmasked_fwd_kernel(const float* __restrict__ series, const int* __restrict__ um, const float* __restrict__ w,
                                                                  const float* __restrict__ b, const float* __restrict__ pos, long S, int L, int Pu,
                                                                  float scale, float p, uint32_t lo, uint32_t hi, uint32_t site, float* __restrict__ x) {
    const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= S * Pu * 24) return;
    const int f4 = (int)(i4 % 24);
    const long st = i4 / 24;
    const int t = (int)(st % Pu);
    const long s = st / Pu;
    const int tok = um[t];
    const float4* pp = (const float4*)(series + s * L + (long)tok * 12);
    const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2];
    const float pv[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
    const float4 b4 = ((const float4*)b)[f4], e4 = ((const float4*)pos)[(long)tok * 24 + f4];
    float v[4] = {b4.x + e4.x, b4.y + e4.y, b4.z + e4.z, b4.w + e4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4* wr = (const float4*)(w + (4 * f4 + j) * 12);
        const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2];
        const float wv[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k) a = __builtin_fmaf(wv[k], pv[k], a);
        v[j] += a;
    }
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (p > 0.f) embed_keep4(lo, hi, site, i4, p, m);
    ((float4*)x)[i4] = make_float4(v[0] * m[0] * scale, v[1] * m[1] * scale, v[2] * m[2] * scale, v[3] * m[3] * scale);
}
// block (t, c): token position t, sequences c, c + gridDim.y, ...; thread = (4 features f4, one of 16 sequence lanes)
__global__ __launch_bounds__(384) void embed_unmasked_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ series, const int* __restrict__ um,
                                                                  long S, int L, int Pu, float scale, float p, uint32_t lo, uint32_t hi, uint32_t site,
                                                                  float* __restrict__ dpos, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[24 * 52];
    const int t = blockIdx.x, f4 = threadIdx.x % 24, part = threadIdx.x / 24;
    const int tok = um[t];
    for (int i = threadIdx.x; i < 24 * 52; i += 384) red[i] = 0.f;
    __syncthreads();
    float aw[4][12], ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 12; ++k) aw[j][k] = 0.f;
    for (long s = (long)blockIdx.y * 16 + part; s < S; s += (long)gridDim.y * 16) {
        const long i4 = (s * Pu + t) * 24 + f4;
        const float4 g4 = ((const float4*)dx)[i4];
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (p > 0.f) embed_keep4(lo, hi, site, i4, p, m);
        const float g[4] = {g4.x * m[0] * scale, g4.y * m[1] * scale, g4.z * m[2] * scale, g4.w * m[3] * scale};
        const float4* pp = (const float4*)(series + s * L + (long)tok * 12);
        const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2];
        const float pv[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ab[j] += g[j];
#pragma unroll
            for (int k = 0; k < 12; ++k) aw[j][k] = __builtin_fmaf(g[j], pv[k], aw[j][k]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        atomicAdd(&red[f4 * 52 + j * 13 + 12], ab[j]);
#pragma unroll
        for (int k = 0; k < 12; ++k) atomicAdd(&red[f4 * 52 + j * 13 + k], aw[j][k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 24 * 52; i += 384) {
        const int f = (i / 52) * 4 + (i % 52) / 13, k = i % 13;
        const float v = red[i];
        if (k < 12) atomicAdd(&dw[f * 12 + k], v);
        else { atomicAdd(&db[f], v); atomicAdd(&dpos[(long)tok * 96 + f], v); }
    }
}

// ---------------------------------------------------------------------------------------------------------------- decoder input, backward
// d z[s][t] = sqrt(96) * d out[s][t] (t < Pu), and for the masked positions j the two parameter gradients straight from d out:
// d pos[midx[j]] += sum_s g, d mask_token += sum_{s, j} g with g = sqrt(96) * keep * d out[s][Pu + j] (the keep decisions of step_pt_dec_input:
// one Philox call per 4 consecutive elements of the [S, P, 96] tensor).  The [S, Pm, 96] scratch tensor of step_pt_dec_input_bwd and the two
// reduction passes over it (step_pt_sum_over_seq, step_colsum) are gone.  Block (t, c): token position t, sequences c, c + gridDim.y, ...
__global__ __launch_bounds__(384) void dec_input_bwd_sums_kernel(const float* __restrict__ dout, long S, int P, int Pu, float scale, float p, uint32_t lo,
                                                                  uint32_t hi, uint32_t site, const int* __restrict__ midx, float* __restrict__ dz,
                                                                  float* __restrict__ dpos, float* __restrict__ dmask) {
    __shared__ float red[96];
    const int t = blockIdx.x, f4 = threadIdx.x % 24, part = threadIdx.x / 24;
    if (t < Pu) {
        for (long s = (long)blockIdx.y * 16 + part; s < S; s += (long)gridDim.y * 16) {
            const float4 g = ((const float4*)dout)[(s * P + t) * 24 + f4];
            ((float4*)dz)[(s * Pu + t) * 24 + f4] = make_float4(g.x * scale, g.y * scale, g.z * scale, g.w * scale);
        }
        return;
    }
    if (threadIdx.x < 96) red[threadIdx.x] = 0.f;
    __syncthreads();
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (long s = (long)blockIdx.y * 16 + part; s < S; s += (long)gridDim.y * 16) {
        const long i4 = (s * P + t) * 24 + f4;
        const float4 g = ((const float4*)dout)[i4];
        float m[4] = {1.f, 1.f, 1.f, 1.f};
        if (p > 0.f) embed_keep4(lo, hi, site, i4, p, m);
        a[0] += g.x * m[0]; a[1] += g.y * m[1]; a[2] += g.z * m[2]; a[3] += g.w * m[3];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(&red[4 * f4 + j], a[j] * scale);
    __syncthreads();
    if (threadIdx.x < 96) {
        const float v = red[threadIdx.x];
        atomicAdd(&dpos[(long)midx[t - Pu] * 96 + threadIdx.x], v);
        atomicAdd(&dmask[threadIdx.x], v);
    }
}

int check_common(const char* who, const float* h1, long R, const void* pack, float p, const uint64_t* pool, long pool_words) {
    STEP_REQUIRE(h1 && pack, "%s: null input", who);
    STEP_REQUIRE(R > 0 && R < (1L << 36), "%s: bad row count %ld", who, R);
    STEP_REQUIRE(p >= 0.f && p < 1.f, "%s: bad dropout %f", who, p);
    if (p > 0.f) {
        STEP_REQUIRE(pool, "%s: dropout needs a keep-mask pool (step_dropout_pool_fill)", who);
        STEP_REQUIRE(pool_words >= 512 && (pool_words & (pool_words - 1)) == 0 && pool_words <= (1L << 31),
                     "%s: pool of %ld words is not a power of two in [512, 2^31]", who, pool_words);
    }
    return STEP_OK;
}

FfnArgs make_args(const float* h1, const float* df2, float* out, long R, const void* pack, const float* b1, float p, const uint64_t* pool,
                  long pool_words, uint64_t seed, uint32_t site, float* ws) {
    FfnArgs a;
    a.h1 = h1; a.df2 = df2; a.out = out; a.R = R; a.pack = (const char*)pack; a.b1 = b1;
    a.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    a.pool = p > 0.f ? (const unsigned long long*)pool : nullptr;
    a.pool_mask = p > 0.f ? (uint32_t)(pool_words - 1) : 0u;
    a.seed = (uint32_t)(seed ^ (seed >> 32)); a.site = site; a.ws = ws;
    return a;
}

constexpr int FR_WAVES = 8;

// the dynamic-LDS limit of a kernel is a per-(kernel, device) attribute: step_raise_lds_once (errors.cpp) sets it once per pair under a lock;
// `done` is the caller's lock-free fast path, valid for the device that set it (done = device index + 1)
template <typename K>
int raise_lds(K kernel, int bytes, int& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (__atomic_load_n(&done, __ATOMIC_ACQUIRE) == dev + 1) return STEP_OK;
    STEP_TRY(step_raise_lds_once((const void*)kernel, bytes, "pretrain_fused"));
    __atomic_store_n(&done, dev + 1, __ATOMIC_RELEASE);
    return STEP_OK;
}

bool ring_flags() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("STEP_FFN_RING_FLAGS"); on = e && e[0] == '1'; }
    return on == 1;
}
int fwd_waves() {
    static int n = 0;
    // (four since the end of round 6: with the attention and the encoder-size LayerNorm kernels shorter it shows in the step, 9.88 -> 9.80 ms
    //  -- profiles/r06_zo_C3_knobs.log; 8 = the one-workgroup-per-unit form)
    if (n == 0) { const char* e = getenv("STEP_FFN_FWD_WAVES"); n = e && atoi(e) == 8 ? 8 : 4; }
    return n;
}
template <bool LN>
int launch_rows_fwd4(const FfnArgs& a, hipStream_t st) {       // four waves per workgroup, two workgroups per compute unit (the default since round 6)
    static int raised[2] = {};
    const int lds = 2 * FF_BLOCK_F + 4 * STG_IN;
    const long npass = (a.R + 127) / 128;
    const int grid = (int)(npass < 512 ? npass : 512);
    if (a.pool) {
        STEP_TRY(raise_lds(ffn_rows_kernel<4, false, true, LN>, lds, raised[1]));
        ffn_rows_kernel<4, false, true, LN><<<grid, 256, lds, st>>>(a);
    } else {
        STEP_TRY(raise_lds(ffn_rows_kernel<4, false, false, LN>, lds, raised[0]));
        ffn_rows_kernel<4, false, false, LN><<<grid, 256, lds, st>>>(a);
    }
    return STEP_OK;
}
int launch_rows_ln(const FfnArgs& a, hipStream_t st) {
    static int raised[4] = {};
    if (fwd_waves() == 4) return launch_rows_fwd4<true>(a, st);
    const long npass = (a.R + 32 * FR_WAVES - 1) / (32 * FR_WAVES);
    const int grid = (int)(npass < 512 ? npass : 512);
    if (ring_flags()) {
        const int lds3 = 3 * FF_BLOCK_F + 64 + FR_WAVES * STG_WAVE;
        if (a.pool) {
            STEP_TRY(raise_lds(ffn_rows_kernel<FR_WAVES, false, true, true, 3, true>, lds3, raised[3]));
            ffn_rows_kernel<FR_WAVES, false, true, true, 3, true><<<grid, FR_WAVES * 64, lds3, st>>>(a);
        } else {
            STEP_TRY(raise_lds(ffn_rows_kernel<FR_WAVES, false, false, true, 3, true>, lds3, raised[2]));
            ffn_rows_kernel<FR_WAVES, false, false, true, 3, true><<<grid, FR_WAVES * 64, lds3, st>>>(a);
        }
        return STEP_OK;
    }
    const int lds = 2 * FF_BLOCK_F + FR_WAVES * STG_WAVE;
    if (a.pool) {
        STEP_TRY(raise_lds(ffn_rows_kernel<FR_WAVES, false, true, true>, lds, raised[1]));
        ffn_rows_kernel<FR_WAVES, false, true, true><<<grid, FR_WAVES * 64, lds, st>>>(a);
    } else {
        STEP_TRY(raise_lds(ffn_rows_kernel<FR_WAVES, false, false, true>, lds, raised[0]));
        ffn_rows_kernel<FR_WAVES, false, false, true><<<grid, FR_WAVES * 64, lds, st>>>(a);
    }
    return STEP_OK;
}

template <bool BWD>
int launch_rows(const FfnArgs& a, hipStream_t st) {
    static int raised[4] = {};
    if constexpr (!BWD) {
        if (fwd_waves() == 4) return launch_rows_fwd4<false>(a, st);
    }
    const int lds = 2 * (BWD ? FF_BLOCK_B : FF_BLOCK_F) + FR_WAVES * STG_WAVE;
    const long npass = (a.R + 32 * FR_WAVES - 1) / (32 * FR_WAVES);
    const int grid = (int)(npass < 512 ? npass : 512);
    if (ring_flags()) {                                 // forward: three slots; backward-data (37 KB blocks): two slots, waves at most one stage apart
        constexpr int NS = BWD ? 2 : 3;
        const int ldsf = NS * (BWD ? FF_BLOCK_B : FF_BLOCK_F) + 64 + FR_WAVES * STG_WAVE;
        if (a.pool) {
            STEP_TRY(raise_lds(ffn_rows_kernel<FR_WAVES, BWD, true, false, NS, true>, ldsf, raised[3]));
            ffn_rows_kernel<FR_WAVES, BWD, true, false, NS, true><<<grid, FR_WAVES * 64, ldsf, st>>>(a);
        } else {
            STEP_TRY(raise_lds(ffn_rows_kernel<FR_WAVES, BWD, false, false, NS, true>, ldsf, raised[2]));
            ffn_rows_kernel<FR_WAVES, BWD, false, false, NS, true><<<grid, FR_WAVES * 64, ldsf, st>>>(a);
        }
        return STEP_OK;
    }
    if (a.pool) {
        STEP_TRY(raise_lds(ffn_rows_kernel<FR_WAVES, BWD, true>, lds, raised[1]));
        ffn_rows_kernel<FR_WAVES, BWD, true><<<grid, FR_WAVES * 64, lds, st>>>(a);
    } else {
        STEP_TRY(raise_lds(ffn_rows_kernel<FR_WAVES, BWD, false>, lds, raised[0]));
        ffn_rows_kernel<FR_WAVES, BWD, false><<<grid, FR_WAVES * 64, lds, st>>>(a);
    }
    return STEP_OK;
}

template <bool W1K>
int launch_wgrad(const FfnArgs& a, int grid, hipStream_t st) {
    static int raised[2] = {};
    const int lds = 2 * FwLayout<W1K>::STAGE;
    if (a.pool) {
        STEP_TRY(raise_lds(ffn_wgrad_kernel<W1K, true>, lds, raised[1]));
        ffn_wgrad_kernel<W1K, true><<<grid, FW_WAVES * 64, lds, st>>>(a);
    } else {
        STEP_TRY(raise_lds(ffn_wgrad_kernel<W1K, false>, lds, raised[0]));
        ffn_wgrad_kernel<W1K, false><<<grid, FW_WAVES * 64, lds, st>>>(a);
    }
    return STEP_OK;
}

}  // namespace

extern "C" long step_pt_ffn_pack_bytes(void) { return FF_PACK_BYTES; }

extern "C" long step_pt_ffn_wgrad_workgroups(long R) {
    const long nstage = ((R + 31) / 32 + FW_TILES - 1) / FW_TILES;
    return nstage < 256 ? nstage : 256;
}
extern "C" long step_pt_ffn_wgrad_ws_floats(long R) { return (long)step_pt_ffn_wgrad_workgroups(R) * (96 * 384 + 384 * 96 + 384); }

extern "C" int step_pt_ffn_pack(const float* w1, const float* b1, const float* w2, const float* b2, void* pack, void* stream) {
    STEP_REQUIRE(w1 && b1 && w2 && b2 && pack, "pt_ffn_pack: null argument");
    STEP_REQUIRE(((uintptr_t)pack & 15) == 0, "pt_ffn_pack: the fragment buffer must be 16-byte aligned");
    ffn_pack_kernel<<<cdiv(FFN_PACK_THREADS, 256), 256, 0, (hipStream_t)stream>>>(w1, b1, w2, b2, (char*)pack);
    STEP_LAUNCH_CHECK("step_pt_ffn_pack");
    return STEP_OK;
}

extern "C" int step_pt_ffn_fused_fwd(const float* h1, long R, const void* pack, float p, const uint64_t* pool, long pool_words, uint64_t seed,
                                     uint32_t site, float* f2, void* stream) {
    STEP_TRY(check_common("pt_ffn_fused_fwd", h1, R, pack, p, pool, pool_words));
    STEP_REQUIRE(f2, "pt_ffn_fused_fwd: null output");
    STEP_TRY(launch_rows<false>(make_args(h1, nullptr, f2, R, pack, nullptr, p, pool, pool_words, seed, site, nullptr), (hipStream_t)stream));
    STEP_LAUNCH_CHECK("step_pt_ffn_fused_fwd");
    return STEP_OK;
}

extern "C" int step_pt_ffn_fused_bwd_data(const float* df2, const float* h1, long R, const void* pack, float p, const uint64_t* pool, long pool_words,
                                          uint64_t seed, uint32_t site, float* dh1, void* stream) {
    STEP_TRY(check_common("pt_ffn_fused_bwd_data", h1, R, pack, p, pool, pool_words));
    STEP_REQUIRE(df2 && dh1, "pt_ffn_fused_bwd_data: null argument");
    STEP_TRY(launch_rows<true>(make_args(h1, df2, dh1, R, pack, nullptr, p, pool, pool_words, seed, site, nullptr), (hipStream_t)stream));
    STEP_LAUNCH_CHECK("step_pt_ffn_fused_bwd_data");
    return STEP_OK;
}

extern "C" int step_pt_ffn_fused_bwd_weights(const float* df2, const float* h1, long R, const void* pack, const float* b1, float p, const uint64_t* pool,
                                             long pool_words, uint64_t seed, uint32_t site, float* ws, float* dw1, float* db1, float* dw2, void* stream) {
    STEP_TRY(check_common("pt_ffn_fused_bwd_weights", h1, R, pack, p, pool, pool_words));
    STEP_REQUIRE(df2 && b1 && ws && dw1 && db1 && dw2, "pt_ffn_fused_bwd_weights: null argument");
    const hipStream_t st = (hipStream_t)stream;
    const int grid = (int)step_pt_ffn_wgrad_workgroups(R);
    float* ws2 = ws;
    float* ws1 = ws + (long)grid * 96 * 384;
    STEP_TRY(launch_wgrad<false>(make_args(h1, df2, nullptr, R, pack, b1, p, pool, pool_words, seed, site, ws2), grid, st));
    STEP_LAUNCH_CHECK("step_pt_ffn_fused_bwd_weights (W2)");
    STEP_TRY(launch_wgrad<true>(make_args(h1, df2, nullptr, R, pack, b1, p, pool, pool_words, seed, site, ws1), grid, st));
    STEP_LAUNCH_CHECK("step_pt_ffn_fused_bwd_weights (W1)");
    const float ks = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    ffn_reduce_kernel<<<dim3(cdiv(96 * 384, 256), 8), 256, 0, st>>>(ws2, grid, 96 * 384, 96 * 384, ks, dw2, 96 * 384, nullptr);
    ffn_reduce_kernel<<<dim3(cdiv(384 * 96 + 384, 256), 8), 256, 0, st>>>(ws1, grid, 384 * 96 + 384, 384 * 96 + 384, ks, dw1, 384 * 96, db1);
    STEP_LAUNCH_CHECK("step_pt_ffn_fused_bwd_weights (reduce)");
    return STEP_OK;
}

extern "C" long step_pt_rows_linear_pack_bytes(int nkc, int nog) { return (long)nog * 3 * nkc * 6 * FF_FRAG + (long)nog * 96 * 4; }

extern "C" int step_pt_rows_linear_pack(const float* w, long swo, long swi, int nkc, int nog, const float* bias, void* pack, void* stream) {
    STEP_REQUIRE(w && pack && nkc >= 1 && nog >= 1 && nkc * nog <= 3, "pt_rows_linear_pack: bad arguments (blocks of 96: %d in x %d out)", nkc, nog);
    STEP_REQUIRE(((uintptr_t)pack & 15) == 0, "pt_rows_linear_pack: the fragment buffer must be 16-byte aligned");
    const int threads = nog * 3 * nkc * 6 * 64 + nog * 96;
    lin_pack_kernel<<<cdiv(threads, 256), 256, 0, (hipStream_t)stream>>>(w, swo, swi, nkc, nog, bias, (char*)pack);
    STEP_LAUNCH_CHECK("step_pt_rows_linear_pack");
    return STEP_OK;
}

namespace {
template <int NKC, int NOG, bool IN_BF16, bool OUT_BF16, bool ACCUM, bool LN, int NW>
int launch_lin_nw(const LinArgs& a, hipStream_t st) {
    static int raised = 0;
    const int wbytes = NOG * 3 * NKC * 6 * FF_FRAG + NOG * 96 * 4;
    const int lds = ((wbytes + 1023) & ~1023) + NW * STG_IN;
    STEP_TRY(raise_lds(rows_linear_kernel<NKC, NOG, IN_BF16, OUT_BF16, ACCUM, LN, NW>, lds, raised));
    const long ntile = (a.R + 31) / 32, nwg = (ntile + NW - 1) / NW;
    rows_linear_kernel<NKC, NOG, IN_BF16, OUT_BF16, ACCUM, LN, NW><<<(int)(nwg < 256 ? nwg : 256), NW * 64, lds, st>>>(a);
    return STEP_OK;
}
template <int NKC, int NOG, bool IN_BF16, bool OUT_BF16, bool ACCUM, bool LN = false>
int launch_lin(const LinArgs& a, hipStream_t st) {
    static const int nw = [] { const char* e = getenv("STEP_PT_LIN_WAVES"); return e ? atoi(e) : 12; }();      // (A/B measurements: 8 = the first version's occupancy)
    if constexpr (LN) return launch_lin_nw<NKC, NOG, IN_BF16, OUT_BF16, ACCUM, LN, 8>(a, st);
    else {
        if (nw == 8) return launch_lin_nw<NKC, NOG, IN_BF16, OUT_BF16, ACCUM, LN, 8>(a, st);
        return launch_lin_nw<NKC, NOG, IN_BF16, OUT_BF16, ACCUM, LN, 12>(a, st);
    }
}
}  // namespace

// y[R, 96 nog] (accumulate ? += : =) x[R, 96 nkc] . M^T + bias, M and bias as packed by step_pt_rows_linear_pack.  Supported forms:
// (nkc 1, nog 1 | 3, x f32, y bf16), (1, 1, x bf16, y f32), (3, 1, x bf16, y f32, accumulate 0 | 1)
extern "C" int step_pt_rows_linear(const void* x, int x_bf16, long R, const void* pack, int nkc, int nog, void* y, int y_bf16, int accumulate,
                                   void* stream) {
    STEP_REQUIRE(x && pack && y && R > 0 && R < (1L << 36), "pt_rows_linear: bad arguments");
    STEP_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)pack) & 15) == 0, "pt_rows_linear: 16-byte aligned tensors expected");
    LinArgs a;
    a.x = x; a.y = y; a.R = R; a.ldx = 96 * nkc; a.ldy = 96 * nog; a.pack = (const char*)pack;
    const hipStream_t st = (hipStream_t)stream;
    int rc = -1;
    if (nkc == 1 && nog == 3 && !x_bf16 && y_bf16 && !accumulate) rc = launch_lin<1, 3, false, true, false>(a, st);
    else if (nkc == 1 && nog == 1 && !x_bf16 && y_bf16 && !accumulate) rc = launch_lin<1, 1, false, true, false>(a, st);
    else if (nkc == 1 && nog == 1 && x_bf16 && !y_bf16 && !accumulate) rc = launch_lin<1, 1, true, false, false>(a, st);
    else if (nkc == 3 && nog == 1 && x_bf16 && !y_bf16) rc = accumulate ? launch_lin<3, 1, true, false, true>(a, st) : launch_lin<3, 1, true, false, false>(a, st);
    STEP_REQUIRE(rc != -1, "pt_rows_linear: unsupported form (%d x 96 %s in, %d x 96 %s out, accumulate %d)", nkc, x_bf16 ? "bf16" : "f32", nog,
                 y_bf16 ? "bf16" : "f32", accumulate);
    STEP_TRY(rc);
    STEP_LAUNCH_CHECK("step_pt_rows_linear");
    return STEP_OK;
}

// all fragment buffers of one transformer layer in one launch (what five step_pt_ffn_pack / step_pt_rows_linear_pack calls write):
// ffn (step_pt_ffn_pack_bytes), qkv (1, 3), o (1, 1), da (1, 1: Wo transposed), dx (3, 1: Wi transposed)
extern "C" int step_pt_layer_pack(const float* wi, const float* bi, const float* wo, const float* bo, const float* w1, const float* b1, const float* w2,
                                  const float* b2, void* ffn, void* qkv, void* o, void* da, void* dx, void* stream) {
    STEP_REQUIRE(wi && bi && wo && bo && w1 && b1 && w2 && b2 && ffn && qkv && o && da && dx, "pt_layer_pack: null argument");
    STEP_REQUIRE((((uintptr_t)ffn | (uintptr_t)qkv | (uintptr_t)o | (uintptr_t)da | (uintptr_t)dx) & 15) == 0, "pt_layer_pack: 16-byte aligned buffers");
    LayerPackArgs a = {wi, bi, wo, bo, w1, b1, w2, b2, (char*)ffn, (char*)qkv, (char*)o, (char*)da, (char*)dx};
    layer_pack_kernel<<<LP_B4, 256, 0, (hipStream_t)stream>>>(a);
    STEP_LAUNCH_CHECK("step_pt_layer_pack");
    return STEP_OK;
}

extern "C" long step_pt_proj_wgrad_ws_floats(long R) {
    const long ntile = (R + 31) / 32;
    return (ntile < 256 ? ntile : 256) * PW_WS;
}
// d Wi [288, 96] += d qkv^T x, d bi [288] += column sums of d qkv, d Wo [96, 96] += d o^T a; x, d o f32 [R, 96], d qkv bf16 [R, 288], a bf16 [R, 96]
extern "C" int step_pt_proj_wgrad(const float* x, const uint16_t* dqkv, const float* dov, const uint16_t* a, long R, float* ws, float* dwi, float* dbi,
                                  float* dwo, void* stream) {
    STEP_REQUIRE(x && dqkv && dov && a && ws && dwi && dbi && dwo && R > 0 && R < (1L << 36), "pt_proj_wgrad: bad arguments");
    STEP_REQUIRE((((uintptr_t)x | (uintptr_t)dqkv | (uintptr_t)dov | (uintptr_t)a) & 15) == 0, "pt_proj_wgrad: 16-byte aligned tensors expected");
    static int raised = 0;
    STEP_TRY(raise_lds(proj_wgrad_kernel, 2 * PW_TILE, raised));
    const hipStream_t st = (hipStream_t)stream;
    const long ntile = (R + 31) / 32;
    const int grid = (int)(ntile < 256 ? ntile : 256);
    ProjWgradArgs pa = {x, dqkv, dov, a, R, ws};
    proj_wgrad_kernel<<<grid, FW_WAVES * 64, 2 * PW_TILE, st>>>(pa);
    STEP_LAUNCH_CHECK("step_pt_proj_wgrad");
    ffn_reduce_kernel<<<dim3(cdiv(288 * 96 + 288, 256), 8), 256, 0, st>>>(ws, grid, 288 * 96 + 288, PW_WS, 1.f, dwi, 288 * 96, dbi);
    ffn_reduce_kernel<<<dim3(cdiv(96 * 96, 256), 8), 256, 0, st>>>(ws + 288 * 96 + 288, grid, 96 * 96, PW_WS, 1.f, dwo, 96 * 96, nullptr);
    STEP_LAUNCH_CHECK("step_pt_proj_wgrad (reduce)");
    return STEP_OK;
}

namespace {
int make_ln(LnEpi& e, const char* who, const float* res, float* pre, float* y, float* stats, const float* gamma, const float* beta, float p, uint64_t seed,
            uint32_t site) {
    STEP_REQUIRE(res && y && stats && gamma && beta, "%s: null LayerNorm argument", who);
    STEP_REQUIRE((((uintptr_t)res | (uintptr_t)pre | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0 && (((uintptr_t)stats) & 7) == 0,
                 "%s: 16-byte aligned row tensors expected", who);
    e.res = res; e.pre = pre; e.y = y; e.stats = stats; e.gamma = gamma; e.beta = beta; e.p = p; e.lo = (uint32_t)seed; e.hi = (uint32_t)(seed >> 32); e.site = site;
    return STEP_OK;
}
}  // namespace

// step_pt_ffn_fused_fwd followed by step_pt_add_layernorm_fwd(h1, f2, ..., site_out) without the f2 round trip through HBM:
// pre [R, 96] (nullable) = h1 + dropout(f2), y = LayerNorm(pre) * gamma + beta, stats [R, 2] = (mean, rstd)
extern "C" int step_pt_ffn_fused_fwd_ln(const float* h1, long R, const void* pack, float p, const uint64_t* pool, long pool_words, uint64_t seed,
                                        uint32_t site_hidden, uint32_t site_out, const float* gamma, const float* beta, float* pre, float* y, float* stats,
                                        void* stream) {
    STEP_TRY(check_common("pt_ffn_fused_fwd_ln", h1, R, pack, p, pool, pool_words));
    FfnArgs a = make_args(h1, nullptr, nullptr, R, pack, nullptr, p, pool, pool_words, seed, site_hidden, nullptr);
    STEP_TRY(make_ln(a.ln, "pt_ffn_fused_fwd_ln", h1, pre, y, stats, gamma, beta, p, seed, site_out));
    STEP_TRY(launch_rows_ln(a, (hipStream_t)stream));
    STEP_LAUNCH_CHECK("step_pt_ffn_fused_fwd_ln");
    return STEP_OK;
}

// step_pt_rows_linear (bf16 in, one block of 96, f32 out) followed by step_pt_add_layernorm_fwd(res, o, ..., site): the out-projection of the attention
// with its residual add, dropout and LayerNorm as the output stage
extern "C" int step_pt_rows_linear_ln(const uint16_t* x, long R, const void* pack, const float* res, float p, uint64_t seed, uint32_t site, const float* gamma,
                                      const float* beta, float* pre, float* y, float* stats, void* stream) {
    STEP_REQUIRE(x && pack && R > 0 && R < (1L << 36) && p >= 0.f && p < 1.f, "pt_rows_linear_ln: bad arguments");
    STEP_REQUIRE((((uintptr_t)x | (uintptr_t)pack) & 15) == 0, "pt_rows_linear_ln: 16-byte aligned tensors expected");
    LinArgs a;
    a.x = x; a.y = nullptr; a.R = R; a.ldx = 96; a.ldy = 96; a.pack = (const char*)pack;
    STEP_TRY(make_ln(a.ln, "pt_rows_linear_ln", res, pre, y, stats, gamma, beta, p, seed, site));
    STEP_TRY((launch_lin<1, 1, true, false, false, true>(a, (hipStream_t)stream)));
    STEP_LAUNCH_CHECK("step_pt_rows_linear_ln");
    return STEP_OK;
}

// x [S, Pu, 96] = sqrt(96) * dropout(patch embedding + positional embedding) of the unmasked tokens um [Pu] (int32, device) of series [S, L] (L a multiple of 12)
extern "C" int step_pt_embed_unmasked_fwd(const float* series, const int* um, const float* w, const float* b, const float* pos, long S, int L, int Pu, float p,
                                          uint64_t seed, uint32_t site, float* x, void* stream) {
    STEP_REQUIRE(series && um && w && b && pos && x && S > 0 && L > 0 && L % 12 == 0 && Pu > 0 && Pu <= L / 12 && p >= 0.f && p < 1.f,
                 "pt_embed_unmasked_fwd: bad arguments");
    STEP_REQUIRE((((uintptr_t)series | (uintptr_t)w | (uintptr_t)b | (uintptr_t)pos | (uintptr_t)x) & 15) == 0, "pt_embed_unmasked_fwd: 16-byte aligned tensors expected");
    embed_unmasked_fwd_kernel<<<cdiv(S * Pu * 24, 256), 256, 0, (hipStream_t)stream>>>(series, um, w, b, pos, S, L, Pu, 9.797958971132712f, p, (uint32_t)seed,
                                                                                      (uint32_t)(seed >> 32), site, x);
    STEP_LAUNCH_CHECK("step_pt_embed_unmasked_fwd");
    return STEP_OK;
}
// from d x [S, Pu, 96]: dpos[um[t]] += sum_s g, dw [96, 12] += sum g patch^T, db [96] += sum g with g = sqrt(96) * keep * d x (same seed / site as the forward)
extern "C" int step_pt_embed_unmasked_bwd(const float* dx, const float* series, const int* um, long S, int L, int Pu, float p, uint64_t seed, uint32_t site,
                                          float* dpos, float* dw, float* db, void* stream) {
    STEP_REQUIRE(dx && series && um && dpos && dw && db && S > 0 && L > 0 && L % 12 == 0 && Pu > 0 && Pu <= L / 12 && p >= 0.f && p < 1.f,
                 "pt_embed_unmasked_bwd: bad arguments");
    STEP_REQUIRE((((uintptr_t)series | (uintptr_t)dx) & 15) == 0, "pt_embed_unmasked_bwd: 16-byte aligned tensors expected");
    const int ny = (int)(S >= 16 * 16 ? 16 : (S + 15) / 16);
    embed_unmasked_bwd_kernel<<<dim3(Pu, ny), 384, 0, (hipStream_t)stream>>>(dx, series, um, S, L, Pu, 9.797958971132712f, p, (uint32_t)seed, (uint32_t)(seed >> 32),
                                                                            site, dpos, dw, db);
    STEP_LAUNCH_CHECK("step_pt_embed_unmasked_bwd");
    return STEP_OK;
}

// step_pt_dec_input_bwd + step_pt_sum_over_seq(midx) + step_colsum in one pass over d out [S, P, 96]: dz [S, Pu, 96] written,
// dpos [*, 96] rows midx[j] and dmask [96] accumulated
extern "C" int step_pt_dec_input_bwd_sums(const float* dout, long S, int P, int Pu, float p, uint64_t seed, uint32_t site, const int* midx, float* dz,
                                          float* dpos, float* dmask, void* stream) {
    STEP_REQUIRE(dout && midx && dz && dpos && dmask && S > 0 && P > Pu && Pu > 0 && p >= 0.f && p < 1.f, "pt_dec_input_bwd_sums: bad arguments");
    STEP_REQUIRE((((uintptr_t)dout | (uintptr_t)dz) & 15) == 0, "pt_dec_input_bwd_sums: 16-byte aligned tensors expected");
    const int ny = (int)(S >= 16 * 16 ? 16 : (S + 15) / 16);
    dec_input_bwd_sums_kernel<<<dim3(P, ny), 384, 0, (hipStream_t)stream>>>(dout, S, P, Pu, 9.797958971132712f, p, (uint32_t)seed, (uint32_t)(seed >> 32), site,
                                                                           midx, dz, dpos, dmask);
    STEP_LAUNCH_CHECK("step_pt_dec_input_bwd_sums");
    return STEP_OK;
}
