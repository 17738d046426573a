// Slice-resident gcn layer of the GraphWaveNet backbone (bf16 mode, graphs of up to SLICE_MAX_N nodes): included by gwnet.hip.
//
// One layer of model.py:183-213 is, per (sample b, output time t), a chain over ALL nodes of that slice:
//     z = tanh(conv_f) * sigmoid(conv_g)            gated TCN, per node                          (model.py:183-189)
//     x1_s = P_s^T z,  x2_s = P_s^T x1_s            two diffusion hops for each of the 3 supports (model.py:10-16,35-45)
//     y = W_mix [z, x1_0, x2_0, ...] + b, dropout, + residual, BatchNorm sums                   (model.py:46-47,206-212)
// As four launches (tcn_fwd, two batched hop GEMMs, mix_fwd) every link of that chain is a kernel boundary plus a round trip of a
// [B N T 32] tensor through L2 -- about 55 us per layer at PEMS04 for ~3 us of arithmetic, eight layers deep, and the same again
// (mix_bwd, two adjoint hops) in the backward.  Here ONE workgroup owns a whole (b, t) slice: the [N, 32] slice lives in LDS as
// bf16 [32][N] (channel-major: a row is the k-contiguous A operand of the hop product), each wave owns one or two 32-node tiles,
// and every product has the shape  D[m = channel][n = node]  so that a lane is a node and an accumulator tile is, by a plain
// f32 -> bf16 pack, the B operand of the next product (k order = chan8h, as everywhere in gwnet.hip):
//     hop:  D[c][w] = sum_v XT[c][v] * Pst[w][v]      A from LDS (ds_read_b128), B straight from the k-contiguous bf16 support stack
//     mix:  Y[o][w] += sum_c Wmix[o][slot c] * X[c][w]  A = pre-packed weight fragment, B = the hop's accumulator tile
// The three supports run side by side (three independent accumulators / operand streams per node tile), operand loads are issued
// four k-steps ahead in two alternating register batches.  What the backward needs (tf, sg, all 7 slots of cat, y, the dropout
// mask, zlast) is written exactly as the four-kernel path writes it, so the backward, the weight-gradient leaves and the tests
// see the same buffers.  The backward's middle -- BatchNorm backward + dropout + mix data gradient + both adjoint hops -- is the
// mirror image (gcn_bwd_slice_kernel); tcn_bwd stays its own launch (its transposed im2col crosses time steps).
//
// Numerics: identical operand roundings to the four-kernel bf16 path (f32 accumulators are rounded to bf16 exactly where that path
// re-reads an f32 tensor as a matrix-core operand); only the f32 summation order of the contractions differs.
#pragma once

constexpr int SLICE_MAX_N = 384;          // LDS: 4 slices of 32 x (N + 8) bf16 + 8 KB (forward), 6 slices (backward: 150.5 KB at 384 nodes);
                                          // every dataset of the reference but PEMS07 (883 nodes, four-kernel path) is below it
constexpr int SL_TCN_FRAGS = 8, SL_MIX_FRAGS = 14;
constexpr int SL_FRAGS_PER_LAYER = SL_TCN_FRAGS + 2 * SL_MIX_FRAGS;      // [tcn 8][mix forward 14][mix backward 14] fragments of 64 lanes x 8 bf16

__device__ __forceinline__ int sl_crow(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }      // accumulator register e of lane half h <-> row
__device__ __forceinline__ bf16x8 sl_frag(const uint16_t* frags, int f, int lane) { return *(const bf16x8*)(frags + ((long)f * 64 + lane) * 8); }
__device__ __forceinline__ bf16x8 sl_pack(const f32x16& v, int q) {
    float t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = v[8 * q + j];
    return pack8(t);
}
__device__ __forceinline__ f32x16 sl_zero() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}
// 4 consecutive channels of one position (16 bytes): accumulator registers 4 g .. 4 g + 3 of lane half h are channels 8 g + 4 h ..
__device__ __forceinline__ void sl_store_row(float* row, const f32x16& v, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *(float4*)(row + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}
// the tile as bf16 into the channel-major LDS slice XT[c][w] (zero for nodes beyond N)
__device__ __forceinline__ void sl_to_lds(uint16_t* XT, int LDP, int w, bool ok, const f32x16& v, int h) {
#pragma unroll
    for (int e = 0; e < 16; ++e) XT[sl_crow(e, h) * LDP + w] = ok ? (uint16_t)f32_to_bf16_bits(v[e]) : (uint16_t)0;
}

// Pre-packed weight fragments of all 8 layers, one launch (blocks = 8 layers x SL_FRAGS_PER_LAYER, 64 threads):
//   tcn  f q (0..3), g 4 + q:   A(m = o, k) = Wcat[o (+32)][16 q + chan(h, j)]            (wcat  [64 out][64 in], in = tap * 32 + c)
//   mix forward  2 slot + q:    A(m = o, k) = Wmix[o][32 slot + 16 q + chan(h, j)]        (Wmix  [32][224])
//   mix backward 2 slot + q:    A(m = c, k) = Wmix[16 q + chan(h, j)][32 slot + c]        (wmixT [224][32])
// with chan(h, j) = 4 h + (j & 3) + 8 (j >> 2), the k order every register-fed tile of gwnet.hip uses.
struct MixPtrs8 { const float* w[NL]; };
__global__ __launch_bounds__(64) void slice_pack_frags_kernel(const float* __restrict__ wcat_all, MixPtrs8 M, uint16_t* __restrict__ frags) {
    const int layer = blockIdx.x / SL_FRAGS_PER_LAYER, f = blockIdx.x % SL_FRAGS_PER_LAYER;
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    float v[8];
    if (f < SL_TCN_FRAGS) {
        const int q = f & 3, part = f >> 2;
        load8h(wcat_all + layer * 4096 + (32 * part + r) * 64 + 16 * q, h, v);
    } else {
        const bool bwd = f >= SL_TCN_FRAGS + SL_MIX_FRAGS;
        const int g = f - SL_TCN_FRAGS - (bwd ? SL_MIX_FRAGS : 0), slot = g >> 1, q = g & 1;
        const float* w = M.w[layer];          // (layer 7's gcn is only evaluated for its dead BatchNorm statistics)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * q + 4 * h + (j & 3) + 8 * (j >> 2);
            v[j] = bwd ? w[kk * CAT + 32 * slot + r] : w[r * CAT + 32 * slot + kk];
        }
    }
    *(bf16x8*)(frags + ((long)blockIdx.x * 64 + lane) * 8) = pack8(v);
}

// Three supports side by side over the k range [0, nks) of 16-node steps: acc[s] += XT_s (LDS, A) x Pst_s rows (global bf16, B).
// NA = 1: the supports share one A slice (the first hop of the forward); 3: one slice each.  SUM: all three accumulate into acc[0]
// (the backward's d_z).  prow[s]: this lane's row of support s (k contiguous, zero padded to N8); operands beyond N8 are zeros.
template <int NA, bool SUM>
__device__ __forceinline__ void sl_hop3(f32x16 (&acc)[3], const uint16_t* xa0, const uint16_t* xa1, const uint16_t* xa2, const uint16_t* p0,
                                        const uint16_t* p1, const uint16_t* p2, int nks, int N8, int LDP, int lane) {
    const int c = lane & 31, h = lane >> 5;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 ba[3][4], bb[3][4];
    auto issue = [&](bf16x8 (&buf)[3][4], int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = 16 * (k0 + i) + 8 * h;
            const bool ok = k0 + i < nks && off < N8;
            buf[0][i] = ok ? *(const bf16x8*)(p0 + off) : zero8;
            buf[1][i] = ok ? *(const bf16x8*)(p1 + off) : zero8;
            buf[2][i] = ok ? *(const bf16x8*)(p2 + off) : zero8;
        }
    };
    auto compute = [&](const bf16x8 (&buf)[3][4], int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (k0 + i < nks) {
                const int off = c * LDP + 16 * (k0 + i) + 8 * h;
                const bf16x8 a0 = *(const bf16x8*)(xa0 + off);
                const bf16x8 a1 = NA == 1 ? a0 : *(const bf16x8*)(xa1 + off);
                const bf16x8 a2 = NA == 1 ? a0 : *(const bf16x8*)(xa2 + off);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, buf[0][i], acc[0], 0, 0, 0);
                acc[SUM ? 0 : 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, buf[1][i], acc[SUM ? 0 : 1], 0, 0, 0);
                acc[SUM ? 0 : 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, buf[2][i], acc[SUM ? 0 : 2], 0, 0, 0);
            }
        }
    };
    issue(ba, 0);
#pragma unroll 1
    for (int k0 = 0; k0 < nks; k0 += 8) {
        issue(bb, k0 + 4);
        compute(ba, k0);
        issue(ba, k0 + 8);
        compute(bb, k0 + 4);
    }
}

struct SliceFwdArgs {
    const float* src; BnFwd bn;                    // layer input (x0 or y[i-1]) and the BatchNorm in front of it (stat == nullptr: layer 0)
    int B, N, Tin, Tout, dil, layer;
    const uint16_t* frags;                         // this layer's SL_FRAGS_PER_LAYER fragments
    const float* bcat;                             // [64] filter | gate bias
    const float* mix_bias;                         // [32]
    const uint16_t* PT16;                          // [3][B][N][N8] bf16, rows = receiving node w, k = v
    float *tf, *sg, *cat, *zlast;
    float drop_p; uint32_t seed_lo, seed_hi;
    float *mask, *y; double* sums;
};

// grid = B * Tout workgroups of (node tiles / TPW) waves; dynamic LDS = 4 slices of 32 x LDP bf16 + 32 x 64 floats
template <int TPW>
__global__ __launch_bounds__(512) void gcn_fwd_slice_kernel(SliceFwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) char sl_smem[];
    __shared__ float st[128];
    __shared__ float bsum[64];
    if (A.bn.stat) bn_fwd_stats(A.bn, st);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, col = lane & 31, h = lane >> 5;
    const int N = A.N, NT = (N + 31) >> 5, NP = NT * 32, LDP = NP + 8, N8 = (N + 7) & ~7, nks = NP >> 4;
    const int b = blockIdx.x / A.Tout, t = blockIdx.x - b * A.Tout;
    uint16_t* zT = (uint16_t*)sl_smem;
    uint16_t* x1T[3] = {zT + 32 * LDP, zT + 2 * 32 * LDP, zT + 3 * 32 * LDP};
    float* red = (float*)(sl_smem + 4L * 32 * LDP * 2);          // [32 values][64 lanes], all waves add into it
    for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x) red[i] = 0.f;          // (ordered before its first use by the barriers below)
    const uint16_t* F = A.frags;
    f32x16 ymix[TPW];
    // ---------------------------------------------------------------- gated TCN of this slice, z into slot 0 / LDS / the mix
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + i * nw;
        ymix[i] = sl_zero();
        if (tile >= NT) continue;
        const int w = tile * 32 + col;
        const bool ok = w < N;
        const long bnw = (long)b * N + (ok ? w : N - 1);
        const long xrow = bnw * A.Tin + t, p = bnw * A.Tout + t;
        bf16x8 xb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a[8];
            load8h(A.src + (xrow + (q >> 1) * A.dil) * C + 16 * (q & 1), h, a);
            if (A.bn.stat) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int c = chan8h(q & 1, h, j); a[j] = a[j] * st[c] + st[32 + c]; }
            }
            xb[q] = pack8(a);
        }
        f32x16 af = sl_zero(), ag = sl_zero();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            af = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, q, lane), xb[q], af, 0, 0, 0);
            ag = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, 4 + q, lane), xb[q], ag, 0, 0, 0);
        }
        f32x16 z;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = sl_crow(e, h);
            const float f = tanhf(af[e] + A.bcat[c]);
            const float g = 1.f / (1.f + __expf(-(ag[e] + A.bcat[32 + c])));
            af[e] = f; ag[e] = g; z[e] = f * g;
        }
        if (ok) {
            sl_store_row(A.tf + p * C, af, h);
            sl_store_row(A.sg + p * C, ag, h);
            sl_store_row(A.cat + p * CAT, z, h);
            if (t == A.Tout - 1) sl_store_row(A.zlast + bnw * CS + A.layer * C, z, h);
        }
        sl_to_lds(zT, LDP, w, ok, z, h);
        ymix[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, SL_TCN_FRAGS + 0, lane), sl_pack(z, 0), ymix[i], 0, 0, 0);
        ymix[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, SL_TCN_FRAGS + 1, lane), sl_pack(z, 1), ymix[i], 0, 0, 0);
    }
    __syncthreads();
    // ---------------------------------------------------------------- first hop of the three supports: slots 1, 3, 5
    const long prow_b = (long)b * N;
    const long sstride = (long)A.B * N * N8;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + i * nw;
        if (tile >= NT) continue;
        const int w = tile * 32 + col;
        const bool ok = w < N;
        const long bnw = (long)b * N + (ok ? w : N - 1);
        const long p = bnw * A.Tout + t;
        const uint16_t* pr = A.PT16 + (prow_b + (ok ? w : N - 1)) * N8;
        f32x16 acc[3] = {sl_zero(), sl_zero(), sl_zero()};
        sl_hop3<1, false>(acc, zT, zT, zT, pr, pr + sstride, pr + 2 * sstride, nks, N8, LDP, lane);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (ok) sl_store_row(A.cat + p * CAT + (1 + 2 * s) * C, acc[s], h);
            sl_to_lds(x1T[s], LDP, w, ok, acc[s], h);
            ymix[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, SL_TCN_FRAGS + 2 * (1 + 2 * s), lane), sl_pack(acc[s], 0), ymix[i], 0, 0, 0);
            ymix[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, SL_TCN_FRAGS + 2 * (1 + 2 * s) + 1, lane), sl_pack(acc[s], 1), ymix[i], 0, 0, 0);
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- second hop: slots 2, 4, 6; then the layer's epilogue
    float s1[16], s2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const float keep_scale = A.drop_p > 0.f ? 1.f / (1.f - A.drop_p) : 1.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + i * nw;
        if (tile >= NT) continue;
        const int w = tile * 32 + col;
        const bool ok = w < N;
        const long bnw = (long)b * N + (ok ? w : N - 1);
        const long p = bnw * A.Tout + t;
        const uint16_t* pr = A.PT16 + (prow_b + (ok ? w : N - 1)) * N8;
        f32x16 acc[3] = {sl_zero(), sl_zero(), sl_zero()};
        sl_hop3<3, false>(acc, x1T[0], x1T[1], x1T[2], pr, pr + sstride, pr + 2 * sstride, nks, N8, LDP, lane);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (ok) sl_store_row(A.cat + p * CAT + (2 + 2 * s) * C, acc[s], h);
            ymix[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, SL_TCN_FRAGS + 2 * (2 + 2 * s), lane), sl_pack(acc[s], 0), ymix[i], 0, 0, 0);
            ymix[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, SL_TCN_FRAGS + 2 * (2 + 2 * s) + 1, lane), sl_pack(acc[s], 1), ymix[i], 0, 0, 0);
        }
        // y = (mix + bias) * dropout + residual (model.py:46-47,206-209), BatchNorm sums
        const float* rsrc = A.src + (bnw * A.Tin + t + A.dil) * C;
        f32x16 res, yv, mk;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 r4 = *(const float4*)(rsrc + 8 * g + 4 * h);
            res[4 * g] = r4.x; res[4 * g + 1] = r4.y; res[4 * g + 2] = r4.z; res[4 * g + 3] = r4.w;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int o = sl_crow(e, h);
            float m = 1.f;
            if (A.drop_p > 0.f) {
                // the keep decision of element (p, o) exactly as mix_fwd_kernel draws it: word p & 3 of the Philox call of the aligned group of four positions
                const long i0 = (p & ~3L) * C + o;
                uint32_t r[4];
                philox4x32((uint32_t)i0, (uint32_t)(i0 >> 32), (uint32_t)A.layer, 0xD409u, A.seed_lo, A.seed_hi, r);
                const uint32_t pick = (p & 3) == 0 ? r[0] : (p & 3) == 1 ? r[1] : (p & 3) == 2 ? r[2] : r[3];
                m = u32_to_unit(pick) >= A.drop_p ? keep_scale : 0.f;
            }
            mk[e] = m;
            const float xs = A.bn.stat ? st[o] : 1.f, xh = A.bn.stat ? st[32 + o] : 0.f;
            const float v = (ymix[i][e] + A.mix_bias[o]) * m + (res[e] * xs + xh);
            yv[e] = v;
            if (ok) { s1[e] += v; s2[e] += v * v; }
        }
        if (ok) {
            sl_store_row(A.y + p * C, yv, h);
            if (A.drop_p > 0.f) sl_store_row(A.mask + p * C, mk, h);
        }
    }
    // per-channel sums over the slice's nodes: lanes of a half hold the same 16 channels for different nodes
#pragma unroll
    for (int e = 0; e < 16; ++e) { atomicAdd(&red[e * 64 + lane], s1[e]); atomicAdd(&red[(16 + e) * 64 + lane], s2[e]); }      // ds_add_f32, one address per lane
    __syncthreads();
    if (threadIdx.x < 64) {
        // thread = (stat, channel o): o = crow(e, hh) <-> e = (o & 3) + 4 (o >> 3), hh = (o >> 2) & 1
        const int stat = threadIdx.x >> 5, o = threadIdx.x & 31, e = (o & 3) + 4 * (o >> 3), hh = (o >> 2) & 1;
        float a = 0.f;
        for (int l = 0; l < 32; ++l) a += red[(16 * stat + e) * 64 + 32 * hh + l];
        bsum[threadIdx.x] = a;
    }
    __syncthreads();
    add_block_sums(A.sums, bsum);
}

struct SliceBwdArgs {
    const float *dy, *y; BnBwd bn;                 // gradient of BatchNorm_i's output, its saved input, the BatchNorm's backward sums
    const float* mask;                             // nullptr: dropout off
    int B, N, Tout;
    const uint16_t* frags;                         // this layer's fragments (the mix-backward ones are used)
    const uint16_t* P16;                           // [3][B][N][N8] bf16, rows = sending node v, k = w
    float *dres, *dh, *dcat;
};

// Backward of BatchNorm_i + dropout + mix data gradient (32 -> 224) + both adjoint hops of one (b, t) slice:
//   d = gamma rstd (dy - m1 - xhat m2) -> dres;  dh = d * mask;  dcat_k = Wmix_k^T dh (k = 0..6)
//   d_x1_s = dcat_{1+2s} + P_s d_x2_s   (d_x2_s = dcat_{2+2s});   d_z = dcat_0 + sum_s P_s d_x1_s
// dcat (all 7 slots, the accumulated ones) is written for tcn_bwd and the adjacency-gradient contraction.  LDS: 6 slices (d_x2_s, d_x1_s).
template <int TPW>
__global__ __launch_bounds__(512) void gcn_bwd_slice_kernel(SliceBwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) char sl_smem[];
    __shared__ float co[96];
    bn_bwd_coef(A.bn, co);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6, col = lane & 31, h = lane >> 5;
    const int N = A.N, NT = (N + 31) >> 5, NP = NT * 32, LDP = NP + 8, N8 = (N + 7) & ~7, nks = NP >> 4;
    const int b = blockIdx.x / A.Tout, t = blockIdx.x - b * A.Tout;
    uint16_t* XT[3] = {(uint16_t*)sl_smem, (uint16_t*)sl_smem + 32 * LDP, (uint16_t*)sl_smem + 2 * 32 * LDP};
    uint16_t* X1[3] = {XT[0] + 3 * 32 * LDP, XT[1] + 3 * 32 * LDP, XT[2] + 3 * 32 * LDP};
    const uint16_t* F = A.frags + (long)(SL_TCN_FRAGS + SL_MIX_FRAGS) * 64 * 8;
    const long prow_b = (long)b * N;
    const long sstride = (long)A.B * N * N8;
    bf16x8 dhb[TPW][2];
    auto mixpart = [&](int slot, const bf16x8 (&d)[2]) -> f32x16 {
        f32x16 a = sl_zero();
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, 2 * slot, lane), d[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl_frag(F, 2 * slot + 1, lane), d[1], a, 0, 0, 0);
        return a;
    };
    // ---------------------------------------------------------------- BatchNorm backward, dh, d_x2_s = slots 2, 4, 6
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + i * nw;
        if (tile >= NT) continue;
        const int w = tile * 32 + col;
        const bool ok = w < N;
        const long p = ((long)b * N + (ok ? w : N - 1)) * A.Tout + t;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float d8[8], y8[8], m8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, a[8];
            load8h(A.dy + p * C + 16 * q, h, d8); load8h(A.y + p * C + 16 * q, h, y8);
            if (A.mask) load8h(A.mask + p * C + 16 * q, h, m8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = chan8h(q, h, j);
                const float xhat = (y8[j] - A.bn.stat[64 + c]) * A.bn.stat[96 + c];
                const float d = ok ? co[64 + c] * (d8[j] - co[c] - xhat * co[32 + c]) : 0.f;
                d8[j] = d;
                a[j] = d * m8[j];
            }
            if (ok) { store8h(A.dres + p * C + 16 * q, h, d8); store8h(A.dh + p * C + 16 * q, h, a); }
            dhb[i][q] = pack8(a);
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const f32x16 d2 = mixpart(2 + 2 * s, dhb[i]);
            if (ok) sl_store_row(A.dcat + p * CAT + (2 + 2 * s) * C, d2, h);
            sl_to_lds(XT[s], LDP, w, ok, d2, h);
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- d_x1_s = slot (1 + 2 s) + P_s d_x2_s
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + i * nw;
        if (tile >= NT) continue;
        const int w = tile * 32 + col;
        const bool ok = w < N;
        const long p = ((long)b * N + (ok ? w : N - 1)) * A.Tout + t;
        const uint16_t* pr = A.P16 + (prow_b + (ok ? w : N - 1)) * N8;
        f32x16 d1[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) d1[s] = mixpart(1 + 2 * s, dhb[i]);
        sl_hop3<3, false>(d1, XT[0], XT[1], XT[2], pr, pr + sstride, pr + 2 * sstride, nks, N8, LDP, lane);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (ok) sl_store_row(A.dcat + p * CAT + (1 + 2 * s) * C, d1[s], h);
            sl_to_lds(X1[s], LDP, w, ok, d1[s], h);
        }
    }
    __syncthreads();
    // ---------------------------------------------------------------- d_z = slot 0 + sum_s P_s d_x1_s
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tile = wave + i * nw;
        if (tile >= NT) continue;
        const int w = tile * 32 + col;
        const bool ok = w < N;
        const long p = ((long)b * N + (ok ? w : N - 1)) * A.Tout + t;
        const uint16_t* pr = A.P16 + (prow_b + (ok ? w : N - 1)) * N8;
        f32x16 acc[3] = {mixpart(0, dhb[i]), sl_zero(), sl_zero()};
        sl_hop3<3, true>(acc, X1[0], X1[1], X1[2], pr, pr + sstride, pr + 2 * sstride, nks, N8, LDP, lane);
        if (ok) sl_store_row(A.dcat + p * CAT, acc[0], h);
    }
}

static inline bool slice_path_ok(int N) {
    static const bool on = []() { const char* e = getenv("STEP_GCN_SLICE"); return !(e && e[0] == '0'); }();      // (A/B knob: 0 = the four-kernel path)
    return on && N >= 32 && N <= SLICE_MAX_N;
}
static inline int slice_tpw(int N) { return ((N + 31) / 32 + 7) / 8; }                    // node tiles per wave (1 or 2): at most 8 waves
static inline int slice_waves(int N) { const int nt = (N + 31) / 32; return (nt + slice_tpw(N) - 1) / slice_tpw(N); }
static inline size_t slice_lds_bytes(int N, int nbuf, bool red) {
    const int NP = (N + 31) / 32 * 32, LDP = NP + 8;
    return (size_t)nbuf * 32 * LDP * 2 + (red ? (size_t)32 * 64 * 4 : 0);
}
