// GraphWaveNet backbone (STEP's fork, reference: step/step_arch/graphwavenet/model.py:132-224),
// forward and hand-written backward, orchestrated on the host side of the C ABI.
//
// Activation layout: X[b][n][t][c] with the 32 channels innermost ("position-major": one
// position = (b, n, t)).  Every channel contraction is then a plain row-major GEMM and the
// K-hop diffusion  (P x)[b,:,w,:] = sum_v x[b,:,v,:] P[b,v,w]  (model.py:10-16) is a batched
// GEMM  Out[b] (N x T*32) = P[b]^T X[b]  that reads/writes 32-channel slots of the concatenated
// gcn buffer cat[b][n][t][7*32] in place (slot order = torch.cat order of model.py:36-45:
// [x, P_f x, P_f^2 x, P_b x, P_b^2 x, P_a x, P_a^2 x]), so the 224->32 mix is one GEMM.
// All GEMMs run on the f32 matrix cores (step_gemm).  The remaining kernels are HBM-bound
// position-wise maps / reductions.
//
// Semantics kept from the reference: skip connections only matter at the last time index
// (every crop at model.py:196 keeps the tail, final T = 1); the last layer's gcn/bn output is
// dead code (model.py:202-213 for i = 7), so gconv.7 / bn.7 / residual_convs.* get no gradient.
#include "common.h"
#include "step_internal.h"
#include <vector>

namespace {

constexpr int C = 32, CAT = 224, CS = 256, CE = 512, OUT = 12, NL = 8, HID = 96;
constexpr int TIN[NL] = {13, 12, 10, 9, 7, 6, 4, 3};
constexpr int DIL[NL] = {1, 2, 1, 2, 1, 2, 1, 2};
constexpr int TOUT[NL] = {12, 10, 9, 7, 6, 4, 3, 1};

// ---------------------------------------------------------------------------- pointwise kernels
// start conv on the left-padded 2-channel input (model.py:143-155): x0[bn][t][c], t = 0..12
__global__ void start_conv_kernel(const float* __restrict__ hist, int B, int N, int Cin, const float* __restrict__ w,
                                  const float* __restrict__ b, float* __restrict__ x0) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)B * N * 13 * C;
    if (idx >= total) return;
    int c = idx % C, t = (idx / C) % 13;
    long bn = idx / (C * 13);
    int n = bn % N, bb = bn / N;
    float v = b[c];
    if (t >= 1) {
        const float* src = hist + (((long)bb * 12 + (t - 1)) * N + n) * Cin;
        v += w[c * 2] * src[0] + w[c * 2 + 1] * src[1];
    }
    x0[idx] = v;
}
__global__ __launch_bounds__(256) void start_conv_bwd_kernel(const float* __restrict__ hist, int B, int N, int Cin,
                                                             const float* __restrict__ dx0, float* __restrict__ dw,
                                                             float* __restrict__ db) {
    __shared__ float red[8][96];
    const int c = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;                       // 8 position sub-rows per block
    long npos = (long)B * N * 13;
    float a0 = 0.f, a1 = 0.f, ab = 0.f;
    for (long p = (long)blockIdx.x * 8 + sub; p < npos; p += (long)gridDim.x * 8) {
        int t = p % 13;
        long bn = p / 13;
        int n = bn % N, bb = bn / N;
        float d = dx0[p * C + c];
        ab += d;
        if (t >= 1) {
            const float* src = hist + (((long)bb * 12 + (t - 1)) * N + n) * Cin;
            a0 += d * src[0]; a1 += d * src[1];
        }
    }
    red[sub][c] = a0; red[sub][32 + c] = a1; red[sub][64 + c] = ab;
    __syncthreads();
    if (threadIdx.x < 96) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][threadIdx.x];
        const int cc = threadIdx.x & 31, kind = threadIdx.x >> 5;
        if (kind == 2) atomicAdd(&db[cc], s); else atomicAdd(&dw[cc * 2 + kind], s);
    }
}

// degree vectors: rs[b][i] = 1 + sum_j A[b][i][j] ; cs[b][j] = 1 + sum_i A[b][i][j]
__global__ __launch_bounds__(256) void row_sums_kernel(const float* __restrict__ A, int N, float* __restrict__ rs, float* __restrict__ cs_init) {
    __shared__ float red[4];
    const long row = blockIdx.x;                            // b*N + i
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) s += A[row * N + j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { rs[row] = 1.f + red[0] + red[1] + red[2] + red[3]; cs_init[row] = 1.f; }      // cs starts at 1 (the + I), see col_sums_kernel
}
// column sums on top of cs = 1 (written by row_sums_kernel, same element count): grid (columns / 64, row slices, B), one atomic per
// slice and column -- a single slice per column read 4096 rows sequentially at N = 4096
__global__ __launch_bounds__(256) void col_sums_kernel(const float* __restrict__ A, int N, float* __restrict__ cs) {
    __shared__ float red[4][64];
    const int b = blockIdx.z;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const int per = (N + gridDim.y - 1) / gridDim.y, ibeg = blockIdx.y * per, iend = min(N, ibeg + per);
    float s = 0.f;
    if (j < N)
        for (int i = ibeg + w; i < iend; i += 4) s += A[((long)b * N + i) * N + j];
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && j < N) atomicAdd(&cs[(long)b * N + j], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// P_f = D^-1 (A + I),  P_b = Dc^-1 (A^T + I)       (model.py:121-130,160); 32x32 tiles, grid (N/32, N/32, B)
// Writes both supports and their transposes (the adjoint hops read the transposed stack so that the
// contraction index is never the contiguous one).
__global__ __launch_bounds__(256) void rw_build_kernel(const float* __restrict__ A, int N, const float* __restrict__ rs,
                                                       const float* __restrict__ cs, float* __restrict__ Pf, float* __restrict__ Pb,
                                                       float* __restrict__ PfT, float* __restrict__ PbT) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long base = (long)b * N * N;
    for (int r = ty; r < 32; r += 8) {
        int i = i0 + r, j = j0 + tx;
        float a = (i < N && j < N) ? A[base + (long)i * N + j] : 0.f;
        tile[r][tx] = a;
        if (i < N && j < N) {
            const float v = a + (i == j ? 1.f : 0.f);
            Pf[base + (long)i * N + j] = v / rs[(long)b * N + i];
            PbT[base + (long)i * N + j] = v / cs[(long)b * N + j];        // P_b^T[i][j] = P_b[j][i]
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                       // transposed tile: element (i, j) = tile[tx][r]
        int j = j0 + r, i = i0 + tx;
        if (i < N && j < N) {
            const float v = tile[tx][r] + (i == j ? 1.f : 0.f);
            Pb[base + (long)j * N + i] = v / cs[(long)b * N + j];
            PfT[base + (long)j * N + i] = v / rs[(long)b * N + i];        // P_f^T[j][i] = P_f[i][j]
        }
    }
}
// adaptive support replicated per sample into stack slot 2 (and its transpose)
__global__ void replicate_adp_kernel(const float* __restrict__ Pa, int N, int B, float* __restrict__ P2, float* __restrict__ PT2) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * N) return;
    int i = idx / N, j = idx % N;
    float v = Pa[idx], vt = Pa[(long)j * N + i];
    for (int b = 0; b < B; ++b) { P2[(long)b * N * N + idx] = v; PT2[(long)b * N * N + idx] = vt; }
}
// bf16 copies of both stacks with the row pitch padded to N8 (pad = 0): the k-contiguous A operand of the bf16 hops
__global__ void stacks_to_bf16_kernel(const float* __restrict__ P, const float* __restrict__ PT, long rows, int N, int N8,
                                      uint16_t* __restrict__ P16, uint16_t* __restrict__ PT16) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * (N8 / 2)) return;
    const long row = idx / (N8 / 2);
    const int j = (int)(idx % (N8 / 2)) * 2;
    const float* src = blockIdx.y ? PT : P;
    uint16_t* dst = blockIdx.y ? PT16 : P16;
    const float a = j < N ? src[row * N + j] : 0.f, b = j + 1 < N ? src[row * N + j + 1] : 0.f;
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    f2 t = {a, b};
    *(uint32_t*)(dst + row * N8 + j) = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, b2));
}
// Hop operand of large graphs: 32-channel slots of the gcn buffer as bf16, transposed to [source][b][column = t * 32 + c][node] with the
// node axis contiguous (pitch N8, zero padded) -- the k-contiguous B operand of the hop product.  Read in place as f32 [node][t][224]
// the staged GEMM re-reads and re-converts the slot once per 128-row tile of the support (32 times at 4096 nodes: 1.2 GB through L2
// per launch, profiles/r03_j_C5_*); this copy is 3 MB per source and is read with plain 16-byte loads.
// grid (node tiles of 64, T, nsrc * B), 256 threads; source i reads slot slot0 + sstep * i.
__global__ __launch_bounds__(256) void slots_to_bf16T_kernel(const float* __restrict__ cat, int slot0, int sstep, int B, int N, int T, int N8,
                                                             uint16_t* __restrict__ XT) {
    __shared__ float tile[64][33];
    const int v0 = blockIdx.x * 64, t = blockIdx.y, src = blockIdx.z / B, b = blockIdx.z % B;
    const int slot = slot0 + sstep * src;
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    for (int i = r; i < 64; i += 8) {
        const int v = v0 + i;
        tile[i][c] = v < N ? cat[(((long)b * N + v) * T + t) * CAT + slot * C + c] : 0.f;
    }
    __syncthreads();
    // thread -> (channel = threadIdx / 8, 8 consecutive nodes): one 16-byte store
    const int cc = threadIdx.x >> 3, g8 = (threadIdx.x & 7) * 8;
    if (v0 + g8 < N8) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = tile[g8 + j][cc];
        *(bf16x8*)(XT + (((long)blockIdx.z * T + t) * C + cc) * N8 + v0 + g8) = pack8(v8);
    }
}
// out[e] = sum_b x[b][e]
__global__ void sum_batches_kernel(const float* __restrict__ x, long n, int B, float* __restrict__ out) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += x[(long)b * n + idx];
    out[idx] = s;
}
// rdot[row] = sum_k dP[row][k] * P[row][k]
__global__ __launch_bounds__(256) void row_dot_kernel(const float* __restrict__ dP, const float* __restrict__ P, int N,
                                                      float* __restrict__ out) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) s += dP[row * N + j] * P[row * N + j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[row] = red[0] + red[1] + red[2] + red[3];
}
// dA[b][i][j] = (dPf[i][j] - rf[i]) / rs[i] + (dPb[j][i] - rb[j]) / cs[j]
__global__ __launch_bounds__(256) void rw_bwd_kernel(const float* __restrict__ dPf, const float* __restrict__ dPb, int N,
                                                     const float* __restrict__ rs, const float* __restrict__ cs,
                                                     const float* __restrict__ rf, const float* __restrict__ rb, float* __restrict__ dA) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long base = (long)b * N * N;
    for (int r = ty; r < 32; r += 8) {                       // load dPb tile rows j, cols i
        int j = j0 + r, i = i0 + tx;
        tile[r][tx] = (i < N && j < N) ? (dPb[base + (long)j * N + i] - rb[(long)b * N + j]) / cs[(long)b * N + j] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int i = i0 + r, j = j0 + tx;
        if (i < N && j < N)
            dA[base + (long)i * N + j] = (dPf[base + (long)i * N + j] - rf[(long)b * N + i]) / rs[(long)b * N + i] + tile[tx][r];
    }
}
// adaptive adjacency: P_a[i][:] = softmax(relu(M[i][:]))   (model.py:165), one block per row
__global__ __launch_bounds__(256) void softmax_relu_rows_kernel(const float* __restrict__ M, int N, float* __restrict__ P) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    float mx = 0.f;                                          // relu output is >= 0
    for (int j = threadIdx.x; j < N; j += 256) mx = fmaxf(mx, M[row * N + j]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int j = threadIdx.x; j < N; j += 256) s += __expf(fmaxf(M[row * N + j], 0.f) - mx);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    for (int j = threadIdx.x; j < N; j += 256) P[row * N + j] = __expf(fmaxf(M[row * N + j], 0.f) - mx) * inv;
}
// dM = [M > 0] * P * (dP - rowdot)
__global__ void softmax_relu_bwd_kernel(const float* __restrict__ M, const float* __restrict__ P, const float* __restrict__ dP,
                                        const float* __restrict__ rdot, int N, float* __restrict__ dM) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * N) return;
    int i = idx / N;
    dM[idx] = M[idx] > 0.f ? P[idx] * (dP[idx] - rdot[i]) : 0.f;
}

// Weight packing, one launch: gated-TCN weights [32,32,1,2] x2 -> Wcat[64 out][64 in] (in = tap*32 + c) + its transpose,
// bias[64] (blocks [0, 128): layer = block / 16); the 8 skip convolutions as one [256 out][8*32 in] matrix and the sum of
// their biases (blocks [128, 384): one output row each); the 7 gcn mix weights transposed [224][32] (blocks [384, 580)).
struct GatePtrs { float *wf[8], *bf[8], *wg[8], *bg[8]; };
struct SkipPtrs { float *w[8], *b[8]; };
struct MixPtrs { const float* w[7]; };
__global__ __launch_bounds__(256) void pack_weights_kernel(GatePtrs P, SkipPtrs K, MixPtrs M, float* __restrict__ wcat_all,
                                                           float* __restrict__ wcatT_all, float* __restrict__ bcat_all,
                                                           float* __restrict__ wskip, float* __restrict__ bsum, float* __restrict__ wmixT) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk < 128) {
        const int L = blk >> 4;
        const int idx = (blk & 15) * 256 + tid;
        const int o = idx / 64, k = idx % 64;
        const float v = (o < 32 ? P.wf[L] : P.wg[L])[(((o & 31) * 32) + (k & 31)) * 2 + (k >> 5)];
        wcat_all[L * 4096 + idx] = v;
        wcatT_all[L * 4096 + k * 64 + o] = v;
        if (idx < 64) bcat_all[L * 64 + idx] = idx < 32 ? P.bf[L][idx] : P.bg[L][idx - 32];
    } else if (blk < 384) {
        const int o = blk - 128, k = tid;
        wskip[o * CS + k] = K.w[k >> 5][o * C + (k & 31)];
        if (o == 0) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += K.b[i][k];
            bsum[k] = sum;
        }
    } else {
        const int L = (blk - 384) / 28;
        const int idx = ((blk - 384) % 28) * 256 + tid;             // = c * 224 + j
        const int c = idx / CAT, j = idx % CAT;
        wmixT[L * (C * CAT) + j * C + c] = M.w[L][idx];
    }
}
__global__ void unpack_gate_grad_kernel(const float* __restrict__ dwcat_all, const float* __restrict__ dbcat_all, GatePtrs G) {
    const int L = blockIdx.y;
    const float* dwcat = dwcat_all + L * 4096;
    const float* dbcat = dbcat_all + L * 64;
    float *dwf = G.wf[L], *dbf = G.bf[L], *dwg = G.wg[L], *dbg = G.bg[L];
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < 64 * 64) {
        int o = idx / 64, k = idx % 64;
        float* dst = o < 32 ? dwf : dwg;
        dst[(((o & 31) * 32) + (k & 31)) * 2 + (k >> 5)] += dwcat[idx];
    }
    if (idx < 64) { if (idx < 32) dbf[idx] += dbcat[idx]; else dbg[idx - 32] += dbcat[idx]; }
}
// ---------------------------------------------------------------------------- fused layer kernels
// The input of layer i >= 1 is BatchNorm_{i-1}(y_{i-1}) and is never materialised: every consumer applies the two per-channel
// numbers scale / shift on the fly (stat == nullptr: layer 0, the start conv's output as is).
struct XIn { const float* src; const float* stat; };

// Operand loads of the register-fed MFMA tiles.  A lane needs 8 of the 16 (32x32x16 tile, two lanes per row) or 32 (16x16x32
// tile, four lanes per row) k values of a step; WHICH 8 is free as long as both operands agree, so the lanes of a row take
// interleaved 16-byte pieces: one load instruction then covers 32 (64) contiguous bytes per row instead of scattered pieces.
//   half h    of a 16-float chunk: floats [4h, 4h+4) and [8+4h, 8+4h+4)      -> element j is float 4h + (j&3) + 8(j>>2)
//   quarter g of a 32-float chunk: floats [4g, 4g+4) and [16+4g, 16+4g+4)    -> element j is float 4g + (j&3) + 16(j>>2)
__device__ __forceinline__ void load8h(const float* chunk, int h, float* v) {
    const float4 a = *(const float4*)(chunk + 4 * h), b = *(const float4*)(chunk + 8 + 4 * h);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8q(const float* chunk, int g, float* v) {
    const float4 a = *(const float4*)(chunk + 4 * g), b = *(const float4*)(chunk + 16 + 4 * g);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8h(float* chunk, int h, const float* v) {
    *(float4*)(chunk + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(chunk + 8 + 4 * h) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ int chan8h(int q, int h, int j) { return 16 * q + 4 * h + (j & 3) + 8 * (j >> 2); }

// One 16-deep step of a 32x32 tile product (lane l: row / column l & 31, half l >> 5).  bf16 mode: one v_mfma_f32_32x32x16_bf16
// (operands rounded to nearest even, like step_gemm's bf16 path); f32 mode: eight v_mfma_f32_32x32x2_f32, step s contracting
// element s of both halves.
template <bool BF16>
__device__ __forceinline__ void mma16(f32x16& acc, const float* a, const float* b) {
    if constexpr (BF16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(a), pack8(b), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
}
// 32-deep step of a 16x16 tile (lane l: row / column l & 15, quarter l >> 4): v_mfma_f32_16x16x32_bf16 / eight
// v_mfma_f32_16x16x4_f32; result D[m = 4 (l >> 4) + e][n = l & 15].
template <bool BF16>
__device__ __forceinline__ void mma32_16(f32x4& acc, const float* a, const float* b) {
    if constexpr (BF16) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack8(a), pack8(b), acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    }
}

// BatchNorm sums without finalize launches and without grid-wide synchronisation inside a kernel: the producing kernel adds
// its 64 block sums (f32) to f64 accumulators [NCOPY][64] with native f64 atomics (copy = block % NCOPY spreads the
// contention; the order-dependence of a few hundred f64 additions is ~1e-13 relative, far below the f32 numbers derived
// from them), and every block of the CONSUMING kernel turns the sums into the 2 x 32 numbers it needs in its prologue
// (block 0 also stores what later kernels and the backward read, and updates the running statistics exactly once).
constexpr int NCOPY = 16;
__device__ __forceinline__ void add_block_sums(double* acc, const float* blocksum64) {
    if (threadIdx.x < 64) unsafeAtomicAdd(acc + (blockIdx.x % NCOPY) * 64 + threadIdx.x, (double)blocksum64[threadIdx.x]);
}
__device__ __forceinline__ double gather_sum(const double* acc, int i) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < NCOPY; ++k) a += acc[k * 64 + i];
    return a;
}
// channel-last BN statistics -> st[128] = scale, shift, mean, rstd in LDS (model.py:212; running stats in train mode)
struct BnFwd { const float *gamma, *beta; float *rm, *rv; int training; float momentum; float* stat; double count; const double* sums; };
__device__ __forceinline__ void bn_fwd_stats(const BnFwd& bn, float* st) {
    __shared__ double tot[64];
    if (bn.training && threadIdx.x < 64) tot[threadIdx.x] = gather_sum(bn.sums, threadIdx.x);
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c = threadIdx.x;
        double mean, var;
        if (bn.training) {
            mean = tot[c] / bn.count;
            var = fmax(tot[32 + c] / bn.count - mean * mean, 0.0);
        } else {
            mean = bn.rm[c]; var = bn.rv[c];
        }
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        const float sc = bn.gamma[c] * rstd, sh = bn.beta[c] - (float)mean * sc;
        st[c] = sc; st[32 + c] = sh; st[64 + c] = (float)mean; st[96 + c] = rstd;
        if (blockIdx.x == 0) {
            bn.stat[c] = sc; bn.stat[32 + c] = sh; bn.stat[64 + c] = (float)mean; bn.stat[96 + c] = rstd;
            if (bn.training) {
                bn.rm[c] = (1.f - bn.momentum) * bn.rm[c] + bn.momentum * (float)mean;
                bn.rv[c] = (1.f - bn.momentum) * bn.rv[c] + bn.momentum * (float)(var * bn.count / fmax(bn.count - 1.0, 1.0));
            }
        }
    }
    __syncthreads();
}
// statistics of a BatchNorm whose output nobody reads (bn.7, model.py:210-213): only the running-statistics update of bn_fwd_stats
__global__ void bn_stats_only_kernel(BnFwd bn) {
    __shared__ float st[128];
    bn_fwd_stats(bn, st);
}
// BN backward sums S1 = sum dy, S2 = sum dy*xhat -> co[96] = [S1/count | S2/count | gamma*rstd] in LDS; block 0 adds the
// gradients of gamma / beta
struct BnBwd { const float *gamma, *stat; float *dgamma, *dbeta; double count; const double* sums; };
__device__ __forceinline__ void bn_bwd_coef(const BnBwd& bn, float* co) {
    __shared__ double tot[64];
    if (threadIdx.x < 64) tot[threadIdx.x] = gather_sum(bn.sums, threadIdx.x);
    __syncthreads();
    if (threadIdx.x < 32) {
        const int c = threadIdx.x;
        co[c] = (float)(tot[c] / bn.count); co[32 + c] = (float)(tot[32 + c] / bn.count); co[64 + c] = bn.gamma[c] * bn.stat[96 + c];
        if (blockIdx.x == 0) { bn.dgamma[c] += (float)tot[32 + c]; bn.dbeta[c] += (float)tot[c]; }
    }
    __syncthreads();
}

// Gated TCN of one layer (model.py:183-189) in one kernel: the two-tap dilated filter / gate convolutions as one 64 -> 64
// contraction per position (k = tap*32 + c, read straight from the layer input: no im2col), tanh * sigmoid, z into slot 0
// of the gcn buffer, and the last time step's z into column block `layer` of zlast [BN][8*32] (the skip convolutions only
// matter there -- they are applied to all 8 layers at once in the head).  Layers >= 1 finalise the previous layer's
// BatchNorm statistics in the prologue (bn.stat != nullptr) and apply them to what they read.  One wave = 32 positions.
template <bool BF16>
__global__ __launch_bounds__(256) void tcn_fwd_kernel(const float* __restrict__ src, BnFwd bn, long npos, int Tin, int Tout, int dil,
                                                      const float* __restrict__ wcat, const float* __restrict__ bcat,
                                                      float* __restrict__ tf, float* __restrict__ sg, float* __restrict__ cat,
                                                      float* __restrict__ zlast, int layer) {
    __shared__ float st[128];
    if (bn.stat) bn_fwd_stats(bn, st);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (r0 >= npos) return;
    long p = r0 + col;
    if (p >= npos) p = npos - 1;
    const long bn_ = p / Tout;
    const long xrow = bn_ * Tin + (p - bn_ * Tout);
    float a[4][8], bf[4][8], bg[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                          // k = 16 q + ... = tap*32 + c
        load8h(src + (xrow + (q >> 1) * dil) * C + 16 * (q & 1), h, a[q]);
        load8h(wcat + col * 64 + 16 * q, h, bf[q]);
        load8h(wcat + (32 + col) * 64 + 16 * q, h, bg[q]);
    }
    if (bn.stat) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int c = chan8h(q & 1, h, j); a[q][j] = a[q][j] * st[c] + st[32 + c]; }
    }
    f32x16 af, ag;
#pragma unroll
    for (int e = 0; e < 16; ++e) { af[e] = 0.f; ag[e] = 0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { mma16<BF16>(af, a[q], bf[q]); mma16<BF16>(ag, a[q], bg[q]); }
    const float b_f = bcat[col], b_g = bcat[32 + col];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long pe = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (pe >= npos) continue;
        const float f = tanhf(af[e] + b_f);
        const float g = 1.f / (1.f + __expf(-(ag[e] + b_g)));
        tf[pe * C + col] = f; sg[pe * C + col] = g;
        cat[pe * CAT + col] = f * g;
        const long bne = pe / Tout;
        if (pe - bne * Tout == Tout - 1) zlast[bne * CS + layer * C + col] = f * g;
    }
}

// gcn mix of one layer: h = cat @ Wmix^T + b (224 -> 32), dropout, + residual x_in[t + dil] (model.py:36-47,206-212),
// y written once, BatchNorm sums added to `sums`.  One wave = 16 positions x 32 channels (two 16x16 tiles), every operand
// load of the 224-deep contraction issued before the first MFMA; one Philox call yields the keep decisions of a lane's four
// positions of one channel.
template <bool BF16>
__global__ __launch_bounds__(256) void mix_fwd_kernel(const float* __restrict__ cat, const float* __restrict__ w,
                                                      const float* __restrict__ bias, XIn x, long npos, int Tin, int Tout, int dil,
                                                      float drop_p, uint32_t seed_lo, uint32_t seed_hi, uint32_t layer,
                                                      float* __restrict__ mask, float* __restrict__ y, double* __restrict__ sums,
                                                      const StepDynState* __restrict__ dyn) {
    if (dyn) { const uint64_t sx = dyn->seed_xor; seed_lo ^= (uint32_t)sx; seed_hi ^= (uint32_t)(sx >> 32); }      // replayed steps (step_hip.h)
    __shared__ float red[4][64];
    __shared__ float bsum[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ln = lane & 15, kg = lane >> 4;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 16;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    if (r0 < npos) {
        long p = r0 + ln;
        if (p >= npos) p = npos - 1;
        float a[7][8], b0[7][8], b1[7][8];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            load8q(cat + p * CAT + 32 * s, kg, a[s]);
            load8q(w + ln * CAT + 32 * s, kg, b0[s]);
            load8q(w + (16 + ln) * CAT + 32 * s, kg, b1[s]);
        }
        // residual rows of this lane's 4 positions (clamped; stores are guarded below)
        float res[2][4];
        long pe[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            long q = r0 + 4 * kg + e;
            pe[e] = q;
            if (q >= npos) q = npos - 1;
            const long bne = q / Tout;
            const float* rsrc = x.src + (bne * Tin + (q - bne * Tout) + dil) * C;
            res[0][e] = rsrc[ln]; res[1][e] = rsrc[16 + ln];
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 7; ++s) { mma32_16<BF16>(acc[0], a[s], b0[s]); mma32_16<BF16>(acc[1], a[s], b1[s]); }
        const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int col = 16 * half + ln;
            const float bv = bias[col];
            const float xs = x.stat ? x.stat[col] : 1.f, xh = x.stat ? x.stat[32 + col] : 0.f;
            uint32_t r[4] = {0u, 0u, 0u, 0u};
            if (drop_p > 0.f) {
                const long i0 = pe[0] * C + col;
                philox4x32((uint32_t)i0, (uint32_t)(i0 >> 32), layer, 0xD409u, seed_lo, seed_hi, r);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (pe[e] >= npos) continue;
                const long idx = pe[e] * C + col;
                float m = 1.f;
                if (drop_p > 0.f) {
                    m = u32_to_unit(r[e]) >= drop_p ? keep_scale : 0.f;
                    mask[idx] = m;
                }
                const float v = (acc[half][e] + bv) * m + (res[half][e] * xs + xh);
                y[idx] = v;
                s1[half] += v; s2[half] += v * v;
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            s1[half] += __shfl_xor(s1[half], 16, 64); s2[half] += __shfl_xor(s2[half], 16, 64);
            s1[half] += __shfl_xor(s1[half], 32, 64); s2[half] += __shfl_xor(s2[half], 32, 64);
        }
    }
    if (kg == 0) { red[wave][ln] = s1[0]; red[wave][16 + ln] = s1[1]; red[wave][32 + ln] = s2[0]; red[wave][48 + ln] = s2[1]; }
    __syncthreads();
    if (threadIdx.x < 64) bsum[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    __syncthreads();
    add_block_sums(sums, bsum);
}

// Backward of BatchNorm_i + dropout + mix data gradient in one kernel (coefficients from the sums the previous backward
// kernel left, see bn_bwd_coef):
//   d = gamma*rstd * (dy - m1 - xhat m2)  (-> dres: the residual branch's gradient),  dh = d * mask,  dcat = dh @ Wmix (32 -> 224)
// dh is kept for the weight-gradient GEMM.  One wave = 32 positions; wT = Wmix transposed [224][32].
template <bool BF16>
__global__ __launch_bounds__(256) void mix_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, BnBwd bn,
                                                      const float* __restrict__ mask, const float* __restrict__ wT, long npos,
                                                      float* __restrict__ dres, float* __restrict__ dh, float* __restrict__ dcat) {
    __shared__ float co[96];
    bn_bwd_coef(bn, co);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (r0 >= npos) return;
    const long p = r0 + col;
    const bool ok = p < npos;
    float a[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float d8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, y8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, m8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        if (ok) {
            load8h(dy + p * C + 16 * q, h, d8); load8h(y + p * C + 16 * q, h, y8);
            if (mask) load8h(mask + p * C + 16 * q, h, m8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chan8h(q, h, j);
            const float xhat = (y8[j] - bn.stat[64 + c]) * bn.stat[96 + c];
            const float d = ok ? co[64 + c] * (d8[j] - co[c] - xhat * co[32 + c]) : 0.f;
            d8[j] = d;
            a[q][j] = d * m8[j];
        }
        if (ok) { store8h(dres + p * C + 16 * q, h, d8); store8h(dh + p * C + 16 * q, h, a[q]); }
    }
#pragma unroll
    for (int slot = 0; slot < 7; ++slot) {
        float b[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q) load8h(wT + (slot * C + col) * C + 16 * q, h, b[q]);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) mma16<BF16>(acc, a[q], b[q]);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long pe = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (pe < npos) dcat[pe * CAT + slot * C + col] = acc[e];
        }
    }
}

// Backward of the gated TCN of one layer in one kernel.  A workgroup owns G = 64 / Tout whole (b, n) pairs:
//   dz = dcat slot 0 (+ d_zlast at the last time step: the skip branch),  dpre = [dz g (1 - f^2) | dz f g (1 - g)]  (kept for
//   the weight-gradient GEMM),  dxcat = dpre @ Wcat (64 -> 64, tile kept in LDS),  then the transposed im2col
//   dx[bn][t'][c] = [t' < Tout] dxcat[t'][c] + [t' >= dil] (dxcat[t'-dil][32+c] + dres[t'-dil][c])
// and, dx being the gradient of BatchNorm_{layer-1}'s output, that BatchNorm's backward sums (xhat from yprev / statprev)
// added to `sums`.  dcat == nullptr: last layer (gradient from the skip branch only).  wcatT = Wcat transposed [64 n][64 k].
// Wave w: rows 32 (w & 1) of the tile, output columns 32 (w >> 1).
constexpr int TB_ROWS = 64;
template <bool BF16>
__global__ __launch_bounds__(256) void tcn_bwd_kernel(const float* __restrict__ dcat, const float* __restrict__ dzlast, int layer,
                                                      const float* __restrict__ tf, const float* __restrict__ sg,
                                                      const float* __restrict__ wcatT, const float* __restrict__ dres, long BN, int Tin,
                                                      int Tout, int dil, float* __restrict__ dpre, float* __restrict__ dx,
                                                      const float* __restrict__ yprev, const float* __restrict__ statprev,
                                                      double* __restrict__ sums) {
    __shared__ float dxs[TB_ROWS][65];
    __shared__ float red[8][64];
    __shared__ float bsum[64];
    const int G = TB_ROWS / Tout, R = G * Tout;
    const long bn0 = (long)blockIdx.x * G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
    {
        const int nt = wave >> 1;
        const int r = (wave & 1) * 32 + col;                 // row of the tile
        const long bnr = bn0 + r / Tout;
        const bool ok = r < R && bnr < BN;
        const long p = bn0 * Tout + r;
        const bool lastt = ok && (r % Tout == Tout - 1);
        float b[4][8];                                       // B[k = part*32 + c][n] = Wcat[k][n]
#pragma unroll
        for (int q = 0; q < 4; ++q) load8h(wcatT + (32 * nt + col) * 64 + 16 * q, h, b[q]);
        float dp[2][2][8];                                   // [part filter | gate][16-channel chunk][j]
#pragma unroll
        for (int qc = 0; qc < 2; ++qc) {
            float dz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, f8[8], g8[8];
            if (ok) {
                if (dcat) load8h(dcat + p * CAT + 16 * qc, h, dz);
                load8h(tf + p * C + 16 * qc, h, f8); load8h(sg + p * C + 16 * qc, h, g8);
                if (lastt) {
                    float s8[8];
                    load8h(dzlast + bnr * CS + layer * C + 16 * qc, h, s8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) dz[j] += s8[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dp[0][qc][j] = ok ? dz[j] * g8[j] * (1.f - f8[j] * f8[j]) : 0.f;
                dp[1][qc][j] = ok ? dz[j] * f8[j] * g8[j] * (1.f - g8[j]) : 0.f;
            }
            if (ok) store8h(dpre + p * 64 + 32 * nt + 16 * qc, h, dp[nt][qc]);     // wave pair (w, w + 2) shares the rows: each stores one part
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) mma16<BF16>(acc, dp[q >> 1][q & 1], b[q]);
#pragma unroll
        for (int e = 0; e < 16; ++e) dxs[(wave & 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h][32 * nt + col] = acc[e];
    }
    __syncthreads();
    const int c = threadIdx.x & 31, jg = threadIdx.x >> 5;
    float s1 = 0.f, s2 = 0.f;
    const float mean = statprev ? statprev[64 + c] : 0.f, rstd = statprev ? statprev[96 + c] : 0.f;
    const int njt = G * Tin;
    for (int j0 = 0; j0 < njt; j0 += 32) {                   // 4 (bn, t') pairs per thread and pass, their loads issued together
        float dr[4], yv[4], v[4];
        long o[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 8 * u + jg;
            const int g = j / Tin, t = j - g * Tin;
            const long bnr = bn0 + g;
            valid[u] = j < njt && bnr < BN;
            const long bc = valid[u] ? bnr : 0;
            const int tc = valid[u] ? t : 0;
            o[u] = (bc * Tin + tc) * C + c;
            const bool tap1 = valid[u] && tc >= dil;
            dr[u] = (dres && tap1) ? dres[(bc * Tout + tc - dil) * C + c] : 0.f;
            yv[u] = yprev ? yprev[o[u]] : 0.f;
            float x = 0.f;
            if (valid[u] && tc < Tout) x += dxs[g * Tout + tc][c];
            if (tap1) x += dxs[g * Tout + tc - dil][32 + c];
            v[u] = x;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!valid[u]) continue;
            const float x = v[u] + dr[u];
            dx[o[u]] = x;
            s1 += x; s2 += x * (yv[u] - mean) * rstd;
        }
    }
    if (!yprev) return;
    red[jg][c] = s1; red[jg][32 + c] = s2;
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) a += red[r][threadIdx.x];
        bsum[threadIdx.x] = a;
    }
    __syncthreads();
    add_block_sums(sums, bsum);
}

// xcat[(bn,t)][tap*32 + c] = x[bn][t + tap*dil][c]   (operand of the gated-TCN weight gradient)
__global__ void im2col_kernel(XIn x, long BN, int Tin, int Tout, int dil, float* __restrict__ xcat) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= BN * Tout * 64) return;
    int k = idx % 64, t = (idx / 64) % Tout;
    long bn = idx / (64L * Tout);
    float v = x.src[(bn * Tin + t + (k >> 5) * dil) * C + (k & 31)];
    if (x.stat) v = v * x.stat[k & 31] + x.stat[32 + (k & 31)];
    xcat[idx] = v;
}
__global__ void unpack_skip_grad_kernel(const float* __restrict__ dwskip, SkipPtrs G) {
    const int o = blockIdx.x, k = threadIdx.x;
    G.w[k >> 5][o * C + (k & 31)] += dwskip[o * CS + k];
}
// xh = relu(skip + h2)   (skip already carries the sum of the 8 skip-conv biases)
__global__ void head_combine_kernel(const float* __restrict__ skip, const float* __restrict__ h2, long n, float* __restrict__ xh) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    xh[idx] = fmaxf(skip[idx] + h2[idx], 0.f);
}
struct MPtr8 { float* p[8]; };
// dst_k[i] += v[i] for the 8 skip-conv bias gradients
__global__ void add_to8_kernel(MPtr8 dst, const float* __restrict__ v, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = v[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) dst.p[k][i] += x;
}
// in place: d *= (y > 0)
__global__ void relu_bwd_kernel(float* __restrict__ d, const float* __restrict__ y, long n) {
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n && !(y[idx] > 0.f)) d[idx] = 0.f;
}

// out[row % mod] += sum_j x[row*cols + j]   (bias gradient of the [B,12,N] prediction)
__global__ __launch_bounds__(256) void rowsum_mod_kernel(const float* __restrict__ x, int cols, int mod, float* __restrict__ out) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < cols; j += 256) s += x[row * cols + j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[row % mod], red[0] + red[1] + red[2] + red[3]);
}

// Adjacency gradients of all 7 gcn layers as ONE segmented contraction per support (instead of 14 read-modify-write passes over
// the [3][B][N][N] gradient stack):  dP_s[b][v][w] = sum_layers sum_t sum_c ( x1_s[v][t][c] d_x2_s[w][t][c] + z[v][t][c] d_x1_s[w][t][c] ).
// k block (layer i, t, pair) -> 32 channels of one slot of cat_i (A side, offsets from `saved`) and of dcat_i (B side, offsets from
// the per-layer gradient buffers); table [3 supports][ADJ_NSEG].
constexpr int ADJ_NSEG = 2 * (12 + 10 + 9 + 7 + 6 + 4 + 3);
struct AdjOffsets { long cat[NL - 1], dcat[NL - 1]; };
__global__ void adj_ktab_kernel(AdjOffsets o, int N, GemmKSeg* __restrict__ tab) {
    const int j = threadIdx.x, s = blockIdx.x;
    if (j >= ADJ_NSEG) return;
    constexpr int tout[NL - 1] = {12, 10, 9, 7, 6, 4, 3};
    int i = 0, rem = j >> 1;
    while (rem >= tout[i]) { rem -= tout[i]; ++i; }
    const int t = rem, pair = j & 1, T = tout[i];
    GemmKSeg g;
    g.a_off = o.cat[i] + (long)t * CAT + (pair == 0 ? (1 + 2 * s) * C : 0);
    g.b_off = o.dcat[i] + (long)t * CAT + (pair == 0 ? (2 + 2 * s) * C : (1 + 2 * s) * C);
    g.a_rs = g.b_rs = T * CAT;
    g.a_bs = g.b_bs = (long)N * T * CAT;
    tab[s * ADJ_NSEG + j] = g;
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

// ---------------------------------------------------------------------------- buffer carving
struct Carver {
    float* base;
    long used = 0;
    explicit Carver(float* b) : base(b) {}
    float* take(long n) {
        float* p = base ? base + used : nullptr;
        used += (n + 3) & ~3L;            // keep 16-byte alignment
        return p;
    }
};

static inline int n8(int N) { return (N + 7) & ~7; }
// graphs from this many nodes on feed the diffusion hops a transposed bf16 copy of their source slots (see slots_to_bf16T_kernel);
// STEP_HOP_XT_MIN_N overrides the threshold (A/B measurements: a huge value = always the in-place f32 operand)
static inline bool hop_transposed_operand(int N) {
    // (768 since the end of round 6: at 883 nodes the copy pays too, 5.35 / 5.32 -> 5.32 / 5.30 ms per step; at 307 it costs, 3.39 -> 3.47:
    //  profiles/r06_zm_hop_xt.log)
    static const int min_n = []() { const char* e = getenv("STEP_HOP_XT_MIN_N"); return e ? atoi(e) : 768; }();
    return N >= min_n;
}

struct Saved {
    float *x0, *cat[NL], *tf[NL], *sg[NL], *y[NL], *mask[NL], *bnstat[NL], *zlast;     // x0: start conv output; layer i >= 1 reads BN(y[i-1])
    float *Pstk, *PTstk, *Pa, *Madp, *rs, *cs;      // stacks: [3 supports f,b,a][B][N][N]
    uint16_t *P16, *PT16;                           // bf16 copies, rows zero-padded to N8 = roundup(N, 8) (bf16 hop operands)
    float *skip, *h1, *h2, *xh, *e1;
    long total;
};
Saved carve_saved(float* base, int B, int N, bool dropout) {
    Carver cv(base);
    Saved s;
    const long BN = (long)B * N;
    s.x0 = cv.take(BN * TIN[0] * C);
    s.zlast = cv.take(BN * CS);
    for (int i = 0; i < NL; ++i) {
        s.cat[i] = cv.take(BN * TOUT[i] * CAT);
        s.tf[i] = cv.take(BN * TOUT[i] * C);
        s.sg[i] = cv.take(BN * TOUT[i] * C);
        s.y[i] = i < NL - 1 ? cv.take(BN * TOUT[i] * C) : nullptr;
        s.mask[i] = (i < NL - 1 && dropout) ? cv.take(BN * TOUT[i] * C) : nullptr;
        s.bnstat[i] = cv.take(128);
    }
    s.Pstk = cv.take(3L * B * N * N);
    s.PTstk = cv.take(3L * B * N * N);
    s.P16 = (uint16_t*)cv.take(3L * B * N * n8(N) / 2);
    s.PT16 = (uint16_t*)cv.take(3L * B * N * n8(N) / 2);
    s.Pa = cv.take((long)N * N);
    s.Madp = cv.take((long)N * N);
    s.rs = cv.take(BN);
    s.cs = cv.take(BN);
    s.skip = cv.take(BN * CS);
    s.h1 = cv.take(BN * CE);
    s.h2 = cv.take(BN * CS);
    s.xh = cv.take(BN * CS);
    s.e1 = cv.take(BN * CE);
    s.total = cv.used;
    return s;
}
struct Work {
    float *wcat, *wcatT, *bcat, *wskip, *wmixT, *dwcat, *dbcat, *dwskip;       // [8][64][64] x2, [8][64], [256][256], [7][224][32]; d* and acc64 contiguous (one memset)
    double* acc64;             // [7 BatchNorms][NCOPY][64] f64 accumulators of the in-kernel BatchNorm sums
    // xcat / dpre / dh: per-layer copies of what the weight-gradient GEMMs read -- those GEMMs are leaves of the backward and run on
    // the auxiliary stream while the data-gradient chain goes on; dcat: one gcn-buffer gradient per layer (all needed at the end)
    float *xcat[NL], *bsum, *y7, *m7;      // y7 / m7: output and dropout mask of the last layer's (dead) gcn, only for its BatchNorm statistics
    uint16_t* xT;              // [3][B][12 * 32][N8] bf16: transposed hop operand of large graphs (slots_to_bf16T_kernel)
    float *dcat[NL - 1], *dpre[NL], *dh[NL - 1], *dres, *dxa, *dxb, *dskip, *dPstk, *dPa, *dM, *rf, *rb;
    GemmKSeg* ktab;
    float *d_e1, *d_xh, *d_h2, *d_h1;
    long total;
};
Work carve_work(float* base, int B, int N, bool backward) {
    Carver cv(base);
    Work w;
    const long BN = (long)B * N;
    w.wcat = cv.take(NL * 64 * 64);
    w.wcatT = cv.take(NL * 64 * 64);
    w.bcat = cv.take(NL * 64);
    w.wskip = cv.take(CS * CS);
    w.wmixT = cv.take(7 * C * CAT);
    w.dwcat = cv.take(NL * 64 * 64);
    w.dbcat = cv.take(NL * 64);
    w.dwskip = cv.take(CS * CS);
    w.acc64 = (double*)cv.take(2L * NL * NCOPY * 64);      // 7 live BatchNorms + the dead bn.7 (only with STEP_GWNET_DEAD_BN7)
    w.xT = hop_transposed_operand(N) ? (uint16_t*)cv.take(3L * B * 12 * C * n8(N) / 2) : nullptr;
    for (int i = 0; i < NL; ++i) w.xcat[i] = backward ? cv.take(BN * TOUT[i] * 64) : nullptr;
    w.bsum = cv.take(CS);
    w.y7 = cv.take(BN * TOUT[NL - 1] * C);
    w.m7 = cv.take(BN * TOUT[NL - 1] * C);
    if (backward) {
        for (int i = 0; i < NL - 1; ++i) w.dcat[i] = cv.take(BN * TOUT[i] * CAT);
        w.ktab = (GemmKSeg*)cv.take(3L * ADJ_NSEG * sizeof(GemmKSeg) / sizeof(float));
        for (int i = 0; i < NL; ++i) w.dpre[i] = cv.take(BN * TOUT[i] * 64);
        for (int i = 0; i < NL - 1; ++i) w.dh[i] = cv.take(BN * TOUT[i] * C);
        w.dres = cv.take(BN * 12 * C);
        w.dxa = cv.take(BN * 13 * C);
        w.dxb = cv.take(BN * 13 * C);
        w.dskip = cv.take(BN * CS);
        w.dPstk = cv.take(3L * B * N * N);
        w.dPa = cv.take((long)N * N);
        w.dM = cv.take((long)N * N);
        w.rf = cv.take(BN > N ? BN : N);
        w.rb = cv.take(BN);
        w.d_e1 = cv.take(BN * CE);
        w.d_xh = cv.take(BN * CS);
        w.d_h2 = cv.take(BN * CS);
        w.d_h1 = cv.take(BN * CE);
    }
    w.total = cv.used;
    return w;
}

// Fork / join between the caller's stream and its auxiliary stream with re-usable events (one small pool per host thread and device:
// a wait captures the state of its event when it is queued, so re-recording an event for a later fork is safe).
int zero(float* p, long n, hipStream_t st) {
    if (hipMemsetAsync(p, 0, (size_t)n * sizeof(float), st) != hipSuccess) {
        step_set_error("gwnet: memset failed");
        return STEP_ERR_HIP;
    }
    return STEP_OK;
}

// One launch = the same diffusion hop for the three supports (two-level batch: i1 = support, i0 = sample).
// Forward hop  Out_s[b][w][n] = sum_v P_s[b][v][w] X_s[b][v][n]  reading slot src0 + sstep*s, writing dst0 + 2*s.
// bf16 mode: A(m=w, k=v) = PT16[w][v] (k contiguous, bf16) -- no transposition on the way into LDS.
int nconv_fwd3(const float* Pstk, const uint16_t* PT16, float* cat, int src0, int sstep, int dst0, int B, int N, int T, int bf16,
               uint16_t* xT, hipStream_t st) {
    StepGemm g = gemm_desc(N, T * C, N, Pstk, 1, N, cat + src0 * C, (long)T * CAT, 1, cat + dst0 * C, (long)T * CAT);
    if (bf16 && xT) {
        // B(k = v, n = column) = xT[source][b][column][v]: k contiguous bf16, one source for all three supports (sstep == 0) or one each
        const int nsrc = sstep == 0 ? 1 : 3;
        slots_to_bf16T_kernel<<<dim3(cdiv(N, 64), T, nsrc * B), 256, 0, st>>>(cat, src0, sstep, B, N, T, n8(N), xT);
        STEP_LAUNCH_CHECK("slots_to_bf16T");
        g.batch = 3 * B; g.batch0 = B;
        g.A = PT16; g.a_bf16 = 1; g.sam = n8(N); g.sak = 1; g.sab = (long)N * n8(N); g.sab1 = (long)B * N * n8(N);
        g.B = xT; g.b_bf16 = 1; g.sbk = 1; g.sbn = n8(N); g.sbb = (long)T * C * n8(N); g.sbb1 = nsrc == 1 ? 0 : (long)B * T * C * n8(N);
        g.scb = (long)N * T * CAT; g.scb1 = 2L * C;
        g.c_nblk = C; g.c_nstride = CAT;
        g.compute_bf16 = 1;
        return step_gemm_launch(g, st);
    }
    g.batch = 3 * B; g.batch0 = B;
    g.sab = (long)N * N; g.sab1 = (long)B * N * N;
    if (bf16) { g.A = PT16; g.a_bf16 = 1; g.sam = n8(N); g.sak = 1; g.sab = (long)N * n8(N); g.sab1 = (long)B * N * n8(N); }
    g.sbb = (long)N * T * CAT; g.sbb1 = (long)sstep * C;
    g.scb = (long)N * T * CAT; g.scb1 = 2L * C;
    g.b_nblk = C; g.b_nstride = CAT; g.c_nblk = C; g.c_nstride = CAT;
    g.compute_bf16 = bf16;
    return step_gemm_launch(g, st);
}
// Adjoint hop  dDst_s[b][v][n] += sum_w P_s[b][v][w] dSrc_s[b][w][n]  (reads the transposed stack: A(m=v,k=w) = PT[w][v]).
// dstep == 0: the three supports accumulate into the same slot -> atomics.
int nconv_bwd_data3(const float* PTstk, const uint16_t* P16, float* dcat, int src0, int dst0, int dstep, int B, int N, int T, int bf16,
                    uint16_t* xT, hipStream_t st) {
    StepGemm g = gemm_desc(N, T * C, N, PTstk, 1, N, dcat + src0 * C, (long)T * CAT, 1, dcat + dst0 * C, (long)T * CAT);
    if (bf16 && xT) {
        // the three sources are the slots src0, src0 + 2, src0 + 4 of the gradient buffer
        slots_to_bf16T_kernel<<<dim3(cdiv(N, 64), T, 3 * B), 256, 0, st>>>(dcat, src0, 2, B, N, T, n8(N), xT);
        STEP_LAUNCH_CHECK("slots_to_bf16T");
        g.batch = 3 * B; g.batch0 = B;
        g.A = P16; g.a_bf16 = 1; g.sam = n8(N); g.sak = 1; g.sab = (long)N * n8(N); g.sab1 = (long)B * N * n8(N);
        g.B = xT; g.b_bf16 = 1; g.sbk = 1; g.sbn = n8(N); g.sbb = (long)T * C * n8(N); g.sbb1 = (long)B * T * C * n8(N);
        g.scb = (long)N * T * CAT; g.scb1 = (long)dstep * C;
        g.c_nblk = C; g.c_nstride = CAT;
        g.accumulate = dstep == 0 ? 2 : 1;
        g.compute_bf16 = 1;
        return step_gemm_launch(g, st);
    }
    g.batch = 3 * B; g.batch0 = B;
    g.sab = (long)N * N; g.sab1 = (long)B * N * N;
    if (bf16) { g.A = P16; g.a_bf16 = 1; g.sam = n8(N); g.sak = 1; g.sab = (long)N * n8(N); g.sab1 = (long)B * N * n8(N); }
    g.sbb = (long)N * T * CAT; g.sbb1 = 2L * C;
    g.scb = (long)N * T * CAT; g.scb1 = (long)dstep * C;
    g.b_nblk = C; g.b_nstride = CAT; g.c_nblk = C; g.c_nstride = CAT;
    g.accumulate = dstep == 0 ? 2 : 1;
    g.compute_bf16 = bf16;
    return step_gemm_launch(g, st);
}
}  // namespace

// =========================================================================================== C ABI
extern "C" long step_gwnet_saved_floats(int B, int N, int dropout) { return carve_saved(nullptr, B, N, dropout != 0).total; }
extern "C" long step_gwnet_work_floats(int B, int N, int backward) { return carve_work(nullptr, B, N, backward != 0).total; }
// element offset (in floats) of a saved item, for tests / the python side: item 0 = dropout mask, 1 = y (pre-BN),
// 2 = bnstat of layer `layer`
extern "C" long step_gwnet_saved_offset(int B, int N, int dropout, int item, int layer) {
    Saved s = carve_saved((float*)16, B, N, dropout != 0);
    float* p = item == 0 ? s.mask[layer] : item == 1 ? s.y[layer] : s.bnstat[layer];
    return p ? (long)(p - (float*)16) : -1;
}

// first_tcn_only: layer 0's gated TCN and nothing else (it reads the start convolution's output, not the adjacency: phase 5);
// skip_first_tcn: everything but that launch (phase 6)
template <bool BF16>
static int gwnet_layers_forward(const StepGwnetParams* p, const Saved& S, const Work& W, int B, int N, bool training, float drop_p,
                                uint64_t seed, float momentum, bool dead_bn7, const StepDynState* dyn, hipStream_t st,
                                bool first_tcn_only = false, bool skip_first_tcn = false) {
    const long BN = (long)B * N;
    for (int i = 0; i < (first_tcn_only ? 1 : NL); ++i) {
        const int Tin = TIN[i], Tout = TOUT[i], dil = DIL[i];
        const long npos = BN * Tout;
        // statistics of BatchNorm_{i-1} (sums left by mix_fwd of the previous layer), finalised by this layer's first kernel
        BnFwd bn = {nullptr, nullptr, nullptr, nullptr, training ? 1 : 0, momentum, nullptr, (double)BN * Tin, nullptr};
        if (i > 0) {
            bn.gamma = p->bn_w[i - 1]; bn.beta = p->bn_b[i - 1]; bn.rm = p->bn_rm[i - 1]; bn.rv = p->bn_rv[i - 1];
            bn.stat = S.bnstat[i - 1]; bn.sums = W.acc64 + (long)(i - 1) * NCOPY * 64;
        }
        if (!(i == 0 && skip_first_tcn)) {
            tcn_fwd_kernel<BF16><<<(unsigned)cdiv(npos, 128), 256, 0, st>>>(i == 0 ? S.x0 : S.y[i - 1], bn, npos, Tin, Tout, dil, W.wcat + i * 4096,
                                                                            W.bcat + i * 64, S.tf[i], S.sg[i], S.cat[i], S.zlast, i);
            STEP_LAUNCH_CHECK("tcn_fwd");
        }
        if (first_tcn_only) break;
        if (i == NL - 1 && !(dead_bn7 && training)) break;
        STEP_TRY(nconv_fwd3(S.Pstk, S.PT16, S.cat[i], 0, 0, 1, B, N, Tout, BF16, W.xT, st));      // slots 1,3,5 = P_s z
        STEP_TRY(nconv_fwd3(S.Pstk, S.PT16, S.cat[i], 1, 2, 2, B, N, Tout, BF16, W.xT, st));      // slots 2,4,6 = P_s (P_s z)
        const XIn xin = {i == 0 ? S.x0 : S.y[i - 1], i == 0 ? nullptr : S.bnstat[i - 1]};
        const bool dead = i == NL - 1;              // the reference evaluates gconv[7] / bn[7] and drops the result (model.py:202-213)
        mix_fwd_kernel<BF16><<<(unsigned)cdiv(npos, 64), 256, 0, st>>>(S.cat[i], p->gconv_w[i], p->gconv_b[i], xin, npos, Tin, Tout, dil, drop_p,
                                                                       (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)i, dead ? W.m7 : S.mask[i],
                                                                       dead ? W.y7 : S.y[i], W.acc64 + (long)i * NCOPY * 64, dyn);
        STEP_LAUNCH_CHECK("mix_fwd");
        if (dead) {
            const BnFwd b7 = {p->bn_w[i], p->bn_b[i], p->bn_rm[i], p->bn_rv[i], 1, momentum, S.bnstat[i], (double)npos, W.acc64 + (long)i * NCOPY * 64};
            bn_stats_only_kernel<<<1, 64, 0, st>>>(b7);
            STEP_LAUNCH_CHECK("bn7_stats");
        }
    }
    return STEP_OK;
}

static int pack_weights(const StepGwnetParams* p, const Work& W, hipStream_t st) {
    GatePtrs gp;
    SkipPtrs sp;
    MixPtrs mp;
    for (int i = 0; i < NL; ++i) {
        gp.wf[i] = p->filter_w[i]; gp.bf[i] = p->filter_b[i]; gp.wg[i] = p->gate_w[i]; gp.bg[i] = p->gate_b[i];
        sp.w[i] = p->skip_w[i]; sp.b[i] = p->skip_b[i];
        if (i < NL - 1) mp.w[i] = p->gconv_w[i];
    }
    pack_weights_kernel<<<384 + 7 * 28, 256, 0, st>>>(gp, sp, mp, W.wcat, W.wcatT, W.bcat, W.wskip, W.bsum, W.wmixT);
    STEP_LAUNCH_CHECK("pack_weights");
    return STEP_OK;
}

extern "C" int step_gwnet_forward(const float* hist, int B, int N, int Cin, const float* hidden_last, const float* adj,
                                  const StepGwnetParams* p, int training, float dropout_p, uint64_t seed, float momentum,
                                  float* saved, float* work, float* pred, void* stream) {
    return step_gwnet_forward_phase(hist, B, N, Cin, hidden_last, adj, p, training, dropout_p, seed, momentum, saved, work, pred, 0, stream);
}

extern "C" int step_gwnet_forward_phase(const float* hist, int B, int N, int Cin, const float* hidden_last, const float* adj,
                                        const StepGwnetParams* p, int training, float dropout_p, uint64_t seed, float momentum,
                                        float* saved, float* work, float* pred, int phase, void* stream) {
    return step_gwnet_forward_phase_dyn(hist, B, N, Cin, hidden_last, adj, p, training, dropout_p, seed, momentum, saved, work, pred, phase, nullptr,
                                        stream);
}
extern "C" int step_gwnet_forward_phase_dyn(const float* hist, int B, int N, int Cin, const float* hidden_last, const float* adj,
                                            const StepGwnetParams* p, int training, float dropout_p, uint64_t seed, float momentum,
                                            float* saved, float* work, float* pred, int phase, const StepDynState* dyn, void* stream) {
    STEP_REQUIRE(p && saved && work && phase >= 0 && phase <= 6, "gwnet_forward: null argument / bad phase");
    // phase 1 in two: 5 = what does not read the sampled adjacency (start convolution, adaptive support, weight packing, layer 0's gated
    // TCN), 6 = the rest (random-walk supports, bf16 stacks, hops and mixes of all layers) -- the caller can run 5 next to the graph learner
    const bool do_prep = phase == 0 || phase == 1 || phase == 5, do_layers = phase == 0 || phase == 1 || phase == 6;
    const bool do_his = phase == 0 || phase == 2 || phase == 3, do_head = phase == 0 || phase == 2 || phase == 4;
    STEP_REQUIRE(!do_prep || hist, "gwnet_forward: the layer phase needs hist");
    STEP_REQUIRE(!do_layers || adj, "gwnet_forward: the layer phase needs adj");
    STEP_REQUIRE(!do_his || hidden_last, "gwnet_forward: the fc_his phase needs hidden_last");
    STEP_REQUIRE(!do_head || pred, "gwnet_forward: the head phase needs pred");
    STEP_REQUIRE(B > 0 && N > 0 && Cin >= 2, "gwnet_forward: bad sizes B=%d N=%d C=%d", B, N, Cin);
    hipStream_t st = (hipStream_t)stream;
    const bool train = (training & 1) != 0;          // bit 1 (value 2): also the dead bn.7 statistics, see the header
    const bool use_drop = train && dropout_p > 0.f;
    Saved S = carve_saved(saved, B, N, use_drop);
    Work W = carve_work(work, B, N, false);
    const long BN = (long)B * N;
    const int allbf16 = p->gemm_bf16;      // bf16 mode: every contraction of this file on the bf16 matrix cores (the K=10 / dpred-transposed ones stay f32)

    const long NN = (long)N * N;
    const bool dead_bn7 = (training & 2) != 0;
    if (do_prep) {
        STEP_TRY(zero((float*)W.acc64, 2L * NL * NCOPY * 64, st));
        start_conv_kernel<<<g1(BN * 13 * C), 256, 0, st>>>(hist, B, N, Cin, p->start_w, p->start_b, S.x0);
        STEP_LAUNCH_CHECK("start_conv");
        {   // adaptive support softmax(relu(E1 E2)) (model.py:165), replicated per sample into stack slot 2
            StepGemm g = gemm_desc(N, N, 10, p->nodevec1, 10, 1, p->nodevec2, N, 1, S.Madp, N);
            STEP_TRY(step_gemm_launch(g, st));
            softmax_relu_rows_kernel<<<N, 256, 0, st>>>(S.Madp, N, S.Pa);
            replicate_adp_kernel<<<g1(NN), 256, 0, st>>>(S.Pa, N, B, S.Pstk + 2 * B * NN, S.PTstk + 2 * B * NN);
            STEP_LAUNCH_CHECK("adp_softmax");
        }
        STEP_TRY(pack_weights(p, W, st));
        if (allbf16) STEP_TRY(gwnet_layers_forward<true>(p, S, W, B, N, train, use_drop ? dropout_p : 0.f, seed, momentum, dead_bn7, dyn, st, true, false));
        else STEP_TRY(gwnet_layers_forward<false>(p, S, W, B, N, train, use_drop ? dropout_p : 0.f, seed, momentum, dead_bn7, dyn, st, true, false));
    }
    if (do_layers) {
        // random-walk supports of the sampled adjacency (model.py:160-166)
        row_sums_kernel<<<(unsigned)BN, 256, 0, st>>>(adj, N, S.rs, S.cs);
        {
            int slices = cdiv(512, cdiv(N, 64) * B);          // ~2 blocks per compute unit
            if (slices > cdiv(N, 64)) slices = cdiv(N, 64);
            if (slices < 1) slices = 1;
            col_sums_kernel<<<dim3(cdiv(N, 64), slices, B), 256, 0, st>>>(adj, N, S.cs);
        }
        rw_build_kernel<<<dim3(cdiv(N, 32), cdiv(N, 32), B), 256, 0, st>>>(adj, N, S.rs, S.cs, S.Pstk, S.Pstk + B * NN, S.PTstk,
                                                                           S.PTstk + B * NN);
        STEP_LAUNCH_CHECK("rw_build");
        if (p->gemm_bf16) {
            const long rows = 3L * B * N;
            stacks_to_bf16_kernel<<<dim3((unsigned)cdiv(rows * (n8(N) / 2), 256), 2), 256, 0, st>>>(S.Pstk, S.PTstk, rows, N, n8(N), S.P16,
                                                                                                   S.PT16);
            STEP_LAUNCH_CHECK("stacks_to_bf16");
        }
        // 8 layers: gated TCN (one kernel; layer 0's ran with the preparation), two diffusion hops (the three supports per launch), gcn mix +
        // dropout + residual + BatchNorm statistics (one kernel); the BatchNorm transform itself is applied by the next layer's reads
        if (allbf16) STEP_TRY(gwnet_layers_forward<true>(p, S, W, B, N, train, use_drop ? dropout_p : 0.f, seed, momentum, dead_bn7, dyn, st, false, true));
        else STEP_TRY(gwnet_layers_forward<false>(p, S, W, B, N, train, use_drop ? dropout_p : 0.f, seed, momentum, dead_bn7, dyn, st, false, true));
    }

    // head (model.py:215-220).  fc_his (phase 3) is the only part that needs the TSFormer's hidden state, and needs nothing else: the
    // caller can queue it behind the encoder while the layers still run on another stream, and the rest (phase 4) after both
    if (do_his) {
        StepGemm g = gemm_desc((int)BN, CE, HID, hidden_last, HID, 1, p->fc_his0_w, 1, HID, S.h1, CE);
        g.bias = p->fc_his0_b; g.relu = 1;
        g.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g, st));
        StepGemm g2 = gemm_desc((int)BN, CS, CE, S.h1, CE, 1, p->fc_his2_w, 1, CE, S.h2, CS);
        g2.bias = p->fc_his2_b; g2.relu = 1;
        g2.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g2, st));
    }
    if (do_head) {
        // skip = sum_i Wskip_i z_i[last step] + sum_i b_i: the 8 skip convolutions as one K = 256 contraction
        StepGemm gs = gemm_desc((int)BN, CS, CS, S.zlast, CS, 1, W.wskip, 1, CS, S.skip, CS);
        gs.bias = W.bsum;
        gs.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gs, st));
        head_combine_kernel<<<g1(BN * CS), 256, 0, st>>>(S.skip, S.h2, BN * CS, S.xh);
        STEP_LAUNCH_CHECK("head_combine");
        StepGemm g3 = gemm_desc((int)BN, CE, CS, S.xh, CS, 1, p->end1_w, 1, CS, S.e1, CE);
        g3.bias = p->end1_b; g3.relu = 1;
        g3.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g3, st));
        // pred[b][o][n] = e1[b,n,:] . W2[o,:] + b2[o]      (written directly as [B,12,N], step.py:65)
        StepGemm g4 = gemm_desc(N, OUT, CE, S.e1, CE, 1, p->end2_w, 1, CE, pred, 1);
        g4.batch = B; g4.sab = (long)N * CE; g4.scb = (long)OUT * N; g4.scn = N;
        g4.bias = p->end2_b;
        g4.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(g4, st));
    }
    return STEP_OK;
}

template <bool BF16>
static int gwnet_layers_backward(const StepGwnetParams* p, const StepGwnetParams* grads, const Saved& S, const Work& W, int B, int N,
                                 float** dx0, hipStream_t st, AuxLane& lane, AuxLane& leaves, hipEvent_t* adj_done) {
    const long BN = (long)B * N;
    float* dx_next = nullptr;
    float* dxbuf[2] = {W.dxa, W.dxb};
    // dP_s = sum over layers, time steps and channels of x1_s (x) d_x2_s + z (x) d_x1_s: a segmented contraction, K = 3264, cut into
    // three layer ranges that leave on the auxiliary stream as soon as their layers' hop gradients exist (they accumulate into the
    // same stack in stream order); only the supports' backward at the very end reads the result
    {
        AdjOffsets o;
        for (int i = 0; i < NL - 1; ++i) { o.cat[i] = S.cat[i] - S.cat[0]; o.dcat[i] = W.dcat[i] - W.dcat[0]; }
        adj_ktab_kernel<<<3, 128, 0, st>>>(o, N, W.ktab);
        STEP_LAUNCH_CHECK("adj_ktab");
    }
    // (STEP_ADJ_PIECES=0: one contraction after the loop, A/B measurements)
    static const bool adj_pieces = []() { const char* e = getenv("STEP_ADJ_PIECES"); return !(e && e[0] == '0'); }();
    bool adj_first = true;
    auto adj_piece = [&](int lo, int hi, hipStream_t on_stream) -> int {       // layers lo .. hi (inclusive) of the table
        int seg0 = 0, nseg = 0;
        for (int i = 0; i < lo; ++i) seg0 += 2 * TOUT[i];
        for (int i = lo; i <= hi; ++i) nseg += 2 * TOUT[i];
        StepGemm g = gemm_desc(N, N, 32 * nseg, S.cat[0], 0, 1, W.dcat[0], 1, 0, W.dPstk, N);
        g.batch = 3 * B; g.batch0 = B;
        g.scb = (long)N * N; g.scb1 = (long)B * N * N;
        g.compute_bf16 = BF16;
        g.accumulate = adj_first ? 0 : 1;
        adj_first = false;
        return step_gemm_segmented_launch(g, W.ktab + seg0, ADJ_NSEG, on_stream);
    };
    for (int i = NL - 1; i >= 0; --i) {
        const int Tin = TIN[i], Tout = TOUT[i], dil = DIL[i];
        const long npos = BN * Tout;
        const float* cat = S.cat[i];
        if (i < NL - 1) {
            // BatchNorm_i backward (sums left by the previous iteration's tcn_bwd) + dropout + mix data gradient
            const BnBwd bn = {p->bn_w[i], S.bnstat[i], grads->bn_w[i], grads->bn_b[i], (double)npos, W.acc64 + (long)i * NCOPY * 64};
            mix_bwd_kernel<BF16><<<(unsigned)cdiv(npos, 128), 256, 0, st>>>(dx_next, S.y[i], bn, S.mask[i], W.wmixT + i * (C * CAT), npos, W.dres, W.dh[i],
                                                                            W.dcat[i]);
            STEP_LAUNCH_CHECK("mix_bwd");
            // diffusion hops, the three supports per launch: slots (1,2) <- P_f, (3,4) <- P_b, (5,6) <- P_a
            // (the adjacency gradients x (x) d_hop of all layers are contracted in one launch after the loop: every dcat[i] is kept)
            STEP_TRY(nconv_bwd_data3(S.PTstk, S.P16, W.dcat[i], 2, 1, 2, B, N, Tout, BF16, W.xT, st));          // d_x1 += P (d_x2)
            STEP_TRY(nconv_bwd_data3(S.PTstk, S.P16, W.dcat[i], 1, 0, 0, B, N, Tout, BF16, W.xT, st));          // d_z  += sum_s P_s (d_x1_s)
            // slots 1..6 of dcat[i] are final here (tcn_bwd only reads dcat).  The LAST adjacency-gradient piece is what the supports' backward
            // waits for after the loop, so it is the smallest possible -- layer 0 alone -- and is queued before layer 0's tcn_bwd (with layers
            // 0..1 as one piece after the loop the main stream waited 111 us for it at PEMS04, profiles/r03_ac_C2_step_timeline.md)
            if (adj_pieces && i == 0) { STEP_TRY(adj_piece(0, 0, lane.fork())); *adj_done = lane.done(); }       // (what the supports' backward waits for)
        }
        // gated TCN (+ the skip branch's gradient at the last step, + col2im, + BatchNorm_{i-1}'s backward sums)
        float* dx = dxbuf[i & 1];
        tcn_bwd_kernel<BF16><<<(unsigned)cdiv(BN, TB_ROWS / Tout), 256, 0, st>>>(i < NL - 1 ? W.dcat[i] : nullptr, W.dskip, i, S.tf[i], S.sg[i],
                                                                                W.wcatT + i * 4096, i < NL - 1 ? W.dres : nullptr, BN, Tin, Tout, dil,
                                                                                W.dpre[i], dx, i > 0 ? S.y[i - 1] : nullptr,
                                                                                i > 0 ? S.bnstat[i - 1] : nullptr,
                                                                                i > 0 ? W.acc64 + (long)(i - 1) * NCOPY * 64 : nullptr);
        STEP_LAUNCH_CHECK("tcn_bwd");
        const XIn xin = {i == 0 ? S.x0 : S.y[i - 1], i == 0 ? nullptr : S.bnstat[i - 1]};
        // ONE event on the main stream per layer (each record costs the dependent chain ~6 us between two kernels): the leaves of this
        // layer -- mix weight gradient, gate / filter weight gradient -- and the adjacency-gradient piece of the layers finished so far
        // all wait for it
        hipEvent_t ev = leaves.on ? leaves.mark() : lane.mark();
        // (the adjacency piece first: the pieces are what the backward waits for at the end, the leaves behind them on the same stream are not)
        if (adj_pieces && i == 4) STEP_TRY(adj_piece(4, 6, lane.after(ev)));
        if (adj_pieces && i == 2) STEP_TRY(adj_piece(2, 3, lane.after(ev)));
        if (adj_pieces && i == 1) STEP_TRY(adj_piece(1, 1, lane.after(ev)));
        hipStream_t leaf = leaves.after(ev);
        if (i < NL - 1) {
            StepGemm gw = gemm_desc(C, CAT, (int)npos, W.dh[i], 1, C, cat, CAT, 1, grads->gconv_w[i], CAT);
            gw.accumulate = 2; gw.splitk = -1;
            gw.a_rowsum = grads->gconv_b[i];
            gw.compute_bf16 = BF16; STEP_TRY(step_gemm_launch(gw, leaf));          // leaf: nothing in the backward reads it
        }
        im2col_kernel<<<g1(npos * 64), 256, 0, leaf>>>(xin, BN, Tin, Tout, dil, W.xcat[i]);
        STEP_LAUNCH_CHECK("im2col");
        StepGemm gw = gemm_desc(64, 64, (int)npos, W.dpre[i], 1, 64, W.xcat[i], 64, 1, W.dwcat + i * 4096, 64);
        gw.accumulate = 2; gw.splitk = -1;
        gw.a_rowsum = W.dbcat + i * 64;
        gw.compute_bf16 = BF16; STEP_TRY(step_gemm_launch(gw, leaf));
        dx_next = dx;
    }
    *dx0 = dx_next;
    if (!adj_pieces) STEP_TRY(adj_piece(0, NL - 2, lane.fork()));
    return STEP_OK;
}

extern "C" int step_gwnet_backward(const float* hist, int B, int N, int Cin, const float* hidden_last, const StepGwnetParams* p,
                                   const float* saved, float* work, const float* dpred, const StepGwnetParams* grads,
                                   float* dadj, int dropout, void* aux_stream, void* leaf_stream, void* stream) {
    STEP_REQUIRE(hist && hidden_last && p && saved && work && dpred && grads && dadj, "gwnet_backward: null argument");
    STEP_REQUIRE(B > 0 && N > 0 && Cin >= 2, "gwnet_backward: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    // two lanes: the adjacency-gradient contractions (`lane`, joined before the supports' backward needs their result) and the pure
    // leaves (`leaves`: parameter gradients only, joined by the caller).  On ONE in-order stream the contractions queued behind the weight
    // gradients of the same layers and the join waited 136 us for them at PEMS04 (profiles/r03_ab_C2_step_timeline.md)
    AuxLane lane(st, (hipStream_t)aux_stream);
    AuxLane leaves(st, leaf_stream ? (hipStream_t)leaf_stream : (hipStream_t)aux_stream, 64);
    Saved S = carve_saved((float*)saved, B, N, dropout != 0);
    Work W = carve_work(work, B, N, true);
    const long BN = (long)B * N;
    const int allbf16 = p->gemm_bf16;
    auto split_for = [](long) { return -1; };      // -1: step_gemm picks a split that fills the chip

    STEP_TRY(pack_weights(p, W, st));
    STEP_TRY(zero(W.dwcat, (long)((float*)W.acc64 - W.dwcat) + 2L * NL * NCOPY * 64, st));       // dwcat, dbcat, dwskip, acc64
    const long NN = (long)N * N;

    // ---------------------------------------------------------------- head (model.py:215-220)
    // Main stream: the data-gradient chain d_e1 -> d_xh -> d zlast (what the layers need).  Everything else here -- the weight and
    // bias gradients and the whole fc_his branch -- is a leaf of the backward and goes to the auxiliary stream (leaves.fork()).
    {
        // d_e1[b,n,:] = sum_o dpred[b][o][n] W2[o,:], masked by relu(e1)
        StepGemm g = gemm_desc(N, CE, OUT, dpred, 1, N, p->end2_w, CE, 1, W.d_e1, CE);
        g.batch = B; g.sab = (long)OUT * N; g.scb = (long)N * CE;
        STEP_TRY(step_gemm_launch(g, st));
        relu_bwd_kernel<<<g1(BN * CE), 256, 0, st>>>(W.d_e1, S.e1, BN * CE);
        StepGemm gx = gemm_desc((int)BN, CS, CE, W.d_e1, CE, 1, p->end1_w, CS, 1, W.d_xh, CS);
        gx.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gx, st));
        relu_bwd_kernel<<<g1(BN * CS), 256, 0, st>>>(W.d_xh, S.xh, BN * CS);       // = d skip = d h2 (pre-mask)
        {   // the 8 skip convolutions, data side: d zlast = d skip @ Wskip (every layer's last-step gradient)
            hipStream_t leaf = leaves.fork();          // ONE event for all leaves of the head (each record is a ~6 us bubble on the main chain)
            StepGemm gz = gemm_desc((int)BN, CS, CS, W.d_xh, CS, 1, W.wskip, CS, 1, W.dskip, CS);
            gz.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gz, st));
            {   // dW2[o,:] += sum_{b,n} dpred[b][o][n] e1[b,n,:]   (batches accumulate atomically)
                StepGemm gw = gemm_desc(OUT, CE, N, dpred, N, 1, S.e1, CE, 1, grads->end2_w, CE);
                gw.batch = B; gw.sab = (long)OUT * N; gw.sbb = (long)N * CE; gw.scb = 0; gw.accumulate = 2;
                STEP_TRY(step_gemm_launch(gw, leaf));
                rowsum_mod_kernel<<<B * OUT, 256, 0, leaf>>>(dpred, N, OUT, grads->end2_b);
                STEP_LAUNCH_CHECK("end2_bias_grad");
                StepGemm gw1 = gemm_desc(CE, CS, (int)BN, W.d_e1, 1, CE, S.xh, CS, 1, grads->end1_w, CS);
                gw1.accumulate = 2; gw1.splitk = split_for(BN);
                gw1.a_rowsum = grads->end1_b;                         // bias gradient = row sums of the same A
                gw1.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw1, leaf));
            }
            // ---- leaves that read d_xh
            // the 8 skip biases all receive colsum(d skip)
            STEP_TRY(zero(W.bsum, CS, leaf));
            STEP_TRY(step_colsum_launch(W.d_xh, BN, CS, CS, W.bsum, leaf));
            MPtr8 gb;
            for (int i = 0; i < NL; ++i) gb.p[i] = grads->skip_b[i];
            add_to8_kernel<<<1, 256, 0, leaf>>>(gb, W.bsum, CS);
            // dWskip = d skip^T zlast
            StepGemm gws = gemm_desc(CS, CS, (int)BN, W.d_xh, 1, CS, S.zlast, CS, 1, W.dwskip, CS);
            gws.accumulate = 2; gws.splitk = split_for(BN);
            gws.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gws, leaf));
            // fc_his (reads the TSFormer's last hidden state; nothing upstream of it takes a gradient)
            if (hipMemcpyAsync(W.d_h2, W.d_xh, (size_t)BN * CS * sizeof(float), hipMemcpyDeviceToDevice, leaf) != hipSuccess) {
                step_set_error("gwnet_backward: copy failed");
                return STEP_ERR_HIP;
            }
            relu_bwd_kernel<<<g1(BN * CS), 256, 0, leaf>>>(W.d_h2, S.h2, BN * CS);
            StepGemm gw2 = gemm_desc(CS, CE, (int)BN, W.d_h2, 1, CS, S.h1, CE, 1, grads->fc_his2_w, CE);
            gw2.accumulate = 2; gw2.splitk = split_for(BN);
            gw2.a_rowsum = grads->fc_his2_b;
            gw2.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw2, leaf));
            StepGemm gh1 = gemm_desc((int)BN, CE, CS, W.d_h2, CS, 1, p->fc_his2_w, CE, 1, W.d_h1, CE);
            gh1.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gh1, leaf));
            relu_bwd_kernel<<<g1(BN * CE), 256, 0, leaf>>>(W.d_h1, S.h1, BN * CE);
            StepGemm gw0 = gemm_desc(CE, HID, (int)BN, W.d_h1, 1, CE, hidden_last, HID, 1, grads->fc_his0_w, HID);
            gw0.accumulate = 2; gw0.splitk = split_for(BN);
            gw0.a_rowsum = grads->fc_his0_b;
            gw0.compute_bf16 = allbf16; STEP_TRY(step_gemm_launch(gw0, leaf));
        }
    }

    // ---------------------------------------------------------------- WaveNet layers, reversed
    float* dx0 = nullptr;
    hipEvent_t adj_done = nullptr;
    if (allbf16) STEP_TRY(gwnet_layers_backward<true>(p, grads, S, W, B, N, &dx0, st, lane, leaves, &adj_done));
    else STEP_TRY(gwnet_layers_backward<false>(p, grads, S, W, B, N, &dx0, st, lane, leaves, &adj_done));
    // the adjacency gradients are complete from here on (the event behind the last piece; leaves queued on that stream after it -- layer
    // 0's gate gradient -- no longer hold the main stream back: 100 us at PEMS07, profiles/r03_aj_C4_step_timeline.md)
    STEP_TRY(lane.wait_done(adj_done));
    // ---------------------------------------------------------------- supports
    // What the caller's next kernels wait for is dadj alone: the random-walk normalisations' backward stays on the main stream.  The
    // rest of the tail only writes parameter gradients -- unpacking the gate / skip gradients and the adaptive adjacency's chain
    // (softmax(relu(E1 E2)) backward + the two node-embedding gradients: 88 us at PEMS04, 180 us at PEMS07 of badly shaped K = N products)
    // -- and goes to the auxiliary stream WITHOUT a join: the caller orders the first reader of `grads` after aux_stream.
    static const bool tail_leaves = []() { const char* e = getenv("STEP_TAIL_LEAVES"); return !(e && e[0] == '0'); }();      // (A/B knob)
    if (!tail_leaves) STEP_TRY(leaves.join());
    hipStream_t leaf = tail_leaves ? leaves.fork() : st;
    row_dot_kernel<<<(unsigned)BN, 256, 0, st>>>(W.dPstk, S.Pstk, N, W.rf);
    row_dot_kernel<<<(unsigned)BN, 256, 0, st>>>(W.dPstk + B * NN, S.Pstk + B * NN, N, W.rb);
    rw_bwd_kernel<<<dim3(cdiv(N, 32), cdiv(N, 32), B), 256, 0, st>>>(W.dPstk, W.dPstk + B * NN, N, S.rs, S.cs, W.rf, W.rb, dadj);
    STEP_LAUNCH_CHECK("rw_bwd");
    {
        GatePtrs gg;
        SkipPtrs sg;
        for (int i = 0; i < NL; ++i) {
            gg.wf[i] = grads->filter_w[i]; gg.bf[i] = grads->filter_b[i]; gg.wg[i] = grads->gate_w[i]; gg.bg[i] = grads->gate_b[i];
            sg.w[i] = grads->skip_w[i]; sg.b[i] = nullptr;
        }
        start_conv_bwd_kernel<<<128, 256, 0, leaf>>>(hist, B, N, Cin, dx0, grads->start_w, grads->start_b);      // (a leaf too: 37 us at PEMS04)
        STEP_LAUNCH_CHECK("start_conv_bwd");
        unpack_gate_grad_kernel<<<dim3(16, NL), 256, 0, leaf>>>(W.dwcat, W.dbcat, gg);
        unpack_skip_grad_kernel<<<CS, CS, 0, leaf>>>(W.dwskip, sg);
    }
    {   // adaptive adjacency softmax(relu(E1 E2), dim=1); its gradient is the sum over samples of stack slot 2
        float* rdot = W.y7;          // [N] row dots (y7 is a forward-only buffer: B N 32 floats, free here; rf / rb belong to the chain above)
        sum_batches_kernel<<<g1(NN), 256, 0, leaf>>>(W.dPstk + 2 * B * NN, NN, B, W.dPa);
        row_dot_kernel<<<N, 256, 0, leaf>>>(W.dPa, S.Pa, N, rdot);
        softmax_relu_bwd_kernel<<<g1(NN), 256, 0, leaf>>>(S.Madp, S.Pa, W.dPa, rdot, N, W.dM);
        STEP_LAUNCH_CHECK("adp_bwd");
        StepGemm g1_ = gemm_desc(N, 10, N, W.dM, N, 1, p->nodevec2, 1, N, grads->nodevec1, 10);
        g1_.accumulate = 1;
        STEP_TRY(step_gemm_launch(g1_, leaf));
        StepGemm g2_ = gemm_desc(10, N, N, p->nodevec1, 1, 10, W.dM, N, 1, grads->nodevec2, N);
        g2_.accumulate = 1;
        STEP_TRY(step_gemm_launch(g2_, leaf));
    }
    STEP_LAUNCH_CHECK("rw_bwd");
    return STEP_OK;
}
