// What one gfx950 SIMD does with the instruction mix of the encoder's key-tile loop (round 5): dependent / independent MFMA chains, v_exp_f32,
// plain and packed vector instructions, alone and with one or two more waves on the SAME SIMD playing another role.  One workgroup of
// 256 * k threads per compute unit: wave w runs on SIMD w % 4 (the k waves w, w + 4, w + 8 share a SIMD) and plays role[w / 4].
//     hipcc --offload-arch=gfx950 -O3 -o simd_probe tools/simd_probe.cpp && ./simd_probe
// Output: shader cycles (s_memtime) per instruction of each role, per configuration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum Role { IDLE = 0, MFMA_DEP, MFMA_2ACC, EXP, CND, PKADD, CVT, LOOP_NODROP, LOOP_DROP, LOOP_DROP_IL, MFMA_4ACC };
static const char* role_name[] = {"idle", "mfma dependent chain", "mfma two accumulators", "v_exp_f32 x16", "v_cndmask x16", "v_pk_add_f32 x8",
                                  "v_cvt_pk x8", "loop (16 exp, 8 cvt, 2+2 dependent mfma)", "loop + 16 cndmask + 8 pk_add",
                                  "loop + masks, mfmas interleaved among the exps", "mfma four accumulators"};

struct Args { int roles[3]; int iters; unsigned long long* out; float* sink; };

__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

#define EXP16(v)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define EXPN(v, lo, hi)                                                                           \
    _Pragma("unroll") for (int i = lo; i < hi; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));

__global__ __launch_bounds__(768) void probe(Args A) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = A.roles[wave >> 2];
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 c0, c1, c2, c3, s, o;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; s[i] = -0.001f * i; o[i] = 0.f; }
    f32x2 l0 = {0.f, 0.f}, l1 = {0.f, 0.f};
    const unsigned long long m0 = 0x5555555555555555ull ^ (unsigned long long)blockIdx.x, m1 = ~m0;
    long ninst = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int N = A.iters;
    if (role == MFMA_DEP) {
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c0 = mfma(a, b, c0);
        }
        ninst = 8L * N;
    } else if (role == MFMA_2ACC) {
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { c0 = mfma(a, b, c0); c1 = mfma(a, b, c1); }
        }
        ninst = 8L * N;
    } else if (role == MFMA_4ACC) {
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k) { c0 = mfma(a, b, c0); c1 = mfma(a, b, c1); c2 = mfma(a, b, c2); c3 = mfma(a, b, c3); }
        }
        ninst = 8L * N;
    } else if (role == EXP) {
#pragma unroll 1
        for (int it = 0; it < N; ++it) { EXP16(s); }
        ninst = 16L * N;
    } else if (role == CND) {
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(s[i]) : "s"((i & 1) ? m0 : m1));
        }
        ninst = 16L * N;
    } else if (role == PKADD) {
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(l0) : "v"(f32x2{s[i], s[i + 1]}));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(l1) : "v"(f32x2{s[i + 2], s[i + 3]}));
            }
        }
        ninst = 8L * N;
    } else if (role == CVT) {
        unsigned int p[8];
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i]) : "v"(s[2 * i]), "v"(s[2 * i + 1]));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(p[i]));
        }
        ninst = 8L * N;
    } else if (role == LOOP_NODROP || role == LOOP_DROP || role == LOOP_DROP_IL) {
        // the shape of the encoder's fast key-tile step: exponentials of the current score tile, the next tile's two (dependent) score products,
        // (keep masks, row sums), packs, the two (dependent) P V products chained through o
        f32x16 cur = s, nxt = s;
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
            if (role == LOOP_DROP_IL) {
                EXPN(cur, 0, 6);
                nxt = mfma(a, b, c1);                  // c1 == 0: no accumulator input dependency
                __builtin_amdgcn_sched_barrier(0);
                EXPN(cur, 6, 12);
                __builtin_amdgcn_sched_barrier(0);
                nxt = mfma(b, a, nxt);
                __builtin_amdgcn_sched_barrier(0);
                EXPN(cur, 12, 16);
            } else {
                EXP16(cur);
            }
            if (role != LOOP_NODROP) {
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(l0) : "v"(f32x2{cur[i], cur[i + 1]}));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(l1) : "v"(f32x2{cur[i + 2], cur[i + 3]}));
                }
            }
            if (role != LOOP_DROP_IL) {
                __builtin_amdgcn_sched_barrier(0);
                nxt = mfma(a, b, c1);
                nxt = mfma(b, a, nxt);
            }
            if (role != LOOP_NODROP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(cur[i]) : "s"((i & 1) ? m0 : m1));
            }
            f16x8 p0, p1;
            {
                unsigned int p[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i]) : "v"(cur[2 * i]), "v"(cur[2 * i + 1]));
                p0 = __builtin_bit_cast(f16x8, *(uint4*)&p[0]);
                p1 = __builtin_bit_cast(f16x8, *(uint4*)&p[4]);
            }
            o = mfma(a, p0, o);
            o = mfma(b, p1, o);
            __builtin_amdgcn_sched_barrier(0);
            // swap roles of the two tiles (register renaming by unrolling would double the code; a move of 16 registers costs 16 VALU: instead
            // the next tile's scores are folded back cheaply)
            cur[0] += nxt[0] * 1e-30f;
        }
        ninst = N;
    } else {
#pragma unroll 1
        for (int it = 0; it < N; ++it) __builtin_amdgcn_s_sleep(8);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i] + c2[i] + c3[i] + s[i] + o[i];
    acc += l0[0] + l0[1] + l1[0] + l1[1];
    if (acc == 12345.678f) A.sink[0] = acc;
    if (lane == 0 && ninst > 0) {
        A.out[(blockIdx.x * 12 + wave) * 2] = t1 - t0;
        A.out[(blockIdx.x * 12 + wave) * 2 + 1] = (unsigned long long)ninst;
    }
}

int main() {
    struct Cfg { int r[3]; const char* what; };
    const std::vector<Cfg> cfgs = {
        {{MFMA_DEP, 0, 0}, "one wave per SIMD"}, {{MFMA_2ACC, 0, 0}, "one wave per SIMD"}, {{MFMA_4ACC, 0, 0}, "one wave per SIMD"},
        {{EXP, 0, 0}, "one wave per SIMD"}, {{CND, 0, 0}, "one wave per SIMD"}, {{PKADD, 0, 0}, "one wave per SIMD"}, {{CVT, 0, 0}, "one wave per SIMD"},
        {{EXP, EXP, 0}, "two waves per SIMD"}, {{EXP, EXP, EXP}, "three waves per SIMD"}, {{CND, CND, CND}, "three waves per SIMD"},
        {{MFMA_DEP, MFMA_DEP, 0}, "two waves per SIMD"}, {{MFMA_DEP, MFMA_DEP, MFMA_DEP}, "three waves per SIMD"},
        {{MFMA_DEP, EXP, 0}, "two waves per SIMD"}, {{MFMA_DEP, CND, 0}, "two waves per SIMD"}, {{MFMA_DEP, EXP, EXP}, "three waves per SIMD"},
        {{MFMA_2ACC, EXP, EXP}, "three waves per SIMD"}, {{EXP, MFMA_DEP, 0}, "two waves per SIMD (the older wave does the exponentials)"},
        {{LOOP_NODROP, 0, 0}, "one wave per SIMD"}, {{LOOP_NODROP, LOOP_NODROP, 0}, "two"}, {{LOOP_NODROP, LOOP_NODROP, LOOP_NODROP}, "three"},
        {{LOOP_DROP, 0, 0}, "one wave per SIMD"}, {{LOOP_DROP, LOOP_DROP, 0}, "two"}, {{LOOP_DROP, LOOP_DROP, LOOP_DROP}, "three"},
        {{LOOP_DROP_IL, 0, 0}, "one wave per SIMD"}, {{LOOP_DROP_IL, LOOP_DROP_IL, 0}, "two"}, {{LOOP_DROP_IL, LOOP_DROP_IL, LOOP_DROP_IL}, "three"},
    };
    unsigned long long* d_out; float* d_sink;
    const int grid = 256;
    hipMalloc(&d_out, grid * 12 * 2 * 8); hipMalloc(&d_sink, 4);
    std::vector<unsigned long long> h(grid * 12 * 2);
    for (const Cfg& c : cfgs) {
        int k = c.r[2] ? 3 : c.r[1] ? 2 : 1;
        Args a; a.roles[0] = c.r[0]; a.roles[1] = c.r[1]; a.roles[2] = c.r[2]; a.iters = 2000; a.out = d_out; a.sink = d_sink;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d_out, 0, grid * 12 * 2 * 8);
            probe<<<grid, 256 * k, 0, 0>>>(a);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        printf("%-28s:", c.what);
        for (int g = 0; g < k; ++g) {
            double cyc = 0, n = 0; int cnt = 0;
            for (int blk = 0; blk < grid; ++blk)
                for (int w = 4 * g; w < 4 * g + 4; ++w)
                    if (h[(blk * 12 + w) * 2 + 1]) { cyc += (double)h[(blk * 12 + w) * 2]; n += (double)h[(blk * 12 + w) * 2 + 1]; ++cnt; }
            if (cnt) printf("  [%s: %.1f cycles each]", role_name[c.r[g]], cyc / n);
        }
        printf("\n");
    }
    return 0;
}
