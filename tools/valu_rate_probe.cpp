// Issue cost of single vector instructions on one gfx950 SIMD, saturated (round 6): k = 1 / 2 / 3 waves per SIMD all running the same
// 16-instruction body over 16 independent registers, so what is measured is the SIMD's issue rate for that opcode, not a dependency chain.
// Second table: the same body on two waves next to one wave issuing a dependent chain of 32x32x16 matrix products (what a product costs
// the vector stream and the other way round).
//     hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe tools/valu_rate_probe.cpp && ./valu_rate_probe
// Output: shader cycles per instruction and SIMD = (cycles until the LAST wave of the SIMD is done) / (instructions of all its waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct Args { int roles[3]; int iters; unsigned long long* out; float* sink; };

#define R16(stmt) _Pragma("unroll") for (int i = 0; i < 16; ++i) { stmt; }
#define R8(stmt) _Pragma("unroll") for (int i = 0; i < 8; ++i) { stmt; }

enum {
    IDLE = 0, MFMA_DEP, EXP32, CND_S, CND_VCC, ADD, MUL, FMA, FMAC, PKADD, PKMUL, PKFMA, CVTBF, CVTF16, CVTRTZ, UNPK, UNPK_SDWA, AND, MOV, MAX, MAX3, PKMAXI16,
    EXP16, MIXLO, ADDU, PERM, PKMULH, PKFMAH, DOT2F16, RCP, LSHLADD, SUB, MED3, BFE, MOV64, EXECMOV, EXECFMAC, EXECMOV_PAIR, NROLES
};
static const char* role_name[NROLES] = {
    "idle", "mfma 32x32x16 dependent", "v_exp_f32", "v_cndmask_b32 (sgpr pair)", "v_cndmask_b32 (vcc, e32)", "v_add_f32", "v_mul_f32", "v_fma_f32",
    "v_fmac_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32", "v_cvt_f32_f16",
    "v_cvt_f32_f16 sdwa WORD_1", "v_and_b32", "v_mov_b32", "v_max_f32", "v_max3_f32", "v_pk_max_i16", "v_exp_f16", "v_fma_mixlo_f16", "v_add_u32",
    "v_perm_b32", "v_pk_mul_f16", "v_pk_fma_f16", "v_dot2_f32_f16", "v_rcp_f32", "v_lshl_add_u32", "v_sub_f32", "v_med3_f32", "v_bfe_i32", "v_mov_b64", "s_not_b64 exec + v_mov_b32 (per pair)", "s_mov_b64 exec + v_fmac_f32 (per pair)",
    "8 x (s_not_b64 exec, v_mov_b32), one restore (per mov)"};

__device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int ROLE>
__device__ __forceinline__ long run_role(int N, float (&s)[16], unsigned int (&p)[8], f32x2 (&q)[8], float (&acc_out)) {
    const unsigned long long m0 = 0x5555555555555555ull ^ (unsigned long long)blockIdx.x, m1 = ~m0;
    const int lane = threadIdx.x & 63;
    float c1 = 1.0001f, c2 = 0.5f;
    asm volatile("" : "+v"(c1), "+v"(c2));
    if constexpr (ROLE == MFMA_DEP) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        f32x16 c0;
        for (int i = 0; i < 16; ++i) c0[i] = 0.f;
#pragma unroll 1
        for (int it = 0; it < N; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) c0 = mfma(a, b, c0);
        }
        for (int i = 0; i < 16; ++i) acc_out += c0[i];
        return 8L * N;
    }
#define BODY16(stmt) _Pragma("unroll 1") for (int it = 0; it < N; ++it) { R16(stmt) } return 16L * N;
#define BODY8(stmt) _Pragma("unroll 1") for (int it = 0; it < N; ++it) { R8(stmt) R8(stmt) } return 16L * N;
    if constexpr (ROLE == EXP32) { BODY16(asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]))) }
    if constexpr (ROLE == CND_S) { BODY16(asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(s[i]) : "s"((i & 1) ? m0 : m1))) }
    if constexpr (ROLE == CND_VCC) {
        asm volatile("s_mov_b64 vcc, %0" :: "s"(m0) : "vcc");
        BODY16(asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc" : "+v"(s[i]) :: "vcc"))
    }
    if constexpr (ROLE == ADD) { BODY16(asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(s[i]) : "v"(c2))) }
    if constexpr (ROLE == SUB) { BODY16(asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(s[i]) : "v"(c2))) }
    if constexpr (ROLE == MUL) { BODY16(asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(s[i]) : "v"(c1))) }
    if constexpr (ROLE == FMA) { BODY16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c1), "v"(c2))) }
    if constexpr (ROLE == FMAC) { BODY16(asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(c1), "v"(c2))) }
    if constexpr (ROLE == PKADD) { BODY8(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]))) }
    if constexpr (ROLE == PKMUL) { BODY8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]))) }
    if constexpr (ROLE == PKFMA) { BODY8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]))) }
    if constexpr (ROLE == CVTBF) { BODY8(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p[i]) : "v"(s[2 * i]), "v"(s[2 * i + 1]))) }
    if constexpr (ROLE == CVTF16) { BODY8(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[i]) : "v"(s[2 * i]), "v"(s[2 * i + 1]))) }
    if constexpr (ROLE == CVTRTZ) { BODY8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(p[i]) : "v"(s[2 * i]), "v"(s[2 * i + 1]))) }
    if constexpr (ROLE == UNPK) { BODY16(asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(s[i]) : "v"(p[i & 7]))) }
    if constexpr (ROLE == UNPK_SDWA) { BODY16(asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(s[i]) : "v"(p[i & 7]))) }
    if constexpr (ROLE == AND) { BODY16(asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(s[i]) : "v"(p[i & 7]))) }
    if constexpr (ROLE == MOV) { BODY16(asm volatile("v_mov_b32_e32 %0, %1" : "=v"(s[i]) : "v"(p[i & 7]))) }
    if constexpr (ROLE == MOV64) { BODY8(asm volatile("v_mov_b64 %0, %1" : "=v"(q[i]) : "v"(q[(i + 1) & 7]))) }
    if constexpr (ROLE == MAX) { BODY16(asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(s[i]) : "v"(c2))) }
    if constexpr (ROLE == MAX3) { BODY16(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c1), "v"(c2))) }
    if constexpr (ROLE == MED3) { BODY16(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c1), "v"(c2))) }
    if constexpr (ROLE == PKMAXI16) { BODY8(asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]))) }
    if constexpr (ROLE == EXP16) { BODY16(asm volatile("v_exp_f16_e32 %0, %0" : "+v"(s[i]))) }
    if constexpr (ROLE == MIXLO) { BODY16(asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(p[i & 7]) : "v"(s[i]), "v"(c1), "v"(c2))) }
    if constexpr (ROLE == ADDU) { BODY16(asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]))) }
    if constexpr (ROLE == PERM) { BODY16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]))) }
    if constexpr (ROLE == PKMULH) { BODY8(asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]))) }
    if constexpr (ROLE == PKFMAH) { BODY8(asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]))) }
    if constexpr (ROLE == DOT2F16) { BODY16(asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(s[i]) : "v"(p[i & 7]), "v"(p[(i + 1) & 7]))) }
    if constexpr (ROLE == RCP) { BODY16(asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(s[i]))) }
    if constexpr (ROLE == LSHLADD) { BODY16(asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]))) }
    if constexpr (ROLE == BFE) { BODY16(asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(p[i & 7]))) }
    if constexpr (ROLE == EXECMOV) {
        BODY16(asm volatile("s_not_b64 exec, %1\n\tv_mov_b32 %0, 0\n\ts_mov_b64 exec, -1" : "+v"(s[i]) : "s"((i & 1) ? m0 : m1) : "scc"))
    }
    if constexpr (ROLE == EXECFMAC) {
        BODY16(asm volatile("s_mov_b64 exec, %3\n\tv_fmac_f32_e32 %0, %1, %2\n\ts_mov_b64 exec, -1" : "+v"(s[i]) : "v"(c1), "v"(c2), "s"((i & 1) ? m0 : m1)))
    }
    if constexpr (ROLE == EXECMOV_PAIR) {
        _Pragma("unroll 1") for (int it = 0; it < N; ++it) {
            asm volatile("s_not_b64 exec, %8\n\tv_mov_b32 %0, 0\n\ts_not_b64 exec, %9\n\tv_mov_b32 %1, 0\n\ts_not_b64 exec, %8\n\tv_mov_b32 %2, 0\n\ts_not_b64 exec, %9\n\tv_mov_b32 %3, 0\n\t"
                         "s_not_b64 exec, %8\n\tv_mov_b32 %4, 0\n\ts_not_b64 exec, %9\n\tv_mov_b32 %5, 0\n\ts_not_b64 exec, %8\n\tv_mov_b32 %6, 0\n\ts_not_b64 exec, %9\n\tv_mov_b32 %7, 0\n\ts_mov_b64 exec, -1"
                         : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]) : "s"(m0), "s"(m1) : "scc");
            asm volatile("s_not_b64 exec, %8\n\tv_mov_b32 %0, 0\n\ts_not_b64 exec, %9\n\tv_mov_b32 %1, 0\n\ts_not_b64 exec, %8\n\tv_mov_b32 %2, 0\n\ts_not_b64 exec, %9\n\tv_mov_b32 %3, 0\n\t"
                         "s_not_b64 exec, %8\n\tv_mov_b32 %4, 0\n\ts_not_b64 exec, %9\n\tv_mov_b32 %5, 0\n\ts_not_b64 exec, %8\n\tv_mov_b32 %6, 0\n\ts_not_b64 exec, %9\n\tv_mov_b32 %7, 0\n\ts_mov_b64 exec, -1"
                         : "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15]) : "s"(m0), "s"(m1) : "scc");
        }
        return 16L * N;
    }
    return 0;
}

template <int ROLE>
__global__ __launch_bounds__(768) void probe(Args A) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = A.roles[wave >> 2];          // 0: this wave's own role (ROLE); 1: the matrix chain; 2: idle
    const int lane = threadIdx.x & 63;
    float s[16];
    unsigned int p[8];
    f32x2 q[8];
    for (int i = 0; i < 16; ++i) s[i] = -0.001f * (i + lane);
    for (int i = 0; i < 8; ++i) { p[i] = 0x3c003c00u + lane * 3 + i; q[i] = f32x2{0.5f + i, 0.25f * lane}; }
    float extra = 0.f;
    long ninst = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (role == 0) ninst = run_role<ROLE>(A.iters, s, p, q, extra);
    else if (role == 1) ninst = run_role<MFMA_DEP>(A.iters, s, p, q, extra);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = extra;
    for (int i = 0; i < 16; ++i) acc += s[i];
    for (int i = 0; i < 8; ++i) acc += (float)p[i] + q[i][0] + q[i][1];
    if (acc == 12345.678f) A.sink[0] = acc;
    if (lane == 0) {
        A.out[(blockIdx.x * 12 + wave) * 2] = t1 - t0;
        A.out[(blockIdx.x * 12 + wave) * 2 + 1] = (unsigned long long)ninst;
    }
}

typedef void (*Kern)(Args);
template <int R> struct Tab { static void fill(Kern* t) { t[R] = probe<R>; Tab<R - 1>::fill(t); } };
template <> struct Tab<1> { static void fill(Kern* t) { t[1] = probe<1>; } };

int main() {
    Kern tab[NROLES] = {};
    Tab<NROLES - 1>::fill(tab);
    unsigned long long* d_out; float* d_sink;
    const int grid = 256;
    hipMalloc(&d_out, grid * 12 * 2 * 8); hipMalloc(&d_sink, 4);
    std::vector<unsigned long long> h(grid * 12 * 2);
    auto run = [&](int R, int r0, int r1, int r2, double* per_role_cycles, double* per_role_inst) {
        const int k = r2 >= 0 ? 3 : r1 >= 0 ? 2 : 1;
        Args a; a.roles[0] = r0; a.roles[1] = r1 < 0 ? 2 : r1; a.roles[2] = r2 < 0 ? 2 : r2; a.iters = 1000; a.out = d_out; a.sink = d_sink;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d_out, 0, grid * 12 * 2 * 8);
            hipLaunchKernelGGL(tab[R], dim3(grid), dim3(256 * k), 0, 0, a);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        // per SIMD (waves w, w + 4, w + 8 of a workgroup): span = the longest of its waves, instructions per role
        double span = 0; int nsimd = 0;
        per_role_inst[0] = per_role_inst[1] = 0; per_role_cycles[0] = per_role_cycles[1] = 0;
        int cnt[2] = {0, 0};
        for (int blk = 0; blk < grid; ++blk)
            for (int sd = 0; sd < 4; ++sd) {
                unsigned long long mx = 0;
                for (int g = 0; g < k; ++g) {
                    const int w = 4 * g + sd;
                    const unsigned long long c = h[(blk * 12 + w) * 2], n = h[(blk * 12 + w) * 2 + 1];
                    if (c > mx) mx = c;
                    const int role = g == 0 ? r0 : g == 1 ? r1 : r2;
                    if (role == 0 || role == 1) { per_role_inst[role] += (double)n; per_role_cycles[role] += (double)c; ++cnt[role]; }
                }
                span += (double)mx; ++nsimd;
            }
        for (int r = 0; r < 2; ++r) if (cnt[r]) { per_role_cycles[r] /= cnt[r]; per_role_inst[r] /= cnt[r]; }
        return span / nsimd;
    };
    printf("%-28s | cycles per instruction and SIMD: 1 wave | 2 waves | 3 waves | 2 waves next to a matrix chain: per vector instr, per matrix product (32.1 alone)\n", "opcode");
    for (int R = 2; R < NROLES; ++R) {
        double pc[2], pi[2];
        double s1 = run(R, 0, -1, -1, pc, pi); double c1 = s1 / pi[0];
        double s2 = run(R, 0, 0, -1, pc, pi);  double c2 = s2 / (2 * pi[0]);
        double s3 = run(R, 0, 0, 0, pc, pi);   double c3 = s3 / (3 * pi[0]);
        double sm = run(R, 1, 0, 0, pc, pi);
        // next to the matrix chain: the vector waves' own time per instruction (two waves share), and the chain's time per product
        double vper = pc[0] / pi[0] / 2.0, mper = pc[1] / pi[1];
        (void)sm;
        printf("%-28s | %6.2f | %6.2f | %6.2f | %6.2f  %6.1f\n", role_name[R], c1, c2, c3, vper, mper);
    }
    return 0;
}
