// Standalone A/B harness for the fused TSFormer encoder (no torch, no python): dlopen a libstep_hip build (ABI 3), check the
// hidden states of 12 full-length sequences against the CPU oracle for bf16 and f16 operand fragments, time the kernel at the
// PEMS04 launch size (S=2456, P=336) with and without dropout, and dump the selftest stream of a dropout generator.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int (*enc_fn)(const float*, int, int, const void*, long, int, int, uint16_t*, float*, float*, float*, float, uint64_t, void*);
typedef int (*dump_fn)(uint32_t, int, int, int, uint32_t*, void*);
typedef const char* (*err_fn)(void);
typedef int (*abi_fn)(void);

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static std::vector<char> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { printf("cannot open %s\n", path); exit(3); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> b(n);
    if (fread(b.data(), 1, n, f) != (size_t)n) { printf("short read %s\n", path); exit(3); }
    fclose(f);
    return b;
}

int main(int argc, char** argv) {
    const char* libpath = argv[1];
    const char* tag = argv[2];
    const int dump_gen = argc > 3 ? atoi(argv[3]) : -1;
    void* h = dlopen(libpath, RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    enc_fn enc = (enc_fn)dlsym(h, "step_tsformer_encode");
    dump_fn dump = (dump_fn)dlsym(h, "step_selftest_dropout_stream");
    err_fn lasterr = (err_fn)dlsym(h, "step_last_error");
    abi_fn abi = (abi_fn)dlsym(h, "step_abi_version");
    if (!enc || !dump || !lasterr || !abi) { printf("missing symbol\n"); return 1; }
    printf("[%s] abi %d\n", tag, abi());
    if (abi() != 3) { printf("needs an ABI 3 build\n"); return 1; }
    const int P = 336, L = 12 * P, S0 = 12, S = 2456, depth = 4;
    std::vector<char> series = slurp("series_small.bin"), want = slurp("want_hidden.bin");
    std::vector<char> pack[2] = {slurp("pack_bf16.bin"), slurp("pack_f16.bin")};
    hipStream_t st;
    HIPCK(hipStreamCreate(&st));
    float *d_series, *d_hid32, *d_last, *d_sqn, *d_big;
    uint16_t* d_hid16;
    void* d_pack[2];
    HIPCK(hipMalloc(&d_series, series.size()));
    HIPCK(hipMemcpy(d_series, series.data(), series.size(), hipMemcpyHostToDevice));
    for (int k = 0; k < 2; ++k) {
        HIPCK(hipMalloc(&d_pack[k], pack[k].size()));
        HIPCK(hipMemcpy(d_pack[k], pack[k].data(), pack[k].size(), hipMemcpyHostToDevice));
    }
    HIPCK(hipMalloc(&d_hid32, (size_t)S0 * P * 96 * 4));
    HIPCK(hipMalloc(&d_hid16, (size_t)S * P * 96 * 2));
    HIPCK(hipMalloc(&d_last, (size_t)S * 96 * 4));
    HIPCK(hipMalloc(&d_sqn, (size_t)S * 16 * 4));
    // ---- correctness vs the CPU oracle, dropout off
    std::vector<float> got((size_t)S0 * P * 96);
    const float* w = (const float*)want.data();
    for (int k = 0; k < 2; ++k) {
        HIPCK(hipMemset(d_hid32, 0xff, got.size() * 4));
        int rc = enc(d_series, S0, L, d_pack[k], (long)pack[k].size(), depth, k, d_hid16, d_hid32, d_last, d_sqn, 0.f, 0, st);
        if (rc) { printf("encode failed: %s\n", lasterr()); return 1; }
        HIPCK(hipStreamSynchronize(st));
        HIPCK(hipMemcpy(got.data(), d_hid32, got.size() * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, worst = 0; int nan = 0;
        for (int s = 0; s < S0; ++s) {
            double n1 = 0, d1 = 0;
            for (long i = (long)s * P * 96; i < (long)(s + 1) * P * 96; ++i) {
                if (!(got[i] == got[i])) ++nan;
                double d = (double)got[i] - w[i]; n1 += d * d; d1 += (double)w[i] * w[i];
            }
            num += n1; den += d1;
            if (sqrt(n1 / d1) > worst) worst = sqrt(n1 / d1);
        }
        printf("[%s] operand %s: hidden rel-L2 vs oracle %.3e (worst sequence %.3e, NaN %d)\n", tag, k ? "f16 " : "bf16", sqrt(num / den), worst, nan);
        // run-to-run determinism
        std::vector<float> again(got.size());
        rc = enc(d_series, S0, L, d_pack[k], (long)pack[k].size(), depth, k, d_hid16, d_hid32, d_last, d_sqn, 0.f, 0, st);
        HIPCK(hipStreamSynchronize(st));
        HIPCK(hipMemcpy(again.data(), d_hid32, got.size() * 4, hipMemcpyDeviceToHost));
        printf("[%s] operand %s: deterministic %d\n", tag, k ? "f16 " : "bf16", (int)(memcmp(again.data(), got.data(), got.size() * 4) == 0));
    }
    // ---- timing at the PEMS04 launch size
    std::vector<float> big((size_t)S * L);
    uint32_t x = 12345u;
    for (size_t i = 0; i < big.size(); ++i) { x = x * 1664525u + 1013904223u; big[i] = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.0f; }
    HIPCK(hipMalloc(&d_big, big.size() * 4));
    HIPCK(hipMemcpy(d_big, big.data(), big.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    for (int k = 0; k < 2; ++k)
        for (int dr = 0; dr < 2; ++dr) {
            const float p = dr ? 0.1f : 0.f;
            for (int i = 0; i < 3; ++i) enc(d_big, S, L, d_pack[k], (long)pack[k].size(), depth, k, d_hid16, nullptr, d_last, d_sqn, p, 7 + i, st);
            HIPCK(hipStreamSynchronize(st));
            float best = 1e9f, sum = 0.f;
            const int reps = 10;
            for (int i = 0; i < reps; ++i) {
                HIPCK(hipEventRecord(e0, st));
                int rc = enc(d_big, S, L, d_pack[k], (long)pack[k].size(), depth, k, d_hid16, nullptr, d_last, d_sqn, p, 100 + i, st);
                HIPCK(hipEventRecord(e1, st));
                HIPCK(hipEventSynchronize(e1));
                if (rc) { printf("encode failed: %s\n", lasterr()); return 1; }
                float ms; HIPCK(hipEventElapsedTime(&ms, e0, e1));
                sum += ms; if (ms < best) best = ms;
            }
            // sanity of the dropout-on output: mean squared hidden of the first sequences must be ~1 per feature (LayerNorm output)
            std::vector<uint16_t> hb((size_t)64 * P * 96);
            HIPCK(hipMemcpy(hb.data(), d_hid16, hb.size() * 2, hipMemcpyDeviceToHost));
            double ss = 0; int bad = 0;
            for (size_t i = 0; i < hb.size(); ++i) { uint32_t u = (uint32_t)hb[i] << 16; float f; memcpy(&f, &u, 4); if (!(f == f) || fabsf(f) > 1e4f) ++bad; ss += (double)f * f; }
            printf("[%s] operand %s dropout %.1f: %.3f ms avg, %.3f ms best (S=%d P=%d); hidden mean-square %.4f, non-finite %d\n", tag,
                   k ? "f16 " : "bf16", p, sum / reps, best, S, P, ss / hb.size(), bad);
        }
    // ---- generator dump
    if (dump_gen >= 0) {
        const int streams = 4096, words = 256;
        uint32_t* d_out;
        HIPCK(hipMalloc(&d_out, (size_t)streams * words * 4));
        int rc = dump(0x1234567u, dump_gen, streams, words, d_out, st);
        if (rc) { printf("dump failed: %s\n", lasterr()); return 1; }
        HIPCK(hipStreamSynchronize(st));
        std::vector<uint32_t> o((size_t)streams * words);
        HIPCK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost));
        char name[256];
        snprintf(name, sizeof(name), "../gpurun_out/dropout_stream_gen%d.bin", dump_gen);
        FILE* f = fopen(name, "wb");
        if (f) { fwrite(o.data(), 4, o.size(), f); fclose(f); printf("[%s] wrote %s\n", tag, name); }
    }
    return 0;
}
