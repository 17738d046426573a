// Standalone A/B harness for the fused TSFormer encoder (no torch, no python).  Usage:
//     ./enc_ab tag1=./libenc_tag1.so tag2=./libenc_tag2.so ...
// Every library is a build of csrc/tsformer_encoder.hip + errors.cpp (ABI 4; an ABI 3 build -- the round-1 kernel -- is
// accepted too and called through its old signature, so the previous kernel can be timed on the same box).
// Per library: hidden states of 12 full-length sequences against the CPU oracle (dropout off, bf16 and f16 operand fragments;
// the re-shift-on-every-maximum test mode), training-mode dropout against the oracle replaying the same keep-mask pool
// (files written by tools/enc_ab_prepare.py), run-to-run determinism.  Then timing at the PEMS04 launch size (S=2456, P=336)
// in INTERLEAVED rounds over all libraries (cdna_hip_programming.md 5.4 rule 24), with and without dropout.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

typedef int (*enc4_fn)(const float*, int, int, const void*, long, int, int, uint16_t*, float*, float*, float*, float, const uint64_t*, long,
                       uint64_t, void*);
typedef int (*enc5_fn)(const float*, int, int, const void*, long, int, int, uint16_t*, float*, float*, float*, float, const uint64_t*, long,
                       uint64_t, unsigned int*, void*);
typedef int (*enc3_fn)(const float*, int, int, const void*, long, int, int, uint16_t*, float*, float*, float*, float, uint64_t, void*);
typedef int (*fill_fn)(uint64_t*, long, float, uint64_t, void*);
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

struct Lib {
    std::string tag;
    std::string env_key, env_val;            // optional `tag=path@KEY=VAL`: KEY is set to VAL around every launch of this entry (the library reads its knobs per launch)
    int abi;
    void* handle = nullptr;
    enc4_fn e4; enc3_fn e3; enc5_fn e5; fill_fn fill; err_fn err;
    unsigned int* counter = nullptr;         // ABI 5: device counter of softmax units on the re-shifting (slow) path
    // unified call: flags bit0 f16, bit1 always-reshift (ABI 4 only)
    int run(const float* x, int S, int L, const void* pk, long pkb, int flags, uint16_t* h16, float* h32, float* last, float* sqn, float p,
            const uint64_t* pool, long words, uint64_t seed, hipStream_t st) const {
        if (!env_key.empty()) setenv(env_key.c_str(), env_val.c_str(), 1);
        const int rc = run_(x, S, L, pk, pkb, flags, h16, h32, last, sqn, p, pool, words, seed, st);
        if (!env_key.empty()) unsetenv(env_key.c_str());
        return rc;
    }
    int run_(const float* x, int S, int L, const void* pk, long pkb, int flags, uint16_t* h16, float* h32, float* last, float* sqn, float p,
             const uint64_t* pool, long words, uint64_t seed, hipStream_t st) const {
        if (abi >= 5) return e5(x, S, L, pk, pkb, 4, flags, h16, h32, last, sqn, p, pool, words, seed, counter, st);
        if (abi >= 4) return e4(x, S, L, pk, pkb, 4, flags, h16, h32, last, sqn, p, pool, words, seed, st);
        return e3(x, S, L, pk, pkb, 4, flags & 1, h16, h32, last, sqn, p, seed, st);
    }
};

static double rel_l2(const float* got, const float* want, size_t n, int* nan) {
    double num = 0, den = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!(got[i] == got[i])) ++*nan;
        double d = (double)got[i] - want[i]; num += d * d; den += (double)want[i] * want[i];
    }
    return sqrt(num / den);
}

int main(int argc, char** argv) {
    std::vector<Lib> libs;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        size_t eq = a.find('=');
        if (eq == std::string::npos) { printf("argument %s is not tag=path\n", argv[i]); return 1; }
        Lib l; l.tag = a.substr(0, eq);
        std::string path = a.substr(eq + 1);
        const size_t at = path.find('@');
        if (at != std::string::npos) {
            const std::string kv = path.substr(at + 1);
            path = path.substr(0, at);
            const size_t e2 = kv.find('=');
            if (e2 == std::string::npos) { printf("argument %s: expected tag=path@KEY=VAL\n", argv[i]); return 1; }
            l.env_key = kv.substr(0, e2); l.env_val = kv.substr(e2 + 1);
        }
        void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
        abi_fn abi = (abi_fn)dlsym(h, "step_abi_version");
        l.err = (err_fn)dlsym(h, "step_last_error");
        if (!abi || !l.err) { printf("[%s] missing symbol\n", l.tag.c_str()); return 1; }
        l.abi = abi();
        l.handle = h;
        l.e4 = (enc4_fn)dlsym(h, "step_tsformer_encode"); l.e3 = (enc3_fn)l.e4; l.e5 = (enc5_fn)l.e4;
        l.fill = (fill_fn)dlsym(h, "step_dropout_pool_fill");
        if (!l.e4 || (l.abi >= 4 && !l.fill)) { printf("[%s] missing symbol\n", l.tag.c_str()); return 1; }
        printf("[%s] abi %d\n", l.tag.c_str(), l.abi);
        libs.push_back(l);
    }
    if (libs.empty()) { printf("no libraries\n"); return 1; }
    // ENC_AB_P / ENC_AB_S: tokens per sequence and sequences of the timed launch (defaults: the PEMS04 launch, 336 x 2456; PEMS07 is
    // 168 x 3532, the 4096-node graph 168 x 4096); the fixtures of another P carry the suffix _p<P> (tools/enc_ab_prepare.py)
    const int P = getenv("ENC_AB_P") ? atoi(getenv("ENC_AB_P")) : 336, L = 12 * P, S0 = 12;
    const int S = getenv("ENC_AB_S") ? atoi(getenv("ENC_AB_S")) : 2456;
    const std::string sfx = P == 336 ? "" : "_p" + std::to_string(P);
    auto fx = [&](const char* stem) { return slurp((std::string(stem) + sfx + ".bin").c_str()); };
    std::vector<char> series = fx("series_small"), want = fx("want_hidden"), wantd = fx("want_hidden_drop");
    std::vector<char> poolf = slurp("drop_pool.bin"), seedf = slurp("drop_seed.bin");
    // [2]: float16 fragments of the UNSCALED initialisation (bench.py's model), used for timing only: the softmax schedule of
    // the kernel is data dependent (heads whose scores outrun the fixed shift are redone), and the sharpened weights of
    // packs [0] / [1] make a whole layer take that path
    std::vector<char> pack[3] = {fx("pack_bf16"), fx("pack_f16"), fx("pack_f16_plain")};
    const long pool_words = (long)poolf.size() / 8;
    uint64_t drop_seed; memcpy(&drop_seed, seedf.data(), 8);
    hipStream_t st;
    HIPCK(hipStreamCreate(&st));
    float *d_series, *d_hid32, *d_last, *d_sqn, *d_big;
    uint16_t* d_hid16;
    uint64_t *d_pool, *d_pool2;
    void* d_pack[3];
    HIPCK(hipMalloc(&d_series, series.size()));
    HIPCK(hipMemcpy(d_series, series.data(), series.size(), hipMemcpyHostToDevice));
    HIPCK(hipMalloc(&d_pool, poolf.size() + 128));          // + the wrap-around copy of the first 16 words (ABI 5 kernels read on into it)
    HIPCK(hipMemcpy(d_pool, poolf.data(), poolf.size(), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy((char*)d_pool + poolf.size(), poolf.data(), 128, hipMemcpyHostToDevice));
    const long words2 = getenv("ENC_AB_POOL_LOG2") ? 1L << atoi(getenv("ENC_AB_POOL_LOG2")) : 1L << 18;
    HIPCK(hipMalloc(&d_pool2, (words2 + 16) * 8));
    unsigned int* d_counter;
    HIPCK(hipMalloc(&d_counter, 256));
    for (Lib& l : libs) l.counter = d_counter;
    printf("timing pool: 2^%d words\n", (int)log2((double)words2));
    for (int k = 0; k < 3; ++k) {
        HIPCK(hipMalloc(&d_pack[k], pack[k].size()));
        HIPCK(hipMemcpy(d_pack[k], pack[k].data(), pack[k].size(), hipMemcpyHostToDevice));
    }
    HIPCK(hipMalloc(&d_hid32, (size_t)S0 * P * 96 * 4));
    HIPCK(hipMalloc(&d_hid16, (size_t)S * P * 96 * 2));
    HIPCK(hipMalloc(&d_last, (size_t)S * 96 * 4));
    HIPCK(hipMalloc(&d_sqn, (size_t)S * 16 * 4));
    const size_t n0 = (size_t)S0 * P * 96;
    std::vector<float> got(n0), again(n0);
    const float* w = (const float*)want.data();
    const float* wd = (const float*)wantd.data();
    // ---------------------------------------------------------------- correctness
    for (const Lib& l : libs) {
        const char* tag = l.tag.c_str();
        for (int k = 0; k < 2; ++k)
            for (int fl = 0; fl < (l.abi >= 4 ? 2 : 1); ++fl) {
                HIPCK(hipMemset(d_hid32, 0xff, n0 * 4));
                int rc = l.run(d_series, S0, L, d_pack[k], (long)pack[k].size(), k | (fl << 1), d_hid16, d_hid32, d_last, d_sqn, 0.f, nullptr, 0, 0, st);
                if (rc) { printf("[%s] encode failed: %s\n", tag, l.err()); return 1; }
                HIPCK(hipStreamSynchronize(st));
                HIPCK(hipMemcpy(got.data(), d_hid32, n0 * 4, hipMemcpyDeviceToHost));
                int nan = 0;
                double e = rel_l2(got.data(), w, n0, &nan);
                rc = l.run(d_series, S0, L, d_pack[k], (long)pack[k].size(), k | (fl << 1), d_hid16, d_hid32, d_last, d_sqn, 0.f, nullptr, 0, 0, st);
                HIPCK(hipStreamSynchronize(st));
                HIPCK(hipMemcpy(again.data(), d_hid32, n0 * 4, hipMemcpyDeviceToHost));
                printf("[%s] operand %s%s: hidden rel-L2 vs oracle %.3e (NaN %d), deterministic %d\n", tag, k ? "f16 " : "bf16",
                       fl ? " re-shift on every maximum" : "", e, nan, (int)(memcmp(again.data(), got.data(), n0 * 4) == 0));
            }
        if (l.abi >= 4)
            for (int k = 0; k < 2; ++k) {
                HIPCK(hipMemset(d_hid32, 0xff, n0 * 4));
                int rc = l.run(d_series, S0, L, d_pack[k], (long)pack[k].size(), k, d_hid16, d_hid32, d_last, d_sqn, 0.1f, d_pool, pool_words, drop_seed, st);
                if (rc) { printf("[%s] encode failed: %s\n", tag, l.err()); return 1; }
                HIPCK(hipStreamSynchronize(st));
                HIPCK(hipMemcpy(got.data(), d_hid32, n0 * 4, hipMemcpyDeviceToHost));
                int nan = 0;
                double e = rel_l2(got.data(), wd, n0, &nan), pert = rel_l2(wd, w, n0, &nan);
                printf("[%s] operand %s DROPOUT 0.1, pool from file: hidden rel-L2 vs oracle replaying the same masks %.3e (NaN %d; dropout moves the states by %.3f)\n",
                       tag, k ? "f16 " : "bf16", e, nan, pert);
            }
    }
    // ---------------------------------------------------------------- timing at the PEMS04 launch size, interleaved rounds
    std::vector<float> big((size_t)S * L), like((size_t)S * L);
    uint32_t x = 12345u;
    for (size_t i = 0; i < big.size(); ++i) { x = x * 1664525u + 1013904223u; big[i] = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.0f; }
    // bench.py's synthetic series: sin(2 pi t / 288 + phase_s) + noise of standard deviation 0.5 (uniform here)
    for (int s = 0; s < S; ++s)
        for (int t = 0; t < L; ++t) {
            x = x * 1664525u + 1013904223u;
            like[(size_t)s * L + t] = sinf(6.2831853f * t / 288.0f + 0.37f * s) + ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 1.7320508f;
        }
    float* d_like;
    HIPCK(hipMalloc(&d_big, big.size() * 4));
    HIPCK(hipMemcpy(d_big, big.data(), big.size() * 4, hipMemcpyHostToDevice));
    HIPCK(hipMalloc(&d_like, like.size() * 4));
    HIPCK(hipMemcpy(d_like, like.data(), like.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    const int rounds = 7;
    const double flop = (double)S * P * (4.0 * (221184 + 384.0 * P) + 2304);
    for (int mode = 0; mode < 5; ++mode) {          // 0: f16 no dropout, 1: f16 dropout, 2: bf16 dropout; 3 / 4: bench.py-like weights and series
        const int k = mode >= 3 ? 2 : mode == 2 ? 0 : 1;
        const int fl = mode == 2 ? 0 : 1;             // operand flag of the pack
        const float p = (mode == 0 || mode == 3) ? 0.f : 0.1f;
        const float* src = mode >= 3 ? d_like : d_big;
        std::vector<std::vector<float>> t(libs.size());
        std::vector<unsigned int> slow(libs.size(), 0u);
        for (int r = -1; r < rounds; ++r)             // round -1 = warm-up
            for (size_t li = 0; li < libs.size(); ++li) {
                const Lib& l = libs[li];
                if (p > 0 && l.abi >= 4) { int rc = l.fill(d_pool2, words2, p, 77 + r, st); if (rc) { printf("fill failed: %s\n", l.err()); return 1; } }
                for (int rep = 0; rep < 3; ++rep) {
                    if (r == rounds - 1 && rep == 2) HIPCK(hipMemsetAsync(d_counter, 0, 256, st));
                    HIPCK(hipEventRecord(e0, st));
                    int rc = l.run(src, S, L, d_pack[k], (long)pack[k].size(), fl, d_hid16, nullptr, d_last, d_sqn, p, d_pool2, words2, 100 + r * 3 + rep, st);
                    HIPCK(hipEventRecord(e1, st));
                    HIPCK(hipEventSynchronize(e1));
                    if (rc) { printf("[%s] encode failed: %s\n", l.tag.c_str(), l.err()); return 1; }
                    float ms; HIPCK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 0) t[li].push_back(ms);
                    if (r == rounds - 1 && rep == 2) { unsigned int c64[64]; HIPCK(hipMemcpy(c64, d_counter, 256, hipMemcpyDeviceToHost)); slow[li] = 0; for (int q = 0; q < 64; ++q) slow[li] += c64[q]; }
                }
            }
        for (size_t li = 0; li < libs.size(); ++li) {
            std::vector<float> v = t[li];
            std::sort(v.begin(), v.end());
            const float med = v[v.size() / 2];
            std::vector<uint16_t> hb((size_t)64 * P * 96);
            HIPCK(hipMemcpy(hb.data(), d_hid16, hb.size() * 2, hipMemcpyDeviceToHost));
            printf("[%s] %s%s dropout %.1f: median %.3f ms, min %.3f, max %.3f (%zu launches, S=%d P=%d) = %.1f TFLOP/s algorithmic, %.1f %% of 2.5 PF; slow-path softmax units %u of %d\n",
                   libs[li].tag.c_str(), fl ? "f16 " : "bf16", mode >= 3 ? " bench-like data" : "", p, med, v.front(), v.back(), v.size(), S, P, flop / med / 1e9, flop / med / 1e9 / 25.0,
                   slow[li], S * 4 * 4 * ((P + 31) / 32));
        }
    }
    // ---------------------------------------------------------------- phase stamps (libraries built with -DTSF_TIMING=1)
    for (const Lib& l : libs) {
        typedef int (*sett_fn)(unsigned long long*);
        typedef int (*nst_fn)(void);
        sett_fn sett = (sett_fn)dlsym(l.handle, "step_tsformer_set_timing");
        nst_fn nst = (nst_fn)dlsym(l.handle, "step_tsformer_timing_stamps");
        if (!sett || !nst) continue;
        const int NST = nst(), groups = (S + 63) / 64;
        const size_t nw = (size_t)groups * 16 * 4 * NST;
        unsigned long long* d_t;
        HIPCK(hipMalloc(&d_t, nw * 8));
        for (int p10 = 0; p10 < 2; ++p10) {           // dropout off / on, bench-like weights and series
            const float p = p10 ? 0.1f : 0.f;
            if (p > 0) l.fill(d_pool2, words2, p, 4242, st);
            for (int rep = 0; rep < 3; ++rep) {       // the third launch is the one that is kept
                HIPCK(hipMemsetAsync(d_t, 0, nw * 8, st));
                HIPCK(hipStreamSynchronize(st));
                if (sett(rep == 2 ? d_t : nullptr)) { printf("set_timing failed\n"); return 1; }
                int rc = l.run(d_like, S, L, d_pack[2], (long)pack[2].size(), 1, d_hid16, nullptr, d_last, d_sqn, p, d_pool2, words2, 900 + rep, st);
                if (rc) { printf("[%s] encode failed: %s\n", l.tag.c_str(), l.err()); return 1; }
                HIPCK(hipStreamSynchronize(st));
            }
            sett(nullptr);
            std::vector<unsigned long long> ht(nw);
            HIPCK(hipMemcpy(ht.data(), d_t, nw * 8, hipMemcpyDeviceToHost));
            const std::string fn = std::string(getenv("ENC_AB_OUT") ? getenv("ENC_AB_OUT") : ".") + "/enc_timing_" + l.tag + (p10 ? "_drop" : "_nodrop") + "_p" + std::to_string(P) + ".bin";
            FILE* f = fopen(fn.c_str(), "wb");
            if (f) {
                const int hdr[4] = {groups, 16, 4, NST};
                fwrite(hdr, 4, 4, f); fwrite(ht.data(), 8, nw, f); fclose(f);
                printf("[%s] phase stamps of %d workgroups -> %s\n", l.tag.c_str(), groups, fn.c_str());
            }
        }
        HIPCK(hipFree(d_t));
    }
    // sanity of the last dropout-on output of the last library: LayerNorm output, mean square ~1 per feature
    {
        std::vector<uint16_t> hb((size_t)64 * P * 96);
        HIPCK(hipMemcpy(hb.data(), d_hid16, hb.size() * 2, hipMemcpyDeviceToHost));
        double ss = 0; int bad = 0;
        for (size_t i = 0; i < hb.size(); ++i) { uint32_t u = (uint32_t)hb[i] << 16; float f; memcpy(&f, &u, 4); if (!(f == f) || fabsf(f) > 1e4f) ++bad; ss += (double)f * f; }
        printf("last dropout-on launch: hidden mean-square %.4f, non-finite %d\n", ss / hb.size(), bad);
    }
    return 0;
}
