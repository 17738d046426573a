// Fused TSFormer encoder forward for gfx950 (forecasting mode, frozen weights).
//
// One workgroup per sequence s = (sample, node); one wave per 32-token tile (P <= 512 tokens,
// 336 for PEMS04).  The residual stream of a wave's 32 tokens lives in f32 accumulator
// registers for the whole kernel: activations are kept TRANSPOSED ([feature][token], token =
// lane & 31), weights are the MFMA A operand, and thanks to the k-slot map in
// tsformer_layout.h every accumulator tile becomes the next MFMA's B operand by a plain
// f32->16-bit pack.  Inter-wave traffic: the per-head K and V operand fragments (2 KB per key tile
// each) through LDS, and the weights, which every workgroup streams ONCE from L2 into an LDS ring of 25 KB stage blocks
// (global_load_lds DMA, prefetched ahead; 2 slots, or 4 with the feed-forward blocks consumed in pairs) shared by all waves.
// No scratch memory in the variants for P <= 384: values derived from the lane id are re-derived where they are used
// rather than kept alive (fresh_lane_id()), and what does not fit the 168 registers is parked in LDS explicitly.
//
//   patch embed + pos-emb (exact f32 on v_mfma_f32_32x32x2_f32)  ->  4 x { per head: Q,K,V (MFMA, K=96) -> per key tile: S^T = K Q^T
//   -> P = exp2(S - shift) -> O^T += V^T P^T  (ONE pass over the keys: online softmax, see below)
//   -> out-proj accumulates onto (x + b_o) ; LN1 ; FFN in 12 chunks of 32 hidden units, never leaving
//   registers ; LN2 }  ->  encoder_norm  ->  hidden (bf16 and/or f32), last-patch state, squared norms.
//
// Softmax.  The score tile comes out of the matrix cores already shifted: head-dim slot 25 (the head dim is 24 of 32)
// carries (1 | -shift) through the contraction, slot 26 (is_padding | -30000) masks the padded keys.  `shift` is a per-query
// running value kept TSF_BIAS = 60 (log2 units) ABOVE the largest score seen so far, so the probabilities of the leading keys
// are ~2^-60 and a later key may exceed the running maximum by up to TSF_THR + TSF_BIAS = 160 before anything has to be
// touched: P and V are bfloat16 operands (8 exponent bits) and O / the denominator are f32, so magnitudes up to 2^100 are
// harmless and the common factor cancels in O / denominator.  Per key tile the only extra work over a plain exp2 is the
// tile maximum (8 x v_max3_f32) and one compare; if any query of the wave saw a score above TSF_THR, the wave takes the
// re-shift path (new shift, O and the denominator scaled by exp2(old - new), the score tiles in flight corrected).  The
// first key tile always takes it (that sets the initial shift).  With the reference's sqrt(96) input scaling and default
// initialisation the scores are in the hundreds, so this path does run (about once per head); tests force it on every
// tile (flags bit 1) and on adversarial inputs.
// Fast schedule (TSF_FAST_ATTN, the default): the per-tile maximum / compare / branch of that loop costs more than its ten
// instructions (it splits the loop body and serialises on a vector compare), and after the first tile it practically never
// fires.  So a head is first run with the shift FIXED at what the first key tile sets -- a branch-free loop of exp2, packs and
// MFMAs -- and checked once at the end: every probability is positive, so a denominator below TSF_LIMIT = 2^110 proves that no
// score went above the shift by more than that and nothing overflowed.  Otherwise (any query of the wave) the head is redone
// from its K / V fragments with the re-shifting loop above, and the remaining heads of that layer skip the attempt.  When no
// later tile would have re-shifted, both schedules execute the same arithmetic in the same order.
//
// Dropout (training-mode TSFormer inside STEP, positional_encoding.py:32 and the four sites of each encoder layer): keep-masks
// are lane masks fetched with scalar loads from the per-step pool (tsformer_device.h), one v_cndmask_b32 per element; all
// survivor scales are folded into existing multiplies (sqrt(d), 1/denominator, the pre-scaled b2, the residual fma).
//
// MFMA work per token-layer: 221 184*1.33(q/k/v/o head padding 24->32 only) ... see DESIGN.md
// for the flop accounting used by bench.py's roofline.
#include "tsformer_device.h"

namespace {

#define TSF_THR 100.0f
#define TSF_BIAS 60.0f
#ifndef TSF_ABLATE
#define TSF_ABLATE 0        // timing experiments only (WRONG results): 1 no exp2, 2 no attention keep-masks, 4 no re-shift check, 16 no denominator adds,
                            // 32 no key-tile loop at all, 64 one FFN stage instead of six, 128 no LayerNorm arithmetic, 256 no matrix products in the fast key-tile loop
#endif
#ifndef TSF_FAST_ATTN
#define TSF_FAST_ATTN 1     // 0: always the re-shifting loop (A/B builds)
#endif
#define TSF_LIMIT 1.2980742e33f   // 2^110: largest softmax denominator the fast schedule accepts
#ifndef TSF_MASK_EARLY
#define TSF_MASK_EARLY 1    // fetch a tile's keep-mask words at the top of its step instead of next to their use
#endif

#ifndef TSF_TAIL_GROUPS
#define TSF_TAIL_GROUPS 1   // 0: the last key tile runs the full step even when most of its keys are padding (A/B builds)
#endif
#ifndef TSF_RING12
#define TSF_RING12 4        // ring slots of the 9..12 tile variant without the parked operand copy (2: one barrier per block, A/B builds)
#endif
template <int MAXW, bool PARK>
constexpr int ring_slots() { return (MAXW == 12 && !PARK) ? TSF_RING12 : 2; }
#ifndef TSF_PARK1
#define TSF_PARK1 1         // 0: leave the allocation to the compiler (it spills one operand fragment to scratch memory; A/B builds)
#endif
// number of operand-copy fragments (of 6) parked in LDS while the attention loops run: all of them (PARK), or just the one the
// register allocator would otherwise put into scratch memory in the 9..12 tile variant
template <int MAXW, bool PARK>
constexpr int parked_frags() { return PARK ? 6 : ((MAXW == 12 || MAXW == 8) && TSF_PARK1) ? 1 : 0; }

#ifndef TSF_SUM_SCALAR
#define TSF_SUM_SCALAR 0    // 1: softmax row sums as four chains of plain v_add_f32 instead of two chains of v_pk_add_f32 (A/B builds; measured 1.986 vs 2.002 ms, within noise)
#endif
#ifndef TSF_STATIC_PRIO
#define TSF_STATIC_PRIO 0   // 1: the later-dispatched half of the workgroup's waves runs at s_setprio 1 for the whole kernel (A/B builds; 1.994 vs 2.002 ms)
#endif
// Persistent launch (EncArgs.grid_limit > 0, flags bits 8..23 of step_tsformer_encode): at most that many workgroups, each looping over the
// sequences blockIdx.x, + gridDim.x, ...; the next sequence's first weight block is requested during the last feed-forward stage of the
// current one.  A workgroup fills its compute unit (704 threads x 168 registers, ~150 KB of LDS), so the limit IS the number of compute units
// the encoder takes.  Round 3 measured it alone and inside one step (profiles/r03_i_persistent_encoder_ab.log: no gain -- the tail of 2456
// workgroups over 256 units is not where the time goes, and within ONE step the rest of the forward needs the encoder's output anyway).
// Round 5: with the frozen branch of the NEXT batch queued next to this batch's backward (STEP.prefetch) the split pays -- the encoder on
// 160 units for 3.1 ms next to a 3.7 ms chain of small kernels on the other 96, instead of 2.1 ms + 2.1 ms one after the other
// (4.26 -> 3.73 ms per step at PEMS04, profiles/r05_k_persist_prefetch.log).
#ifndef TSF_BATCH_FRAGS
#define TSF_BATCH_FRAGS 0   // 1: scheduling fences around the batched weight-fragment reads (measured: forces the operand copy into scratch, 1.93 -> 2.22 ms)
#endif
#ifndef TSF_RELU_PACKED
#define TSF_RELU_PACKED 1   // 0: f32 clamp before the pack (A/B builds)
#endif
// ---- round 5: what profiles/r05_a_encoder_phase_table.md (s_memtime stamps per wave and phase) and profiles/r05_b_simd_probe.log (what one
// SIMD does with this instruction mix) say, turned into defaults.  A SIMD arbitrates its waves by priority, then AGE; one wave alone issues a
// vector instruction every ~7.5 cycles (three together one every ~3), and a matrix product costs every other wave of the SIMD ~16 cycles of
// vector issue.  Left alone, the first-dispatched wave of a SIMD runs each phase almost unimpeded and then waits at the barrier (49 % of its
// time) while the last-dispatched one finishes ALONE at the single-wave rate: key-tile loop 5.6k / 9k / 12.7k cycles for the three waves.
#ifndef TSF_FFN_PIPE
#define TSF_FFN_PIPE 1      // 1: the feed-forward chunks request their weight fragments one matrix-product group AHEAD, pinned with scheduling fences
#endif                      // (left to itself the compiler sinks every fragment read to one or two products in front of its use).  0: the round-4 loop
#ifndef TSF_QKV_PIPE
#define TSF_QKV_PIPE 3      // Q / K / V projections: 0 the compiler's order; 1 fences around the rolling three-fragment batches; 2 two six-fragment
#endif                      // buffers (spills); 3 a ring of six fragment registers refilled one at a time + the out-projection's fragments requested early
#ifndef TSF_PROGRESS_PRIO
#define TSF_PROGRESS_PRIO 1 // 1: a wave's issue priority FALLS as it advances through a phase (key-tile loop: by key-tile pair; feed-forward: by chunk):
#endif                      // a wave that is ahead yields to the ones behind it (key-tile loop 7.5k .. 11.9k instead of 5.6k .. 12.7k).  2 / 3 / 4:
                            // per-step schemes, measured slower (profiles/r05_d_*).  0: off
#ifndef TSF_FILL_OLD
#define TSF_FILL_OLD 1      // 1: the weight ring's DMA pieces are requested by waves 0..3 only (see issue_fill)
#endif
#ifndef TSF_ATTN_IL
#define TSF_ATTN_IL 0       // 1: the fast key-tile step issues its four matrix products interleaved with the vector work (see fast_step)
#endif
#ifndef TSF_TIMING
#define TSF_TIMING 0        // 1: s_memtime stamps at the phase boundaries of sampled workgroups (tools/enc_ab.cpp dumps them, tools/enc_phase_table.py reads them)
#endif
#if TSF_TIMING
#define TSF_NSTAMP 40       // stamps per layer and wave
__device__ unsigned long long* g_tsf_timing = nullptr;      // [sampled workgroup][wave 16][layer 4][TSF_NSTAMP]
#define TSF_STAMP(idx) do { if (tbuf && layer < 4) { const unsigned long long t_ = __builtin_readcyclecounter(); if (fresh_lane_id() == 0) tbuf[layer * TSF_NSTAMP + (idx)] = t_; } } while (0)
#else
#define TSF_STAMP(idx) do { } while (0)
#endif
typedef __attribute__((ext_vector_type(8))) short s16x8;
template <bool F16>
__device__ __forceinline__ typename Opnd<F16>::v8 relu_packed(typename Opnd<F16>::v8 v) {
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(typename Opnd<F16>::v8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
}

// level 0..3 -> s_setprio (an immediate operand: the wave-uniform level is branched on)
__device__ __forceinline__ void prio_level(int lvl) {
    if (lvl >= 3) __builtin_amdgcn_s_setprio(3);
    else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
    else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

struct Yes { static constexpr bool value = true; };
struct No { static constexpr bool value = false; };
struct G1 { static constexpr int value = 1; };
struct G2 { static constexpr int value = 2; };
struct G3 { static constexpr int value = 3; };

// waves per SIMD the register allocation has to admit: the unparked 5..8 tile variant is launched with <= 6-7 waves per workgroup and
// wants two workgroups per compute unit (12-14 waves = 3-4 per SIMD -> 168 registers, like the twelve-wave variant gets by itself)
template <int MAXW, bool PARK>
constexpr int min_waves_per_simd() { return (MAXW == 8 && !PARK) ? 3 : 1; }

// TG: groups of 8 keys that exist in the LAST key tile when the launch is known to have that many (P = 168: 1, P = 336: 2); 4 = any P
template <int MAXW, bool DROP, bool PARK, bool F16, int PIPE, int TG>
__global__ __launch_bounds__(MAXW * 64, (min_waves_per_simd<MAXW, PARK>())) void tsformer_encoder_kernel(EncArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool drop = DROP;
    // LayerNorm's last step as two fmas (tsformer_device.h) -- except in the unparked eight-tile variants without dropout, where the form costs
    // the allocation 20 bytes of scratch per lane
    constexpr bool LN2F = TSF_LN_TWO_FMA && !(MAXW == 8 && !DROP && !PARK);
    typedef typename Opnd<F16>::v8 op8;            // one MFMA operand: 8 x bfloat16 or 8 x float16
    typedef typename Opnd<F16>::elem ope;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int P = A.P, nkt = A.nkt;
    // Two sequences per workgroup (A.nseq == 2, round 6: P <= 192, six token tiles each): waves 0 .. nkt - 1 own sequence 2 b, the next nkt waves
    // sequence 2 b + 1; they share the weight ring, its barriers and the DMA issue, and each half has its own K / V fragment area.  At 168
    // tokens a six-wave workgroup left the compute unit at 1.4 waves per SIMD (the second workgroup the register / LDS budget admits only fits
    // when its two-wave SIMDs happen to be the other's one-wave SIMDs; profiles/r06_y_encoder_occupancy.txt), and one wave alone issues a vector
    // instruction every ~7.5 cycles; twelve waves are three per SIMD by construction.  wsq / wv: this wave's sequence slot and its tile in it.
    const int nw = nkt * A.nseq;
    const int wsq = (A.nseq == 2 && wave >= nkt) ? 1 : 0;
    const int wv = wave - wsq * nkt;
    const int tok = wv * 32 + c;
    const bool tok_ok = tok < P;
    const int tokc = tok_ok ? tok : 0;
    const char* W = A.wpack;
    const mask_ptr pool = (mask_ptr)(uintptr_t)A.pool;
    const uint32_t pmask = A.pool_mask;
    const DropLayout dl(nkt);
    const float keep = A.keep, inv_keep = A.inv_keep;
    auto mask_words = [&](uint32_t chunk, uint32_t off) -> mask_ptr { return pool + ((chunk + off) & pmask); };
    // 48 h, re-derived from a fresh lane id at every use: as a hoisted loop invariant it (or h) is the value the register allocator
    // sends to scratch memory
    auto h48 = [&]() -> int { return (fresh_lane_id() >> 5) * 48; };

    // LDS: [K frags nkt*2 KB][V frags nkt*2 KB][weight ring: 2 stage blocks of 25 KB]
    char* kbuf = smem + wsq * (nkt * 4 * TSF_FRAG);
    char* vbuf = kbuf + nkt * 2 * TSF_FRAG;
    char* ring = smem + nw * 4 * TSF_FRAG;
    // Ring of NSLOT stage blocks.  With four slots (where the LDS budget allows: the 9..12 tile variant without the parked
    // operand copy) the six feed-forward blocks are consumed in PAIRS -- one barrier per two blocks, fills issued two blocks ahead.
    constexpr int NSLOT = ring_slots<MAXW, PARK>();
    constexpr bool PAIR = NSLOT == 4;
    // PARK: the 16-bit operand copy of the residual stream (6 fragments per wave) lives in a wave-private LDS
    // area while the attention loops run, which frees 24 VGPRs for the software-pipelined score tiles
    constexpr int NPARK = parked_frags<MAXW, PARK>();      // fragments 6 - NPARK .. 5 of the operand copy
    char* xpark = ring + NSLOT * TSF_BLOCK + wave * NPARK * TSF_FRAG;
    const uint32_t ring_addr = __builtin_amdgcn_readfirstlane(LDS_ADDR(ring));
    const int nstage = A.depth * TSF_STAGES;

    auto slot_of = [&](int g) -> const char* { return ring + (g & (NSLOT - 1)) * TSF_BLOCK; };
    auto issue_fill = [&](int g) {            // stage block g -> ring slot g mod NSLOT, pieces spread over the waves
        const char* src = W + TSF_LAYER0 + (long)g * TSF_BLOCK + fresh_lane_id() * 16;
        const uint32_t dst = ring_addr + (uint32_t)(g & (NSLOT - 1)) * TSF_BLOCK;
        if (TSF_FILL_OLD) {
            // only the four first-dispatched waves (one per SIMD) request the pieces: they are the ones that reach every barrier early and wait
            // there (49 % of their time, profiles/r05_a_encoder_phase_table.md), while a piece costs the last-arriving wave 100+ cycles of the
            // workgroup's critical path behind every stage boundary
            int nf = nw < 4 ? nw : 4;                    // (workgroups of fewer than four waves: all of them)
            asm volatile("" : "+s"(nf));                 // (compared at every call: hoisted, `wave < nf` is kept as a 0 / 1 vector register in scratch memory)
            if (wave < nf)
                for (int pc = wave; pc < 25; pc += nf) dma_1k(src + pc * TSF_FRAG, dst + (uint32_t)pc * TSF_FRAG);
            return;
        }
        for (int pc = wave; pc < 25; pc += nw) dma_1k(src + pc * TSF_FRAG, dst + (uint32_t)pc * TSF_FRAG);
    };
    // Stage boundary: my DMA pieces of every block requested so far have landed (vmcnt), everybody's have (barrier), and
    // everybody is done with the blocks before g -- whose slots are then refilled, up to `ahead` blocks beyond g (block b lands
    // in the slot of block b - NSLOT, and ahead < NSLOT).
    int issued = 1;                           // next block to request (wave-uniform)
    auto stage_begin = [&](int g, int ahead) -> const char* {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int upto = g + ahead < nstage ? g + ahead : nstage - 1;
        for (; issued <= upto; ++issued) issue_fill(issued);
        return slot_of(g);
    };

    if (TSF_STATIC_PRIO == 1 && wave * 2 >= nw) __builtin_amdgcn_s_setprio(1);
    if (TSF_STATIC_PRIO == 2) {               // (A/B builds) the later a wave was dispatched onto its SIMD, the higher its priority: reverses the age order
        if (wave >= 8) __builtin_amdgcn_s_setprio(2);
        else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    bool first_block_requested = false;       // (persistent launch) block 0 of this sequence was requested during the previous one
#pragma unroll 1
    for (int seq0 = blockIdx.x * A.nseq; seq0 < A.S; seq0 += gridDim.x * A.nseq) {          // one iteration unless the launch is persistent (grid_limit)
    // (an odd number of sequences: the second half of the last workgroup computes its partner's sequence again and stores nothing)
    const bool seq_ok = seq0 + wsq < A.S;
    const int seq = seq_ok ? seq0 + wsq : seq0;
    // no barrier between sequences: the first stage_begin() of a sequence is one, and ring slot 0 was released two stages before the end
    if (!first_block_requested) issue_fill(0);
    issued = 1;
    first_block_requested = false;

    // ------------------------------------------------------------------ patch embedding + pos
    f32x16 xT[3];
    {
        float xin[12];
        // (per-lane addresses of this block are derived from a fresh lane id: as loop invariants of the persistent variant they
        //  would be kept alive -- in scratch memory -- across the whole sequence)
        const int lane_p = fresh_lane_id(), h_p = lane_p >> 5, tok_p = wv * 32 + (lane_p & 31);
        const bool tok_ok_p = tok_p < P;
        const int tokc_p = tok_ok_p ? tok_p : 0;
        const float4* src = (const float4*)(A.series + (long)seq * A.L + (long)tokc_p * TSF_PATCH);
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            float4 t4 = tok_ok_p ? src[v] : make_float4(0.f, 0.f, 0.f, 0.f);
            xin[4 * v] = t4.x; xin[4 * v + 1] = t4.y; xin[4 * v + 2] = t4.z; xin[4 * v + 3] = t4.w;
        }
        // x W_pe^T on the matrix cores in full f32 (v_mfma_f32_32x32x2_f32, K = 12 in six steps): A = W_pe rows (features) of
        // block t, stored per lane by the packer ([3][6][64] floats, one coalesced dword load each), B = this token's inputs
        // 2s + h; the accumulators start from pos + b_pe (pre-added by the packer) and come out in the layout of every other tile
        const float4* pos = (const float4*)(fresh_uniform(W) + TSF_POS_OFF(A.depth)) + ((long)tokc_p * 2 + h_p) * 12;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 p4 = pos[t * 4 + q4];
                xT[t][4 * q4] = p4.x; xT[t][4 * q4 + 1] = p4.y; xT[t][4 * q4 + 2] = p4.z; xT[t][4 * q4 + 3] = p4.w;
            }
        const float* wpe = (const float*)(W + TSF_G_WPE) + lane_p;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            // (a plain `h ? odd : even` is turned into a dynamically indexed private array, i.e. scratch memory)
            const uint32_t hm = 0u - (uint32_t)h_p;
            const float xb = __uint_as_float((__float_as_uint(xin[2 * ks]) & ~hm) | (__float_as_uint(xin[2 * ks + 1]) & hm));
#pragma unroll
            for (int t = 0; t < 3; ++t) xT[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wpe[(t * 6 + ks) * 64], xb, xT[t], 0, 0, 0);
        }
        float sc = 9.797958971132712f;   // sqrt(96), transformer_layers.py:15
        if constexpr (drop) {            // positional_encoding.py:32; the survivor scale rides on sqrt(d)
            const uint32_t chunk = drop_chunk_base(A.seed, (uint32_t)seq, (uint32_t)A.depth, pmask);
#pragma unroll
            for (int t = 0; t < 3; ++t) keep16(xT[t], mask_words(chunk, dl.d1 + (uint32_t)(wv * 3 + t) * 16u));
            sc *= fresh_uniform(inv_keep);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) xT[t][i] *= sc;
    }

    // ------------------------------------------------------------------ encoder layers
#if TSF_TIMING
    unsigned long long* tbuf = (g_tsf_timing && (seq & 63) == 7) ? g_tsf_timing + ((long)(seq >> 6) * 16 + wave) * (4 * TSF_NSTAMP) : nullptr;
#endif
    int g = 0;                                  // global stage index (10 per layer)
    int slow_units = 0;                         // (wave-uniform) heads of this wave that ran the re-shifting softmax loop
#pragma unroll 1
    for (int layer = 0; layer < A.depth; ++layer) {
        const uint32_t chunk = drop ? drop_chunk_base(A.seed, (uint32_t)seq, (uint32_t)layer, pmask) : 0u;
        op8 xb[6];
        f32x16 acc[3];

        // the first head's stage is opened outside the loop so that the f32 residual stream xT is
        // dead (folded into acc / xb) while the head loop runs
        TSF_STAMP(0);
        const char* blk = stage_begin(g, 1);
        const float* tail = (const float*)(blk + TSF_TAIL);
        {
#pragma unroll
            for (int t = 0; t < 3; ++t) { xb[2 * t] = pack_half<F16>(xT[t], 0); xb[2 * t + 1] = pack_half<F16>(xT[t], 1); }
            if constexpr (NPARK > 0) {
#pragma unroll
                for (int f = 6 - NPARK; f < 6; ++f) *(op8*)(xpark + (f - (6 - NPARK)) * TSF_FRAG + lane * 16) = xb[f];
            }
            const float* bo = tail + 64 + h48();
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = (drop ? 0.f : xT[t][i]) + bo[t * 16 + i];
        }
        bool skip_fast = false;                 // wave-uniform: a head of this layer overflowed the fixed-shift schedule
#pragma unroll 1
        for (int hd = 0; hd < TSF_HEADS; ++hd, ++g) {
            if (hd > 0) {
                blk = stage_begin(g, PAIR && hd == TSF_HEADS - 1 ? 2 : 1);      // the last head also requests the second ffn block
                tail = (const float*)(blk + TSF_TAIL);
            }
            TSF_STAMP(1 + 5 * hd);
            // the lane's K / V fragment addresses are derived per head from a fresh lane id: as kernel-lifetime values the register allocator
            // keeps them in scratch memory
            const int lane_a = fresh_lane_id();
            if constexpr (NPARK > 0) {
#pragma unroll
                for (int f = 6 - NPARK; f < 6; ++f) xb[f] = lfrag<F16>(xpark, f - (6 - NPARK), lane);
            }
            // ---- Q^T (kept in registers as the B operand of S^T = K Q^T), K^T, V.  Left to itself the compiler cycles a single
            // 4-register buffer through read -> wait -> MFMA, one exposed LDS latency per MFMA.
            TSF_PRIO_CHAIN(1);
            op8 qb[2];
#if TSF_QKV_PIPE == 3
            {
                // a ring of six fragment registers refilled one at a time: fragment i + 6 is requested right behind product i (static indices, the
                // scheduling fences pin each read there), so every product finds its fragment requested six products = ~190 cycles earlier
                op8 wr[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) wr[k] = lfrag<F16>(blk, k, lane);
                f32x16 q, kk, vv;
                const float* bq = tail + h * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = bq[i];           // bias rides in the accumulator
                const float bv = tail[32 + (fresh_lane_id() & 31)];      // (c kept alive across the sequence loop lands in scratch memory)
#pragma unroll
                for (int i = 0; i < 16; ++i) { kk[i] = 0.f; vv[i] = bv; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    if (i < 6) q = mfma16<F16>(wr[i % 6], xb[i], q);
                    else if (i < 12) kk = mfma16<F16>(wr[i % 6], xb[i - 6], kk);
                    else vv = mfma16<F16>(xb[i - 12], wr[i % 6], vv);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i + 6 < 18) {
                        wr[i % 6] = lfrag<F16>(blk, i + 6, lane);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (i == 8) {
                        qb[0] = pack_half<F16>(q, 0);
                        qb[1] = pack_half<F16>(q, 1);
                        if (h == 0) { qb[1][5] = (ope)0.0f; qb[1][6] = (ope)(-30000.0f); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (i == 14) {
                        if (h == 0) { kk[13] = 1.0f; kk[14] = (wv * 32 + (fresh_lane_id() & 31) < P) ? 0.0f : 1.0f; }     // slots 25 / 26 of this key (rows 25, 26)
                        *(op8*)(kbuf + (wv * 2 + 0) * TSF_FRAG + lane * 16) = pack_half<F16>(kk, 0);
                        *(op8*)(kbuf + (wv * 2 + 1) * TSF_FRAG + lane * 16) = pack_half<F16>(kk, 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                *(bf16x8*)(vbuf + (wv * 2 + 0) * TSF_FRAG + lane * 16) = pack_half<false>(vv, 0);
                *(bf16x8*)(vbuf + (wv * 2 + 1) * TSF_FRAG + lane * 16) = pack_half<false>(vv, 1);
            }
#elif TSF_QKV_PIPE == 2
            {
                // six-fragment buffers, each requested a whole chain ahead and pinned there: A = Wq, then Wv; B = Wk
                op8 wA[6], wB[6];
                auto load6 = [&](op8 (&w)[6], int f0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) w[k] = lfrag<F16>(blk, f0 + k, lane);
                };
                load6(wA, 0);
                f32x16 q;
                const float* bq = tail + h * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = bq[i];           // bias rides in the accumulator
                const float bv = tail[32 + (fresh_lane_id() & 31)];      // (c kept alive across the sequence loop lands in scratch memory)
                __builtin_amdgcn_sched_barrier(0);
                q = mfma16<F16>(wA[0], xb[0], q);
                __builtin_amdgcn_sched_barrier(0);
                load6(wB, 6);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 1; k < 6; ++k) q = mfma16<F16>(wA[k], xb[k], q);
                __builtin_amdgcn_sched_barrier(0);
                load6(wA, 12);
                __builtin_amdgcn_sched_barrier(0);
                f32x16 kk;
#pragma unroll
                for (int i = 0; i < 16; ++i) kk[i] = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) kk = mfma16<F16>(wB[k], xb[k], kk);
                qb[0] = pack_half<F16>(q, 0);
                qb[1] = pack_half<F16>(q, 1);
                if (h == 0) { qb[1][5] = (ope)0.0f; qb[1][6] = (ope)(-30000.0f); }
                __builtin_amdgcn_sched_barrier(0);
                f32x16 vv;
#pragma unroll
                for (int i = 0; i < 16; ++i) vv[i] = bv;
#pragma unroll
                for (int k = 0; k < 6; ++k) vv = mfma16<F16>(xb[k], wA[k], vv);
                if (h == 0) { kk[13] = 1.0f; kk[14] = (wv * 32 + (fresh_lane_id() & 31) < P) ? 0.0f : 1.0f; }     // slots 25 / 26 of this key (rows 25, 26)
                *(op8*)(kbuf + (wv * 2 + 0) * TSF_FRAG + lane * 16) = pack_half<F16>(kk, 0);
                *(op8*)(kbuf + (wv * 2 + 1) * TSF_FRAG + lane * 16) = pack_half<F16>(kk, 1);
                *(bf16x8*)(vbuf + (wv * 2 + 0) * TSF_FRAG + lane * 16) = pack_half<false>(vv, 0);
                *(bf16x8*)(vbuf + (wv * 2 + 1) * TSF_FRAG + lane * 16) = pack_half<false>(vv, 1);
            }
#else
            {
                // the 18 weight fragments are written as two rolling 3-fragment buffers (reads of the batch after the next issued
                // behind each batch's MFMAs); the order is a hint -- pinning it with scheduling fences (TSF_BATCH_FRAGS=1) costs
                // registers the attention loop needs (the operand copy lands in scratch), so the final schedule is the compiler's
                op8 wa[3], wb[3];
                auto load3 = [&](op8 (&w)[3], int f0) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) w[k] = lfrag<F16>(blk, f0 + k, lane);
                };
                auto fence = [&]() { if (TSF_BATCH_FRAGS || TSF_QKV_PIPE == 1) __builtin_amdgcn_sched_barrier(0); };
                load3(wa, 0);
                load3(wb, 3);
                f32x16 q;
                const float* bq = tail + h * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = bq[i];           // bias rides in the accumulator
                const float bv = tail[32 + (fresh_lane_id() & 31)];      // (c kept alive across the sequence loop lands in scratch memory)
                fence();
#pragma unroll
                for (int k = 0; k < 3; ++k) q = mfma16<F16>(wa[k], xb[k], q);
                load3(wa, 6);
                fence();
#pragma unroll
                for (int k = 0; k < 3; ++k) q = mfma16<F16>(wb[k], xb[3 + k], q);
                load3(wb, 9);
                fence();
                // ---- K^T -> this tile's A-operand fragments (bias dropped: it cancels in softmax)
                f32x16 kk;
#pragma unroll
                for (int i = 0; i < 16; ++i) kk[i] = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) kk = mfma16<F16>(wa[k], xb[k], kk);
                load3(wa, 12);
                qb[0] = pack_half<F16>(q, 0);
                qb[1] = pack_half<F16>(q, 1);
                // head-dim slots 25 / 26 (the head dim is 24 of 32) carry the softmax shift and the key-padding mask
                // through the contraction: the key side holds (1, is_padding), the query side (-shift, -30000)
                if (h == 0) { qb[1][5] = (ope)0.0f; qb[1][6] = (ope)(-30000.0f); }
                fence();
#pragma unroll
                for (int k = 0; k < 3; ++k) kk = mfma16<F16>(wb[k], xb[3 + k], kk);
                load3(wb, 15);
                fence();
                // ---- V (tokens as rows) -> this tile's A-operand fragments of V^T; always bfloat16 (like P): see the header
                f32x16 vv;
#pragma unroll
                for (int i = 0; i < 16; ++i) vv[i] = bv;
#pragma unroll
                for (int k = 0; k < 3; ++k) vv = mfma16<F16>(xb[k], wa[k], vv);
                if (h == 0) { kk[13] = 1.0f; kk[14] = (wv * 32 + (fresh_lane_id() & 31) < P) ? 0.0f : 1.0f; }     // slots 25 / 26 of this key (rows 25, 26)
                *(op8*)(kbuf + (wv * 2 + 0) * TSF_FRAG + lane * 16) = pack_half<F16>(kk, 0);
                *(op8*)(kbuf + (wv * 2 + 1) * TSF_FRAG + lane * 16) = pack_half<F16>(kk, 1);
                fence();
#pragma unroll
                for (int k = 0; k < 3; ++k) vv = mfma16<F16>(xb[3 + k], wb[k], vv);
                *(bf16x8*)(vbuf + (wv * 2 + 0) * TSF_FRAG + lane * 16) = pack_half<false>(vv, 0);
                *(bf16x8*)(vbuf + (wv * 2 + 1) * TSF_FRAG + lane * 16) = pack_half<false>(vv, 1);
            }
#endif
            TSF_PRIO_CHAIN(0);
            TSF_STAMP(2 + 5 * hd);
            // K/V fragments visible to every wave; the in-flight weight DMA is NOT drained here
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TSF_STAMP(3 + 5 * hd);

            // ---- one pass over the key tiles: S^T = K Q^T - shift, P = exp2(S^T), O^T += V^T P^T
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;
            auto score_tile = [&](int kt) -> f32x16 {
                TSF_PRIO_ATTN(1);
                f32x16 s = mfma16<F16>(lfrag<F16>(kbuf, kt * 2, lane_a), qb[0], zero);
                s = mfma16<F16>(lfrag<F16>(kbuf, kt * 2 + 1, lane_a), qb[1], s);
                TSF_PRIO_ATTN(0);
                return s;
            };
            f32x16 o;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = 0.f;
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            f32x2 lsum2 = {0.f, 0.f}, lsum2b = {0.f, 0.f};
            auto row_sums = [&](const f32x16& pr) {
                if (TSF_SUM_SCALAR) {
                    // plain adds, kept unpacked (the optimiser would re-fuse adjacent f32 adds into v_pk_add_f32)
                    float a0 = lsum2[0], a1 = lsum2[1], a2 = lsum2b[0], a3 = lsum2b[1];
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        asm("v_add_f32 %0, %1, %2" : "=v"(a0) : "v"(a0), "v"(pr[i]));
                        asm("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(a1), "v"(pr[i + 1]));
                        asm("v_add_f32 %0, %1, %2" : "=v"(a2) : "v"(a2), "v"(pr[i + 2]));
                        asm("v_add_f32 %0, %1, %2" : "=v"(a3) : "v"(a3), "v"(pr[i + 3]));
                    }
                    lsum2 = f32x2{a0, a1};
                    lsum2b = f32x2{a2, a3};
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {                          // v_pk_add_f32, two independent chains
                        lsum2 += f32x2{pr[i], pr[i + 1]};
                        lsum2b += f32x2{pr[i + 2], pr[i + 3]};
                    }
                }
            };
            float shift = 0.f;                       // what slot 25 currently subtracts (a value of the operand type)
            // test hook: re-shift whenever a tile holds a new running maximum (classic online softmax) instead of only
            // when the head room is used up
            const float thr = A.always_rescale ? -TSF_BIAS : TSF_THR;
            const uint32_t att_off = (uint32_t)((hd * nkt + wv) * nkt) * 16u;

            // Is a re-shift due?  (wave-uniform answer.)  tmax: this lane's largest score of the tile.
            auto tile_max = [&](const f32x16& sc) -> float {          // two interleaved v_max3_f32 chains (dependency depth 5)
                float m0 = fmaxf(fmaxf(sc[0], sc[1]), sc[2]), m1 = fmaxf(fmaxf(sc[3], sc[4]), sc[5]);
                m0 = fmaxf(fmaxf(m0, sc[6]), sc[7]);    m1 = fmaxf(fmaxf(m1, sc[8]), sc[9]);
                m0 = fmaxf(fmaxf(m0, sc[10]), sc[11]);  m1 = fmaxf(fmaxf(m1, sc[12]), sc[13]);
                return fmaxf(fmaxf(m0, sc[14]), fmaxf(m1, sc[15]));
            };
            // Re-shift: queries whose tile maximum exceeds the threshold (all queries on the first tile) move their shift to
            // TSF_BIAS above that maximum; O and the denominator follow by exp2(old - new) and the tile itself, computed against
            // the old shift, is corrected.  It runs BEFORE the next tile's score MFMAs are issued, so those see the new shift.
            auto reshift = [&](f32x16& cur, float tmax, bool first) {
                float lo, hi;
                both_halves(tmax, lo, hi);
                const float t = fmaxf(lo, hi);                        // maximum over the 32 keys of the tile, per query
                const bool upd = first || t > thr;
                const float ns = round_to_operand<F16>(shift + t + TSF_BIAS);
                const float d = upd ? ns - shift : 0.f;               // exact: both are 16-bit-significand values; > 0 unless first
                shift = upd ? ns : shift;
                const float a = first ? 1.0f : __builtin_amdgcn_exp2f(-d);      // first tile: O and the denominator are still zero
#pragma unroll
                for (int i = 0; i < 16; ++i) { o[i] *= a; cur[i] -= d; }
                lsum2 *= a;
                lsum2b *= a;
                if (h == 0) qb[1][5] = (ope)(-shift);
            };
            // keep-mask words of one score tile, fetched at the top of the tile's step so that the scalar-load latency hides
            // behind the tile maximum, the next tile's score MFMAs and the exponentials
            struct TileMask { unsigned long long w[16]; };
            auto load_mask = [&](int kt) -> TileMask {
                TileMask m;
                const mask_ptr mp = mask_words(chunk, att_off + (uint32_t)kt * 16u);
#pragma unroll
                for (int i = 0; i < 16; ++i) m.w[i] = mp[i];
                return m;
            };
            // the tail of a tile's step: (drop: keep-masks), pack to bfloat16, O^T += V^T P^T
            auto finish_tile = [&](f32x16& pr, int kt, const TileMask& tm) {
                const bf16x8 v0 = lfrag<false>(vbuf, kt * 2, lane_a), v1 = lfrag<false>(vbuf, kt * 2 + 1, lane_a);
                if constexpr (drop) {
                    if (!(TSF_ABLATE & 2)) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) pr[i] = __builtin_amdgcn_inverse_ballot_w64(tm.w[i]) ? pr[i] : 0.f;
                    }
                }
                const bf16x8 p0 = pack_half<false>(pr, 0), p1 = pack_half<false>(pr, 1);
                TSF_PRIO_ATTN(1);
                o = mfma16<false>(v0, p0, o);
                o = mfma16<false>(v1, p1, o);
                TSF_PRIO_ATTN(0);
            };
            auto exp_tile = [&](f32x16& pr, const f32x16& sc) {
                if (TSF_ABLATE & 1) { pr = sc; return; }
#pragma unroll
                for (int i = 0; i < 16; ++i) pr[i] = __builtin_amdgcn_exp2f(sc[i]);
            };
            // One key tile: decide / re-shift on `cur`, put the next tile's score MFMAs in flight, then the VALU work on `cur`.
            auto tile_step = [&](f32x16& cur, f32x16& nxt, int kt) {
                TileMask tm;
                if constexpr (drop && TSF_MASK_EARLY) {
                    tm = load_mask(kt);
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    if (!(TSF_ABLATE & 4)) {
                        const float tmax = tile_max(cur);
                        if (kt == 0 || __builtin_amdgcn_ballot_w64(tmax > thr) != 0) reshift(cur, tmax, kt == 0);
                    } else if (kt == 0) {
                        reshift(cur, tile_max(cur), true);
                    }
                    if (PIPE != 0 && kt + 1 < nkt) nxt = score_tile(kt + 1);
                    if constexpr (drop && !TSF_MASK_EARLY) tm = load_mask(kt);
                    exp_tile(cur, cur);
                    if constexpr (drop) {
                        if (!(TSF_ABLATE & 16)) row_sums(cur);
                    }
                    finish_tile(cur, kt, tm);
                }
            };
            // fast schedule: one tile with the shift already fixed -- no maximum, no branch.  LDS and scalar loads share one
            // wait counter and scalar loads return out of order, so a wait for ANY LDS read also waits for every scalar load in
            // flight.  Hence the order: all four operand fragments of the step are requested at its top and waited for once
            // (behind the exponentials); the keep-mask words are loop carried, the next tile's being fetched into the same
            // scalar registers right after this tile's selects have consumed them -- nothing waits on that counter again
            // before the next step's exponentials are done.
            // (Measured and not kept, round 4: issuing score(k-step 0), P V (half 0), score(k-step 1), P V (half 1) with each half's selects and
            //  packs in between -- no matrix instruction right behind the one it depends on -- costs 20 spilled registers in the unparked
            //  variants and is 1 % SLOWER in the parked ones: 1.922 vs 1.905 ms at P = 336, 1.648 vs 1.652 at P = 168,
            //  profiles/r04_a_enc_ab_*.log.)
            auto fast_step = [&](f32x16& cur, f32x16& nxt, int kt, TileMask& tm, auto has_next) {
                constexpr bool NEXT = decltype(has_next)::value;
#if TSF_ATTN_IL
                // Interleaved order (round 5): a wave issues in order, so a matrix product right behind the product it depends on, or right behind the
                // packs that feed it, stalls the wave's whole instruction stream for the product's 32 cycles.  Here every product has >= 32 cycles of
                // independent vector work behind it: score product 1 | second half of the exponentials + row sums | score product 2 (needs 1) |
                // first eight selects + packs | P V product 1 | last eight selects + packs | P V product 2 (needs 1).  Same arithmetic, same order
                // inside every chain: bit-identical results.
                if constexpr (NEXT) {
                    const op8 k0 = lfrag<F16>(kbuf, kt * 2 + 2, lane_a), k1 = lfrag<F16>(kbuf, kt * 2 + 3, lane_a);
                    const bf16x8 v0 = lfrag<false>(vbuf, kt * 2, lane_a), v1 = lfrag<false>(vbuf, kt * 2 + 1, lane_a);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) cur[i] = __builtin_amdgcn_exp2f(cur[i]);
                    __builtin_amdgcn_sched_barrier(0);
                    nxt = mfma16<F16>(k0, qb[0], zero);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 8; i < 16; ++i) cur[i] = __builtin_amdgcn_exp2f(cur[i]);
                    if constexpr (drop) row_sums(cur);
                    __builtin_amdgcn_sched_barrier(0);
                    nxt = mfma16<F16>(k1, qb[1], nxt);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (drop) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) cur[i] = __builtin_amdgcn_inverse_ballot_w64(tm.w[i]) ? cur[i] : 0.f;
                    }
                    const bf16x8 p0 = pack_half<false>(cur, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    o = mfma16<false>(v0, p0, o);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (drop) {
#pragma unroll
                        for (int i = 8; i < 16; ++i) cur[i] = __builtin_amdgcn_inverse_ballot_w64(tm.w[i]) ? cur[i] : 0.f;
                        // (the last fragment read of the step is waited for HERE: behind the scalar loads below, a wait for it would be a wait
                        //  for them as well -- they return out of order -- and expose their whole latency in every step)
                        asm volatile("" :: "v"(v1));
                        __builtin_amdgcn_sched_barrier(0);
                        tm = load_mask(kt + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const bf16x8 p1 = pack_half<false>(cur, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    o = mfma16<false>(v1, p1, o);
                    return;
                }
#endif
                const bf16x8 v0 = lfrag<false>(vbuf, kt * 2, lane_a), v1 = lfrag<false>(vbuf, kt * 2 + 1, lane_a);
                op8 k0, k1;
                if constexpr (NEXT) {
                    k0 = lfrag<F16>(kbuf, kt * 2 + 2, lane_a);
                    k1 = lfrag<F16>(kbuf, kt * 2 + 3, lane_a);
                }
                __builtin_amdgcn_sched_barrier(0);
                exp_tile(cur, cur);
                if constexpr (drop) row_sums(cur);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NEXT) {
                    if (TSF_ABLATE & 256) { asm volatile("" :: "v"(k0), "v"(k1)); nxt = cur; }
                    else {
                        nxt = mfma16<F16>(k0, qb[0], zero);
                        nxt = mfma16<F16>(k1, qb[1], nxt);
                    }
                }
                if constexpr (drop) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) cur[i] = __builtin_amdgcn_inverse_ballot_w64(tm.w[i]) ? cur[i] : 0.f;
                    if constexpr (NEXT) {
                        __builtin_amdgcn_sched_barrier(0);
                        tm = load_mask(kt + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                const bf16x8 p0 = pack_half<false>(cur, 0), p1 = pack_half<false>(cur, 1);
                if (TSF_ABLATE & 256) { asm volatile("" :: "v"(p0), "v"(p1), "v"(v0), "v"(v1)); }
                else {
                    o = mfma16<false>(v0, p0, o);
                    o = mfma16<false>(v1, p1, o);
                }
            };
            // The LAST key tile when only its first 8 * NG keys exist (P = 168: 8 of 32, NG = 1; P = 336: 16, NG = 2).  Keys 8 g .. 8 g + 7 are
            // accumulator registers 4 g .. 4 g + 3 of both lane halves (tsformer_layout.h), and the probability of a padded key is exactly 0
            // (its score carries the -30000 of slot 26), so their exponentials, row sums, selects and packs -- and, from NG <= 2, the whole second
            // k-step of the P V product -- are skipped: same bits, 3/4 (1/2) of this tile's vector work less.
            auto fast_last = [&](f32x16& cur, int kt, TileMask& tm, auto groups) {
                constexpr int NG = decltype(groups)::value;
                const int lane_l = fresh_lane_id();          // (as a value kept alive from the kernel's start the address below lands in scratch memory)
                const bf16x8 v0 = lfrag<false>(vbuf, kt * 2, lane_l);
                bf16x8 v1;
                if constexpr (NG > 2) v1 = lfrag<false>(vbuf, kt * 2 + 1, lane_l);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) cur[i] = i < 4 * NG ? ((TSF_ABLATE & 1) ? cur[i] : __builtin_amdgcn_exp2f(cur[i])) : 0.f;
                if constexpr (drop) {
#pragma unroll
                    for (int i = 0; i < 4 * NG; i += 4) {
                        lsum2 += f32x2{cur[i], cur[i + 1]};
                        lsum2b += f32x2{cur[i + 2], cur[i + 3]};
                    }
#pragma unroll
                    for (int i = 0; i < 4 * NG; ++i) cur[i] = __builtin_amdgcn_inverse_ballot_w64(tm.w[i]) ? cur[i] : 0.f;
                }
                o = mfma16<false>(v0, pack_half<false>(cur, 0), o);
                if constexpr (NG > 2) o = mfma16<false>(v1, pack_half<false>(cur, 1), o);
            };
            auto last_step = [&](f32x16& cur, f32x16& other, int kt, TileMask& tm) {
                if constexpr (TG == 1) fast_last(cur, kt, tm, G1{});
                else if constexpr (TG == 2) fast_last(cur, kt, tm, G2{});
                else if constexpr (TG == 3) fast_last(cur, kt, tm, G3{});
                else fast_step(cur, other, kt, tm, No{});
            };
            auto denominator = [&]() -> float {
                if constexpr (drop) {
                    float lo, hi;
                    both_halves((lsum2[0] + lsum2[1]) + (lsum2b[0] + lsum2b[1]), lo, hi);
                    return (lo + hi) * keep;
                } else {
                    float lo, hi;
                    both_halves(o[12], lo, hi);      // V^T row 24 == ones: lane-half 0, register 12
                    return lo;
                }
            };
            float den = 1.0f;
            bool redo = true;                        // wave-uniform
            if (TSF_FAST_ATTN && !(TSF_ABLATE & 32) && !A.always_rescale && !skip_fast) {
                f32x16 sa = score_tile(0), sb;
                TileMask tm;
                if constexpr (drop) tm = load_mask(0);
                reshift(sa, tile_max(sa), true);
                int kt = 0;
#pragma unroll 1
                for (; kt + 2 < nkt; kt += 2) {             // both tiles of the pair have a successor
                    if (TSF_PROGRESS_PRIO == 1) prio_level(3 - (kt * 4) / nkt);
                    if (TSF_PROGRESS_PRIO == 2) prio_level(((kt + (wave >> 2)) % 3 == 0) ? 1 : 0);                 // round-robin token per step
                    if (TSF_PROGRESS_PRIO == 3) prio_level(2 * (kt * 2 < nkt ? 1 : 0) + (((kt + (wave >> 2)) % 3 == 0) ? 1 : 0));
                    if (TSF_PROGRESS_PRIO == 4) prio_level(3 - (kt * 4) / nkt);
                    fast_step(sa, sb, kt, tm, Yes{});
                    if (TSF_PROGRESS_PRIO == 2) prio_level(((kt + 1 + (wave >> 2)) % 3 == 0) ? 1 : 0);
                    if (TSF_PROGRESS_PRIO == 3) prio_level(2 * (kt * 2 < nkt ? 1 : 0) + (((kt + 1 + (wave >> 2)) % 3 == 0) ? 1 : 0));
                    if (TSF_PROGRESS_PRIO == 4) prio_level(3 - ((kt + 1) * 4) / nkt);
                    fast_step(sb, sa, kt + 1, tm, Yes{});
                }
                if (kt + 1 < nkt) {
                    fast_step(sa, sb, kt, tm, Yes{});
                    last_step(sb, sa, kt + 1, tm);
                } else {
                    last_step(sa, sb, kt, tm);
                }
                if (TSF_PROGRESS_PRIO) __builtin_amdgcn_s_setprio(0);
                den = denominator();
                redo = (TSF_ABLATE & 256) ? false : __builtin_amdgcn_ballot_w64(!(den < TSF_LIMIT)) != 0;
                if (redo) {
                    skip_fast = true;                // the other heads of this layer see the same tokens: straight to the re-shifting loop
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[i] = 0.f;
                    lsum2 = f32x2{0.f, 0.f};
                    lsum2b = f32x2{0.f, 0.f};
                    shift = 0.f;
                    if (h == 0) qb[1][5] = (ope)0.0f;
                }
            }
            slow_units = __builtin_amdgcn_readfirstlane(slow_units + (redo ? 1 : 0));
            if (!redo) {
            } else if (TSF_ABLATE & 32) {
            } else if constexpr (PIPE == 2) {
                // software pipeline over two alternating score tiles (no register copies)
                f32x16 sa = score_tile(0), sb;
                int kt = 0;
#pragma unroll 1
                for (; kt + 1 < nkt; kt += 2) {
                    tile_step(sa, sb, kt);
                    tile_step(sb, sa, kt + 1);
                }
                if (kt < nkt) tile_step(sa, sb, kt);
            } else if constexpr (PIPE == 1) {
                f32x16 sa = score_tile(0);
#pragma unroll 1
                for (int kt = 0; kt < nkt; ++kt) {
                    f32x16 sb;
                    tile_step(sa, sb, kt);
                    sa = sb;
                }
            } else {
#pragma unroll 1
                for (int kt = 0; kt < nkt; ++kt) {
                    f32x16 sa = score_tile(kt);
                    tile_step(sa, sa, kt);
                }
            }
            if (redo) den = denominator();
            TSF_STAMP(4 + 5 * hd);
            op8 wo[6];
            if (TSF_QKV_PIPE >= 2) {           // the out-projection's fragments are requested in front of the normalisation's vector work
#pragma unroll
                for (int f = 0; f < 6; ++f) wo[f] = lfrag<F16>(blk, 18 + f, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
            const float inv = 1.0f / den;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] *= inv;
            op8 ob0 = pack_half<F16>(o, 0), ob1 = pack_half<F16>(o, 1);
            // ---- out-projection of this head accumulates onto the residual
            TSF_PRIO_CHAIN(1);
            {
                if (TSF_QKV_PIPE < 2) {
#pragma unroll
                    for (int f = 0; f < 6; ++f) wo[f] = lfrag<F16>(blk, 18 + f, lane);
                }
                if (TSF_BATCH_FRAGS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t] = mfma16<F16>(wo[t * 2], ob0, acc[t]);
                    acc[t] = mfma16<F16>(wo[t * 2 + 1], ob1, acc[t]);
                }
            }
            TSF_PRIO_CHAIN(0);
            TSF_STAMP(5 + 5 * hd);
        }  // heads
        if constexpr (drop) {
            // dropout1 on (attention output + b_o); residual re-read from its 16-bit operand copy
            if constexpr (NPARK > 0) {
#pragma unroll
                for (int f = 6 - NPARK; f < 6; ++f) xb[f] = lfrag<F16>(xpark, f - (6 - NPARK), lane);
            }
            const mask_ptr w1[3] = {mask_words(chunk, dl.d1 + (uint32_t)(wv * 3) * 16u), mask_words(chunk, dl.d1 + (uint32_t)(wv * 3 + 1) * 16u),
                                    mask_words(chunk, dl.d1 + (uint32_t)(wv * 3 + 2) * 16u)};
            add_residual_op<F16>(acc, xb, w1, inv_keep);
        }
        layer_norm96<LN2F>(acc, tail + 64 + h48(), tail + 160 + h48());      // LN1 params ride in head 3's block

        // ---- FFN 96 -> 384 -> 96: 6 stages of two 32-unit chunks, hidden units never leave registers
        TSF_STAMP(21);
        blk = stage_begin(g, PAIR ? 3 : 1);
        TSF_STAMP(22);
        tail = (const float*)(blk + TSF_TAIL);
        {
#pragma unroll
            for (int t = 0; t < 3; ++t) { xb[2 * t] = pack_half<F16>(acc[t], 0); xb[2 * t + 1] = pack_half<F16>(acc[t], 1); }
            const float* b2 = tail + 64 + h48();
            // training: the hidden-unit survivor scale is NOT applied per element; b2 is pre-multiplied by keep instead and the
            // whole sub-layer output gets 1/keep^2 in the residual fma (one scale for the FFN dropout, one for dropout2)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = drop ? b2[t * 16 + i] * keep : acc[t][i] + b2[t * 16 + i];
        }
#if TSF_FFN_PIPE
        // Groups of NC chunks between two stage boundaries (PAIR: four chunks = two blocks, else two).  Inside a group the fragment reads run one
        // matrix-product group ahead of their use: U (the six W1 fragments of chunk q + 1) is requested behind the W1 chain of chunk q and has
        // the chunk's vector work and its W2 products to arrive; D (W2 of chunk q) is requested behind the first product of the chunk's W1 chain
        // and has the rest of the chain and the vector work.  The scheduling fences keep the compiler from sinking the reads back to their uses.
        {
        constexpr int NBLK = PAIR ? 2 : 1, NC = 2 * NBLK;
#pragma unroll 1
        for (int j = 0; j < 6; j += NBLK, g += NBLK) {
            if (j > 0) {
                TSF_STAMP(21 + 2 * j);
                blk = stage_begin(g, PAIR ? 3 : 1);
                TSF_STAMP(22 + 2 * j);
                if (PAIR && j == 4 && layer == A.depth - 1 && seq0 + (int)gridDim.x * A.nseq < A.S) {
                    issue_fill(0);
                    first_block_requested = true;
                }
            }
            const char* blk2 = PAIR ? slot_of(g + 1) : blk;
            tail = (const float*)(blk2 + TSF_TAIL);                 // after the loop: the LN2 parameters ride in the last block
            op8 U[6], D[6];
            f32x16 hh;
            unsigned long long mw[16];
            auto cblk = [&](int q) -> const char* { return q < 2 ? blk : blk2; };
            auto load_U = [&](int q) {
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) U[ks] = lfrag<F16>(cblk(q), (q & 1) * 12 + ks, lane);
            };
            auto load_D = [&](int q) {
#pragma unroll
                for (int f = 0; f < 6; ++f) D[f] = lfrag<F16>(cblk(q), (q & 1) * 12 + 6 + f, lane);
            };
            auto load_b1 = [&](int q) {
                const float* b1 = (const float*)(cblk(q) + TSF_TAIL) + ((q & 1) * 2 + h) * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) hh[i] = b1[i];
            };
            auto load_mw = [&](int q) {
                if constexpr (drop) {
                    const mask_ptr mp = mask_words(chunk, dl.ffn + (uint32_t)(wv * 12 + j * 2 + q) * 16u);
#pragma unroll
                    for (int i = 0; i < 16; ++i) mw[i] = mp[i];
                }
            };
            load_U(0);
            load_b1(0);
            load_mw(0);
            __builtin_amdgcn_sched_barrier(0);
            TSF_PRIO_CHAIN(1);
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                if (TSF_PROGRESS_PRIO) prio_level(3 - (j * 2 + q) / 3);
                // (scalar loads return out of order, so the wait in front of the chain's first product is for EVERYTHING in flight: the reads of
                //  D are issued behind that product, not in front of it)
                hh = mfma16<F16>(U[0], xb[0], hh);
                __builtin_amdgcn_sched_barrier(0);
                load_D(q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 1; ks < 6; ++ks) hh = mfma16<F16>(U[ks], xb[ks], hh);
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < NC) load_U(q + 1);
                // every fragment of D has to be here BEFORE the next scalar load is issued: with scalar loads in flight any later wait for an LDS
                // read becomes a wait for everything (they return out of order).  The empty statement below makes the compiler wait for D's
                // last fragment now -- LDS reads return in order, so that is lgkmcnt(6) with U still in flight.
                asm volatile("" :: "v"(D[5]));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (drop) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) hh[i] = __builtin_amdgcn_inverse_ballot_w64(mw[i]) ? hh[i] : 0.f;
                    if (q + 1 < NC) {
                        __builtin_amdgcn_sched_barrier(0);
                        load_mw(q + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                const op8 hb0 = relu_packed<F16>(pack_half<F16>(hh, 0)), hb1 = relu_packed<F16>(pack_half<F16>(hh, 1));
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < NC) {
                    load_b1(q + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t] = mfma16<F16>(D[t * 2], hb0, acc[t]);
                    acc[t] = mfma16<F16>(D[t * 2 + 1], hb1, acc[t]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TSF_PRIO_CHAIN(0);
        }
        }
#else
#pragma unroll 1
        for (int j = 0; j < ((TSF_ABLATE & 64) ? 1 : 6); ++j, g += ((TSF_ABLATE & 64) ? 6 : 1)) {
            if (j > 0) {
                if (!(PAIR && (j & 1))) TSF_STAMP(21 + 2 * j);
                blk = (PAIR && (j & 1)) ? slot_of(g) : stage_begin(g, PAIR ? 3 : 1);      // PAIR: odd blocks arrived with their predecessor
                if (!(PAIR && (j & 1))) TSF_STAMP(22 + 2 * j);
                tail = (const float*)(blk + TSF_TAIL);
                if (PAIR && j == 4 && layer == A.depth - 1 && seq0 + (int)gridDim.x * A.nseq < A.S) {
                    issue_fill(0);            // slot 0 (block g - 2) is free since this stage's barrier: the next sequence's first block
                    first_block_requested = true;
                }
            }
            TSF_PRIO_CHAIN(1);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                if (TSF_PROGRESS_PRIO) prio_level(3 - (j * 2 + cc) / 3);
                // both weight fragment sets of the chunk and its keep-mask words are requested up front
                op8 wu[6], wd[6];
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) wu[ks] = lfrag<F16>(blk, cc * 12 + ks, lane);
                f32x16 hh;
                const float* b1 = tail + (cc * 2 + h) * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) hh[i] = b1[i];
#pragma unroll
                for (int f = 0; f < 6; ++f) wd[f] = lfrag<F16>(blk, cc * 12 + 6 + f, lane);
                unsigned long long mw[16];
                if constexpr (drop) {
                    const mask_ptr mp = mask_words(chunk, dl.ffn + (uint32_t)(wv * 12 + j * 2 + cc) * 16u);
#pragma unroll
                    for (int i = 0; i < 16; ++i) mw[i] = mp[i];
                }
                if (TSF_BATCH_FRAGS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) hh = mfma16<F16>(wu[ks], xb[ks], hh);
                op8 hb0, hb1;
                if (TSF_RELU_PACKED) {
                    // ReLU after the pack, on 16-bit pairs: a negative float16 / bfloat16 is a negative int16, so one
                    // v_pk_max_i16 against zero per two hidden units replaces two f32 clamps (rounding commutes with the clamp)
                    if constexpr (drop) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) hh[i] = __builtin_amdgcn_inverse_ballot_w64(mw[i]) ? hh[i] : 0.f;
                    }
                    hb0 = relu_packed<F16>(pack_half<F16>(hh, 0));
                    hb1 = relu_packed<F16>(pack_half<F16>(hh, 1));
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) hh[i] = __builtin_amdgcn_fmed3f(hh[i], 0.f, 3.0e38f);      // relu, one VALU op
                    if constexpr (drop) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) hh[i] = __builtin_amdgcn_inverse_ballot_w64(mw[i]) ? hh[i] : 0.f;
                    }
                    hb0 = pack_half<F16>(hh, 0);
                    hb1 = pack_half<F16>(hh, 1);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc[t] = mfma16<F16>(wd[t * 2], hb0, acc[t]);
                    acc[t] = mfma16<F16>(wd[t * 2 + 1], hb1, acc[t]);
                }
            }
            TSF_PRIO_CHAIN(0);
        }
#endif
        if (TSF_PROGRESS_PRIO) __builtin_amdgcn_s_setprio(0);
        TSF_STAMP(34);
        if constexpr (drop) {
            const mask_ptr w2[3] = {mask_words(chunk, dl.d2 + (uint32_t)(wv * 3) * 16u), mask_words(chunk, dl.d2 + (uint32_t)(wv * 3 + 1) * 16u),
                                    mask_words(chunk, dl.d2 + (uint32_t)(wv * 3 + 2) * 16u)};
            add_residual_op<F16>(acc, xb, w2, A.inv_keep2);
        }
        layer_norm96<LN2F>(acc, tail + 64 + h48(), tail + 160 + h48());      // LN2 params ride in the last ffn block
        TSF_STAMP(35);
#pragma unroll
        for (int t = 0; t < 3; ++t) xT[t] = acc[t];
    }  // layers

    // one atomic per wave that took the slow path at all, spread over 64 counters (100 000 adds to one address cost 0.1 ms)
    if (A.fallback != nullptr && slow_units > 0 && seq_ok && fresh_lane_id() == 0) atomicAdd(A.fallback + (blockIdx.x & 63), (unsigned)slow_units);

    // ------------------------------------------------------------------ encoder_norm + outputs
    // Lane-derived indices are re-derived here from a fresh lane id: kept alive from the kernel's start they sit in scratch memory
    // for its whole duration (64 bytes per lane and wave of HBM write-back for nothing).
    const int lane_e = fresh_lane_id();
    const int h_e = lane_e >> 5;
    const int tok_e = wv * 32 + (lane_e & 31);
    layer_norm96<LN2F>(xT, (const float*)(W + TSF_G_NORM_G) + h_e * 48, (const float*)(W + TSF_G_NORM_B) + h_e * 48);
    float sq = 0.f;
    if (tok_e < P && seq_ok) {
        const long row = ((long)seq * P + tok_e) * TSF_D;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int f0 = t * 32 + 8 * g4 + 4 * h_e;
                float v0 = xT[t][4 * g4], v1 = xT[t][4 * g4 + 1], v2 = xT[t][4 * g4 + 2], v3 = xT[t][4 * g4 + 3];
                u32x2 pk;
                pk[0] = pack_bf16x2(v0, v1);
                pk[1] = pack_bf16x2(v2, v3);
                if (A.hid_bf16) *(u32x2*)(A.hid_bf16 + row + f0) = pk;
                if (A.hid_f32) *(float4*)(A.hid_f32 + row + f0) = make_float4(v0, v1, v2, v3);
                if (A.last_f32 && tok_e == P - 1) *(float4*)(A.last_f32 + (long)seq * TSF_D + f0) = make_float4(v0, v1, v2, v3);
                float r0 = bf16_bits_to_f32(pk[0] & 0xffffu), r1 = bf16_bits_to_f32(pk[0] >> 16);
                float r2 = bf16_bits_to_f32(pk[1] & 0xffffu), r3 = bf16_bits_to_f32(pk[1] >> 16);
                sq += r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
            }
    }
    if (A.sqn && seq_ok) {
        sq = wave_sum_swz(sq);
        if (lane_e == 0) {
            A.sqn[(long)seq * 16 + wv] = sq;
            // float16 operands overflow at 65 504 (round-to-nearest packs give inf, the matrix cores carry it into the residual stream, the
            // final LayerNorm turns it into NaN): the squared norm this wave just formed says so for free -- one compare per wave
            // (STEP_ENC_RANGE_FLAG; the host passes sqnorm_part whenever it asks for the flag)
            if (A.range_word && !(sq <= 3.0e38f)) atomicOr(A.range_word, 1u);
        }
        if (wv == 0 && lane_e >= nkt && lane_e < 16) A.sqn[(long)seq * 16 + lane_e] = 0.f;
    }
    }   // sequences of this workgroup
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

#ifndef TSF_PIPE
#define TSF_PIPE 2          // attention loop schedule (A/B knob): 0 plain, 1 next tile's score MFMAs in flight during this tile's
#endif                      // VALU work (one tile copy per iteration), 2 the same over two alternating tiles (no copies)

template <int MAXW, bool DROP, bool PARK, bool F16, int TG>
int launch_enc_tg(const EncArgs& a, hipStream_t st) {
    const int nw = a.nkt * a.nseq;
    size_t lds = (size_t)nw * 4 * TSF_FRAG + (size_t)ring_slots<MAXW, PARK>() * TSF_BLOCK + (size_t)nw * parked_frags<MAXW, PARK>() * TSF_FRAG;
    // once per device and instantiation (the attribute is per device): not on every launch, so that a launch inside a stream capture
    // (step_amd.GraphedTrainStep, after its eager warm-up steps) is a plain kernel node
    static bool raised[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!__atomic_load_n(&raised[dev & 15], __ATOMIC_ACQUIRE)) {          // (two host threads racing here both set the same attribute: harmless)
        hipError_t e = hipFuncSetAttribute((const void*)tsformer_encoder_kernel<MAXW, DROP, PARK, F16, TSF_PIPE, TG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            step_set_error("tsformer_encode: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
            return STEP_ERR_HIP;
        }
        __atomic_store_n(&raised[dev & 15], true, __ATOMIC_RELEASE);
    }
    const int units = (a.S + a.nseq - 1) / a.nseq, limit = (a.grid_limit + a.nseq - 1) / a.nseq;      // (the limit counts one-sequence workgroups)
    const int grid = (a.grid_limit > 0 && units > limit) ? limit : units;
    tsformer_encoder_kernel<MAXW, DROP, PARK, F16, TSF_PIPE, TG><<<grid, nw * 64, lds, st>>>(a);
    STEP_LAUNCH_CHECK("step_tsformer_encode");
    return STEP_OK;
}
// The last key tile's shortcut is instantiated where the reference's configurations land: 8 keys (P = 168: METR-LA, PEMS-BAY, PEMS07) in
// the 5..8 tile variants, 16 keys (P = 336: PEMS03 / 04 / 08) in the 9..12 tile variants; every other P runs the generic last step.
template <int MAXW, bool DROP, bool PARK, bool F16>
int launch_enc_t(const EncArgs& a, hipStream_t st) {
    constexpr int TGX = !TSF_TAIL_GROUPS ? 4 : MAXW == 8 ? 1 : MAXW == 12 ? 2 : 4;
    const int tail_keys = a.P - (a.nkt - 1) * 32;
    if constexpr (MAXW == 12 && !PARK && TSF_TAIL_GROUPS) {         // two sequences of <= 192 tokens per workgroup: P = 168 ends on 8 keys
        if (a.nseq == 2 && tail_keys == 8) return launch_enc_tg<MAXW, DROP, PARK, F16, 1>(a, st);
    }
    if (TGX != 4 && tail_keys == 8 * TGX) return launch_enc_tg<MAXW, DROP, PARK, F16, TGX>(a, st);
    return launch_enc_tg<MAXW, DROP, PARK, F16, 4>(a, st);
}
template <int MAXW, bool DROP, bool PARK>
int launch_enc(const EncArgs& a, hipStream_t st) {
    return a.f16 ? launch_enc_t<MAXW, DROP, PARK, true>(a, st) : launch_enc_t<MAXW, DROP, PARK, false>(a, st);
}

// ---------------------------------------------------------------------------------------
// Dropout pool: word w = 64 Bernoulli(keep) bits, bit l <- Philox4x32-10(counter (w, l / 4), key seed) word l % 4 >= drop * 2^32.
// One thread per Philox call (4 bits), 16 threads per word; the nibbles are OR-combined with four xor-shuffles.
__global__ __launch_bounds__(256) void dropout_pool_kernel(unsigned long long* __restrict__ pool, long words, uint32_t thresh,
                                                           uint32_t k0, uint32_t k1, const StepDynState* __restrict__ dyn) {
    if (dyn) { const uint64_t x = dyn->seed_xor; k0 ^= (uint32_t)x; k1 ^= (uint32_t)(x >> 32); }       // replayed steps: the key moves on the device
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long w = id >> 4;
    const int q = (int)(id & 15);
    uint32_t r[4] = {0u, 0u, 0u, 0u};
    if (w < words) philox4x32((uint32_t)w, (uint32_t)(w >> 32), (uint32_t)q, 0x5EEDD80Fu, k0, k1, r);
    const uint32_t nib = (r[0] >= thresh ? 1u : 0u) | (r[1] >= thresh ? 2u : 0u) | (r[2] >= thresh ? 4u : 0u) | (r[3] >= thresh ? 8u : 0u);
    unsigned long long m = (unsigned long long)nib << (4 * q);
    uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { lo |= __shfl_xor(lo, o, 64); hi |= __shfl_xor(hi, o, 64); }
    if (q == 0 && w < words) {
        const unsigned long long v = ((unsigned long long)hi << 32) | lo;
        pool[w] = v;
        if (w < 16) pool[words + w] = v;          // the wrap-around copy behind the pool (see tsformer_device.h)
    }
}

}  // namespace

extern "C" int step_dropout_pool_fill(uint64_t* pool, long words, float dropout_p, uint64_t seed, void* stream) {
    return step_dropout_pool_fill_dyn(pool, words, dropout_p, seed, nullptr, stream);
}
extern "C" int step_dropout_pool_fill_dyn(uint64_t* pool, long words, float dropout_p, uint64_t seed, const StepDynState* dyn, void* stream) {
    STEP_REQUIRE(pool, "dropout_pool_fill: null pool");
    STEP_REQUIRE(words >= 16 && (words & (words - 1)) == 0, "dropout_pool_fill: %ld words is not a power of two >= 16", words);
    STEP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout_pool_fill: bad dropout %f", dropout_p);
    const uint32_t thresh = (uint32_t)((double)dropout_p * 4294967296.0);
    dropout_pool_kernel<<<cdiv(words, 16), 256, 0, (hipStream_t)stream>>>((unsigned long long*)pool, words, thresh, (uint32_t)seed,
                                                                         (uint32_t)(seed >> 32), dyn);
    STEP_LAUNCH_CHECK("step_dropout_pool_fill");
    return STEP_OK;
}

#if TSF_TIMING
// timing builds only (tools/enc_ab.cpp): buffer of [ceil(S / 64)][16 waves][4 layers][TSF_NSTAMP] s_memtime stamps, written by the workgroups
// whose sequence index is 7 mod 64; NULL switches the stamps off
extern "C" int step_tsformer_set_timing(unsigned long long* buf) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_tsf_timing), &buf, sizeof(buf));
    return e == hipSuccess ? STEP_OK : STEP_ERR_HIP;
}
extern "C" int step_tsformer_timing_stamps(void) { return TSF_NSTAMP; }
#endif

extern "C" long step_tsformer_dropout_words(int L, int depth) {
    if (L <= 0 || L % TSF_PATCH != 0 || depth < 1) return 0;
    const int nkt = (L / TSF_PATCH + 31) / 32;
    return (long)DropLayout(nkt).words;
}

extern "C" int step_tsformer_encode(const float* series, int S, int L, const void* wpack, long wpack_bytes,
                                    int depth, int flags, uint16_t* hidden_bf16, float* hidden_f32, float* last_f32,
                                    float* sqnorm_part, float dropout_p, const uint64_t* drop_pool, long pool_words, uint64_t seed,
                                    unsigned int* fallback_count, void* stream) {
    STEP_REQUIRE(series && wpack, "tsformer_encode: null input");
    STEP_REQUIRE(S > 0 && L > 0 && L % TSF_PATCH == 0, "tsformer_encode: L=%d must be a positive multiple of %d", L, TSF_PATCH);
    const int P = L / TSF_PATCH;
    STEP_REQUIRE(P <= 512, "tsformer_encode: %d tokens exceed the 512-token workgroup limit", P);
    STEP_REQUIRE(depth >= 1 && depth <= 16, "tsformer_encode: bad depth %d", depth);
    STEP_REQUIRE(wpack_bytes >= TSF_TOTAL_BYTES(depth, P), "tsformer_encode: packed weights too small (%ld < %ld)",
                 wpack_bytes, (long)TSF_TOTAL_BYTES(depth, P));
    STEP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "tsformer_encode: bad dropout %f", dropout_p);
    STEP_REQUIRE((flags & ~(7 | (0xffff << 8))) == 0, "tsformer_encode: unknown flag bits 0x%x", flags);
    STEP_REQUIRE(!(flags & STEP_ENC_RANGE_FLAG) || (fallback_count && sqnorm_part), "tsformer_encode: STEP_ENC_RANGE_FLAG needs fallback_count (uint32 [65]) and sqnorm_part");
    const bool dr = dropout_p > 0.f;
    EncArgs a;
    a.series = series; a.S = S; a.L = L; a.P = P; a.depth = depth; a.nkt = (P + 31) / 32;
    if (dr) {
        STEP_REQUIRE(drop_pool, "tsformer_encode: dropout needs a keep-mask pool (step_dropout_pool_fill)");
        STEP_REQUIRE(pool_words >= 16 && (pool_words & (pool_words - 1)) == 0 && pool_words <= (1L << 31),
                     "tsformer_encode: pool of %ld words is not a power of two in [16, 2^31]", pool_words);
        STEP_REQUIRE(pool_words >= 2 * (long)DropLayout(a.nkt).words, "tsformer_encode: pool of %ld words is smaller than two chunks of %ld",
                     pool_words, (long)DropLayout(a.nkt).words);
    }
    a.wpack = (const char*)wpack; a.hid_bf16 = hidden_bf16; a.hid_f32 = hidden_f32; a.last_f32 = last_f32;
    a.sqn = sqnorm_part; a.keep = 1.0f - dropout_p; a.inv_keep = 1.0f / a.keep; a.inv_keep2 = a.inv_keep * a.inv_keep; a.seed = (uint32_t)(seed ^ (seed >> 32));
    a.pool = dr ? (const unsigned long long*)drop_pool : nullptr;
    a.pool_mask = dr ? (uint32_t)(pool_words - 1) : 0u;
    a.fallback = fallback_count;
    a.f16 = (flags & STEP_ENC_F16) != 0;
    a.always_rescale = (flags & STEP_ENC_ALWAYS_RESHIFT) != 0;
    a.range_word = (flags & STEP_ENC_RANGE_FLAG) ? fallback_count + 64 : nullptr;
    a.grid_limit = (flags >> 8) & 0xffff;               // STEP_ENC_WORKGROUPS(n): persistent launch of at most n workgroups
    a.nseq = 1;
    hipStream_t st = (hipStream_t)stream;
    if (a.nkt >= 5 && a.nkt <= 6 && S >= 2) {
        // 5 / 6 token tiles (P = 168: METR-LA, PEMS-BAY, PEMS07): two sequences per twelve-wave workgroup, three waves per SIMD (see the kernel's
        // prologue).  STEP_ENC_NSEQ=1 keeps one sequence per workgroup (A/B measurements).
        static const bool one = [] { const char* e = getenv("STEP_ENC_NSEQ"); return e && e[0] == '1'; }();
        if (!one) {
            a.nseq = 2;
            return dr ? launch_enc<12, true, false>(a, st) : launch_enc<12, false, false>(a, st);
        }
    }
    // parking the operand copy needs nkt * 10 KB + 50 KB of LDS (<= 160 KB up to 11 token tiles = 352 tokens)
    if (a.nkt <= 4) return dr ? launch_enc<4, true, true>(a, st) : launch_enc<4, false, true>(a, st);
    if (a.nkt <= 8) {
        // 5..8 token tiles (P = 168: six).  Unparked, a workgroup needs nkt * 5 KB + 50 KB of LDS -- 80 KB at P = 168, so TWO
        // workgroups (sequences) share a compute unit (12 waves x 168 registers fit the register file) and run out of phase; parked
        // it is 110 KB, one workgroup of six waves per compute unit.  STEP_ENC_PARK=1 / 0 forces either (A/B measurements).
        const char* e = getenv("STEP_ENC_PARK");
        const bool park = e ? e[0] == '1' : a.nkt * (4 + parked_frags<8, false>()) * TSF_FRAG + 2 * TSF_BLOCK > 80 * 1024;
        if (!park) return dr ? launch_enc<8, true, false>(a, st) : launch_enc<8, false, false>(a, st);
        return dr ? launch_enc<8, true, true>(a, st) : launch_enc<8, false, true>(a, st);
    }
    if (a.nkt > 8 && a.nkt <= 11) {
        // 9..11 token tiles (P = 336): the operand copy stays in registers -- same kernel time as parking it in LDS (2.54 vs 2.55 ms at
        // C2), but 66 KB of LDS per compute unit stay free for the second stream's kernels (step 5.39 -> 5.37 ms).  STEP_ENC_PARK=1
        // selects the parked variant (A/B measurements).
        const char* e = getenv("STEP_ENC_PARK");
        if (!(e && e[0] == '1')) return dr ? launch_enc<12, true, false>(a, st) : launch_enc<12, false, false>(a, st);
    }
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
