// Byte layout of the packed TSFormer weight buffer consumed by tsformer_encoder.hip.
// The python packer (step_amd/tsformer_pack.py) mirrors these constants.
//
// MFMA "chain" k-slot map (v_mfma_f32_32x32x16_bf16, lane = 32*h + r, operand slot j in 0..8):
//     F(s, h, j) = 16*s + 8*(j >> 2) + 4*h + (j & 3)            (s = which K=16 step of a 32-wide block)
// is exactly the accumulator row held in register 8*s+j of lane-half h, so an accumulator
// tile converted to bf16 is directly the next MFMA's operand (no cross-lane movement).
// Weight fragments are stored so that slot (h, j) of k-step s holds input feature
// 32*block + F(s, h, j).  One fragment = 64 lanes * 8 bf16 = 1024 bytes, lane-major.
//
// Each encoder layer is 10 "stage blocks" of 25 KB (24 operand fragments + 1 KB of f32 vectors):
// the kernel streams them through an LDS ring (2 or 4 slots) with global_load_lds, one block per stage,
// so every workgroup reads each weight byte from L2 exactly once.
//   head block hd (stage hd):   frags  0..5  Wq(hd) k-steps (A operand, pre-scaled by log2(e)/sqrt(24))
//                                      6..11 Wk(hd) k-steps (A operand)
//                                     12..17 Wv(hd) k-steps (B operand)
//                                     18..23 Wo(hd) [tile 3][s 2] (A operand)
//                               tail f32 [0..31] bq [2][16] (pre-scaled)   [32..63] bv [32] (slot 24 = 1.0)
//                                        hd==0: [64..159] bo [2][48]       hd==3: [64..159] LN1 gamma, [160..255] LN1 beta
//   ffn block j (stage 4+j), chunks c = 2j, 2j+1 of 32 hidden units:
//                               frags 12*(c&1) + 0..5  W1(c) k-steps,  12*(c&1) + 6..11 W2(c) [tile 3][s 2]
//                               tail f32 [0..63] b1 of both chunks [2][2][16]
//                                        j==0: [64..159] b2 [2][48]        j==5: [64..159] LN2 gamma, [160..255] LN2 beta
#pragma once

#define TSF_D 96
#define TSF_HEADS 4
#define TSF_HDIM 24
#define TSF_FFN 384
#define TSF_PATCH 12
#define TSF_FRAG 1024

#define TSF_MAGIC 0x54534633 /* "TSF3": W_pe as f32 MFMA operands, b_pe folded into the positional table */

// header: int32 magic, P (tokens), depth, reserved
#define TSF_HDR_BYTES 64
// --- global section (f32) ---
#define TSF_G_WPE (TSF_HDR_BYTES)                       /* [3][6][64]: A operands of v_mfma_f32_32x32x2_f32, value (t, s, lane) = W_pe[32 t + lane % 32][2 s + lane / 32] */
#define TSF_G_BPE (TSF_G_WPE + 2 * 48 * 12 * 4)         /* [2][48] b_pe (informative: the kernel reads it pre-added to the positional table) */
#define TSF_G_NORM_G (TSF_G_BPE + 2 * 48 * 4)           /* encoder_norm weight [2][48] */
#define TSF_G_NORM_B (TSF_G_NORM_G + 2 * 48 * 4)
#define TSF_LAYER0 ((TSF_G_NORM_B + 2 * 48 * 4 + 1023) / 1024 * 1024)   /* 1 KB aligned */
// --- stage blocks ---
#define TSF_BLOCK (25 * TSF_FRAG)
#define TSF_STAGES 10
#define TSF_LAYER_BYTES (TSF_STAGES * TSF_BLOCK)
#define TSF_TAIL (24 * TSF_FRAG)
// --- positional table + b_pe follows the last layer: f32 [P][2][48] ---
#define TSF_POS_OFF(depth) (TSF_LAYER0 + (long)(depth) * TSF_LAYER_BYTES)
#define TSF_TOTAL_BYTES(depth, P) (TSF_POS_OFF(depth) + (long)(P) * 2 * 48 * 4)
