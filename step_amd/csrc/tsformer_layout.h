// Byte layout of the packed TSFormer weight buffer consumed by tsformer_encoder.hip.
// The python packer (step_amd/tsformer_pack.py) mirrors these constants; a self-describing
// header at the start of the buffer is checked by the kernel launcher.
//
// MFMA "chain" k-slot map (v_mfma_f32_32x32x16_bf16, lane = 32*h + r, operand slot j in 0..8):
//     F(s, h, j) = 16*s + 8*(j >> 2) + 4*h + (j & 3)            (s = which K=16 step of a 32-wide block)
// is exactly the accumulator row held in register 8*s+j of lane-half h, so an accumulator
// tile converted to bf16 is directly the next MFMA's operand (no cross-lane movement).
// Weight fragments are stored so that slot (h, j) of k-step s holds input feature
// 32*block + F(s, h, j).  One fragment = 64 lanes * 8 bf16 = 1024 bytes, lane-major.
#pragma once

#define TSF_D 96
#define TSF_HEADS 4
#define TSF_HDIM 24
#define TSF_FFN 384
#define TSF_PATCH 12
#define TSF_FRAG 1024

#define TSF_MAGIC 0x54534631 /* "TSF1" */

// header: int32 magic, P (tokens), depth, reserved
#define TSF_HDR_BYTES 64
// --- global section (f32) ---
#define TSF_G_WPE (TSF_HDR_BYTES)                       /* [2][48][12] */
#define TSF_G_BPE (TSF_G_WPE + 2 * 48 * 12 * 4)         /* [2][48] */
#define TSF_G_NORM_G (TSF_G_BPE + 2 * 48 * 4)           /* encoder_norm weight [2][48] */
#define TSF_G_NORM_B (TSF_G_NORM_G + 2 * 48 * 4)
#define TSF_LAYER0 (TSF_G_NORM_B + 2 * 48 * 4)
// --- per layer section ---
#define TSF_L_WQ 0                                      /* [4 heads][6 ksteps] frags (A operand) */
#define TSF_L_WK (TSF_L_WQ + 24 * TSF_FRAG)
#define TSF_L_WV (TSF_L_WK + 24 * TSF_FRAG)             /* B operand frags */
#define TSF_L_WO (TSF_L_WV + 24 * TSF_FRAG)             /* [4 heads][3 tiles][2 s] */
#define TSF_L_W1 (TSF_L_WO + 24 * TSF_FRAG)             /* [12 chunks][6 ksteps] */
#define TSF_L_W2 (TSF_L_W1 + 72 * TSF_FRAG)             /* [12 chunks][3 tiles][2 s] */
#define TSF_L_BQ (TSF_L_W2 + 72 * TSF_FRAG)             /* f32 [4][2][16] (pre-scaled) */
#define TSF_L_BV (TSF_L_BQ + 4 * 2 * 16 * 4)            /* f32 [4][32]  (slot 24 = 1.0) */
#define TSF_L_BO (TSF_L_BV + 4 * 32 * 4)                /* f32 [2][48] */
#define TSF_L_LN1G (TSF_L_BO + 2 * 48 * 4)
#define TSF_L_LN1B (TSF_L_LN1G + 2 * 48 * 4)
#define TSF_L_B1 (TSF_L_LN1B + 2 * 48 * 4)              /* f32 [12][2][16] */
#define TSF_L_B2 (TSF_L_B1 + 12 * 2 * 16 * 4)           /* f32 [2][48] */
#define TSF_L_LN2G (TSF_L_B2 + 2 * 48 * 4)
#define TSF_L_LN2B (TSF_L_LN2G + 2 * 48 * 4)
#define TSF_LAYER_BYTES (TSF_L_LN2B + 2 * 48 * 4)
// --- positional table follows the last layer: f32 [P][2][48] ---
#define TSF_POS_OFF(depth) (TSF_LAYER0 + (long)(depth) * TSF_LAYER_BYTES)
#define TSF_TOTAL_BYTES(depth, P) (TSF_POS_OFF(depth) + (long)(P) * 2 * 48 * 4)
