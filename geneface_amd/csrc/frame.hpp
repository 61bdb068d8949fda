// Workspace carve-out and packed-weight layouts of the fused frame kernels (gf_frame_t itself is declared in
// include/geneface_hip.h).
#pragma once
#include <stdint.h>

#include "geneface_hip.h"  // gf_frame_t

namespace gf {

// ---------------- workspace carve-out (all offsets in bytes, 256-aligned) ----------------
struct FrameWs {
    float *nears, *fars, *rays_t, *weights_sum, *depth, *image, *rays_o, *rays_d;
    float* far_occ;  // min(far, exit of the occupancy bounding box): where marching may stop (no sample can lie beyond it)
    int32_t *alive_a, *alive_b;  // survivor list / hit list
    uint32_t* ctrl;  // [kCtrlWords]
    // torso pass (frame_torso.hip, round 5): the masked pixels as a dense list, its inverse, and the field's outputs per list entry
    uint32_t *torso_list, *torso_dense_of;   // [N] pixel of list entry j;  [N] list entry of pixel n, or kTorsoNone
    float* torso_out;                        // [6][N] SoA by list entry: alpha, r, g, b, deform x, deform y
    size_t bytes;
};
constexpr uint32_t kTorsoNone = 0xFFFFFFFFu;
constexpr uint32_t kMaxSteps = 1024;  // largest max_steps the fused path accepts: the reference's own default (renderer.py:263) and the top of
                                      // its viewer's slider (radnerf_gui.py:466-471).  (64 until round 4: the size of the LDS histogram, which now
                                      // only holds the first kHistLds bins; later terminal indices go to the control block directly.)
constexpr uint32_t kHistLds = 66;     // terminal indices d < kHistLds are counted per workgroup in LDS and flushed once
// Control block (uint32 words), zeroed by a memset node at the head of every frame.  Three groups on separate 128-byte lines (round 3): L2
// retires atomics on one line at ~10 ns each, and the queue heads -- on the critical path of every pool refill -- used to share a line with
// the statistics and the histogram that every retiring workgroup adds to.
// line 0: work distribution
constexpr uint32_t kCtrlQHead0 = 0;    // phase 0: next unclaimed entry of the hit list
constexpr uint32_t kCtrlNHit = 1;      // rays with >= 1 sample (length of the hit list, alive_b)
constexpr uint32_t kCtrlNSurv = 2;     // rays still alive after max_steps samples (length of the survivor list, alive_a)
constexpr uint32_t kCtrlQHead1 = 3;    // phase 1: next unclaimed entry of the survivor list
constexpr uint32_t kCtrlBudget = 10;   // total per-ray sample budget B the reference's n_step schedule arrives at
constexpr uint32_t kCtrlTorsoCount = 16;   // torso pass: pixels the torso mask selects (length of torso_list); written after the head kernels are done
// line 1: statistics (two 64-bit adds per workgroup and phase: [samples | tiles << 32], [rounds | composited << 32])
constexpr uint32_t kCtrlStatA = 32;    // [2 phases] uint64: field evaluations (low word) | 32-sample MFMA tiles executed (high word)
constexpr uint32_t kCtrlStatB = 36;    // [2 phases] uint64: workgroup rounds (low) | samples the compositor consumed (high; <= evaluations: a ray
                                       //     that terminates inside a round leaves the rest of its slots of that round evaluated but unused)
// lines 2..4: terminal-index histogram
constexpr uint32_t kCtrlHist = 64;     // [kMaxSteps + 2] rays that terminate at cumulative sample index d (d = 1 .. max_steps)
constexpr uint32_t kCtrlWords = 1152;  // 64 + kMaxSteps + 2, rounded up to whole 128-byte lines; a frame zeroes only the words its max_steps needs
inline uint32_t ctrl_words_used(uint32_t max_steps) { return kCtrlHist + max_steps + 2; }

inline FrameWs carve_workspace(void* base, uint32_t n_rays) {
    FrameWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return (char*)base + o; };
    const size_t N = n_rays;
    w.nears = (float*)take(N * 4);
    w.fars = (float*)take(N * 4);
    w.rays_t = (float*)take(N * 4);
    w.weights_sum = (float*)take(N * 4);
    w.depth = (float*)take(N * 4);
    w.image = (float*)take(N * 12);
    w.rays_o = (float*)take(N * 12);
    w.rays_d = (float*)take(N * 12);
    w.far_occ = (float*)take(N * 4);
    w.alive_a = (int32_t*)take(N * 4);
    w.alive_b = (int32_t*)take(N * 4);
    w.ctrl = (uint32_t*)take(kCtrlWords * 4);
    w.torso_list = (uint32_t*)take(N * 4);
    w.torso_dense_of = (uint32_t*)take(N * 4);
    w.torso_out = (float*)take(N * 24);
    w.bytes = off;
    return w;
}

// ---------------- packed head weights (floats) ----------------
// Wave w of a head workgroup owns output features [32w, 32w+32) of every 128-wide layer.  Its weights are ONE contiguous
// stream of MFMA A operands in consumption order, read straight from L2 into registers:
//   stream[w][g][lane][i]  (g = 4-step group, i = step within the group)  =  W[row0 + 32w + (lane&31)][col0 + 8u + 4*(lane>>5) + i]
// with u = g - (first group of the layer): step 4u+i of a layer consumes input features (8u+i, 8u+4+i), lane half h
// supplying 8u+4h+i -- which are 4 consecutive floats of a sample's activation row in LDS (one ds_read_b128 per group).
constexpr uint32_t kHidden = 128;
constexpr uint32_t G_AMB1 = 0;              // ambient L1, 3-D grid columns 0..31 (cond columns fold into amb_bias)     4 groups
constexpr uint32_t G_SIG1A = G_AMB1 + 4;    // density L1, 3-D grid columns 0..31 (runs beside ambient L1)                4
constexpr uint32_t G_AMB2 = G_SIG1A + 4;    // ambient L2                                                                 16
constexpr uint32_t G_SIG1B = G_AMB2 + 16;   // density L1, 2-D grid columns 32..63                                        4
constexpr uint32_t G_SIG2 = G_SIG1B + 4;    // density L2                                                                 16
constexpr uint32_t G_SIG3 = G_SIG2 + 16;    // density L3 rows 1..128 (geometry feature)                                  16
constexpr uint32_t G_COL1S = G_SIG3 + 16;   // colour L1, SH columns 0..15                                                2
constexpr uint32_t G_COL1G = G_COL1S + 2;   // colour L1, geometry columns 16..143                                        16
constexpr uint32_t G_TOTAL = G_COL1G + 16;  // 78 groups = 78 KiB per wave and round
constexpr uint32_t HP_STREAM = 0;                                  // [4][G_TOTAL][64][4]
constexpr uint32_t HP_SMALL = HP_STREAM + 4 * G_TOTAL * 256;       // VALU layers + constant bias
constexpr uint32_t HS_AMB3 = 0;        // [2][128]  ambient L3 rows, natural feature order
constexpr uint32_t HS_SIGROW = 256;    // [128]     density row (row 0 of density L3)
constexpr uint32_t HS_COL2 = 384;      // [3][128]  colour L2 rows
constexpr uint32_t HS_COLBIAS = 768;   // [128]     W_color0[:, 144:148] @ individual_code, accumulator-layout order [ob][half][16]
constexpr uint32_t HS_TOTAL = 896;
constexpr uint32_t HP_TOTAL = HP_SMALL + HS_TOTAL;

// ---------------- packed head weights, fast path (f16 MFMA operands, fp32 accumulate; BASELINE.md section 4 "fast") ----------------
// Same ownership (wave w = output features [32w, 32w+32)), one v_mfma_f32_32x32x16_f16 per group and tile:
//   stream16[w][g][lane][i]  (i = 0..7)  =  half( W[row0 + 32w + (lane&31)][col0 + 16u + 8*(lane>>5) + i] )
// lane half h supplies input features 16u + 8h + 0..7: eight consecutive halves of a sample's f16 activation row (one ds_read_b128).
constexpr uint32_t H16_AMB1 = 0;               // 2 groups (K = 32): ambient L1 over the 3-D grid features
constexpr uint32_t H16_AMB2 = H16_AMB1 + 2;    // 8
constexpr uint32_t H16_SIG1A = H16_AMB2 + 8;   // 2: density L1, 3-D grid columns (the features stay in their own LDS buffer)
constexpr uint32_t H16_SIG1B = H16_SIG1A + 2;  // 2: density L1, 2-D grid columns
constexpr uint32_t H16_SIG2 = H16_SIG1B + 2;   // 8
constexpr uint32_t H16_SIG3 = H16_SIG2 + 8;    // 8
constexpr uint32_t H16_COL1S = H16_SIG3 + 8;   // 1 (SH, K = 16)
constexpr uint32_t H16_COL1G = H16_COL1S + 1;  // 8
constexpr uint32_t H16_TOTAL = H16_COL1G + 8;  // 39 groups = 39 KiB per wave and round
constexpr uint32_t HP16_HALVES = 4 * H16_TOTAL * 64 * 8;   // the VALU rows / constant bias stay fp32 (HP_SMALL of the fp32 pack)

// ---------------- packed head weights, split path (gf_frame_t.precision = 2): fp32 values as two-term f16 splits ----------------
// Every fp32 weight w travels as hi = half(w) and lo' = half((w - hi) * 2^11): w = hi + lo' * 2^-11 to 2^-24 relative (both terms are
// rounded to nearest, the products of two halves are exact in the fp32 accumulators).  The activations are split the same way in LDS,
// and one product term set is three v_mfma_f32_32x32x16_f16 per 16 input features and tile:
//   acc1 += hi_w * hi_x;   acc2 += lo'_w * hi_x + hi_w * lo'_x;   result = acc1 + acc2 * 2^-11      (lo'_w * lo'_x * 2^-22 is dropped: 2^-24 relative)
// Same ownership as the other two streams; per group and lane 16 halves = [8 x hi | 8 x lo'] of
//   W[row0 + 32w + (lane&31)][col0 + 16u + 8*(lane>>5) + i]   (two 16-byte loads per lane and group).
// Density L1 is ONE K = 64 layer here: the lane pair that gathered a sample's 3-D features keeps them in registers across the ambient net
// and writes them beside the 2-D features, so the activation buffer is the only large LDS array (same 528-byte rows as the fp32 path).
constexpr uint32_t SP_AMB1 = 0;              // 2 groups (K = 32): ambient L1 over the 3-D grid features
constexpr uint32_t SP_AMB2 = SP_AMB1 + 2;    // 8
constexpr uint32_t SP_SIG1 = SP_AMB2 + 8;    // 4 (K = 64): density L1, columns [3-D 0..31 | 2-D 32..63]
constexpr uint32_t SP_SIG2 = SP_SIG1 + 4;    // 8
constexpr uint32_t SP_SIG3 = SP_SIG2 + 8;    // 8
constexpr uint32_t SP_COL1S = SP_SIG3 + 8;   // 1 (SH, K = 16)
constexpr uint32_t SP_COL1G = SP_COL1S + 1;  // 8
constexpr uint32_t SP_TOTAL = SP_COL1G + 8;  // 39 groups = 78 KiB per wave and round (the fp32 stream's size)
constexpr uint32_t HPS_HALVES = 4 * SP_TOTAL * 64 * 16;
constexpr float kSplitScale = 2048.0f, kSplitInv = 1.0f / 2048.0f;

// ---------------- packed torso weights (floats) ----------------
// MFMA streams as above with NOB out-blocks.  Frequency-encoded pixel coordinate enc(x) has 42 entries, padded to 48:
// lane half h supplies enc index 24h + t (t < 24; indices >= 42 are zero on both sides).
constexpr uint32_t TP_D1 = 0;                        // deform L1, enc(x) columns: NOB=2, 24 steps
constexpr uint32_t TP_D2 = TP_D1 + 2 * 24 * 64;      // deform L2 (64->64): NOB=2, 32 steps
constexpr uint32_t TP_C1 = TP_D2 + 2 * 32 * 64;      // canonical L1, [2-D grid 32 | enc(x) 48]: NOB=1, 16 + 24 steps
constexpr uint32_t TP_C2 = TP_C1 + 1 * 40 * 64;      // canonical L2 (32->32): NOB=1, 16 steps
constexpr uint32_t TP_D3 = TP_C2 + 1 * 16 * 64;      // VALU rows [2][64]  (accumulator-layout order)
constexpr uint32_t TP_C3 = TP_D3 + 2 * 64;           // VALU rows [4][32]
constexpr uint32_t TP_TOTAL = TP_C3 + 4 * 32;
// per-frame torso bias vector: [64 deform L1 | 32 canonical L1], accumulator-layout order
constexpr uint32_t TB_TOTAL = 96;
// torso_head_aware extension (radnerf_torso.py:36-46): the 16 encoder outputs are 8 more steps of both first layers (lane half h supplies
// encoder output 8h + t), as two extra streams in the layout above, followed by head_color_weights_encoder itself in nn.Linear layout
constexpr uint32_t TH_D1E = 0;                       // deform L1, encoder columns 104..119: NOB=2, 8 steps
constexpr uint32_t TH_C1E = TH_D1E + 2 * 8 * 64;     // canonical L1, encoder columns 136..151: NOB=1, 8 steps
constexpr uint32_t TH_W0 = TH_C1E + 1 * 8 * 64;      // [16][4]
constexpr uint32_t TH_B0 = TH_W0 + 64;               // [16]
constexpr uint32_t TH_W1 = TH_B0 + 16;               // [32][16]
constexpr uint32_t TH_B1 = TH_W1 + 512;              // [32]
constexpr uint32_t TH_W2 = TH_B1 + 32;               // [16][32]
constexpr uint32_t TH_B2 = TH_W2 + 512;              // [16]
constexpr uint32_t TH_TOTAL = TH_B2 + 16;

}  // namespace gf
