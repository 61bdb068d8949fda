// Host-side weight packing for the fused field kernels: turns the reference's nn.Linear weights
// (row-major [out][in], modules/radnerfs/cond_encoder.py:92-111; layer shapes radnerf.py:44,53,59 and
// radnerf_torso.py:48-49) into the streams frame_head.hip / frame_torso.hip consume.  Pure host code.
#include "common.hpp"
#include "frame.hpp"
#include <string.h>

namespace {

// MFMA 32x32 accumulator row held by (register r, lane half h): the "C-layout" feature order
inline uint32_t c_row(uint32_t r, uint32_t h) { return (r & 3u) + 8u * (r >> 2) + 4u * h; }

// hidden -> hidden: step t = ob_in*16 + r consumes C-layout feature ob_in*32 + c_row(r, half)
inline uint32_t hidden_col(uint32_t t, uint32_t h) { return (t / 16) * 32 + c_row(t % 16, h); }

}  // namespace

GF_EXPORT uint32_t gf_head_pack_floats(void) { return gf::HP_TOTAL; }
GF_EXPORT uint32_t gf_torso_pack_floats(void) { return gf::TP_TOTAL; }
// float offset of the 128 folded identity-code biases (W_color0[:, 144:148] @ code, gf_clayout_perm row order) inside the head pack
GF_EXPORT uint32_t gf_head_pack_colbias_offset(void) { return gf::HP_SMALL + gf::HS_COLBIAS; }

// perm[i] (i = ob*32 + half*16 + r) = feature index ob*32 + c_row(r, half): the order in which per-frame bias
// vectors (amb_bias) must be handed to the kernels.
GF_EXPORT int gf_clayout_perm(uint32_t* perm128_host) {
    for (uint32_t ob = 0; ob < 4; ob++)
        for (uint32_t h = 0; h < 2; h++)
            for (uint32_t r = 0; r < 16; r++) perm128_host[ob * 32 + h * 16 + r] = ob * 32 + c_row(r, h);
    return GF_OK;
}

// All pointers are HOST pointers.  Shapes (May config, asserted by the Python side):
//   amb0 [128,96] amb1 [128,128] amb2 [2,128] | sig0 [128,64] sig1 [128,128] sig2 [129,128] | col0 [128,148] col1 [3,128]
//   ind_code [4] or NULL.   out [gf_head_pack_floats()]   (layout: frame.hpp)
GF_EXPORT int gf_head_pack(const float* amb0, const float* amb1, const float* amb2, const float* sig0, const float* sig1,
                           const float* sig2, const float* col0, const float* col1, const float* ind_code, float* out) {
    using namespace gf;
    if (!amb0 || !amb1 || !amb2 || !sig0 || !sig1 || !sig2 || !col0 || !col1 || !out) return gf_set_error(GF_ERR_INVALID, "head_pack: null pointer");
    memset(out, 0, sizeof(float) * HP_TOTAL);
    // one layer = `groups` 4-step groups of every wave's stream, starting at group g0
    auto layer = [&](uint32_t g0, uint32_t groups, const float* W, uint32_t ld, uint32_t row0, uint32_t col0) {
        for (uint32_t w = 0; w < 4; w++)
            for (uint32_t u = 0; u < groups; u++)
                for (uint32_t l = 0; l < 64; l++)
                    for (uint32_t i = 0; i < 4; i++)
                        out[HP_STREAM + (((size_t)w * G_TOTAL + g0 + u) * 64 + l) * 4 + i] =
                            W[(size_t)(row0 + 32 * w + (l & 31u)) * ld + col0 + 8 * u + 4 * (l >> 5) + i];
    };
    layer(G_AMB1, 4, amb0, 96, 0, 0);      // the 64 cond columns (32..95) fold into amb_bias per frame
    layer(G_SIG1A, 4, sig0, 64, 0, 0);
    layer(G_AMB2, 16, amb1, 128, 0, 0);
    layer(G_SIG1B, 4, sig0, 64, 0, 32);
    layer(G_SIG2, 16, sig1, 128, 0, 0);
    layer(G_SIG3, 16, sig2, 128, 1, 0);    // rows 1..128: geometry feature (row 0 = log-density, VALU block)
    layer(G_COL1S, 2, col0, 148, 0, 0);    // columns [SH 0..15 | geo 16..143 | id 144..147]
    layer(G_COL1G, 16, col0, 148, 0, 16);
    float* s = out + HP_SMALL;
    for (uint32_t c = 0; c < 2; c++) memcpy(s + HS_AMB3 + c * 128, amb2 + (size_t)c * 128, 128 * sizeof(float));
    memcpy(s + HS_SIGROW, sig2, 128 * sizeof(float));
    for (uint32_t c = 0; c < 3; c++) memcpy(s + HS_COL2 + c * 128, col1 + (size_t)c * 128, 128 * sizeof(float));
    for (uint32_t ob = 0; ob < 4; ob++)
        for (uint32_t h = 0; h < 2; h++)
            for (uint32_t r = 0; r < 16; r++) {
                const uint32_t row = ob * 32 + c_row(r, h);
                float b = 0.0f;
                if (ind_code) for (uint32_t k = 0; k < 4; k++) b += col0[(size_t)row * 148 + 144 + k] * ind_code[k];
                s[HS_COLBIAS + ob * 32 + h * 16 + r] = b;
            }
    return GF_OK;
}

// Fast path: the same eight matrices as f16 MFMA A-operand streams (layout: frame.hpp, H16_*).  out_halves [gf_head_pack16_halves()]
// receives IEEE binary16 bit patterns (round to nearest even).  The VALU rows and the identity bias are taken from gf_head_pack().
GF_EXPORT uint32_t gf_head_pack16_halves(void) { return gf::HP16_HALVES; }

GF_EXPORT int gf_head_pack16(const float* amb0, const float* amb1, const float* sig0, const float* sig1, const float* sig2, const float* col0,
                             uint16_t* out_halves) {
    using namespace gf;
    if (!amb0 || !amb1 || !sig0 || !sig1 || !sig2 || !col0 || !out_halves) return gf_set_error(GF_ERR_INVALID, "head_pack16: null pointer");
    memset(out_halves, 0, sizeof(uint16_t) * HP16_HALVES);
    auto layer = [&](uint32_t g0, uint32_t groups, const float* W, uint32_t ld, uint32_t row0, uint32_t col0) {
        for (uint32_t w = 0; w < 4; w++)
            for (uint32_t u = 0; u < groups; u++)
                for (uint32_t l = 0; l < 64; l++)
                    for (uint32_t i = 0; i < 8; i++) {
                        const _Float16 h = (_Float16)W[(size_t)(row0 + 32 * w + (l & 31u)) * ld + col0 + 16 * u + 8 * (l >> 5) + i];
                        memcpy(out_halves + (((size_t)w * H16_TOTAL + g0 + u) * 64 + l) * 8 + i, &h, sizeof(uint16_t));
                    }
    };
    layer(H16_AMB1, 2, amb0, 96, 0, 0);
    layer(H16_AMB2, 8, amb1, 128, 0, 0);
    layer(H16_SIG1A, 2, sig0, 64, 0, 0);
    layer(H16_SIG1B, 2, sig0, 64, 0, 32);
    layer(H16_SIG2, 8, sig1, 128, 0, 0);
    layer(H16_SIG3, 8, sig2, 128, 1, 0);
    layer(H16_COL1S, 1, col0, 148, 0, 0);
    layer(H16_COL1G, 8, col0, 148, 0, 16);
    return GF_OK;
}

// Where every half of gf_head_pack16's output comes from: 1-based flat index into cat(amb0, amb1, sig0, sig1, sig2, col0) (row-major), 0 = a
// zero slot.  With it a training step re-gathers the f16 streams from the current fp32 master weights ON THE DEVICE (one cat + half() +
// gather) instead of a device -> host -> pack -> device round trip: the AMP tier's forward (gf_field_forward_train16) needs them every step.
GF_EXPORT int gf_head_pack16_index(uint32_t* out_index) {
    using namespace gf;
    if (!out_index) return gf_set_error(GF_ERR_INVALID, "head_pack16_index: null pointer");
    memset(out_index, 0, sizeof(uint32_t) * HP16_HALVES);
    const uint32_t base_amb0 = 1, base_amb1 = base_amb0 + 128 * 96, base_sig0 = base_amb1 + 128 * 128, base_sig1 = base_sig0 + 128 * 64,
                   base_sig2 = base_sig1 + 128 * 128, base_col0 = base_sig2 + 129 * 128;
    auto layer = [&](uint32_t g0, uint32_t groups, uint32_t base, uint32_t ld, uint32_t row0, uint32_t col0) {
        for (uint32_t w = 0; w < 4; w++)
            for (uint32_t u = 0; u < groups; u++)
                for (uint32_t l = 0; l < 64; l++)
                    for (uint32_t i = 0; i < 8; i++)
                        out_index[(((size_t)w * H16_TOTAL + g0 + u) * 64 + l) * 8 + i] = base + (row0 + 32 * w + (l & 31u)) * ld + col0 + 16 * u + 8 * (l >> 5) + i;
    };
    layer(H16_AMB1, 2, base_amb0, 96, 0, 0);
    layer(H16_AMB2, 8, base_amb1, 128, 0, 0);
    layer(H16_SIG1A, 2, base_sig0, 64, 0, 0);
    layer(H16_SIG1B, 2, base_sig0, 64, 0, 32);
    layer(H16_SIG2, 8, base_sig1, 128, 0, 0);
    layer(H16_SIG3, 8, base_sig2, 128, 1, 0);
    layer(H16_COL1S, 1, base_col0, 148, 0, 0);
    layer(H16_COL1G, 8, base_col0, 148, 0, 16);
    return GF_OK;
}

// Split path (gf_frame_t.precision = 2): the same matrices as two-term f16 splits (layout and arithmetic: frame.hpp, SP_*).
// out_halves [gf_head_pack_split_halves()].  A weight beyond the f16 range cannot be split: GF_ERR_UNSUPPORTED (the caller stays on fp32).
GF_EXPORT uint32_t gf_head_pack_split_halves(void) { return gf::HPS_HALVES; }

GF_EXPORT int gf_head_pack_split(const float* amb0, const float* amb1, const float* sig0, const float* sig1, const float* sig2, const float* col0,
                                 uint16_t* out_halves) {
    using namespace gf;
    if (!amb0 || !amb1 || !sig0 || !sig1 || !sig2 || !col0 || !out_halves) return gf_set_error(GF_ERR_INVALID, "head_pack_split: null pointer");
    memset(out_halves, 0, sizeof(uint16_t) * HPS_HALVES);
    bool ok = true;
    auto layer = [&](uint32_t g0, uint32_t groups, const float* W, uint32_t ld, uint32_t row0, uint32_t col0) {
        for (uint32_t w = 0; w < 4; w++)
            for (uint32_t u = 0; u < groups; u++)
                for (uint32_t l = 0; l < 64; l++)
                    for (uint32_t i = 0; i < 8; i++) {
                        const float v = W[(size_t)(row0 + 32 * w + (l & 31u)) * ld + col0 + 16 * u + 8 * (l >> 5) + i];
                        if (!(v >= -65504.0f && v <= 65504.0f)) ok = false;
                        const _Float16 hi = (_Float16)v;       // round to nearest; may be an f16 denormal (honoured by the matrix pipe: split_f16, frame_head.hip)
                        const _Float16 lo = (_Float16)((v - (float)hi) * kSplitScale);
                        uint16_t* dst = out_halves + (((size_t)w * SP_TOTAL + g0 + u) * 64 + l) * 16;
                        memcpy(dst + i, &hi, sizeof(uint16_t));
                        memcpy(dst + 8 + i, &lo, sizeof(uint16_t));
                    }
    };
    layer(SP_AMB1, 2, amb0, 96, 0, 0);
    layer(SP_AMB2, 8, amb1, 128, 0, 0);
    layer(SP_SIG1, 4, sig0, 64, 0, 0);
    layer(SP_SIG2, 8, sig1, 128, 0, 0);
    layer(SP_SIG3, 8, sig2, 128, 1, 0);
    layer(SP_COL1S, 1, col0, 148, 0, 0);
    layer(SP_COL1G, 8, col0, 148, 0, 16);
    if (!ok) return gf_set_error(GF_ERR_UNSUPPORTED, "head_pack_split: a weight is outside the f16 range (|w| > 65504 or not finite): use precision 0");
    return GF_OK;
}

// Axis-aligned world-space box around every occupied cell of the Morton-ordered occupancy bitfield (HOST pointer):
// cell (x,y,z) of cascade c covers ((v + {0,1}) / H * 2 - 1) * min(2^c, bound) per axis (raymarching.cu:883-892).  A sample
// position outside this box lies in an unoccupied cell at every cascade, so the marcher can never emit a sample beyond
// the ray's exit from it.  out6 = {xmin,ymin,zmin,xmax,ymax,zmax}; an empty grid gives an inverted box (min > max).
GF_EXPORT int gf_occupancy_aabb(const uint8_t* bitfield_host, uint32_t cascade, uint32_t H, float bound, float* out6_host) {
    if (!bitfield_host || !out6_host || cascade == 0 || H == 0 || H > 1024) return gf_set_error(GF_ERR_INVALID, "occupancy_aabb: bad argument");
    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
    const uint64_t H3 = (uint64_t)H * H * H;
    for (uint32_t c = 0; c < cascade; c++) {
        double mb = (double)(1u << c);
        if (mb > bound) mb = bound;
        for (uint64_t i = 0; i < H3; i++) {
            const uint64_t idx = c * H3 + i;
            if (!(bitfield_host[idx >> 3] & (1u << (idx & 7u)))) continue;
            const uint32_t m = (uint32_t)i;
            const uint32_t v[3] = {gf::morton3d_invert(m), gf::morton3d_invert(m >> 1), gf::morton3d_invert(m >> 2)};
            for (int d = 0; d < 3; d++) {
                const double a = ((double)v[d] / H * 2 - 1) * mb, b = ((double)(v[d] + 1) / H * 2 - 1) * mb;
                if (a < lo[d]) lo[d] = a;
                if (b > hi[d]) hi[d] = b;
            }
        }
    }
    for (int d = 0; d < 3; d++) { out6_host[d] = (float)lo[d]; out6_host[3 + d] = (float)hi[d]; }
    return GF_OK;
}

//   d0 [64,104] d1 [64,64] d2 [2,64] | c0 [32,136] c1 [32,32] c2 [4,32].  out [gf_torso_pack_floats()]
// d0 columns: [enc(x) 0..41 | enc(pose) 42..95 | code 96..103]; c0 columns: [grid 0..31 | enc(x) 32..73 | enc(pose)+code 74..135].
// The pose/code columns are per-frame constants: the host folds them into torso_bias.
static int torso_pack_impl(const float* d0, uint32_t ld_d0, const float* d1, const float* d2, const float* c0, uint32_t ld_c0, const float* c1,
                           const float* c2, float* out) {
    using namespace gf;
    if (!d0 || !d1 || !d2 || !c0 || !c1 || !c2 || !out) return gf_set_error(GF_ERR_INVALID, "torso_pack: null pointer");
    memset(out, 0, sizeof(float) * TP_TOTAL);
    auto stream = [&](float* dst, const float* W, uint32_t ld, uint32_t nob, uint32_t nsteps, auto col) {
        for (uint32_t ob = 0; ob < nob; ob++)
            for (uint32_t t = 0; t < nsteps; t++)
                for (uint32_t l = 0; l < 64; l++) {
                    const int c = col(t, l >> 5);
                    dst[((ob * (nsteps / 4) + t / 4) * 64 + l) * 4 + (t & 3u)] = c < 0 ? 0.0f : W[(size_t)(ob * 32 + (l & 31u)) * ld + c];
                }
    };
    auto enc_col = [](uint32_t t, uint32_t h) { const uint32_t e = 24 * h + t; return e < 42 ? (int)e : -1; };
    stream(out + TP_D1, d0, ld_d0, 2, 24, enc_col);
    stream(out + TP_D2, d1, 64, 2, 32, [](uint32_t t, uint32_t h) { return (int)hidden_col(t, h); });
    stream(out + TP_C1, c0, ld_c0, 1, 40, [&](uint32_t t, uint32_t h) {
        if (t < 16) return (int)(16 * h + t);            // 2-D grid features: levels 8h..8h+7
        const int e = enc_col(t - 16, h);
        return e < 0 ? -1 : 32 + e;
    });
    stream(out + TP_C2, c1, 32, 1, 16, [](uint32_t t, uint32_t h) { return (int)hidden_col(t, h); });
    for (uint32_t c = 0; c < 2; c++)
        for (uint32_t ob = 0; ob < 2; ob++)
            for (uint32_t h = 0; h < 2; h++)
                for (uint32_t r = 0; r < 16; r++) out[TP_D3 + c * 64 + ob * 32 + h * 16 + r] = d2[(size_t)c * 64 + ob * 32 + c_row(r, h)];
    for (uint32_t c = 0; c < 4; c++)
        for (uint32_t h = 0; h < 2; h++)
            for (uint32_t r = 0; r < 16; r++) out[TP_C3 + c * 32 + h * 16 + r] = c2[(size_t)c * 32 + c_row(r, h)];
    return GF_OK;
}

// One hidden -> hidden weight matrix W [nob * 32][ld] as an A-operand stream (the layout of the second layers above): the training backward's
// transposed blocks (frame_torso.hip, k_torso_train_bwd) are packed with it, one call each.
GF_EXPORT int gf_mlp_stream_pack(const float* W, uint32_t ld, uint32_t nob, uint32_t nsteps, float* out) {
    if (!W || !out || nob == 0 || nsteps == 0 || nsteps % 4) return gf_set_error(GF_ERR_INVALID, "mlp_stream_pack: bad argument");
    for (uint32_t ob = 0; ob < nob; ob++)
        for (uint32_t t = 0; t < nsteps; t++)
            for (uint32_t l = 0; l < 64; l++) {
                const uint32_t c = hidden_col(t, l >> 5);
                out[((ob * (nsteps / 4) + t / 4) * 64 + l) * 4 + (t & 3u)] = c < ld ? W[(size_t)(ob * 32 + (l & 31u)) * ld + c] : 0.0f;
            }
    return GF_OK;
}

// Row p = c_row(r, h) of the transposed grid block must hold grid feature 16 h + r: then accumulator register r of lane half h is the gradient
// of the lane's own level 8 h + r / 2, channel r & 1 -- the layout encode8_grad2 and the level-major gradient store want.
GF_EXPORT int gf_torso_bwd_grid_row_perm(uint32_t* perm32) {
    if (!perm32) return gf_set_error(GF_ERR_INVALID, "torso_bwd_grid_row_perm: null pointer");
    for (uint32_t h = 0; h < 2; h++)
        for (uint32_t r = 0; r < 16; r++) perm32[c_row(r, h)] = 16 * h + r;
    return GF_OK;
}

GF_EXPORT int gf_torso_pack(const float* d0, const float* d1, const float* d2, const float* c0, const float* c1, const float* c2, float* out) {
    return torso_pack_impl(d0, 104, d1, d2, c0, 136, c1, c2, out);
}

// torso_head_aware models (radnerf_torso.py:36-46): d0 [64,120], c0 [32,152]; out_ha layout: frame.hpp TH_*.
GF_EXPORT uint32_t gf_torso_ha_pack_floats(void) { return gf::TH_TOTAL; }

GF_EXPORT int gf_torso_pack_ha(const float* d0, const float* d1, const float* d2, const float* c0, const float* c1, const float* c2, const float* e0w,
                               const float* e0b, const float* e1w, const float* e1b, const float* e2w, const float* e2b, float* out_main, float* out_ha) {
    using namespace gf;
    if (!e0w || !e0b || !e1w || !e1b || !e2w || !e2b || !out_ha) return gf_set_error(GF_ERR_INVALID, "torso_pack_ha: null pointer");
    const int rc = torso_pack_impl(d0, 120, d1, d2, c0, 152, c1, c2, out_main);
    if (rc) return rc;
    memset(out_ha, 0, sizeof(float) * TH_TOTAL);
    auto ext = [&](float* dst, const float* W, uint32_t ld, uint32_t nob, uint32_t col0) {
        for (uint32_t ob = 0; ob < nob; ob++)
            for (uint32_t t = 0; t < 8; t++)
                for (uint32_t l = 0; l < 64; l++)
                    dst[((ob * 2 + t / 4) * 64 + l) * 4 + (t & 3u)] = W[(size_t)(ob * 32 + (l & 31u)) * ld + col0 + 8 * (l >> 5) + t];
    };
    ext(out_ha + TH_D1E, d0, 120, 2, 104);
    ext(out_ha + TH_C1E, c0, 152, 1, 136);
    memcpy(out_ha + TH_W0, e0w, 64 * sizeof(float)); memcpy(out_ha + TH_B0, e0b, 16 * sizeof(float));
    memcpy(out_ha + TH_W1, e1w, 512 * sizeof(float)); memcpy(out_ha + TH_B1, e1b, 32 * sizeof(float));
    memcpy(out_ha + TH_W2, e2w, 512 * sizeof(float)); memcpy(out_ha + TH_B2, e2b, 16 * sizeof(float));
    return GF_OK;
}
