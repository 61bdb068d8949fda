// Torso pass + final composite of one RAD-NeRF frame (gfx950): per pixel, sample the 2-D torso occupancy, evaluate
// the deformable 2-D torso field for the masked pixels on f32 MFMA, blend torso over background and the marched head
// over that, clamp, normalise depth, optionally emit uint8.
//
// Replaces RADNeRFTorso.render's tail (/root/reference/modules/radnerfs/radnerf_torso.py:156-198) and forward_torso
// (:51-84): grid_sample + boolean-mask gather/scatter (host sync on mask.any()), 2 freq encodes, 6 GEMMs, a grid
// encode, cats/sigmoids and five elementwise blends -- ~25 launches.
//
// Three launches since round 5 (one until then):
//   k_torso_mask   one lane per pixel: the mask (bilinear sample of the occupancy > threshold) and a DENSE LIST of the masked pixels
//                  (ballot / popcount inside a workgroup, one atomic per workgroup for its range of the list) + the list's inverse;
//   k_torso_field  the field over the list: a wave owns 32 consecutive list entries (one MFMA tile, the register-chained layer scheme of
//                  mfma_mlp.hpp), a workgroup 128, all torso weights (44 KB) LDS-resident; workgroups beyond the list's end leave at once;
//   k_torso_blend  one lane per pixel: torso over background, head over that, clamp, depth, uint8.
// Why: the masked pixels are the lower part of the picture.  With one workgroup per 256 consecutive pixels the field ran on the ~40 % of
// the workgroups that cover those rows, two tiles deep on each of their waves (8 tiles on 4 waves) while the others only blended; over the
// dense list every wave of every launched workgroup has exactly one tile.  Same values: a pixel's field is a column of the MFMA tiles,
// independent of which pixels share its tile, so the restructure itself left frames bit-identical to the one-kernel form (tools/frame_digests.py);
// the branch-free sine that replaced sinf in the encodings in the same round (sin_reduced, <= 9.3e-8 absolute) did change last-ulp bits -- the
// tolerance against the oracle is unchanged (2e-6 .. 3e-5 measured, bar 1e-4).
#include "common.hpp"
#include "frame.hpp"
#include "grid_core.hpp"
#include "mfma_mlp.hpp"
#include "sh_core.hpp"

namespace {

using gf::floatx16;
constexpr int kThreads = 256;

// Trace build only (-DGF_TRACE, tools/trace_torso.py): lane 0 of every wave of the first kTTraceWGs field workgroups stamps s_memtime at the
// segment boundaries of its first tile.  The product library carries none of it.
#ifdef GF_TRACE
constexpr int kTTraceWGs = 64, kTTraceSlots = 16;
static unsigned long long* g_ttrace_buf = nullptr;
#define GF_TSTAMP(i)                                                                                                     \
    do {                                                                                                                 \
        if (a.ttrace && blockIdx.x < kTTraceWGs && lane == 0 && first_tile)                                              \
            a.ttrace[((size_t)blockIdx.x * 4 + wave) * kTTraceSlots + (i)] = __builtin_amdgcn_s_memtime();              \
    } while (0)
#else
#define GF_TSTAMP(i) do { } while (0)
#endif

struct TorsoArgs {
    uint32_t N, G;
    const float *image, *weights_sum, *depth, *nears, *fars;  // head accumulators (workspace)
    const float *bg_coords, *bg, *occ;
    float thresh, shrink;
    const float *pack, *bias, *table; const int* offsets;
    gf::GridLevels lv;
    float *out_rgb, *out_depth, *out_alpha, *out_torso_rgb, *out_deform; uint8_t* out_rgb8;
    const float* ha;       // head-aware extension pack (frame.hpp TH_*), HA launches only
    const float* ha_enc;   // [N,16] encoder outputs (k_head_aware_encode), HA launches only
    const uint32_t *list, *dense_of, *count;   // the masked pixels as a dense list, its inverse, its length (k_torso_mask)
    uint32_t* count_reset; // == count, for the blend's final clear
#ifdef GF_TRACE
    unsigned long long* ttrace;
#endif
    float* tout;           // [6][N] field outputs by list entry
};

__device__ __forceinline__ float leaky02(float x) { return x > 0.0f ? x : 0.02f * x; }

// head_color_weights_encoder (radnerf_torso.py:38-44: Linear(4,16) LeakyReLU(0.02) Linear(16,32) LeakyReLU Linear(32,16)) of every pixel's
// accumulated head colour + opacity (radnerf_torso.py:68-74), one lane per pixel, into enc16 [N,16].  A launch of its own in front of
// k_torso_finish<true>: inside that kernel its ~50 temporaries pushed 170 registers into scratch.  Weights are read as wave-uniform LDS
// broadcasts.  1 088 FMAs per pixel; only head-aware models (base.yaml:90 default: off) ever launch it.
__global__ void __launch_bounds__(kThreads) k_head_aware_encode(uint32_t N, const float* __restrict__ image, const float* __restrict__ weights_sum,
                                                                const float* __restrict__ ha_pack, float* __restrict__ enc16) {
    __shared__ float w[gf::TH_TOTAL - gf::TH_W0];
    for (int i = threadIdx.x; i < (int)(gf::TH_TOTAL - gf::TH_W0); i += kThreads) w[i] = ha_pack[gf::TH_W0 + i];
    __syncthreads();
    const uint32_t n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= N) return;
    const float* W0 = w; const float* B0 = w + (gf::TH_B0 - gf::TH_W0); const float* W1 = w + (gf::TH_W1 - gf::TH_W0);
    const float* B1 = w + (gf::TH_B1 - gf::TH_W0); const float* W2 = w + (gf::TH_W2 - gf::TH_W0); const float* B2 = w + (gf::TH_B2 - gf::TH_W0);
    const float in4[4] = {image[(size_t)n * 3], image[(size_t)n * 3 + 1], image[(size_t)n * 3 + 2], weights_sum[n]};
    float h0[16], h1[32];
#pragma unroll
    for (int o = 0; o < 16; o++) {
        float v = B0[o];
#pragma unroll
        for (int k = 0; k < 4; k++) v = __builtin_fmaf(W0[o * 4 + k], in4[k], v);
        h0[o] = leaky02(v);
    }
#pragma unroll
    for (int o = 0; o < 32; o++) {
        float v = B1[o];
#pragma unroll
        for (int k = 0; k < 16; k++) v = __builtin_fmaf(W1[o * 16 + k], h0[k], v);
        h1[o] = leaky02(v);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = B2[4 * q + i];
#pragma unroll
            for (int k = 0; k < 32; k++) v = __builtin_fmaf(W2[(4 * q + i) * 32 + k], h1[k], v);
            e[i] = v;
        }
        reinterpret_cast<float4*>(enc16 + (size_t)n * 16)[q] = float4{e[0], e[1], e[2], e[3]};
    }
}

// F.grid_sample(input[1,1,G,G], grid[..., (x,y)], bilinear, zeros padding, align_corners=True) at one location:
// x indexes the last axis, y the one before it.
__device__ __forceinline__ float sample_occ(const float* __restrict__ occ, int G, float x, float y) {
    const float ix = ((x + 1.0f) / 2.0f) * (float)(G - 1), iy = ((y + 1.0f) / 2.0f) * (float)(G - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = (fx + 1.0f) - ix, wy0 = (fy + 1.0f) - iy;
    auto at = [&](int xx, int yy) { return (xx >= 0 && xx < G && yy >= 0 && yy < G) ? occ[yy * G + xx] : 0.0f; };
    float out = at(x0, y0) * (wx0 * wy0);
    out += at(x1, y0) * (wx1 * wy0);
    out += at(x0, y1) * (wx0 * wy1);
    out += at(x1, y1) * (wx1 * wy1);
    return out;
}

// entry e (0..47) of the zero-padded frequency encoding of a 2-vector (42 real entries, freqencoder.cu:30-58 layout)
// Branch-free: the sine is evaluated for every entry (on a harmless argument for the two pass-through and the six padding entries) and
// selected afterwards, so the 24 entries of a lane are 24 independent chains the scheduler interleaves.  Arguments are 2^f x + {0, pi/2} with
// f <= 9 and x = bg_coords * torso_shrink, |x| <= 1 for coordinates that are pixel coordinates: sin_reduced's domain (sh_core.hpp).
// BOUNDED = false: the same entry through sin_bounded (sinf beyond |argument| = 8192, like the op seam's frequency encoder): taken by a whole
// wave when ANY of its pixels carries a coordinate with |bg_coords * shrink| > 15.9 -- never the case for pixel coordinates, but gf_render_torso
// is a public entry and such inputs must not give finite nonsense where the op path gives the sine.  Where both apply they agree
// bit for bit (sin_bounded IS sin_reduced inside the range).
template <bool BOUNDED = true>
__device__ __forceinline__ float enc_entry(float x0, float x1, int e) {
    const int ee = e < 2 ? 2 : (e >= 42 ? 2 : e);
    const int col = ee / 2 - 1, d = ee & 1, freq = col >> 1;
    const float phase = (col & 1) ? (3.141592653589793f / 2) : 0.0f;
    const float arg = scalbnf(d ? x1 : x0, freq) + phase;
    const float sv = BOUNDED ? gf::sin_reduced(arg) : gf::sin_bounded(arg);
    return e >= 42 ? 0.0f : (e < 2 ? (e ? x1 : x0) : sv);
}

// ---- (1) mask + dense list.  list order: workgroups in the order their atomics retire, pixels in order inside a workgroup -- any order gives
// the same frame (see the header); dense_of is the inverse the blend reads.
__global__ void __launch_bounds__(kThreads) k_torso_mask(uint32_t N, uint32_t G, const float* __restrict__ bg_coords, const float* __restrict__ occ,
                                                         float thresh, uint32_t* __restrict__ list, uint32_t* __restrict__ dense_of,
                                                         uint32_t* __restrict__ count) {
    __shared__ uint32_t wcnt[kThreads / 64];
    __shared__ uint32_t wg_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = blockIdx.x * kThreads + tid;
    bool masked = false;
    if (n < N) masked = sample_occ(occ, (int)G, bg_coords[(size_t)n * 2], bg_coords[(size_t)n * 2 + 1]) > thresh;
    const unsigned long long bal = __ballot(masked);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < kThreads / 64; w++) { const uint32_t c = wcnt[w]; if (w < wave) before += c; total += c; }
    if (tid == 0) wg_base = total ? atomicAdd(count, total) : 0u;     // (L2 retires same-address atomics at ~10 ns each: ~5 us for a 512 x 512 frame's
                                                                      //  masked workgroups -- why frame loops build the list once, gf_torso_mask_list)
    __syncthreads();
    if (n < N) {
        const uint32_t j = wg_base + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (masked) list[j] = n;
        dense_of[n] = masked ? j : gf::kTorsoNone;
    }
}

// ---- (2) the field over the list
template <bool HA>
__global__ void __launch_bounds__(kThreads, 2) k_torso_field(const TorsoArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* ha = reinterpret_cast<float*>(smem_raw);     // [TH_W0]: the two extra weight streams (HA launches only)
    float* pack = ha + (HA ? gf::TH_W0 : 0);            // [TP_TOTAL]
    float* bias = pack + gf::TP_TOTAL;                  // [TB_TOTAL]
    float* meta = bias + gf::TB_TOTAL;                  // [64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    [[maybe_unused]] bool first_tile = true;
    GF_TSTAMP(0);
    const uint32_t Mt = *a.count;
    if ((uint32_t)blockIdx.x * 128u >= Mt) return;      // workgroup-uniform: nothing of the list is ours (no weight copy either)
    GF_TSTAMP(1);

    // The 44 KB of weights travel L2 -> LDS as asynchronous 1-KiB wave instructions that touch no register (global_load ... lds), and the
    // first tile's list entry, coordinates and 24 frequency encodings -- which need none of them -- are computed while they fly; the barrier
    // in front of the first MFMA waits for both.  (Through registers and with the barrier first, the copy was 8.7 K of a tile's 50.5 K
    // cycles and the encodings another 11.4 K: tools/trace_torso.py, profiles/round5/r5e_torso_field_timeline_before.txt.)
    static_assert(gf::TP_TOTAL % 256 == 0 && gf::TH_W0 % 256 == 0, "whole 1-KiB DMA instructions");
    gf::dma_to_lds(pack, a.pack, (int)gf::TP_TOTAL, wave, lane);
    if constexpr (HA) gf::dma_to_lds(ha, a.ha, (int)gf::TH_W0, wave, lane);
    if (tid < (int)gf::TB_TOTAL) bias[tid] = a.bias[tid];
    if (tid < 16) {
        meta[tid * 4 + 0] = a.lv.scale[tid];
        meta[tid * 4 + 1] = __uint_as_float(a.lv.resolution[tid]);
        meta[tid * 4 + 2] = __uint_as_float((uint32_t)a.offsets[tid]);
        meta[tid * 4 + 3] = __uint_as_float((uint32_t)(a.offsets[tid + 1] - a.offsets[tid]));
    }
    GF_TSTAMP(2);
    bool weights_ready = false;
    for (uint32_t base = (uint32_t)blockIdx.x * 128u; base < Mt; base += gridDim.x * 128u) {
        const uint32_t tile0 = base + wave * 32;
        const bool have_tile = tile0 < Mt;    // wave-uniform
        uint32_t j = 0, pix = 0;
        bool valid = false;
        float x0 = 0.0f, x1 = 0.0f;
        float e8[8];
        float enc[24];
        if (have_tile) {
            j = tile0 + (lane & 31);
            valid = j < Mt;
            pix = a.list[valid ? j : tile0];
            x0 = a.bg_coords[(size_t)pix * 2] * a.shrink; x1 = a.bg_coords[(size_t)pix * 2 + 1] * a.shrink;
            GF_TSTAMP(3);
            if constexpr (HA) {   // encoder output 8 * half + t of this pixel (k_head_aware_encode): the B operand of the 8 extra steps of both first layers
                const float4* e4 = reinterpret_cast<const float4*>(a.ha_enc + (size_t)pix * 16 + 8 * half);
                const float4 u = e4[0], v = e4[1];
                e8[0] = u.x; e8[1] = u.y; e8[2] = u.z; e8[3] = u.w; e8[4] = v.x; e8[5] = v.y; e8[6] = v.z; e8[7] = v.w;
            }
            // 2^9 |x| + pi/2 <= 8192 <=> |x| <= 15.99...: wave-uniform choice (a ballot), so the common case keeps its 24 branch-free chains
            if (__builtin_expect(__any(!(fabsf(x0) <= 15.9f && fabsf(x1) <= 15.9f)), 0)) {
#pragma unroll 1
                for (int t = 0; t < 24; t++) enc[t] = enc_entry<false>(x0, x1, 24 * half + t);
            } else {
#pragma unroll
                for (int t = 0; t < 24; t++) enc[t] = enc_entry<true>(x0, x1, 24 * half + t);
            }
        }
        if (!weights_ready) {     // workgroup-uniform (first trip of every wave, whether it has a tile or not): the DMA has landed for everybody
            __builtin_amdgcn_s_waitcnt(0);     // this wave's asynchronous copies (vmcnt) and LDS stores
            __syncthreads();
            weights_ready = true;
        }
        if (!have_tile) continue;
        GF_TSTAMP(4);

        floatx16 h2[2];
        float act2[32];
        if constexpr (HA) {
            gf::mfma_layer<2, 24, false, false>(pack + gf::TP_D1, lane, enc, bias, h2);
            gf::mfma_part<2, 0, 2, 8, true, true>(ha + gf::TH_D1E, lane, e8, nullptr, h2);
        } else {
            gf::mfma_layer<2, 24, true, false>(pack + gf::TP_D1, lane, enc, bias, h2);
        }
        gf::unpack<2>(h2, act2);
        GF_TSTAMP(5);
        gf::mfma_layer<2, 32, true, false>(pack + gf::TP_D2, lane, act2, nullptr, h2);
        gf::unpack<2>(h2, act2);
        GF_TSTAMP(6);
        float dx[2];
        gf::valu_rows<2, 2>(pack + gf::TP_D3, half, act2, dx);
        GF_TSTAMP(7);

        const float xc[2] = {(fminf(fmaxf(x0 + dx[0], -1.0f), 1.0f) + 1.0f) / 2.0f, (fminf(fmaxf(x1 + dx[1], -1.0f), 1.0f) + 1.0f) / 2.0f};
        float in[40];
        {
            float g[16];
            gf::encode_half<2>(a.table, meta, half, 1u /*tiled*/, 0u /*linear*/, xc, g);
#pragma unroll
            for (int t = 0; t < 16; t++) in[t] = g[t];
#pragma unroll
            for (int t = 0; t < 24; t++) in[16 + t] = enc[t];
        }
        GF_TSTAMP(8);
        floatx16 h1[1];
        float act1[16];
        if constexpr (HA) {
            gf::mfma_layer<1, 40, false, false>(pack + gf::TP_C1, lane, in, bias + 64, h1);
            gf::mfma_part<1, 0, 1, 8, true, true>(ha + gf::TH_C1E, lane, e8, nullptr, h1);
        } else {
            gf::mfma_layer<1, 40, true, false>(pack + gf::TP_C1, lane, in, bias + 64, h1);
        }
        gf::unpack<1>(h1, act1);
        GF_TSTAMP(9);
        gf::mfma_layer<1, 16, true, false>(pack + gf::TP_C2, lane, act1, nullptr, h1);
        gf::unpack<1>(h1, act1);
        GF_TSTAMP(10);
        float o4[4];
        gf::valu_rows<4, 1>(pack + gf::TP_C3, half, act1, o4);
        GF_TSTAMP(11);
        if (valid && half == 0) {      // SoA by list entry: 32 consecutive floats per array and tile
            const size_t N = a.N;
            a.tout[j] = 1.0f / (1.0f + __expf(-o4[0]));
            a.tout[N + j] = 1.0f / (1.0f + __expf(-o4[1]));
            a.tout[2 * N + j] = 1.0f / (1.0f + __expf(-o4[2]));
            a.tout[3 * N + j] = 1.0f / (1.0f + __expf(-o4[3]));
            a.tout[4 * N + j] = dx[0];
            a.tout[5 * N + j] = dx[1];
        }
        GF_TSTAMP(12);
        first_tile = false;
    }
}

// ---- (3) radnerf_torso.py:186-193: torso over background, head over that, clamp, depth normalisation
__global__ void __launch_bounds__(kThreads) k_torso_blend(const TorsoArgs a) {
    const uint32_t n = blockIdx.x * kThreads + threadIdx.x;
    if (n >= a.N) return;
    const uint32_t j = a.dense_of[n];
    const bool masked = j != gf::kTorsoNone;
    const size_t N = a.N;
    const float alpha = masked ? a.tout[j] : 0.0f;
    const float tc[3] = {masked ? a.tout[N + j] : 0.0f, masked ? a.tout[2 * N + j] : 0.0f, masked ? a.tout[3 * N + j] : 0.0f};
    const float ws = a.weights_sum[n];
    if (a.out_alpha) a.out_alpha[n] = alpha;
    if (a.out_deform && masked) { a.out_deform[(size_t)n * 2] = a.tout[4 * N + j]; a.out_deform[(size_t)n * 2 + 1] = a.tout[5 * N + j]; }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float bgc = tc[c] * alpha + a.bg[(size_t)n * 3 + c] * (1 - alpha);
        if (a.out_torso_rgb) a.out_torso_rgb[(size_t)n * 3 + c] = bgc;
        float v = a.image[(size_t)n * 3 + c] + (1 - ws) * bgc;
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        a.out_rgb[(size_t)n * 3 + c] = v;
        if (a.out_rgb8) a.out_rgb8[(size_t)n * 3 + c] = (uint8_t)(v * 255.0f);
    }
    a.out_depth[n] = fmaxf(a.depth[n] - a.nears[n], 0.0f) / (a.fars[n] - a.nears[n]);
    if (n == 0 && a.count_reset) *a.count_reset = 0u;     // the list is consumed (k_torso_field finished before this launch started): a second torso pass on the same head starts a new one
}

// ---------------------------------------------------------------------------------------------------- training field (round 6)
// forward_torso (radnerf_torso.py:51-84) for the torso task's step (tasks/radnerfs/radnerf_torso.py:74-122): the same chain as k_torso_field
// over a plain list of M pixel coordinates, every layer's activations saved row-major for the backward pass, and the input-gradient chain
// of that field in a second launch.  Default architecture only (torso_head_aware = false); include/geneface_hip.h, gf_torso_train_t.
struct TorsoTrainArgs {
    uint32_t M; float shrink;
    const float* x;
    const float *pack, *bias, *table; const int* offsets;
    gf::GridLevels lv;
    float *out, *dx;
    float *enc, *h_d1, *h_d2, *x01, *g, *h_c1, *h_c2;
    const float *bwd, *g_out, *g_dx;
    float *dz_c3, *dz_c2, *dz_c1, *dz_d3, *dz_d2, *dz_d1, *g_grid; uint32_t* level_max;
};

// this lane's NOB * 16 accumulator-order values <-> row j of a row-major [M, NOB * 32] matrix: registers 4q..4q+3 of block ob are the four
// consecutive features ob * 32 + 8q + 4 half + 0..3 (mfma_mlp.hpp)
template <int NOB>
__device__ __forceinline__ void rows_store(float* __restrict__ G, size_t pt, int half, const float (&a)[NOB * 16]) {
    float* row = G + pt * (NOB * 32) + 4 * half;
#pragma unroll
    for (int ob = 0; ob < NOB; ob++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            *reinterpret_cast<float4*>(row + ob * 32 + 8 * q) = float4{a[ob * 16 + 4 * q], a[ob * 16 + 4 * q + 1], a[ob * 16 + 4 * q + 2], a[ob * 16 + 4 * q + 3]};
}
template <int NOB>
__device__ __forceinline__ void rows_load(const float* __restrict__ G, size_t pt, int half, float (&a)[NOB * 16]) {
    const float* row = G + pt * (NOB * 32) + 4 * half;
#pragma unroll
    for (int ob = 0; ob < NOB; ob++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(row + ob * 32 + 8 * q);
            a[ob * 16 + 4 * q] = v.x; a[ob * 16 + 4 * q + 1] = v.y; a[ob * 16 + 4 * q + 2] = v.z; a[ob * 16 + 4 * q + 3] = v.w;
        }
}

__global__ void __launch_bounds__(kThreads, 2) k_torso_train_fwd(const TorsoTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* pack = reinterpret_cast<float*>(smem_raw);   // [TP_TOTAL]
    float* bias = pack + gf::TP_TOTAL;                  // [TB_TOTAL]
    float* meta = bias + gf::TB_TOTAL;                  // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    gf::dma_to_lds(pack, a.pack, (int)gf::TP_TOTAL, wave, lane);
    if (tid < (int)gf::TB_TOTAL) bias[tid] = a.bias[tid];
    if (tid < 16) {
        meta[tid * 4 + 0] = a.lv.scale[tid];
        meta[tid * 4 + 1] = __uint_as_float(a.lv.resolution[tid]);
        meta[tid * 4 + 2] = __uint_as_float((uint32_t)a.offsets[tid]);
        meta[tid * 4 + 3] = __uint_as_float((uint32_t)(a.offsets[tid + 1] - a.offsets[tid]));
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (uint32_t base = (uint32_t)blockIdx.x * 128u; base < a.M; base += gridDim.x * 128u) {
        const uint32_t tile0 = base + (uint32_t)wave * 32u;
        if (tile0 >= a.M) continue;                       // wave-uniform; no barrier inside the loop
        const uint32_t j = tile0 + (uint32_t)(lane & 31);
        const bool valid = j < a.M;
        const size_t pt = valid ? j : tile0;
        const float x0 = a.x[pt * 2] * a.shrink, x1 = a.x[pt * 2 + 1] * a.shrink;
        float enc[24];
        if (__builtin_expect(__any(!(fabsf(x0) <= 15.9f && fabsf(x1) <= 15.9f)), 0)) {
#pragma unroll 1
            for (int t = 0; t < 24; t++) enc[t] = enc_entry<false>(x0, x1, 24 * half + t);
        } else {
#pragma unroll
            for (int t = 0; t < 24; t++) enc[t] = enc_entry<true>(x0, x1, 24 * half + t);
        }
        if (valid) {
            float* row = a.enc + pt * 48 + 24 * half;
#pragma unroll
            for (int q = 0; q < 6; q++) *reinterpret_cast<float4*>(row + 4 * q) = float4{enc[4 * q], enc[4 * q + 1], enc[4 * q + 2], enc[4 * q + 3]};
        }
        floatx16 h2[2];
        float act2[32];
        gf::mfma_layer<2, 24, true, false>(pack + gf::TP_D1, lane, enc, bias, h2);
        gf::unpack<2>(h2, act2);
        if (valid) rows_store<2>(a.h_d1, pt, half, act2);
        gf::mfma_layer<2, 32, true, false>(pack + gf::TP_D2, lane, act2, nullptr, h2);
        gf::unpack<2>(h2, act2);
        if (valid) rows_store<2>(a.h_d2, pt, half, act2);
        float dx[2];
        gf::valu_rows<2, 2>(pack + gf::TP_D3, half, act2, dx);
        const float xc[2] = {(fminf(fmaxf(x0 + dx[0], -1.0f), 1.0f) + 1.0f) / 2.0f, (fminf(fmaxf(x1 + dx[1], -1.0f), 1.0f) + 1.0f) / 2.0f};
        float in[40];
        {
            float g[16];
            gf::encode_half<2>(a.table, meta, half, 1u /*tiled*/, 0u /*linear*/, xc, g);
            if (valid) {
                float* row = a.g + pt * 32 + 16 * half;
#pragma unroll
                for (int q = 0; q < 4; q++) *reinterpret_cast<float4*>(row + 4 * q) = float4{g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]};
            }
#pragma unroll
            for (int t = 0; t < 16; t++) in[t] = g[t];
#pragma unroll
            for (int t = 0; t < 24; t++) in[16 + t] = enc[t];
        }
        floatx16 h1[1];
        float act1[16];
        gf::mfma_layer<1, 40, true, false>(pack + gf::TP_C1, lane, in, bias + 64, h1);
        gf::unpack<1>(h1, act1);
        if (valid) rows_store<1>(a.h_c1, pt, half, act1);
        gf::mfma_layer<1, 16, true, false>(pack + gf::TP_C2, lane, act1, nullptr, h1);
        gf::unpack<1>(h1, act1);
        if (valid) rows_store<1>(a.h_c2, pt, half, act1);
        float o4[4];
        gf::valu_rows<4, 1>(pack + gf::TP_C3, half, act1, o4);
        if (valid && half == 0) {
            *reinterpret_cast<float4*>(a.out + pt * 4) = float4{1.0f / (1.0f + __expf(-o4[0])), 1.0f / (1.0f + __expf(-o4[1])),
                                                                1.0f / (1.0f + __expf(-o4[2])), 1.0f / (1.0f + __expf(-o4[3]))};
            *reinterpret_cast<float2*>(a.dx + pt * 2) = float2{dx[0], dx[1]};
            *reinterpret_cast<float2*>(a.x01 + pt * 2) = float2{xc[0], xc[1]};
        }
    }
}

// Backward streams (floats): W_c2^T [1 block x 16 steps] | W_c1[:, grid]^T [1 x 16] | W_d2^T [2 x 32], each [ob][step/4][lane][step%4]
constexpr uint32_t TBW_C2T = 0, TBW_C1GT = TBW_C2T + 16 * 64, TBW_D2T = TBW_C1GT + 16 * 64, TBW_TOTAL = TBW_D2T + 2 * 32 * 64;
constexpr size_t kTorsoBwdSmem = (TBW_TOTAL + 2 * 64 + 4 * 32) * sizeof(float) + 16 * sizeof(gf::LevelMeta) + 16 * sizeof(uint32_t);

__global__ void __launch_bounds__(kThreads, 2) k_torso_train_bwd(const TorsoTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* bw = reinterpret_cast<float*>(smem_raw);     // [TBW_TOTAL] transposed streams
    float* d3 = bw + TBW_TOTAL;                         // [2][64] deform L3 rows, accumulator-layout order (the forward pack's)
    float* c3 = d3 + 2 * 64;                            // [4][32] canonical L3 rows
    gf::LevelMeta* meta = reinterpret_cast<gf::LevelMeta*>(c3 + 4 * 32);
    uint32_t* lmax = reinterpret_cast<uint32_t*>(meta + 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    static_assert(TBW_TOTAL % 256 == 0, "whole 1-KiB DMA instructions");
    gf::dma_to_lds(bw, a.bwd, (int)TBW_TOTAL, wave, lane);
    if (tid < 128) d3[tid] = a.pack[gf::TP_D3 + tid];
    if (tid < 128) c3[tid] = a.pack[gf::TP_C3 + tid];
    if (tid < 16) {
        meta[tid] = gf::make_level_meta<2>(a.lv.scale[tid], a.lv.resolution[tid], a.offsets, (uint32_t)tid, 1u);
        lmax[tid] = 0u;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (uint32_t base = (uint32_t)blockIdx.x * 128u; base < a.M; base += gridDim.x * 128u) {
        const uint32_t tile0 = base + (uint32_t)wave * 32u;
        if (tile0 >= a.M) continue;
        const uint32_t j = tile0 + (uint32_t)(lane & 31);
        const bool valid = j < a.M;
        const size_t pt = valid ? j : tile0;
        // ---- outputs: d z = g * s (1 - s)
        const float4 o = *reinterpret_cast<const float4*>(a.out + pt * 4), go = *reinterpret_cast<const float4*>(a.g_out + pt * 4);
        float dz3[4] = {go.x * o.x * (1.0f - o.x), go.y * o.y * (1.0f - o.y), go.z * o.z * (1.0f - o.z), go.w * o.w * (1.0f - o.w)};
        if (!valid) { dz3[0] = dz3[1] = dz3[2] = dz3[3] = 0.0f; }
        if (valid && half == 0) *reinterpret_cast<float4*>(a.dz_c3 + pt * 4) = float4{dz3[0], dz3[1], dz3[2], dz3[3]};
        // ---- canonical net, back to front
        float act[16], dz[16];
        rows_load<1>(a.h_c2, pt, half, act);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 s4 = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float4 w = *reinterpret_cast<const float4*>(c3 + c * 32 + half * 16 + 4 * q);
                s4.x = __builtin_fmaf(dz3[c], w.x, s4.x); s4.y = __builtin_fmaf(dz3[c], w.y, s4.y);
                s4.z = __builtin_fmaf(dz3[c], w.z, s4.z); s4.w = __builtin_fmaf(dz3[c], w.w, s4.w);
            }
            dz[4 * q] = act[4 * q] > 0.0f ? s4.x : 0.0f; dz[4 * q + 1] = act[4 * q + 1] > 0.0f ? s4.y : 0.0f;
            dz[4 * q + 2] = act[4 * q + 2] > 0.0f ? s4.z : 0.0f; dz[4 * q + 3] = act[4 * q + 3] > 0.0f ? s4.w : 0.0f;
        }
        if (valid) rows_store<1>(a.dz_c2, pt, half, dz);
        floatx16 h1[1];
        gf::mfma_layer<1, 16, false, false>(bw + TBW_C2T, lane, dz, nullptr, h1);
        rows_load<1>(a.h_c1, pt, half, act);
#pragma unroll
        for (int r = 0; r < 16; r++) dz[r] = act[r] > 0.0f ? h1[0][r] : 0.0f;
        if (valid) rows_store<1>(a.dz_c1, pt, half, dz);
        gf::mfma_layer<1, 16, false, false>(bw + TBW_C1GT, lane, dz, nullptr, h1);
        // h1[0][r] = d loss / d grid feature 16 half + r  (level 8 half + r / 2, channel r & 1: the rows of the stream are permuted for this)
        float gg[16];
#pragma unroll
        for (int r = 0; r < 16; r++) gg[r] = valid ? h1[0][r] : 0.0f;
        if (valid) {
#pragma unroll
            for (int l = 0; l < 8; l++) *reinterpret_cast<float2*>(a.g_grid + ((size_t)(8 * half + l) * a.M + pt) * 2) = float2{gg[2 * l], gg[2 * l + 1]};
            if (a.level_max) {
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const uint32_t m0 = __float_as_uint(gg[2 * l]) & 0x7fffffffu, m1 = __float_as_uint(gg[2 * l + 1]) & 0x7fffffffu;
                    atomicMax(&lmax[8 * half + l], m0 > m1 ? m0 : m1);
                }
            }
        }
        // ---- through the lookup (input gradient, re-gathered) and the clamp to d dx
        const float2 xv = *reinterpret_cast<const float2*>(a.x + pt * 2), dxs = *reinterpret_cast<const float2*>(a.dx + pt * 2);
        const float2 x01 = *reinterpret_cast<const float2*>(a.x01 + pt * 2);
        const float pre[2] = {xv.x * a.shrink + dxs.x, xv.y * a.shrink + dxs.y};
        const float xq[2] = {x01.x, x01.y};
        float dxc[2];
        gf::encode8_grad2(a.table, meta + 8 * half, 1u, 0u, xq, gg, dxc);
        dxc[0] += __shfl_xor(dxc[0], 32);
        dxc[1] += __shfl_xor(dxc[1], 32);
        float ddx[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            ddx[i] = (pre[i] >= -1.0f && pre[i] <= 1.0f) ? 0.5f * dxc[i] : 0.0f;       // x01 = (clamp(x + dx, -1, 1) + 1) / 2
            if (a.g_dx) ddx[i] += a.g_dx[pt * 2 + i];
            if (!valid) ddx[i] = 0.0f;
        }
        if (valid && half == 0) *reinterpret_cast<float2*>(a.dz_d3 + pt * 2) = float2{ddx[0], ddx[1]};
        // ---- deform net, back to front
        float act2[32], dz2[32];
        rows_load<2>(a.h_d2, pt, half, act2);
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 w0 = *reinterpret_cast<const float4*>(d3 + ob * 32 + half * 16 + 4 * q), w1 = *reinterpret_cast<const float4*>(d3 + 64 + ob * 32 + half * 16 + 4 * q);
                const float v[4] = {__builtin_fmaf(ddx[1], w1.x, ddx[0] * w0.x), __builtin_fmaf(ddx[1], w1.y, ddx[0] * w0.y),
                                    __builtin_fmaf(ddx[1], w1.z, ddx[0] * w0.z), __builtin_fmaf(ddx[1], w1.w, ddx[0] * w0.w)};
#pragma unroll
                for (int i = 0; i < 4; i++) dz2[ob * 16 + 4 * q + i] = act2[ob * 16 + 4 * q + i] > 0.0f ? v[i] : 0.0f;
            }
        if (valid) rows_store<2>(a.dz_d2, pt, half, dz2);
        floatx16 h2[2];
        gf::mfma_layer<2, 32, false, false>(bw + TBW_D2T, lane, dz2, nullptr, h2);
        rows_load<2>(a.h_d1, pt, half, act2);
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int r = 0; r < 16; r++) dz2[ob * 16 + r] = act2[ob * 16 + r] > 0.0f ? h2[ob][r] : 0.0f;
        if (valid) rows_store<2>(a.dz_d1, pt, half, dz2);
    }
    __syncthreads();
    if (a.level_max && tid < 16 && lmax[tid]) atomicMax(&a.level_max[tid], lmax[tid]);
}

constexpr size_t kTorsoSmem = (gf::TP_TOTAL + gf::TB_TOTAL + 64) * sizeof(float);

}  // namespace

// Torso pass + final blend.  Requires gf_render_head(f, stream) to have been enqueued before on the same stream
// (it reads the head accumulators from the workspace).
GF_EXPORT int gf_render_torso(const gf_frame_t* f, void* stream) {
    if (!f || !f->workspace) return gf_set_error(GF_ERR_INVALID, "torso: null descriptor / workspace");
    if (!f->torso_pack || !f->torso_bias || !f->torso_table || !f->torso_offsets || !f->torso_occ || !f->bg_coords || !f->bg_color ||
        !f->out_rgb || !f->out_depth)
        return gf_set_error(GF_ERR_INVALID, "torso: null pointer in the torso description");
    const gf::FrameWs w = gf::carve_workspace(f->workspace, f->n_rays);
    TorsoArgs a;
    a.N = f->n_rays; a.G = f->grid_size;
    a.image = w.image; a.weights_sum = w.weights_sum; a.depth = w.depth; a.nears = w.nears; a.fars = w.fars;
    a.bg_coords = f->bg_coords; a.bg = f->bg_color; a.occ = f->torso_occ;
    a.thresh = f->torso_thresh; a.shrink = f->torso_shrink;
    a.pack = f->torso_pack; a.bias = f->torso_bias; a.table = f->torso_table; a.offsets = f->torso_offsets;
    if (gf::fill_grid_levels(a.lv, 16, f->torso_S, f->base_res)) return gf_set_error(GF_ERR_INVALID, "torso: bad grid levels");
    a.out_rgb = f->out_rgb; a.out_depth = f->out_depth; a.out_alpha = f->out_torso_alpha; a.out_torso_rgb = f->out_torso_rgb;
    a.out_deform = f->out_deform; a.out_rgb8 = f->out_rgb8;
    const bool ha = f->torso_ha_pack && f->torso_ha_branch;
    a.ha = ha ? f->torso_ha_pack : nullptr;
    a.ha_enc = nullptr;
    if (ha) {
        if (!f->torso_ha_ws) return gf_set_error(GF_ERR_INVALID, "torso: torso_ha_branch = 1 needs torso_ha_ws ([n_rays, 16] floats)");
        hipLaunchKernelGGL(k_head_aware_encode, dim3(gf_div_up(f->n_rays, (uint32_t)kThreads)), dim3(kThreads), 0, gf_stream(stream), f->n_rays, w.image,
                           w.weights_sum, f->torso_ha_pack, f->torso_ha_ws);
        a.ha_enc = f->torso_ha_ws;
    }
    const bool given = f->torso_mask_list != nullptr;
    if (given != (f->torso_mask_dense_of != nullptr) || given != (f->torso_mask_count != nullptr))
        return gf_set_error(GF_ERR_INVALID, "torso: torso_mask_list / torso_mask_dense_of / torso_mask_count must be given together");
    a.list = given ? f->torso_mask_list : w.torso_list;
    a.dense_of = given ? f->torso_mask_dense_of : w.torso_dense_of;
    a.count = given ? f->torso_mask_count : w.ctrl + gf::kCtrlTorsoCount;
    a.count_reset = given ? nullptr : w.ctrl + gf::kCtrlTorsoCount;
    a.tout = w.torso_out;
#ifdef GF_TRACE
    a.ttrace = g_ttrace_buf;
#endif
    static GfLdsAttr lds[2];
    const size_t smem = kTorsoSmem + (ha ? gf::TH_W0 * sizeof(float) : 0);
    const void* fn = ha ? reinterpret_cast<const void*>(k_torso_field<true>) : reinterpret_cast<const void*>(k_torso_field<false>);
    if (const int e = gf_raise_lds_limit(lds[ha], fn, (int)smem, "torso")) return e;
    hipStream_t st = gf_stream(stream);
    const dim3 per_pixel(gf_div_up(f->n_rays, (uint32_t)kThreads));
    // the count word was cleared with the rest of the control block at the head of the frame (gf_render_head) and nothing has touched it since
    if (!given)
        hipLaunchKernelGGL(k_torso_mask, per_pixel, dim3(kThreads), 0, st, f->n_rays, f->grid_size, f->bg_coords, f->torso_occ, f->torso_thresh,
                           w.torso_list, w.torso_dense_of, w.ctrl + gf::kCtrlTorsoCount);
    // one workgroup per 128 list entries up to 512 workgroups (two per CU), which then stride over a longer list; the list's length is only
    // known on the device: workgroups beyond it return before they copy a weight
    const dim3 field_grid(gf_div_up(f->n_rays, 128u) < 512u ? gf_div_up(f->n_rays, 128u) : 512u);
    if (ha) hipLaunchKernelGGL(k_torso_field<true>, field_grid, dim3(kThreads), smem, st, a);
    else hipLaunchKernelGGL(k_torso_field<false>, field_grid, dim3(kThreads), smem, st, a);
    hipLaunchKernelGGL(k_torso_blend, per_pixel, dim3(kThreads), 0, st, a);
    return gf_check_launch("render_torso");
}

// The torso mask as a dense list for a whole frame loop (gf_frame_t.torso_mask_*): k_torso_mask into caller-owned buffers.
GF_EXPORT int gf_torso_mask_list(const float* bg_coords, const float* torso_occ, uint32_t n_rays, uint32_t grid_size, float torso_thresh, uint32_t* list,
                                 uint32_t* dense_of, uint32_t* count, void* stream) {
    if (!bg_coords || !torso_occ || !list || !dense_of || !count) return gf_set_error(GF_ERR_INVALID, "torso_mask_list: null pointer");
    hipStream_t st = gf_stream(stream);
    if (hipMemsetAsync(count, 0, sizeof(uint32_t), st) != hipSuccess) return gf_set_error(GF_ERR_HIP, "torso_mask_list: hipMemsetAsync failed");
    if (n_rays == 0) return GF_OK;
    hipLaunchKernelGGL(k_torso_mask, dim3(gf_div_up(n_rays, (uint32_t)kThreads)), dim3(kThreads), 0, st, n_rays, grid_size, bg_coords, torso_occ, torso_thresh,
                       list, dense_of, count);
    return gf_check_launch("torso_mask_list");
}

// ---- training field (gf_torso_train_t): forward with saves, input-gradient chain
static int torso_train_args(const gf_torso_train_t* t, TorsoTrainArgs& a, bool backward) {
    if (!t) return gf_set_error(GF_ERR_INVALID, "torso_train: null descriptor");
    if (!t->x || !t->torso_pack || !t->torso_table || !t->torso_offsets || !t->out || !t->dx || !t->h_d1 || !t->h_d2 || !t->x01 || !t->h_c1 || !t->h_c2)
        return gf_set_error(GF_ERR_INVALID, "torso_train: null pointer");
    if (!backward && (!t->torso_bias || !t->enc || !t->g)) return gf_set_error(GF_ERR_INVALID, "torso_train_forward: null pointer");
    if (backward && (!t->bwd_streams || !t->g_out || !t->dz_c3 || !t->dz_c2 || !t->dz_c1 || !t->dz_d3 || !t->dz_d2 || !t->dz_d1 || !t->g_grid))
        return gf_set_error(GF_ERR_INVALID, "torso_train_backward: null pointer");
    a.M = t->M; a.shrink = t->torso_shrink; a.x = t->x;
    a.pack = t->torso_pack; a.bias = t->torso_bias; a.table = t->torso_table; a.offsets = t->torso_offsets;
    if (gf::fill_grid_levels(a.lv, 16, t->torso_S, t->base_res)) return gf_set_error(GF_ERR_INVALID, "torso_train: bad grid levels");
    a.out = t->out; a.dx = t->dx; a.enc = t->enc; a.h_d1 = t->h_d1; a.h_d2 = t->h_d2; a.x01 = t->x01; a.g = t->g; a.h_c1 = t->h_c1; a.h_c2 = t->h_c2;
    a.bwd = t->bwd_streams; a.g_out = t->g_out; a.g_dx = t->g_dx;
    a.dz_c3 = t->dz_c3; a.dz_c2 = t->dz_c2; a.dz_c1 = t->dz_c1; a.dz_d3 = t->dz_d3; a.dz_d2 = t->dz_d2; a.dz_d1 = t->dz_d1; a.g_grid = t->g_grid;
    a.level_max = t->level_max;
    return GF_OK;
}

GF_EXPORT int gf_torso_train_forward(const gf_torso_train_t* t, void* stream) {
    TorsoTrainArgs a = {};
    if (const int e = torso_train_args(t, a, false)) return e;
    if (a.M == 0) return GF_OK;
    static GfLdsAttr lds;
    if (const int e = gf_raise_lds_limit(lds, reinterpret_cast<const void*>(k_torso_train_fwd), (int)kTorsoSmem, "torso_train_forward")) return e;
    const uint32_t wgs = gf_div_up(a.M, 128u);
    hipLaunchKernelGGL(k_torso_train_fwd, dim3(wgs < 512u ? wgs : 512u), dim3(kThreads), kTorsoSmem, gf_stream(stream), a);
    return gf_check_launch("torso_train_forward");
}

GF_EXPORT int gf_torso_train_backward(const gf_torso_train_t* t, void* stream) {
    TorsoTrainArgs a = {};
    if (const int e = torso_train_args(t, a, true)) return e;
    if (a.M == 0) return GF_OK;
    static GfLdsAttr lds;
    if (const int e = gf_raise_lds_limit(lds, reinterpret_cast<const void*>(k_torso_train_bwd), (int)kTorsoBwdSmem, "torso_train_backward")) return e;
    const uint32_t wgs = gf_div_up(a.M, 128u);
    hipLaunchKernelGGL(k_torso_train_bwd, dim3(wgs < 512u ? wgs : 512u), dim3(kThreads), kTorsoBwdSmem, gf_stream(stream), a);
    return gf_check_launch("torso_train_backward");
}

GF_EXPORT uint32_t gf_torso_bwd_stream_floats(void) { return TBW_TOTAL; }

#ifdef GF_TRACE
// trace build only: device buffer of 64 workgroups x 4 waves x 16 uint64 slots (tools/trace_torso.py)
GF_EXPORT void gf_torso_trace_set(void* dev_buf) { g_ttrace_buf = reinterpret_cast<unsigned long long*>(dev_buf); }
#endif
