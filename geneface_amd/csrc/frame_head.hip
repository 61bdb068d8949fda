// Fused head pass of one RAD-NeRF frame for gfx950.
//
// Replaces, in the reference (paths relative to /root/reference/modules/radnerfs), the whole inference loop of
// renderer.py:316-351: per iteration raymarching.cu:828-929 (march) + radnerf.py:73-105 (~40 launches: 2 grid encodes,
// 8 GEMMs, SH, cats, activations) + raymarching.cu:943-1029 (composite) + `rays_alive[rays_alive >= 0]` (a host sync),
// repeated up to 16 times -- by three launches and no host synchronisation.
//
// EXACT TWO-PHASE SCHEDULING.  The reference gives every still-alive ray n_step = clamp(N // n_alive, 1, 8) more samples
// per iteration until the cumulative count c reaches max_steps (renderer.py:338,351).  Two facts make the global
// iteration barrier unnecessary:
//   (1) what a ray accumulates depends only on its TOTAL sample budget, not on how the budget is cut into chunks (the
//       marcher resumes from the t it stopped at; the compositor's termination tests are per sample);
//   (2) the total budget B = last c is a function of alive(c) for c < max_steps only (the loop stops once c >= max_steps),
//       and alive(c) = N - #{rays whose terminal sample index d <= c}, d = the sample at which T dropped below T_thresh,
//       or (number of samples the ray has) + 1.
// So: phase 0 gives every ray its first max_steps samples (or fewer, if it terminates) and histograms d; phase 1 replays
// the reference's schedule from the histogram, obtains B exactly, and gives the rays still alive the remaining
// B - max_steps samples.  tests/test_gpu_render.py checks B, the alive counts and the sample totals against the oracle's
// iteration trace.
//
// WORK DECOMPOSITION.  k_frame_init: one lane per ray -- ray generation, slab test against aabb_infer and against the box
// around the occupied cells (beyond which no sample can exist), and the march through empty space up to the first occupied
// sample (a latency-bound, high-occupancy kernel; rays that never hit anything are finished here).
// k_head_phase: persistent 256-thread workgroups (2 per CU) each keep a pool of up to 128 live rays -- state in the owning
// lane's registers -- refilled from a global queue, and loop over rounds of <= 128 samples:
//   A. march    : one lane per pooled ray (bounded slice of the traversal), samples to LDS, a wave scan packs them densely
//   B. field    : the 128 samples' activations live in ONE LDS buffer H[128][128]; each wave OWNS 32 output features of
//                 every 128-wide layer and computes them for all (up to four) 32-sample tiles on v_mfma_f32_32x32x2_f32:
//                 A operand = its slice of the weights, streamed L2 -> registers (78 KiB per wave and round, consumption
//                 order, one 1-KiB global_load_dwordx4 per 16 MFMAs, reused by the four tiles), B operand = activations, one
//                 ds_read_b128 per tile and 4 MFMAs.  A layer ends with barrier / in-place write-back of the accumulators /
//                 barrier.  No weight ever touches LDS, no activation ever leaves the CU, accumulators are the only large
//                 register arrays (no spills); the three skinny layers (->2, ->1, ->3) and both grid lookups run on the
//                 VALU with one lane pair per sample
//   C. composite: the owning lane consumes its samples in order; finished rays write their accumulators once.
#include "common.hpp"
#include "frame.hpp"
#include "march_core.hpp"
#include "grid_core.hpp"
#include "sh_core.hpp"
#include "mfma_mlp.hpp"
#include <type_traits>
#include <stdlib.h>

namespace {

using gf::floatx16;

constexpr int kThreads = 256;
constexpr int kPass = 128;            // sample slots per round (four 32-column MFMA tiles)
constexpr int kPool = 128;            // live rays per workgroup
constexpr int kHS = 132;              // floats per activation row (128 + 4: rows 16 B apart in bank space -> conflict-free b128)
constexpr int kPFloats = gf::HS_TOTAL + 128 /*amb bias*/ + 2 * 16 * 8 /*level meta*/;
constexpr int kHistBins = gf::kHistLds;   // LDS bins of the terminal-index histogram (d < kHistBins; larger d: global atomics, rare)
#ifndef GF_MARCH_SLACK
#define GF_MARCH_SLACK 12
#endif
constexpr uint32_t kMarchSlack = GF_MARCH_SLACK;   // empty-space skips a ray may add to its requested samples per round (A/B 3 / 6 / 12 / 24: 12)

// ---- optional per-round timeline (built only with -DGF_TRACE into libgeneface_hip_trace.so; see tools/trace_head.py) ----
#ifdef GF_TRACE
constexpr int kTraceSlots = 48, kTraceRounds = 48, kTraceWGs = 16;
static uint32_t* g_trace_buf = nullptr;
static unsigned long long* g_span_buf = nullptr;
#define GF_STAMP(i)                                                              \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        if (threadIdx.x == 0) s.tr[(i)] = (uint32_t)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
#else
#define GF_STAMP(i) do { } while (0)
#endif

// ---- optional per-sample record (built only with -DGF_DIAG into libgeneface_hip_diag.so; see tools/fast_diag.py) ----
// One record per composited sample, keyed by (ray, cumulative sample index): lets two renders of the same frame be compared
// quantity by quantity (marcher -> grid features -> ambient -> density -> colour -> compositor inputs).
#ifdef GF_DIAG
constexpr int kDiagWords = 60;
// 0 sigma 1 r 2 g 3 b 4 dt 5 t_after 6 tag 7 workgroup 8 round 9 dense index 10 Mv 11 n_pool<<8|n
// 12 ambient0 13 ambient1 14 h0 15 x 16 checksum(3-D features as stored) 17 checksum(2-D features as stored) 18 phase 19 raw slot
// 20 x2[0] 21 x2[1] 22 sum of the fp32 2-D features 23 HW_ID 24, 25 exp(-2 ambient) (fast kernel)
// 26..57 the 32 fp32 2-D features [level][channel] (fast kernel) 58 XCC_ID
#endif

#ifdef GF_DIAG
struct DiagReg { const void* ws; float* buf; uint32_t stride, last_tag; };
static DiagReg g_diag_reg[8] = {};
static uint32_t g_diag_tag = 0;
static uint32_t g_diag_cfg[4] = {0, 0, 0, 0};   // start poison, per-round poison, grid override, pool-cap override
#endif

struct HeadArgs {
    gf::MarchParams mp;
    gf::GridLevels lv3, lv2;
    const float* pos_table; const int* pos_offsets;
    const float* amb_table; const int* amb_offsets;
    const float* head_pack; const float* amb_bias;
    const uint16_t* head_pack16;   // fast path only
    const uint16_t* head_pack_split;   // split path only
    const float* rays_o; const float* rays_d; const float* far_occ;
    float* rays_t; float* weights_sum; float* depth; float* image;
    const int* queue;   // phase 0: hit list, phase 1: survivor list
    int* survivors;     // phase 0 output
    uint32_t* ctrl;
    uint32_t N, phase, max_steps, gridtype, interp;
    float T_thresh, bound;
    float cam_o[3]; uint32_t pose_mode;   // rays generated from a pose share one origin: it travels by value, no per-ray origin array
#ifdef GF_TRACE
    uint32_t* trace;
    unsigned long long* spans;   // [2 phases][512 workgroups][start tick, end tick, rounds | samples << 32, XCC id, start realtime, end realtime (100 MHz), exit tick, exit realtime]; 'end' = end of the last round
#endif
#ifdef GF_DIAG
    float* diag;                 // [N][diag_stride][kDiagWords] per-sample record, or NULL
    uint32_t diag_tag, diag_stride, poison, poison_round, pool_cap_override;
#endif
};

// ---------------------------------------------------------------------------------------------------- LDS carve
struct Smem {
    float* H;        // [kPass][kHS] activations of the round's samples
    float* P;        // VALU-layer rows, colour bias, ambient bias, per-level grid meta
    float *p_dx, *p_dy, *p_dz;   // [kPool] ray directions by pool slot (read by the SH evaluation of the slot's samples)
    // per-round sample staging (raw slot = rank * n + s); field outputs alias the positions
    float *sx, *sy, *sz, *sdt, *st, *ob;
    uint8_t *d2r, *rcnt, *rbase, *rrank;
    uint32_t* hist;  // [kHistBins]
    uint32_t* misc;  // [16]
#ifdef GF_TRACE
    uint32_t* tr;    // [kTraceSlots]
#endif
#ifdef GF_DIAG
    uint32_t* dkey;  // [kPass] record index of the sample in raw slot r (0xFFFFFFFF: none)
#endif
};
constexpr int kSmemBase = (kPass * kHS + kPFloats + 3 * kPool + 6 * kPass + kHistBins + 16) * 4 + 4 * kPass;
#if defined(GF_TRACE)
constexpr int kSmemBytes = kSmemBase + 4 * kTraceSlots;
#elif defined(GF_DIAG)
constexpr int kSmemBytes = kSmemBase + 4 * kPass;
#else
constexpr int kSmemBytes = kSmemBase;
#endif
static_assert(2 * kSmemBytes <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ Smem carve(char* base) {
    Smem s;
    float* f = reinterpret_cast<float*>(base);
    s.H = f; f += kPass * kHS;
    s.P = f; f += kPFloats;
    s.p_dx = f; f += kPool; s.p_dy = f; f += kPool; s.p_dz = f; f += kPool;
    s.sx = f; f += kPass; s.sy = f; f += kPass; s.sz = f; f += kPass; s.sdt = f; f += kPass; s.st = f; f += kPass; s.ob = f; f += kPass;
    s.hist = reinterpret_cast<uint32_t*>(f); f += kHistBins;
    s.misc = reinterpret_cast<uint32_t*>(f); f += 16;
    uint8_t* b = reinterpret_cast<uint8_t*>(f);
    s.d2r = b; s.rcnt = b + kPass; s.rbase = b + 2 * kPass; s.rrank = b + 3 * kPass;
#ifdef GF_TRACE
    s.tr = reinterpret_cast<uint32_t*>(b + 4 * kPass);
#endif
#ifdef GF_DIAG
    s.dkey = reinterpret_cast<uint32_t*>(b + 4 * kPass);
#endif
    return s;
}
constexpr int P_SMALL = 0, P_AMBBIAS = gf::HS_TOTAL, P_META = gf::HS_TOTAL + 128;

// The reference's schedule replayed from the terminal-index histogram: returns the total budget B.
__device__ uint32_t replay_budget(const uint32_t* __restrict__ hist, uint32_t N, uint32_t max_steps) {
    uint32_t c = 0, dead = 0, d = 0;
    while (c < max_steps) {
        while (d < c) { d++; dead += hist[d]; }  // dead = #{rays with terminal index <= c}
        const uint32_t n_alive = N - dead;
        if (n_alive == 0) break;
        uint32_t n = N / n_alive;
        n = n > 8u ? 8u : (n < 1u ? 1u : n);
        c += n;
    }
    return c;
}

// ---------------------------------------------------------------------------------------------------- field building blocks
// One 4-step group of A operands: wave-uniform stream base (SGPRs) + 32-bit lane offset -> global_load_dwordx4 v, v_off, s[base]
__device__ __forceinline__ float4 load_group(const char* __restrict__ Ws, uint32_t g, uint32_t lane16) {
    return *reinterpret_cast<const float4*>(Ws + (size_t)g * 1024 + (size_t)lane16);
}

// The stream is consumed strictly in order (frame.hpp: G_AMB1 .. G_COL1G), so a few register quads form a FIFO that runs kAhead groups
// (of 16 MFMAs = 1 024 cycles with four sample tiles) ahead of the MFMAs, across layer boundaries and barriers: q[g % kAhead] holds group g
// from the moment group g - kAhead has been issued.  Three ahead in the frame kernels; two in the training kernels, which need the four
// registers more than the slack (with three they spill, and a launch that uses scratch at all pays for it: NOTES.md 4.7).
template <int A>
struct WPipeT { static constexpr int kAhead = A; float4 q[A]; };
constexpr int kWAhead = 3, kWAheadTrain = 2;
using WPipe = WPipeT<kWAhead>;

template <int G, class WP>
__device__ __forceinline__ float4 wpipe_take(WP& wp, const char* __restrict__ Ws, uint32_t lane16) {
    return wp.q[G % WP::kAhead];
}
// GEND: one past the last group this launch consumes (the density-only field stops before density L3)
template <int G, int GEND = (int)gf::G_TOTAL, class WP>
__device__ __forceinline__ void wpipe_refill(WP& wp, const char* __restrict__ Ws, uint32_t lane16) {
    if constexpr (G + WP::kAhead < GEND) wp.q[G % WP::kAhead] = load_group(Ws, G + WP::kAhead, lane16);
}

// U 4-step groups (first one = group G0 of this wave's stream Ws) of this wave's output block over NT sample tiles.
// Hb = &H[lane & 31][col0 + 4 * (lane >> 5)]: tile t is 32 rows further, group u eight floats further.
// Software pipeline, pinned with sched_barrier so the scheduler cannot sink the loads next to their use: while the 16 MFMAs
// of group u issue, the B operands of group u+1 (LDS) and the A operands of group u+3 (L2) are in flight.
template <int NT, int G0, int U, int u, int GEND = (int)gf::G_TOTAL, class WP>
__device__ __forceinline__ void obw_step(WP& wp, const char* __restrict__ Ws, uint32_t lane16, const float* Hb, floatx16 (&acc)[4],
                                         const float4 (&b)[NT]) {
    if constexpr (u < U) {
        float4 bn[NT];
        if constexpr (u + 1 < U) {
#pragma unroll
            for (int t = 0; t < NT; t++) bn[t] = *reinterpret_cast<const float4*>(Hb + t * 32 * kHS + 8 * (u + 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        const float4 a = wpipe_take<G0 + u>(wp, Ws, lane16);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
        wpipe_refill<G0 + u, GEND>(wp, Ws, lane16);
        __builtin_amdgcn_sched_barrier(0);
        obw_step<NT, G0, U, u + 1, GEND>(wp, Ws, lane16, Hb, acc, bn);
    }
}
template <int NT, int G0, int U, int GEND = (int)gf::G_TOTAL, class WP>
__device__ __forceinline__ void obw_mfma(WP& wp, const char* __restrict__ Ws, uint32_t lane16, const float* Hb, floatx16 (&acc)[4]) {
    float4 b[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) b[t] = *reinterpret_cast<const float4*>(Hb + t * 32 * kHS);
    __builtin_amdgcn_sched_barrier(0);
    obw_step<NT, G0, U, 0, GEND>(wp, Ws, lane16, Hb, acc, b);
}

template <int NT>
__device__ __forceinline__ void obw_bias_n(const float* bias16, floatx16 (&acc)[NT]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 b = reinterpret_cast<const float4*>(bias16)[q];
#pragma unroll
        for (int t = 0; t < NT; t++) { acc[t][4 * q + 0] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w; }
    }
}

template <int NT>
__device__ __forceinline__ void obw_zero(floatx16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
}

// bias16 = this lane's 16 entries of a bias vector stored in accumulator-layout order [out_block][half][16]
template <int NT>
__device__ __forceinline__ void obw_bias(const float* bias16, floatx16 (&acc)[4]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 b = reinterpret_cast<const float4*>(bias16)[q];
#pragma unroll
        for (int t = 0; t < NT; t++) { acc[t][4 * q + 0] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w; }
    }
}

// max(x, 0) as ONE v_max_f32: fmaxf / fmed3 lower to a canonicalising v_max_f32 x, x, x followed by the max (IEEE sNaN quieting),
// which doubles the VALU work of every write-back; MFMA results are never signalling NaNs.
__device__ __forceinline__ float relu1(float x) {
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}

// Accumulator registers 4q..4q+3 of (lane half h) are features 32w + 8q + 4h + 0..3 of sample (lane & 31): one ds_write_b128.
// Hw = &H[lane & 31][32 * wave + 4 * (lane >> 5)].
template <int NT, bool RELU>
__device__ __forceinline__ void obw_store(float* Hw, const floatx16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 v = {acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
            if (RELU) { v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w); }
            *reinterpret_cast<float4*>(Hw + t * 32 * kHS + 8 * q) = v;
        }
}

// Training forward (gf_field_forward with save buffers): what the backward pass needs of every layer, as [M, width] fp32 in global memory.
struct SaveBufs {
    float *f3, *ha1, *ha2, *f2, *hs1, *hs2, *geo, *hc1;   // [M,32] [M,128] [M,128] [M,32] [M,128] [M,128] [M,128] [M,128]
    uint16_t *m_ha1, *m_ha2, *m_hs1, *m_hs2, *m_hc1;      // ReLU masks in ACCUMULATOR layout: [chunk of 128][tile 4][wave 4][lane 64], bit r = register r > 0
    float* sh;                                            // or NULL: [M,16] the SH basis of the direction (colour L1's other input)
};

// The values obw_store writes to LDS, also to row (gbase + sample) of a [M,128] matrix: this lane's 4 consecutive floats per (tile, q) --
// the wave's four stores per tile complete one 128-byte line of every sample row.
template <int NT, bool RELU>
__device__ __forceinline__ void obw_save(float* __restrict__ G, uint32_t gbase, uint32_t Mv, int wave, int lane, const floatx16 (&acc)[4],
                                         uint16_t* __restrict__ mask = nullptr) {
    const int half = lane >> 5, j = lane & 31;
    if (mask) {   // the backward pass applies the ReLU derivative from these bits instead of re-reading the activations
#pragma unroll
        for (int t = 0; t < NT; t++) {
            uint32_t bits = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) bits |= (acc[t][r] > 0.0f ? 1u : 0u) << r;
            mask[(((size_t)(gbase / kPass) * 4 + t) * 4 + wave) * 64 + lane] = (uint16_t)bits;
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        if ((uint32_t)(t * 32 + j) < Mv) {
            float* row = G + (size_t)(gbase + (uint32_t)(t * 32 + j)) * 128 + 32 * wave + 4 * half;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 v = {acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                if (RELU) { v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w); }
                *reinterpret_cast<float4*>(row + 8 * q) = v;
            }
        }
    }
}

// The ReLU mask alone (the row itself leaves through rows32_to_global).
template <int NT>
__device__ __forceinline__ void obw_mask(uint16_t* __restrict__ mask, uint32_t gbase, int wave, int lane, const floatx16 (&acc)[4]) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) bits |= (acc[t][r] > 0.0f ? 1u : 0u) << r;
        mask[(((size_t)(gbase / kPass) * 4 + t) * 4 + wave) * 64 + lane] = (uint16_t)bits;
    }
}

// A layer's 128 x 128 fp32 rows from the LDS activation buffer to their [M,128] matrix (round 6; see rows16_to_global for the why): after the
// barrier that follows the layer's write-back, 32 consecutive lanes move one whole 512-byte row, a wave instruction two complete rows.  The
// matrix pointer stays a kernel argument and the lane adds one 32-bit byte offset (the training kernels have no register to spare): hence
// M * 512 B < 4 GiB on this tier.  Same bits: H holds the same (ReLU'd) accumulator values.
__device__ __forceinline__ void rows32_to_global(const float* H, float* __restrict__ G, uint32_t gbase, uint32_t Mv, int tid) {
#ifdef GF_PROBE_NO_SAVES
    return;
#endif
    __builtin_amdgcn_sched_barrier(0);
    char* __restrict__ Gc = reinterpret_cast<char*>(G);
    const int row0 = tid >> 5, col = (tid & 31) * 4;
    const float* src = H + row0 * kHS + col;
    const uint32_t dst = ((gbase + (uint32_t)row0) * 128u + (uint32_t)col) * 4u;
#pragma unroll 1
    for (int i = 0; i < kPass * 32 / kThreads; i++) {
        if ((uint32_t)(row0 + 8 * i) < Mv)
            *reinterpret_cast<float4*>(Gc + (dst + (uint32_t)i * (8u * 512u))) = *reinterpret_cast<const float4*>(src + i * 8 * kHS);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// NOUT skinny outputs of one sample on a lane pair: lane half h sums features 64h .. 64h+63 of the sample's row, the halves
// are added with one cross-half exchange (both lanes end up with the same value).
template <int NOUT>
__device__ __forceinline__ void rows_from_lds(const float* Hrow, const float* rows, int half, float (&res)[NOUT]) {
    float sum[NOUT];
#pragma unroll
    for (int c = 0; c < NOUT; c++) sum[c] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float4 x = *reinterpret_cast<const float4*>(Hrow + 64 * half + 4 * i);
#pragma unroll
        for (int c = 0; c < NOUT; c++) {
            const float4 w = *reinterpret_cast<const float4*>(rows + c * 128 + 64 * half + 4 * i);
            sum[c] = __builtin_fmaf(w.x, x.x, sum[c]);
            sum[c] = __builtin_fmaf(w.y, x.y, sum[c]);
            sum[c] = __builtin_fmaf(w.z, x.z, sum[c]);
            sum[c] = __builtin_fmaf(w.w, x.w, sum[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NOUT; c++) res[c] = sum[c] + __shfl_xor(sum[c], 32);
}

// The same NOUT skinny outputs WITHOUT a trip of the 128-wide activations through LDS (round 4).  The activations a skinny layer reads
// are the ReLU'd accumulators of the layer before it, and they sit in registers when that layer's MFMAs end: every wave sums its own 32
// features for all NT tiles (skinny_partials: 16 x NOUT FMAs per tile and lane, the lane halves added with one exchange), lane half 0
// leaves the wave's partials in 16 bytes of the sample's row ([wave][c], skinny_publish), and after a barrier the sample's lane pair adds
// the four waves' partials in wave order (skinny_collect).  Against rows_from_lds per layer and round: no 16 ds_write_b128 + 16 ds_read_b128
// of activations and 4 x NOUT instead of 16 x NOUT weight reads per lane -- a third of the kernel's LDS traffic came from the two write-backs
// nobody else read (ambient L2, colour L1) and the three row passes.  Fixed summation order (features in register order within a wave, waves
// 0..3): run-to-run identical; it is a different order from rows_from_lds, i.e. last-ulp different sums -- the tolerance is the oracle's.
// rows: [NOUT][128] in natural feature order; acc registers 4q..4q+3 of lane half h = features 32 wave + 8q + 4h + 0..3.
template <int NT, int NOUT>
__device__ __forceinline__ void skinny_partials(const floatx16 (&acc)[4], const float* rows, int wave, int half, float (&part)[NT][NOUT]) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int c = 0; c < NOUT; c++) part[t][c] = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float4 w4[NOUT];
#pragma unroll
        for (int c = 0; c < NOUT; c++) w4[c] = *reinterpret_cast<const float4*>(rows + c * 128 + 32 * wave + 8 * q + 4 * half);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float x0 = relu1(acc[t][4 * q + 0]), x1 = relu1(acc[t][4 * q + 1]), x2 = relu1(acc[t][4 * q + 2]), x3 = relu1(acc[t][4 * q + 3]);
#pragma unroll
            for (int c = 0; c < NOUT; c++) {
                part[t][c] = __builtin_fmaf(w4[c].x, x0, part[t][c]);
                part[t][c] = __builtin_fmaf(w4[c].y, x1, part[t][c]);
                part[t][c] = __builtin_fmaf(w4[c].z, x2, part[t][c]);
                part[t][c] = __builtin_fmaf(w4[c].w, x3, part[t][c]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int c = 0; c < NOUT; c++) part[t][c] += __shfl_xor(part[t][c], 32);
}
// dst = &H[lane & 31][col0 + 4 * wave]: tile t is 32 rows further; one 16-byte (NOUT > 1) or 4-byte write per tile, lane half 0 only
template <int NT, int NOUT, int RS = kHS /* floats per activation row */>
__device__ __forceinline__ void skinny_publish(float* dst, int half, const float (&part)[NT][NOUT]) {
    if (half == 0) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if constexpr (NOUT == 1) dst[t * 32 * RS] = part[t][0];
            else *reinterpret_cast<float4*>(dst + t * 32 * RS) = float4{part[t][0], part[t][1], NOUT > 2 ? part[t][NOUT > 2 ? 2 : 0] : 0.0f, 0.0f};
        }
    }
}
// src = &H[sample][col0]: [wave][c] (NOUT > 1: 4 floats per wave) or [wave] (NOUT == 1)
template <int NOUT>
__device__ __forceinline__ void skinny_collect(const float* src, float (&res)[NOUT]) {
    if constexpr (NOUT == 1) {
        const float4 p = *reinterpret_cast<const float4*>(src);
        res[0] = ((p.x + p.y) + p.z) + p.w;
    } else {
        float4 p[4];
#pragma unroll
        for (int w = 0; w < 4; w++) p[w] = *reinterpret_cast<const float4*>(src + 4 * w);
        res[0] = ((p[0].x + p[1].x) + p[2].x) + p[3].x;
        res[1] = ((p[0].y + p[1].y) + p[2].y) + p[3].y;
        if constexpr (NOUT > 2) res[2] = ((p[0].z + p[1].z) + p[2].z) + p[3].z;
    }
}

__device__ __forceinline__ void store16(float* dst, const float (&f)[16]) {
#pragma unroll
    for (int q = 0; q < 4; q++) reinterpret_cast<float4*>(dst)[q] = float4{f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
}

// The field (radnerf.py:73-105) for the round's Mv densely packed samples, NT = ceil(Mv / 32) tiles.
// DENSITY_ONLY = RADNeRF.density (radnerf.py:107-126): the same layers up to the density row; sigma is left in s.sx[raw].
// AMB_OUT (dense point lists only: the staging arrays are free there): tanh(ambient) is left in s.sdt / s.st [raw].
// SAVE (dense point lists only, dense index == point index - gbase): every layer's activations go to `sv` as well (training forward).
template <int NT, bool DENSITY_ONLY = false, bool AMB_OUT = false, bool SAVE = false>
__device__ __forceinline__ void field_round(const HeadArgs& a, const Smem& s, uint32_t Mv, int wave, int lane, const SaveBufs* sv = nullptr,
                                            uint32_t gbase = 0) {
    constexpr int GEND = DENSITY_ONLY ? (int)gf::G_SIG3 : (int)gf::G_TOTAL;
    const int half = lane >> 5, j = lane & 31;
    // lane-pair view (grid lookups, skinny layers): sample sI of this wave's tile
    const uint32_t sI = (uint32_t)(wave * 32 + j);
    const bool tile_on = wave < NT;   // wave-uniform: this wave's tile holds samples
    const bool valid = sI < Mv;
    const uint32_t sC = valid ? sI : (uint32_t)(wave * 32);   // a valid stand-in inside an active tile
    const uint32_t raw = tile_on ? s.d2r[sC] : 0u;
    float* Hrow = s.H + sI * kHS;
    // MFMA view: B-operand / write-back bases of sample column j
    const float* Hb = s.H + j * kHS + 4 * half;
    float* Hw = s.H + j * kHS + 32 * wave + 4 * half;
    const char* Ws = reinterpret_cast<const char*>(a.head_pack + gf::HP_STREAM) + (size_t)wave * gf::G_TOTAL * 1024;   // wave-uniform
    uint32_t lane16 = (uint32_t)lane * 16u;
    // Opaque to the optimiser: otherwise it folds base + lane into one 64-bit VGPR address per group (78 of them), hoists
    // them all out of the round loop and spills them.  Kept symbolic, every load is `global_load v, v_lane16, s[base] offset`.
    asm volatile("" : "+v"(lane16));
    const gf::LevelMeta* meta = reinterpret_cast<const gf::LevelMeta*>(s.P + P_META);

    // Wave priority: the short VALU / LDS / gather segments run at high priority, the MFMA segments at low priority.  The
    // co-resident workgroup is usually inside an MFMA segment (one issue slot per 64 cycles): letting this wave's scalar work
    // through first shortens the stretch in which BOTH workgroups are off the matrix pipe, the only time it idles.
#ifndef GF_NO_SETPRIO
#define GF_PRIO_HI() __builtin_amdgcn_s_setprio(3)
#define GF_PRIO_LO() __builtin_amdgcn_s_setprio(0)
#else
#define GF_PRIO_HI() do { } while (0)
#define GF_PRIO_LO() do { } while (0)
#endif
#ifdef GF_DIAG
    uint32_t dkey = 0xFFFFFFFFu;
#endif
    floatx16 A[4], S[4];
    WPipeT<(SAVE ? kWAheadTrain : kWAhead)> wp;
#pragma unroll
    for (int g = 0; g < decltype(wp)::kAhead; g++) wp.q[g] = load_group(Ws, g, lane16);   // lands while the grid lookups run

    // ---- 3-D grid features -> H[:, 0:32]
    if (tile_on) {
        const float b2 = 2 * a.bound;
        const float x3[3] = {(s.sx[raw] + a.bound) / b2, (s.sy[raw] + a.bound) / b2, (s.sz[raw] + a.bound) / b2};
        float pf[16];
        gf::encode8<3>(a.pos_table, meta + half * 8, a.gridtype, a.interp, x3, pf);
        store16(Hrow + 16 * half, pf);
        if constexpr (SAVE) {
            if (valid) store16(sv->f3 + (size_t)(gbase + sI) * 32 + 16 * half, pf);
        }
#ifdef GF_DIAG
        dkey = valid ? s.dkey[raw] : 0xFFFFFFFFu;
        float c = 0.0f;
        for (int i = 0; i < 16; i++) c += pf[i];
        c += __shfl_xor(c, 32);
        if (a.diag && dkey != 0xFFFFFFFFu && half == 0) a.diag[(size_t)dkey * kDiagWords + 16] = c;
#endif
    }
    GF_STAMP(7);
    __syncthreads();
    GF_STAMP(8);
    GF_PRIO_LO();
    // ---- ambient L1 (cond_feat folded into the bias) and the 3-D half of density L1, both from H[:, 0:32]
    obw_bias<NT>(s.P + P_AMBBIAS + wave * 32 + half * 16, A);
    obw_zero<NT>(S);
    obw_mfma<NT, gf::G_AMB1, 4, GEND>(wp, Ws, lane16, Hb, A);
    obw_mfma<NT, gf::G_SIG1A, 4, GEND>(wp, Ws, lane16, Hb, S);
    GF_PRIO_HI();
    GF_STAMP(9);
    __syncthreads();
    GF_STAMP(10);
    obw_store<NT, true>(Hw, A);
    if constexpr (SAVE) obw_mask<NT>(sv->m_ha1, gbase, wave, lane, A);
    GF_STAMP(11);
    __syncthreads();
    GF_STAMP(12);
    if constexpr (SAVE) rows32_to_global(s.H, sv->ha1, gbase, Mv, wave * 64 + lane);
    // ---- ambient L2
    GF_PRIO_LO();
    obw_zero<NT>(A);
    obw_mfma<NT, gf::G_AMB2, 16, GEND>(wp, Ws, lane16, Hb, A);
    GF_PRIO_HI();
    GF_STAMP(13);
#ifndef GF_SKINNY_FROM_LDS
    {   // ambient L3 from the accumulators: relu(ambient L2) is read by nothing else and never reaches LDS
        float part[NT][2];
        skinny_partials<NT, 2>(A, s.P + P_SMALL + gf::HS_AMB3, wave, half, part);
        __syncthreads();   // every wave has read its last ambient-L1 activation
        GF_STAMP(14);
        skinny_publish<NT, 2>(s.H + j * kHS + 32 + 4 * wave, half, part);   // columns 32..47: the 2-D features go to 0..31 below
    }
    if constexpr (SAVE) obw_save<NT, true>(sv->ha2, gbase, Mv, wave, lane, A, sv->m_ha2);
#else
    __syncthreads();
    GF_STAMP(14);
    obw_store<NT, true>(Hw, A);
    if constexpr (SAVE) obw_save<NT, true>(sv->ha2, gbase, Mv, wave, lane, A, sv->m_ha2);
#endif
    GF_STAMP(15);
    __syncthreads();
    GF_STAMP(16);
    // ---- ambient L3 + tanh -> 2-D grid features -> H[:, 0:32]  (a lane pair only touches its own sample's row)
    if (tile_on) {
        float ambient[2];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<2>(Hrow + 32, ambient);
#else
        rows_from_lds<2>(Hrow, s.P + P_SMALL + gf::HS_AMB3, half, ambient);
#endif
        const float th[2] = {tanhf(ambient[0]), tanhf(ambient[1])};
        const float x2[2] = {(th[0] + 1.0f) / 2.0f, (th[1] + 1.0f) / 2.0f};
        if constexpr (AMB_OUT) {
            if (valid && half == 0) { s.sdt[raw] = th[0]; s.st[raw] = th[1]; }
        }
        float af[16];
        gf::encode8<2>(a.amb_table, meta + 16 + half * 8, a.gridtype, a.interp, x2, af);
        store16(Hrow + 16 * half, af);
        if constexpr (SAVE) {
            if (valid) store16(sv->f2 + (size_t)(gbase + sI) * 32 + 16 * half, af);
        }
#ifdef GF_DIAG
        float c = 0.0f;
        for (int i = 0; i < 16; i++) c += af[i];
        c += __shfl_xor(c, 32);
        if (a.diag && dkey != 0xFFFFFFFFu && half == 0) {
            float* rec = a.diag + (size_t)dkey * kDiagWords;
            rec[12] = ambient[0]; rec[13] = ambient[1]; rec[17] = c;
            rec[20] = x2[0]; rec[21] = x2[1]; rec[22] = c;
            reinterpret_cast<uint32_t*>(rec)[23] = __builtin_amdgcn_s_getreg(63492 /* HW_REG_HW_ID, 32 bits */);
        }
#endif
    }
    GF_STAMP(17);
    __syncthreads();
    GF_STAMP(18);
    // ---- density L1, 2-D half
    GF_PRIO_LO();
    obw_mfma<NT, gf::G_SIG1B, 4, GEND>(wp, Ws, lane16, Hb, S);
    GF_PRIO_HI();
    GF_STAMP(19);
    __syncthreads();
    GF_STAMP(20);
    obw_store<NT, true>(Hw, S);
    if constexpr (SAVE) obw_mask<NT>(sv->m_hs1, gbase, wave, lane, S);
    GF_STAMP(21);
    __syncthreads();
    GF_STAMP(22);
    if constexpr (SAVE) rows32_to_global(s.H, sv->hs1, gbase, Mv, wave * 64 + lane);
    // ---- density L2
    GF_PRIO_LO();
    obw_zero<NT>(A);
    obw_mfma<NT, gf::G_SIG2, 16, GEND>(wp, Ws, lane16, Hb, A);
    GF_PRIO_HI();
    GF_STAMP(23);
    __syncthreads();
    GF_STAMP(24);
    obw_store<NT, true>(Hw, A);
#ifndef GF_SKINNY_FROM_LDS
    {   // the density row's partial sums from the same registers; they travel in the four pad floats of the sample's row
        float part[NT][1];
        skinny_partials<NT, 1>(A, s.P + P_SMALL + gf::HS_SIGROW, wave, half, part);
        skinny_publish<NT, 1>(s.H + j * kHS + 128 + wave, half, part);
    }
#endif
    if constexpr (SAVE) obw_mask<NT>(sv->m_hs2, gbase, wave, lane, A);
    GF_STAMP(25);
    __syncthreads();
    GF_STAMP(26);
    if constexpr (SAVE) rows32_to_global(s.H, sv->hs2, gbase, Mv, wave * 64 + lane);
    // ---- density L3: row 0 on the VALU (sigma = exp, trunc_exp forward has no clamp: utils.py:41), rows 1..128 = geometry feature
    float sigma = 0.0f;
    if (tile_on) {
        float h0[1];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<1>(Hrow + 128, h0);
#else
        rows_from_lds<1>(Hrow, s.P + P_SMALL + gf::HS_SIGROW, half, h0);
#endif
        sigma = expf(h0[0]);
#ifdef GF_DIAG
        if (a.diag && dkey != 0xFFFFFFFFu && half == 0) a.diag[(size_t)dkey * kDiagWords + 14] = h0[0];
#endif
    }
    if constexpr (DENSITY_ONLY) {
        if (tile_on && valid && half == 0) s.sx[raw] = sigma;
        __syncthreads();
        return;
    }
    GF_PRIO_LO();
    obw_zero<NT>(A);
    obw_mfma<NT, gf::G_SIG3, 16>(wp, Ws, lane16, Hb, A);
    GF_PRIO_HI();
    GF_STAMP(27);
    __syncthreads();
    GF_STAMP(28);
    obw_store<NT, false>(Hw, A);   // no activation
    GF_STAMP(29);
    __syncthreads();
    GF_STAMP(30);
    if constexpr (SAVE) rows32_to_global(s.H, sv->geo, gbase, Mv, wave * 64 + lane);
    // ---- colour L1: [SH(dir) 16 | geometry 128 | identity code -> bias]
    obw_bias<NT>(s.P + P_SMALL + gf::HS_COLBIAS + wave * 32 + half * 16, A);
    {
        float shb[NT][8];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            uint32_t d = (uint32_t)(t * 32 + j);
            d = d < Mv ? d : Mv - 1u;
            const uint32_t slot = s.rrank[d];
            float sh[16];
            gf::sh4(s.p_dx[slot], s.p_dy[slot], s.p_dz[slot], sh);
            if constexpr (SAVE) {   // every wave evaluates every tile's directions: wave t writes tile t's rows
                if (sv->sh && t == wave && half == 0 && (uint32_t)(t * 32 + j) < Mv) {
                    float4* row = reinterpret_cast<float4*>(sv->sh + (size_t)(gbase + (uint32_t)(t * 32 + j)) * 16);
#pragma unroll
                    for (int q = 0; q < 4; q++) row[q] = float4{sh[4 * q], sh[4 * q + 1], sh[4 * q + 2], sh[4 * q + 3]};
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int i = 0; i < 4; i++) shb[t][4 * u + i] = half ? sh[8 * u + 4 + i] : sh[8 * u + i];
        }
        auto sh_group = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const float4 w4 = wpipe_take<gf::G_COL1S + u>(wp, Ws, lane16);
#pragma unroll
            for (int t = 0; t < NT; t++) A[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, shb[t][4 * u + 0], A[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) A[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, shb[t][4 * u + 1], A[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) A[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, shb[t][4 * u + 2], A[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) A[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, shb[t][4 * u + 3], A[t], 0, 0, 0);
            wpipe_refill<gf::G_COL1S + u>(wp, Ws, lane16);
            __builtin_amdgcn_sched_barrier(0);
        };
        GF_PRIO_LO();
        sh_group(std::integral_constant<int, 0>{});
        sh_group(std::integral_constant<int, 1>{});
    }
    obw_mfma<NT, gf::G_COL1G, 16>(wp, Ws, lane16, Hb, A);
    GF_PRIO_HI();
    GF_STAMP(31);
#ifndef GF_SKINNY_FROM_LDS
    {   // colour L2 from the accumulators: relu(colour L1) never reaches LDS
        float part[NT][3];
        skinny_partials<NT, 3>(A, s.P + P_SMALL + gf::HS_COL2, wave, half, part);
        __syncthreads();   // every wave has read its last geometry feature
        GF_STAMP(32);
        skinny_publish<NT, 3>(s.H + j * kHS + 4 * wave, half, part);
    }
    if constexpr (SAVE) obw_save<NT, true>(sv->hc1, gbase, Mv, wave, lane, A, sv->m_hc1);
#else
    __syncthreads();
    GF_STAMP(32);
    obw_store<NT, true>(Hw, A);
    if constexpr (SAVE) obw_save<NT, true>(sv->hc1, gbase, Mv, wave, lane, A, sv->m_hc1);
#endif
    GF_STAMP(33);
    __syncthreads();
    GF_STAMP(34);
    // ---- colour L2 + sigmoid; outputs reuse the position slots (read before the first barrier of this function)
    if (tile_on) {
        float c[3];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<3>(Hrow, c);
#else
        rows_from_lds<3>(Hrow, s.P + P_SMALL + gf::HS_COL2, half, c);
#endif
        if (valid && half == 0) {
            s.sx[raw] = sigma;
            s.sy[raw] = 1.0f / (1.0f + __expf(-c[0]));
            s.sz[raw] = 1.0f / (1.0f + __expf(-c[1]));
            s.ob[raw] = 1.0f / (1.0f + __expf(-c[2]));
        }
    }
    GF_STAMP(35);
    __syncthreads();
    GF_STAMP(36);
}

// ---------------------------------------------------------------------------------------------------- fast path (f16 operands)
// Same decomposition as field_round, with the activations of the round kept as f16 in the first 34 KiB of H (rows of
// 128 + 8 halves: 272 B, again 16 B apart in bank space), the weights streamed as f16 and one v_mfma_f32_32x32x16_f16 per group
// (K = 16) and tile.  Accumulation, biases, the three VALU layers' sums, exp / tanh / sigmoid and the compositor stay fp32.
// This is the arithmetic of the reference under torch autocast / model.half() (its training and viewer paths,
// tasks/radnerfs/radnerf.py:359, inference/nerfs/radnerf_gui.py:604-605); BASELINE.md section 4 sets its parity bar (PSNR >= 40 dB).
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
constexpr int kHS16 = 136;            // halves per activation row
constexpr int kFS16 = 72;             // halves per row of the 3-D feature buffer (32 used; 144 B: rows 16 B apart in bank space)
static_assert((kPass * kHS16 + kPass * kFS16 + kPool * 16) * 2 <= kPass * kHS * 4, "f16 activations + 3-D features + SH table fit the fp32 activation buffer");
// A-operand FIFO depth of the f16 kernel.  Two, not more: with four the kernel needed 80 bytes of scratch per lane, and a launch that uses scratch
// at all costs ~45 us more when it runs alone (0.57 -> 0.49 ms of head kernel per frame; the same was seen on a build of the fp32 kernel that
// spilled 12 bytes: +55 us per launch) -- with frames in flight the cost hides, the frame rate is the same either way, and this kernel spends
// 5 % of a round on the matrix pipe, so the shallower prefetch costs nothing.
#ifndef GF_WAHEAD16
#define GF_WAHEAD16 2
#endif
constexpr int kWAhead16 = GF_WAHEAD16;
struct WPipe16 { float4 q[kWAhead16]; };

template <int G>
__device__ __forceinline__ half8 wpipe16_take(const WPipe16& wp) {
    return __builtin_bit_cast(half8, wp.q[G % kWAhead16]);
}
template <int G, int GEND = (int)gf::H16_TOTAL>
__device__ __forceinline__ void wpipe16_refill(WPipe16& wp, const char* __restrict__ Ws, uint32_t lane16) {
    if constexpr (G + kWAhead16 < GEND) wp.q[G % kWAhead16] = load_group(Ws, G + kWAhead16, lane16);
}

// Hb = &H16[lane & 31][8 * (lane >> 5)]: tile t is 32 rows further, group u sixteen halves further.
// The tile count nt (1..4, wave-uniform, an SGPR) is a RUN-TIME bound here, unlike the strict path's NT template parameter: one
// instantiation means one instruction selection for every fp32 -> f16 conversion and fma, so a sample's result does not depend on
// how full the round it happens to land in is.  (With four instantiations the frames differed from run to run in a few
// hundredths of a percent of the pixels: the dynamic ray queue decides which instantiation evaluates which sample.)
template <int G0, int U, int u, int RS, int GEND = (int)gf::H16_TOTAL>
__device__ __forceinline__ void obw16_step(WPipe16& wp, const char* __restrict__ Ws, uint32_t lane16, const _Float16* Hb, floatx16 (&acc)[4],
                                           const half8 (&b)[4], int nt) {
    if constexpr (u < U) {
        half8 bn[4];
        if constexpr (u + 1 < U) {
#pragma unroll
            for (int t = 0; t < 4; t++)
                if (t < nt) bn[t] = *reinterpret_cast<const half8*>(Hb + t * 32 * RS + 16 * (u + 1));
            __builtin_amdgcn_sched_barrier(0);
        }
        const half8 a = wpipe16_take<G0 + u>(wp);
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[t], acc[t], 0, 0, 0);
        wpipe16_refill<G0 + u, GEND>(wp, Ws, lane16);
        __builtin_amdgcn_sched_barrier(0);
        obw16_step<G0, U, u + 1, RS, GEND>(wp, Ws, lane16, Hb, acc, bn, nt);
    }
}
template <int G0, int U, int RS = kHS16, int GEND = (int)gf::H16_TOTAL>
__device__ __forceinline__ void obw16_mfma(WPipe16& wp, const char* __restrict__ Ws, uint32_t lane16, const _Float16* Hb, floatx16 (&acc)[4], int nt) {
    half8 b[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (t < nt) b[t] = *reinterpret_cast<const half8*>(Hb + t * 32 * RS);
    __builtin_amdgcn_sched_barrier(0);
    obw16_step<G0, U, 0, RS, GEND>(wp, Ws, lane16, Hb, acc, b, nt);
}

// Hw = &H16[lane & 31][32 * wave + 4 * (lane >> 5)]: registers 4q..4q+3 -> four consecutive halves (one ds_write_b64)
template <bool RELU>
__device__ __forceinline__ void obw16_store(_Float16* Hw, const floatx16 (&acc)[4], int nt) {
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (t < nt) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 v = {acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                if (RELU) { v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w); }
                const half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                *reinterpret_cast<half4*>(Hw + t * 32 * kHS16 + 8 * q) = h;
            }
        }
}

template <int NOUT>
__device__ __forceinline__ void rows_from_lds16(const _Float16* Hrow, const float* rows, int half, float (&res)[NOUT]) {
    float sum[NOUT];
#pragma unroll
    for (int c = 0; c < NOUT; c++) sum[c] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const half8 x = *reinterpret_cast<const half8*>(Hrow + 64 * half + 8 * i);
#pragma unroll
        for (int c = 0; c < NOUT; c++) {
            const float4 w0 = *reinterpret_cast<const float4*>(rows + c * 128 + 64 * half + 8 * i);
            const float4 w1 = *reinterpret_cast<const float4*>(rows + c * 128 + 64 * half + 8 * i + 4);
            sum[c] = __builtin_fmaf(w0.x, (float)x[0], sum[c]);
            sum[c] = __builtin_fmaf(w0.y, (float)x[1], sum[c]);
            sum[c] = __builtin_fmaf(w0.z, (float)x[2], sum[c]);
            sum[c] = __builtin_fmaf(w0.w, (float)x[3], sum[c]);
            sum[c] = __builtin_fmaf(w1.x, (float)x[4], sum[c]);
            sum[c] = __builtin_fmaf(w1.y, (float)x[5], sum[c]);
            sum[c] = __builtin_fmaf(w1.z, (float)x[6], sum[c]);
            sum[c] = __builtin_fmaf(w1.w, (float)x[7], sum[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NOUT; c++) res[c] = sum[c] + __shfl_xor(sum[c], 32);
}

__device__ __forceinline__ void store16h(_Float16* dst, const float (&f)[16]) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
        half8 h;
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = (_Float16)f[8 * q + i];
        reinterpret_cast<half8*>(dst)[q] = h;
    }
}

// Training forward on the f16 tier (round 6, gf_field_forward_train16): what the backward pass needs of every layer as binary16 [M, width]
// matrices -- the values the MFMAs consumed, i.e. the activations the reference's autocast run keeps -- plus the ReLU masks (same layout as
// the fp32 SaveBufs).
struct SaveBufs16 {
    _Float16 *f3, *ha1, *ha2, *f2, *hs1, *hs2, *geo, *hc1, *sh;   // [M,32] [M,128] [M,128] [M,32] [M,128] [M,128] [M,128] [M,128] [M,16]
    uint16_t *m_ha1, *m_ha2, *m_hs1, *m_hs2, *m_hc1;
};

template <bool RELU>
__device__ __forceinline__ void obw16_save(_Float16* __restrict__ G, uint32_t gbase, uint32_t Mv, int wave, int lane, const floatx16 (&acc)[4], int nt,
                                           uint16_t* __restrict__ mask = nullptr) {
#ifdef GF_PROBE_NO_SAVES      // timing probe only (garbage gradients): what the save stream costs the training forward / the dX chain
    return;
#endif
    const int half = lane >> 5, j = lane & 31;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < nt) {
            if (mask) {
                uint32_t bits = 0;
#pragma unroll
                for (int r = 0; r < 16; r++) bits |= (acc[t][r] > 0.0f ? 1u : 0u) << r;
                mask[(((size_t)(gbase / kPass) * 4 + t) * 4 + wave) * 64 + lane] = (uint16_t)bits;
            }
            if ((uint32_t)(t * 32 + j) < Mv) {
                _Float16* row = G + (size_t)(gbase + (uint32_t)(t * 32 + j)) * 128 + 32 * wave + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float4 v = {acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                    if (RELU) { v.x = relu1(v.x); v.y = relu1(v.y); v.z = relu1(v.z); v.w = relu1(v.w); }
                    *reinterpret_cast<half4*>(row + 8 * q) = half4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                }
            }
        }
    }
}

// The ReLU mask alone (the row itself leaves through rows16_to_global).
__device__ __forceinline__ void obw16_mask(uint16_t* __restrict__ mask, uint32_t gbase, int wave, int lane, const floatx16 (&acc)[4], int nt) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < nt) {
            uint32_t bits = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) bits |= (acc[t][r] > 0.0f ? 1u : 0u) << r;
            mask[(((size_t)(gbase / kPass) * 4 + t) * 4 + wave) * 64 + lane] = (uint16_t)bits;
        }
    }
}

// A layer's 128 x 128 binary16 rows from the LDS activation buffer to their [M,128] matrix (round 6): after the barrier that follows the layer's
// write-back the rows sit in H16 row-major, so 16 consecutive lanes move one whole 256-byte row with 16-byte loads and stores -- a wave
// instruction writes four complete rows.  Until then every lane stored the four 8-byte pieces of ITS sample's row per tile straight from the
// accumulators: 16 store instructions per layer and wave, each touching 32 different rows, and the whole save stream showed up on top of the
// kernel's time (tools/visits/r6s.sh: forward 1.00 -> 0.63 ms, dX chain 1.16 -> 0.85 ms with the stores removed).  Same bits: H16 holds the
// (_Float16) of the same ReLU'd accumulator.  Called right after the barrier, before the next layer's accumulators exist (few live registers).
template <bool WIDE = true>
__device__ __forceinline__ void rows16_to_global(const _Float16* H16, _Float16* __restrict__ G, uint32_t gbase, uint32_t Mv, int tid) {
#ifdef GF_PROBE_NO_SAVES
    return;
#endif
    // one per-lane offset for all eight pieces (they are 16 rows = 4 KiB apart), the chunk's base a workgroup-uniform pointer: two more live
    // registers than the kernel had, not two per piece (k_field_backward16 stands at 256 VGPRs)
    __builtin_amdgcn_sched_barrier(0);       // nothing of the next layer is hoisted above the copy (its operands would be live across it)
    // the matrix pointer stays the kernel argument (scalar registers) and the lane adds ONE 32-bit byte offset: k_field_backward16 stands at 256
    // VGPRs with its scalar registers already spilling into vector lanes -- a 64-bit per-lane address, or one more scalar pair per matrix, and
    // the thread index went to scratch.  Hence the entry points' bound M * 256 B < 4 GiB on this tier.
    char* __restrict__ Gc = reinterpret_cast<char*>(G);
    if constexpr (WIDE) {
        const int row0 = tid >> 4, col = (tid & 15) * 8;
        const _Float16* src = H16 + row0 * kHS16 + col;
        const uint32_t dst = ((gbase + (uint32_t)row0) * 128u + (uint32_t)col) * 2u;
#pragma unroll 1
        for (int i = 0; i < kPass * 16 / kThreads; i++) {
            if ((uint32_t)(row0 + 16 * i) < Mv)
                *reinterpret_cast<float4*>(Gc + (dst + (uint32_t)i * (16u * 256u))) = *reinterpret_cast<const float4*>(src + i * 16 * kHS16);
        }
    } else {      // 8-byte pieces, 32 lanes per row: two transient registers instead of four
        const int row0 = tid >> 5, col = (tid & 31) * 4;
        const _Float16* src = H16 + row0 * kHS16 + col;
        const uint32_t dst = ((gbase + (uint32_t)row0) * 128u + (uint32_t)col) * 2u;
#pragma unroll 1
        for (int i = 0; i < kPass * 32 / kThreads; i++) {
            if ((uint32_t)(row0 + 8 * i) < Mv)
                *reinterpret_cast<float2*>(Gc + (dst + (uint32_t)i * (8u * 256u))) = *reinterpret_cast<const float2*>(src + i * 8 * kHS16);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// TRAIN (dense point lists only): tanh(ambient) is left in s.sdt / s.st [raw] and every layer's activations go to `sv`.
template <bool TRAIN = false>
__device__ __forceinline__ void field_round16(const HeadArgs& a, const Smem& s, uint32_t Mv, int nt, int wave, int lane, const SaveBufs16* sv = nullptr,
                                              uint32_t gbase = 0) {
    const int half = lane >> 5, j = lane & 31;
    const uint32_t sI = (uint32_t)(wave * 32 + j);
    const bool tile_on = wave < nt;
    const bool valid = sI < Mv;
    const uint32_t sC = valid ? sI : (uint32_t)(wave * 32);
    const uint32_t raw = tile_on ? s.d2r[sC] : 0u;
    _Float16* H16 = reinterpret_cast<_Float16*>(s.H);
    _Float16* Hrow = H16 + sI * kHS16;
    const _Float16* Hb = H16 + j * kHS16 + 8 * half;
    _Float16* Hw = H16 + j * kHS16 + 32 * wave + 4 * half;
    const char* Ws = reinterpret_cast<const char*>(a.head_pack16) + (size_t)wave * gf::H16_TOTAL * 1024;   // wave-uniform
    uint32_t lane16 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(lane16));
    const gf::LevelMeta* meta = reinterpret_cast<const gf::LevelMeta*>(s.P + P_META);
    // the 3-D grid features keep their own buffer (LDS is plentiful at f16), so density L1 runs as ONE K = 64 layer after the ambient
    // net and a single accumulator set suffices
    _Float16* F3 = H16 + kPass * kHS16;
    _Float16* Frow = F3 + sI * kFS16;
    const _Float16* Fb = F3 + j * kFS16 + 8 * half;
    // SH(dir) is a property of the ray: one evaluation per pool slot and round (owner lanes) instead of one per sample and tile; the colour
    // layer's SH operand is then a plain 16-byte LDS read
    _Float16* SHT = F3 + kPass * kFS16;
    if (wave * 64 + lane < kPool) {
        const int slot = wave * 64 + lane;
        float sh[16];
        gf::sh4(s.p_dx[slot], s.p_dy[slot], s.p_dz[slot], sh);
        store16h(SHT + slot * 16, sh);
        if constexpr (TRAIN) {
            if ((uint32_t)slot < Mv) store16h(sv->sh + (size_t)(gbase + (uint32_t)slot) * 16, sh);
        }
    }
#ifdef GF_DIAG
    uint32_t dkey = 0xFFFFFFFFu;
#endif
    floatx16 A[4];
    WPipe16 wp;
#pragma unroll
    for (int g = 0; g < kWAhead16; g++) wp.q[g] = load_group(Ws, g, lane16);

    // ---- 3-D grid features -> F3[:, 0:32]
    if (tile_on) {
        const float b2 = 2 * a.bound;
        const float x3[3] = {(s.sx[raw] + a.bound) / b2, (s.sy[raw] + a.bound) / b2, (s.sz[raw] + a.bound) / b2};
        float pf[16];
        gf::encode8<3>(a.pos_table, meta + half * 8, a.gridtype, a.interp, x3, pf);
        store16h(Frow + 16 * half, pf);
        if constexpr (TRAIN) {
            if (valid) store16h(sv->f3 + (size_t)(gbase + sI) * 32 + 16 * half, pf);
        }
#ifdef GF_DIAG
        dkey = valid ? s.dkey[raw] : 0xFFFFFFFFu;
        float c = 0.0f;
        for (int i = 0; i < 16; i++) c += (float)(_Float16)pf[i];
        c += __shfl_xor(c, 32);
        if (a.diag && dkey != 0xFFFFFFFFu && half == 0) a.diag[(size_t)dkey * kDiagWords + 16] = c;
#endif
    }
    GF_STAMP(7);
    __syncthreads();
    GF_STAMP(8);
    // ---- ambient L1 (cond_feat folded into the bias)
    obw_bias<4>(s.P + P_AMBBIAS + wave * 32 + half * 16, A);
    obw16_mfma<gf::H16_AMB1, 2, kFS16>(wp, Ws, lane16, Fb, A, nt);
    GF_STAMP(9);
    GF_STAMP(10);
    obw16_store<true>(Hw, A, nt);          // H is not read by this layer: no barrier before the write-back
    if constexpr (TRAIN) obw16_mask(sv->m_ha1, gbase, wave, lane, A, nt);
    GF_STAMP(11);
    __syncthreads();
    GF_STAMP(12);
    if constexpr (TRAIN) rows16_to_global(H16, sv->ha1, gbase, Mv, wave * 64 + lane);
    // ---- ambient L2
    obw_zero<4>(A);
    obw16_mfma<gf::H16_AMB2, 8>(wp, Ws, lane16, Hb, A, nt);
    GF_STAMP(13);
#ifndef GF_SKINNY_FROM_LDS
    // the three skinny layers from the fp32 accumulators, as in the other two tiers (skinny_partials): relu(ambient L2) and relu(colour L1)
    // are never rounded to f16 nor written to LDS; the rows are 68 floats here
    constexpr int kHF16 = kHS16 / 2;
    float* Hf = reinterpret_cast<float*>(s.H);
    {
        float part[4][2];
        skinny_partials<4, 2>(A, s.P + P_SMALL + gf::HS_AMB3, wave, half, part);
        __syncthreads();
        GF_STAMP(14);
        skinny_publish<4, 2, kHF16>(Hf + j * kHF16 + 16 + 4 * wave, half, part);   // floats 16..31 = halves 32..63: behind the 2-D features (halves 0..31)
    }
    if constexpr (TRAIN) obw16_save<true>(sv->ha2, gbase, Mv, wave, lane, A, nt, sv->m_ha2);
#else
    __syncthreads();
    GF_STAMP(14);
    obw16_store<true>(Hw, A, nt);
#endif
    GF_STAMP(15);
    __syncthreads();
    GF_STAMP(16);
    // ---- ambient L3 + tanh -> 2-D grid features -> H[:, 0:32]
    if (tile_on) {
        float ambient[2];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<2>(Hf + sI * kHF16 + 16, ambient);
#else
        rows_from_lds16<2>(Hrow, s.P + P_SMALL + gf::HS_AMB3, half, ambient);
#endif
        // (tanh(v) + 1) / 2 = 1 / (1 + exp(-2 v))
        const float e2[2] = {__expf(-2.0f * ambient[0]), __expf(-2.0f * ambient[1])};
        const float x2[2] = {1.0f / (1.0f + e2[0]), 1.0f / (1.0f + e2[1])};
        if constexpr (TRAIN) {
            if (valid && half == 0) { s.sdt[raw] = 2.0f * x2[0] - 1.0f; s.st[raw] = 2.0f * x2[1] - 1.0f; }     // tanh(ambient)
        }
        float af[16];
        gf::encode8<2>(a.amb_table, meta + 16 + half * 8, a.gridtype, a.interp, x2, af);
        store16h(Hrow + 16 * half, af);
        if constexpr (TRAIN) {
            if (valid) store16h(sv->f2 + (size_t)(gbase + sI) * 32 + 16 * half, af);
        }
#ifdef GF_DIAG
        float c = 0.0f, cf = 0.0f;
        for (int i = 0; i < 16; i++) { c += (float)(_Float16)af[i]; cf += af[i]; }
        c += __shfl_xor(c, 32);
        cf += __shfl_xor(cf, 32);
        if (a.diag && dkey != 0xFFFFFFFFu) {
            float* rec = a.diag + (size_t)dkey * kDiagWords;
            for (int i = 0; i < 16; i++) rec[26 + 16 * half + i] = af[i];
            if (half == 0) {
                rec[12] = ambient[0]; rec[13] = ambient[1]; rec[17] = c;
                rec[20] = x2[0]; rec[21] = x2[1]; rec[22] = cf; rec[24] = e2[0]; rec[25] = e2[1];
                reinterpret_cast<uint32_t*>(rec)[23] = __builtin_amdgcn_s_getreg(63492 /* HW_REG_HW_ID, 32 bits */);
                reinterpret_cast<uint32_t*>(rec)[58] = __builtin_amdgcn_s_getreg(63508 /* HW_REG_XCC_ID */);
            }
        }
#endif
    }
    GF_STAMP(17);
    __syncthreads();
    GF_STAMP(18);
    // ---- density L1: [3-D features (F3) | 2-D features (H)]
    obw_zero<4>(A);
    obw16_mfma<gf::H16_SIG1A, 2, kFS16>(wp, Ws, lane16, Fb, A, nt);
    obw16_mfma<gf::H16_SIG1B, 2>(wp, Ws, lane16, Hb, A, nt);
    GF_STAMP(19);
    __syncthreads();
    GF_STAMP(20);
    obw16_store<true>(Hw, A, nt);
    if constexpr (TRAIN) obw16_mask(sv->m_hs1, gbase, wave, lane, A, nt);
    GF_STAMP(21);
    __syncthreads();
    GF_STAMP(22);
    if constexpr (TRAIN) rows16_to_global(H16, sv->hs1, gbase, Mv, wave * 64 + lane);
    // ---- density L2
    obw_zero<4>(A);
    obw16_mfma<gf::H16_SIG2, 8>(wp, Ws, lane16, Hb, A, nt);
    GF_STAMP(23);
    __syncthreads();
    GF_STAMP(24);
    obw16_store<true>(Hw, A, nt);
#ifndef GF_SKINNY_FROM_LDS
    {
        float part[4][1];
        skinny_partials<4, 1>(A, s.P + P_SMALL + gf::HS_SIGROW, wave, half, part);
        skinny_publish<4, 1, kHF16>(Hf + j * kHF16 + 64 + wave, half, part);       // the row's 16 pad bytes (halves 128..135)
    }
#endif
    if constexpr (TRAIN) obw16_mask(sv->m_hs2, gbase, wave, lane, A, nt);
    GF_STAMP(25);
    __syncthreads();
    GF_STAMP(26);
    if constexpr (TRAIN) rows16_to_global(H16, sv->hs2, gbase, Mv, wave * 64 + lane);
    // ---- density L3: row 0 on the VALU, rows 1..128 = geometry feature
    float sigma = 0.0f;
    if (tile_on) {
        float h0[1];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<1>(Hf + sI * kHF16 + 64, h0);
#else
        rows_from_lds16<1>(Hrow, s.P + P_SMALL + gf::HS_SIGROW, half, h0);
#endif
        sigma = expf(h0[0]);
#ifdef GF_DIAG
        if (a.diag && dkey != 0xFFFFFFFFu && half == 0) a.diag[(size_t)dkey * kDiagWords + 14] = h0[0];
#endif
    }
    obw_zero<4>(A);
    obw16_mfma<gf::H16_SIG3, 8>(wp, Ws, lane16, Hb, A, nt);
    GF_STAMP(27);
    __syncthreads();
    GF_STAMP(28);
    obw16_store<false>(Hw, A, nt);
    GF_STAMP(29);
    __syncthreads();
    GF_STAMP(30);
    if constexpr (TRAIN) rows16_to_global(H16, sv->geo, gbase, Mv, wave * 64 + lane);
    // ---- colour L1: [SH(dir) 16 | geometry 128 | identity code -> bias]
    obw_bias<4>(s.P + P_SMALL + gf::HS_COLBIAS + wave * 32 + half * 16, A);
    {
        half8 shb[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t d = (uint32_t)(t * 32 + j);
            d = d < Mv ? d : Mv - 1u;
            const uint32_t slot = s.rrank[d];
            shb[t] = *reinterpret_cast<const half8*>(SHT + slot * 16 + 8 * half);
        }
        const half8 w8 = wpipe16_take<gf::H16_COL1S>(wp);
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (t < nt) A[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w8, shb[t], A[t], 0, 0, 0);
        wpipe16_refill<gf::H16_COL1S>(wp, Ws, lane16);
        __builtin_amdgcn_sched_barrier(0);
    }
    obw16_mfma<gf::H16_COL1G, 8>(wp, Ws, lane16, Hb, A, nt);
    GF_STAMP(31);
#ifndef GF_SKINNY_FROM_LDS
    {
        float part[4][3];
        skinny_partials<4, 3>(A, s.P + P_SMALL + gf::HS_COL2, wave, half, part);
        __syncthreads();
        GF_STAMP(32);
        skinny_publish<4, 3, kHF16>(Hf + j * kHF16 + 4 * wave, half, part);
    }
    if constexpr (TRAIN) obw16_save<true>(sv->hc1, gbase, Mv, wave, lane, A, nt, sv->m_hc1);
#else
    __syncthreads();
    GF_STAMP(32);
    obw16_store<true>(Hw, A, nt);
#endif
    GF_STAMP(33);
    __syncthreads();
    GF_STAMP(34);
    // ---- colour L2 + sigmoid
    if (tile_on) {
        float c[3];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<3>(Hf + sI * kHF16, c);
#else
        rows_from_lds16<3>(Hrow, s.P + P_SMALL + gf::HS_COL2, half, c);
#endif
        if (valid && half == 0) {
            s.sx[raw] = sigma;
            s.sy[raw] = 1.0f / (1.0f + __expf(-c[0]));
            s.sz[raw] = 1.0f / (1.0f + __expf(-c[1]));
            s.ob[raw] = 1.0f / (1.0f + __expf(-c[2]));
        }
    }
    GF_STAMP(35);
    __syncthreads();
    GF_STAMP(36);
}


// ---------------------------------------------------------------------------------------------------- split path (fp32 values on the f16 matrix pipe)
// precision = 2.  The fp32 kernel is bound by the f32 matrix rate (64 cycles per 32x32x2 MFMA: 80 K cycles of pipe per wave and round); the
// f16 instruction moves eight times the k-depth in half the cycles.  Here every fp32 value -- weight or activation -- is carried as
//   v = hi + lo' * 2^-11,   hi = half(v) (round to nearest),   lo' = half((v - hi) * 2^11)
// which represents v to 2^-24 relative (the scaled lo' keeps the residual of every |v| >= 2^-14 in the f16 normal range; below that hi is an f16
// denormal with 2^-25 absolute resolution and lo' refines it further -- the matrix pipe and v_cvt_f16_f32 honour f16 denormals on gfx950,
// tools/mfma_denorm_probe.hip), products of two halves are exact in the fp32 accumulators, and
//   w * x = hi_w hi_x + (lo'_w hi_x + hi_w lo'_x) 2^-11 + O(2^-22)
// costs three v_mfma_f32_32x32x16_f16 per 16 input features and tile (acc1 for the first term, acc2 for the two cross terms): 468 MFMAs of 32
// cycles per wave and round instead of 1 248 of 64.  Errors against the fp32 kernel are of the size of its own rounding (different
// summation order, 2^-24 per product): the strict tolerance of BASELINE.md section 4 (max|d rgb| <= 1e-4) holds with the same margin
// (tests/test_gpu_render.py, every strict test runs on this tier too).  NOT fp32 bit patterns: the default stays precision 0.
// Values beyond the f16 range: a weight is refused at pack time (gf_head_pack_split); an activation saturates its hi term at 65504 and
// overflows lo' -> inf / NaN in the frame (visible, never silent) -- hidden activations of these networks are O(1..100).
// Layout: the activation buffer holds rows of [128 x hi | 128 x lo' | 8 pad] halves = 528 bytes, the fp32 rows' size and bank pattern.
constexpr int kHSS = 264;             // halves per split activation row
static_assert(kHSS * 2 == kHS * 4, "split rows are exactly the fp32 rows");
struct WPipeS { float4 q[2][2]; };    // two groups ahead, [hi | lo'] each

__device__ __forceinline__ void load_group_s(float4 (&dst)[2], const char* __restrict__ Ws, uint32_t g, uint32_t lane32) {
    const float4* p = reinterpret_cast<const float4*>(Ws + (size_t)g * 2048 + (size_t)lane32);
    dst[0] = p[0];
    dst[1] = p[1];
}
template <int G>
__device__ __forceinline__ void wpipes_refill(WPipeS& wp, const char* __restrict__ Ws, uint32_t lane32) {
    if constexpr (G + 2 < (int)gf::SP_TOTAL) load_group_s(wp.q[G % 2], Ws, G + 2, lane32);
}

// v -> (hi, lo'): see the header comment.  RELU folds max(v, 0) and the saturation into one v_med3_f32.
template <bool RELU>
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
    const float c = __builtin_amdgcn_fmed3f(v, RELU ? 0.0f : -65504.0f, 65504.0f);
    hi = (_Float16)c;                                               // may be an f16 denormal: v_cvt_f16_f32 produces them and the MFMA honours them
    // (c - hi) * 2^11, exactly (the difference has at most 13 significant bits), as ONE fused op on the f16 register: v_fma_mix_f32
    lo = (_Float16)__builtin_fmaf((float)hi, -gf::kSplitScale, c * gf::kSplitScale);   // (tools/mfma_denorm_probe.hip, profiles/round3/mfma_denorm_probe.txt)
}
// Two values at once, the same arithmetic: ONE v_cvt_pk_f16_f32 makes both hi halves, and the two fused ops read them out of that packed
// register (op_sel).  Written value by value, the compiler converts every hi twice -- once alone for the fused op, once more inside the
// packing conversion: 5.5 instead of 4.5 VALU ops per written value, in segments that are nothing but these ops (the opaque move between
// the conversion and its uses is what keeps the packed register the only copy).
template <bool RELU>
__device__ __forceinline__ void split_f16x2(float v0, float v1, half2v& hi, half2v& lo) {
    const float c0 = __builtin_amdgcn_fmed3f(v0, RELU ? 0.0f : -65504.0f, 65504.0f);
    const float c1 = __builtin_amdgcn_fmed3f(v1, RELU ? 0.0f : -65504.0f, 65504.0f);
    half2v hp = {(_Float16)c0, (_Float16)c1};
    uint32_t bits = __builtin_bit_cast(uint32_t, hp);
    asm("" : "+v"(bits));
    hp = __builtin_bit_cast(half2v, bits);
    hi = hp;
    lo[0] = (_Float16)__builtin_fmaf((float)hp[0], -gf::kSplitScale, c0 * gf::kSplitScale);
    lo[1] = (_Float16)__builtin_fmaf((float)hp[1], -gf::kSplitScale, c1 * gf::kSplitScale);
}

// U groups (K = 16 each) of this wave's output block over NT tiles -- NT is a COMPILE-TIME count (4, or 2 for the thin rounds of phase 1): with
// a run-time tile count every tile's MFMAs and loads sat behind their own branch, the LDS reads were serialised by waits and a K = 128
// layer took 5 000 cycles of a wave where its 96 MFMAs need 3 072 (profiles/round3/r3r_*).  A round with 3 (or 1) tiles runs the 4- (2-)
// tile code: the surplus tile multiplies whatever the activation rows beyond the round's samples hold, and nothing reads its results
// (columns of an MFMA are independent) -- 2.7 % of all tiles.  Hb = &H[lane & 31][8 * (lane >> 5)] (hi part; lo' 128 halves on).
// B operands run ONE tile ahead of the MFMAs that consume them, across group boundaries, in two register sets that alternate with the tile
// index (the accumulators of the two product terms already take 128 of the 256 registers a lane has at two workgroups per CU).
__device__ __forceinline__ void bset_load(half8& h, half8& l, const _Float16* p) {
    h = *reinterpret_cast<const half8*>(p);
    l = *reinterpret_cast<const half8*>(p + 128);
}
// one tile of one group: MFMAs from set (bh, bl) while the other set (oh, ol) receives the next tile's operands (behind the last tile of a
// group -- NT is even, so that tile reads set 1 -- set 0 takes tile 0 of the next group).  ZC1 / ZC2: first group of a layer, the accumulator
// holds nothing yet: a literal zero as C operand instead of a register clear.
template <int NT, int T, int u, int U, bool ZC1, bool ZC2>
__device__ __forceinline__ void obws_tile(const half8& wh, const half8& wl, const _Float16* Hb, floatx16& a1, floatx16& a2, const half8& bh,
                                          const half8& bl, half8& oh, half8& ol) {
    if constexpr (T + 1 < NT) bset_load(oh, ol, Hb + (T + 1) * 32 * kHSS + 16 * u);
    else if constexpr (u + 1 < U) bset_load(oh, ol, Hb + 16 * (u + 1));
    __builtin_amdgcn_sched_barrier(0);     // the next tile's two LDS reads stay IN FRONT of this tile's three MFMAs (96 cycles of issue to land in);
                                           // left alone the scheduler sinks them to just before their use and every tile waits for LDS
    const floatx16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // acc2's two MFMAs are separated by acc1's: no MFMA waits for the result of the one issued just before it
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, ZC2 ? zero : a2, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, ZC1 ? zero : a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, a2, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}
template <int NT, int G0, int U, int u, bool FIRST, bool BIAS>
__device__ __forceinline__ void obws_step(WPipeS& wp, const char* __restrict__ Ws, uint32_t lane32, const _Float16* Hb, floatx16 (&a1)[NT],
                                          floatx16 (&a2)[NT], half8& b0h, half8& b0l, half8& b1h, half8& b1l) {
    static_assert(NT == 2 || NT == 4, "an even tile count: the two B-operand sets alternate with the tile index");
    if constexpr (u < U) {
        const half8 wh = __builtin_bit_cast(half8, wp.q[(G0 + u) % 2][0]), wl = __builtin_bit_cast(half8, wp.q[(G0 + u) % 2][1]);
        constexpr bool Z1 = FIRST && u == 0 && !BIAS, Z2 = FIRST && u == 0;
        obws_tile<NT, 0, u, U, Z1, Z2>(wh, wl, Hb, a1[0], a2[0], b0h, b0l, b1h, b1l);
        obws_tile<NT, 1, u, U, Z1, Z2>(wh, wl, Hb, a1[1], a2[1], b1h, b1l, b0h, b0l);
        if constexpr (NT == 4) {
            obws_tile<NT, 2, u, U, Z1, Z2>(wh, wl, Hb, a1[2], a2[2], b0h, b0l, b1h, b1l);
            obws_tile<NT, 3, u, U, Z1, Z2>(wh, wl, Hb, a1[3], a2[3], b1h, b1l, b0h, b0l);
        }
        wpipes_refill<G0 + u>(wp, Ws, lane32);
        __builtin_amdgcn_sched_barrier(0);
        obws_step<NT, G0, U, u + 1, FIRST, BIAS>(wp, Ws, lane32, Hb, a1, a2, b0h, b0l, b1h, b1l);
    }
}
// FIRST: the layer starts with this call (acc2 from zero; acc1 from zero, or from the bias the caller loaded when BIAS)
template <int NT, int G0, int U, bool FIRST, bool BIAS = false>
__device__ __forceinline__ void obws_mfma(WPipeS& wp, const char* __restrict__ Ws, uint32_t lane32, const _Float16* Hb, floatx16 (&a1)[NT],
                                          floatx16 (&a2)[NT]) {
    half8 b0h, b0l, b1h, b1l;
    bset_load(b0h, b0l, Hb);
    b1h = b0h; b1l = b0l;
    obws_step<NT, G0, U, 0, FIRST, BIAS>(wp, Ws, lane32, Hb, a1, a2, b0h, b0l, b1h, b1l);
}

// accumulators -> split activations.  Hw = &H[lane & 31][32 * wave + 4 * (lane >> 5)]: registers 4q..4q+3 -> four consecutive halves, twice.
template <int NT, bool RELU>
__device__ __forceinline__ void obws_store(_Float16* Hw, const floatx16 (&a1)[NT], const floatx16 (&a2)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            half4 h, l;
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                half2v hh, ll;
                split_f16x2<RELU>(__builtin_fmaf(a2[t][4 * q + i], gf::kSplitInv, a1[t][4 * q + i]),
                                  __builtin_fmaf(a2[t][4 * q + i + 1], gf::kSplitInv, a1[t][4 * q + i + 1]), hh, ll);
                h[i] = hh[0]; h[i + 1] = hh[1]; l[i] = ll[0]; l[i + 1] = ll[1];
            }
            *reinterpret_cast<half4*>(Hw + t * 32 * kHSS + 8 * q) = h;
            *reinterpret_cast<half4*>(Hw + t * 32 * kHSS + 8 * q + 128) = l;
        }
    }
}

template <int NOUT>
__device__ __forceinline__ void rows_from_lds_split(const _Float16* Hrow, const float* rows, int half, float (&res)[NOUT]) {
    float sum[NOUT];
#pragma unroll
    for (int c = 0; c < NOUT; c++) sum[c] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const half8 xh = *reinterpret_cast<const half8*>(Hrow + 64 * half + 8 * i);
        const half8 xl = *reinterpret_cast<const half8*>(Hrow + 64 * half + 8 * i + 128);
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = __builtin_fmaf((float)xl[k], gf::kSplitInv, (float)xh[k]);
#pragma unroll
        for (int c = 0; c < NOUT; c++) {
            const float4 w0 = *reinterpret_cast<const float4*>(rows + c * 128 + 64 * half + 8 * i);
            const float4 w1 = *reinterpret_cast<const float4*>(rows + c * 128 + 64 * half + 8 * i + 4);
            sum[c] = __builtin_fmaf(w0.x, x[0], sum[c]);
            sum[c] = __builtin_fmaf(w0.y, x[1], sum[c]);
            sum[c] = __builtin_fmaf(w0.z, x[2], sum[c]);
            sum[c] = __builtin_fmaf(w0.w, x[3], sum[c]);
            sum[c] = __builtin_fmaf(w1.x, x[4], sum[c]);
            sum[c] = __builtin_fmaf(w1.y, x[5], sum[c]);
            sum[c] = __builtin_fmaf(w1.z, x[6], sum[c]);
            sum[c] = __builtin_fmaf(w1.w, x[7], sum[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NOUT; c++) res[c] = sum[c] + __shfl_xor(sum[c], 32);
}

// 16 fp32 features of a lane pair's sample -> split halves at dst (hi) and dst + 128 (lo')
__device__ __forceinline__ void store16s(_Float16* dst, const float (&f)[16]) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
        half8 h, l;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            half2v hh, ll;
            split_f16x2<false>(f[8 * q + i], f[8 * q + i + 1], hh, ll);
            h[i] = hh[0]; h[i + 1] = hh[1]; l[i] = ll[0]; l[i + 1] = ll[1];
        }
        reinterpret_cast<half8*>(dst)[q] = h;
        reinterpret_cast<half8*>(dst + 128)[q] = l;
    }
}

// NT: compile-time MFMA tile count of the round (4, or 2 when the round holds <= 64 samples); nt = ceil(Mv / 32) <= NT gates the per-sample work.
// skinny_partials for the split tier: the activation is the fp32 value the write-back would have split, relu(a1 + a2 * 2^-11), taken from the
// accumulators (no f16 round trip at all for these two layers: ambient L2 -> L3 and colour L1 -> L2 lose their split write-backs, the
// density row its 32 row reads).  Explicit builtins only, like the rest of the split round.
template <int NT, int NOUT>
__device__ __forceinline__ void skinny_partials_split(const floatx16 (&a1)[NT], const floatx16 (&a2)[NT], const float* rows, int wave, int half,
                                                      float (&part)[NT][NOUT]) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int c = 0; c < NOUT; c++) part[t][c] = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float4 w4[NOUT];
#pragma unroll
        for (int c = 0; c < NOUT; c++) w4[c] = *reinterpret_cast<const float4*>(rows + c * 128 + 32 * wave + 8 * q + 4 * half);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float x[4];
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = relu1(__builtin_fmaf(a2[t][4 * q + i], gf::kSplitInv, a1[t][4 * q + i]));
#pragma unroll
            for (int c = 0; c < NOUT; c++) {
                part[t][c] = __builtin_fmaf(w4[c].x, x[0], part[t][c]);
                part[t][c] = __builtin_fmaf(w4[c].y, x[1], part[t][c]);
                part[t][c] = __builtin_fmaf(w4[c].z, x[2], part[t][c]);
                part[t][c] = __builtin_fmaf(w4[c].w, x[3], part[t][c]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int c = 0; c < NOUT; c++) part[t][c] += __shfl_xor(part[t][c], 32);
}

template <int NT>
__device__ __forceinline__ void field_round_split(const HeadArgs& a, const Smem& s, uint32_t Mv, int nt, int wave, int lane) {
    const int half = lane >> 5, j = lane & 31;
    const uint32_t sI = (uint32_t)(wave * 32 + j);
    const bool tile_on = wave < nt;
    const bool valid = sI < Mv;
    const uint32_t sC = valid ? sI : (uint32_t)(wave * 32);
    const uint32_t raw = tile_on ? s.d2r[sC] : 0u;
    _Float16* H16 = reinterpret_cast<_Float16*>(s.H);
    _Float16* Hrow = H16 + sI * kHSS;
    const _Float16* Hb = H16 + j * kHSS + 8 * half;
    _Float16* Hw = H16 + j * kHSS + 32 * wave + 4 * half;
    const char* Ws = reinterpret_cast<const char*>(a.head_pack_split) + (size_t)wave * gf::SP_TOTAL * 2048;   // wave-uniform
    uint32_t lane32 = (uint32_t)lane * 32u;
    asm volatile("" : "+v"(lane32));
    const gf::LevelMeta* meta = reinterpret_cast<const gf::LevelMeta*>(s.P + P_META);
    floatx16 A1[NT], A2[NT];
    WPipeS wp;
    load_group_s(wp.q[0], Ws, 0, lane32);
    load_group_s(wp.q[1], Ws, 1, lane32);

    // ---- 3-D grid features -> H[:, 0:32]; the lane pair keeps them (fp32) for density L1
    float pf[16];
#pragma unroll
    for (int i = 0; i < 16; i++) pf[i] = 0.0f;
    if (tile_on) {
        const float b2 = 2 * a.bound;
        const float x3[3] = {(s.sx[raw] + a.bound) / b2, (s.sy[raw] + a.bound) / b2, (s.sz[raw] + a.bound) / b2};
        gf::encode8<3>(a.pos_table, meta + half * 8, a.gridtype, a.interp, x3, pf);
        store16s(Hrow + 16 * half, pf);
    }
    GF_STAMP(7);
    __syncthreads();
    GF_STAMP(8);
    // ---- ambient L1 (cond_feat folded into the bias)
    obw_bias_n<NT>(s.P + P_AMBBIAS + wave * 32 + half * 16, A1);
    obws_mfma<NT, gf::SP_AMB1, 2, true, true>(wp, Ws, lane32, Hb, A1, A2);
    GF_STAMP(9);
    __syncthreads();
    GF_STAMP(10);
    obws_store<NT, true>(Hw, A1, A2);
    GF_STAMP(11);
    __syncthreads();
    GF_STAMP(12);
    // ---- ambient L2
    obws_mfma<NT, gf::SP_AMB2, 8, true>(wp, Ws, lane32, Hb, A1, A2);
    GF_STAMP(13);
#ifndef GF_SKINNY_FROM_LDS
    float* Hf = reinterpret_cast<float*>(s.H);     // the same rows as 132 floats (kHSS halves == kHS floats)
    {
        float part[NT][2];
        skinny_partials_split<NT, 2>(A1, A2, s.P + P_SMALL + gf::HS_AMB3, wave, half, part);
        __syncthreads();
        GF_STAMP(14);
        skinny_publish<NT, 2>(Hf + j * kHS + 32 + 4 * wave, half, part);   // floats 32..47 = halves 64..95: between the feature columns and their lo' parts
    }
#else
    __syncthreads();
    GF_STAMP(14);
    obws_store<NT, true>(Hw, A1, A2);
#endif
    GF_STAMP(15);
    __syncthreads();
    GF_STAMP(16);
    // ---- ambient L3 + tanh -> 2-D grid features -> H[:, 32:64]; the kept 3-D features -> H[:, 0:32]
    if (tile_on) {
        float ambient[2];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<2>(Hf + sI * kHS + 32, ambient);
#else
        rows_from_lds_split<2>(Hrow, s.P + P_SMALL + gf::HS_AMB3, half, ambient);
#endif
        const float th[2] = {tanhf(ambient[0]), tanhf(ambient[1])};
        const float x2[2] = {(th[0] + 1.0f) / 2.0f, (th[1] + 1.0f) / 2.0f};
        float af[16];
        gf::encode8<2>(a.amb_table, meta + 16 + half * 8, a.gridtype, a.interp, x2, af);
        store16s(Hrow + 16 * half, pf);
        store16s(Hrow + 32 + 16 * half, af);
    }
    GF_STAMP(17);
    __syncthreads();
    GF_STAMP(18);
    // ---- density L1: K = 64
    obws_mfma<NT, gf::SP_SIG1, 4, true>(wp, Ws, lane32, Hb, A1, A2);
    GF_STAMP(19);
    __syncthreads();
    GF_STAMP(20);
    obws_store<NT, true>(Hw, A1, A2);
    GF_STAMP(21);
    __syncthreads();
    GF_STAMP(22);
    // ---- density L2
    obws_mfma<NT, gf::SP_SIG2, 8, true>(wp, Ws, lane32, Hb, A1, A2);
    GF_STAMP(23);
    __syncthreads();
    GF_STAMP(24);
    obws_store<NT, true>(Hw, A1, A2);
#ifndef GF_SKINNY_FROM_LDS
    {
        float part[NT][1];
        skinny_partials_split<NT, 1>(A1, A2, s.P + P_SMALL + gf::HS_SIGROW, wave, half, part);
        skinny_publish<NT, 1>(Hf + j * kHS + 128 + wave, half, part);       // the row's 16 pad bytes
    }
#endif
    GF_STAMP(25);
    __syncthreads();
    GF_STAMP(26);
    // ---- density L3: row 0 on the VALU, rows 1..128 = geometry feature
    if (tile_on) {
        float h0[1];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<1>(Hf + sI * kHS + 128, h0);
#else
        rows_from_lds_split<1>(Hrow, s.P + P_SMALL + gf::HS_SIGROW, half, h0);
#endif
        if (valid && half == 0) s.sx[raw] = expf(h0[0]);     // the position slots were consumed before the first barrier of this function
    }
    obws_mfma<NT, gf::SP_SIG3, 8, true>(wp, Ws, lane32, Hb, A1, A2);
    GF_STAMP(27);
    __syncthreads();
    GF_STAMP(28);
    obws_store<NT, false>(Hw, A1, A2);
    GF_STAMP(29);
    __syncthreads();
    GF_STAMP(30);
    // ---- colour L1: [SH(dir) 16 | geometry 128 | identity code -> bias]
    obw_bias_n<NT>(s.P + P_SMALL + gf::HS_COLBIAS + wave * 32 + half * 16, A1);
    {
        const half8 wh = __builtin_bit_cast(half8, wp.q[gf::SP_COL1S % 2][0]), wl = __builtin_bit_cast(half8, wp.q[gf::SP_COL1S % 2][1]);
        const floatx16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NT; t++) {
            uint32_t d = (uint32_t)(t * 32 + j);
            d = d < Mv ? d : Mv - 1u;
            const uint32_t slot = s.rrank[d];
            float sh[16];
            gf::sh4(s.p_dx[slot], s.p_dy[slot], s.p_dz[slot], sh);
            half8 sh_h, sh_l;
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                // both candidates pass through an opaque move first: otherwise the select of two array elements becomes ONE element at a
                // lane-dependent index, and the 16-entry array moves to scratch
                float v0 = sh[i], v1 = sh[8 + i], u0 = sh[i + 1], u1 = sh[9 + i];
                asm("" : "+v"(v0));
                asm("" : "+v"(v1));
                asm("" : "+v"(u0));
                asm("" : "+v"(u1));
                half2v hh, ll;
                split_f16x2<false>(half ? v1 : v0, half ? u1 : u0, hh, ll);
                sh_h[i] = hh[0]; sh_h[i + 1] = hh[1]; sh_l[i] = ll[0]; sh_l[i + 1] = ll[1];
            }
            A2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, sh_h, zero, 0, 0, 0);
            A1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, sh_h, A1[t], 0, 0, 0);
            A2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, sh_l, A2[t], 0, 0, 0);
        }
        wpipes_refill<gf::SP_COL1S>(wp, Ws, lane32);
        __builtin_amdgcn_sched_barrier(0);
    }
    obws_mfma<NT, gf::SP_COL1G, 8, false>(wp, Ws, lane32, Hb, A1, A2);
    GF_STAMP(31);
#ifndef GF_SKINNY_FROM_LDS
    {
        float part[NT][3];
        skinny_partials_split<NT, 3>(A1, A2, s.P + P_SMALL + gf::HS_COL2, wave, half, part);
        __syncthreads();
        GF_STAMP(32);
        skinny_publish<NT, 3>(Hf + j * kHS + 4 * wave, half, part);
    }
#else
    __syncthreads();
    GF_STAMP(32);
    obws_store<NT, true>(Hw, A1, A2);
#endif
    GF_STAMP(33);
    __syncthreads();
    GF_STAMP(34);
    // ---- colour L2 + sigmoid
    if (tile_on) {
        float c[3];
#ifndef GF_SKINNY_FROM_LDS
        skinny_collect<3>(Hf + sI * kHS, c);
#else
        rows_from_lds_split<3>(Hrow, s.P + P_SMALL + gf::HS_COL2, half, c);
#endif
        if (valid && half == 0) {
            s.sy[raw] = 1.0f / (1.0f + __expf(-c[0]));
            s.sz[raw] = 1.0f / (1.0f + __expf(-c[1]));
            s.ob[raw] = 1.0f / (1.0f + __expf(-c[2]));
        }
    }
    GF_STAMP(35);
    __syncthreads();
    GF_STAMP(36);
}

// MODE: 0 = fp32 (strict, bit-reproducible fp32 arithmetic), 1 = fast (f16 operands), 2 = split (fp32 values as two-term f16 splits)
template <int MODE>
__global__ void __launch_bounds__(kThreads, 2) k_head_phase(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: per-wave weight-stream bases stay in SGPRs
    const bool owner = tid < kPool;
#ifdef GF_TRACE
    const unsigned long long span_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long span_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- phase set-up (uniform) ----
    uint32_t budget, limit;
    const uint32_t qhead = a.phase ? gf::kCtrlQHead1 : gf::kCtrlQHead0;
    if (a.phase == 0) {
        budget = a.max_steps;
        limit = a.ctrl[gf::kCtrlNHit];
    } else {
        // the histogram of phase 0 (complete: this launch follows it in stream order) is staged in LDS by all lanes -- one parallel read instead
        // of max_steps dependent global loads on lane 0 -- in the activation buffer, which nothing uses yet
        uint32_t* hs = reinterpret_cast<uint32_t*>(s.H);
        for (uint32_t i = tid; i <= a.max_steps; i += kThreads) hs[i] = a.ctrl[gf::kCtrlHist + i];
        __syncthreads();
        if (tid == 0) {
            const uint32_t B = replay_budget(hs, a.N, a.max_steps);
            s.misc[8] = B;
            if (blockIdx.x == 0) a.ctrl[gf::kCtrlBudget] = B;
        }
        __syncthreads();
        const uint32_t B = s.misc[8];
        budget = B > a.max_steps ? B - a.max_steps : 0u;
        limit = a.ctrl[gf::kCtrlNSurv];
    }
    if (budget == 0 || limit == 0) return;
    // Rays per pool: a full pool (128 rays, 1 sample per ray and round) when there is plenty of work, but when the queue is short
    // (phase 1: a few thousand survivors that each need B - max_steps more samples) spread it over the whole grid and give
    // every ray up to 8 samples per round instead of walking 128 rays through B - max_steps rounds on a handful of CUs.
    // (Round 5 tried a floor of 128 / budget rays when fewer than 8 samples remain of the budget -- phase 1 of the default scene has 7, so
    // sixteen rays can never use the eighth slot each -- : byte-identical frames, 309 instead of 348 phase-1 workgroups, +0.1 % / +0.3 % fps
    // (fp32 / split, noise level) and the kernel ALONE 1.5 % slower, because the fuller rounds are longer and phase 1 is one round deep
    // whatever its width (profiles/round5/r5g_phase1_pool_floor_same_box_ab.txt).  Not kept.)
    uint32_t pool_cap = (limit + gridDim.x - 1) / gridDim.x;
    pool_cap = pool_cap < 16u ? 16u : (pool_cap > (uint32_t)kPool ? (uint32_t)kPool : pool_cap);
#ifdef GF_DIAG
    if (a.pool_cap_override) pool_cap = a.pool_cap_override;
#endif
    if ((uint32_t)blockIdx.x * pool_cap >= limit) return;  // not even one refill's worth of work for this workgroup
    // the three loop-invariant scalars live in LDS from here on (s.misc[13..15]): with ~100 SGPRs already parked in VGPR lanes the allocator
    // put them into scratch instead, and a launch that touches scratch at all pays for it (NOTES.md 4.7)
#ifdef GF_DIAG
    if (a.poison) {   // any read of LDS this workgroup has not written itself now returns NaN (fp32 and f16 views alike)
        __syncthreads();
        uint32_t* w = reinterpret_cast<uint32_t*>(smem_raw);
        for (int i = tid; i < kSmemBytes / 4; i += kThreads) w[i] = a.poison;
        __syncthreads();
    }
    uint32_t dg_round = 0;
#endif
    if (tid == 0) { s.misc[13] = budget; s.misc[14] = limit; s.misc[15] = pool_cap; }   // (after the diagnostic poison fill, which covers all of LDS)

    for (int i = tid; i < (int)gf::HS_TOTAL; i += kThreads) s.P[P_SMALL + i] = a.head_pack[gf::HP_SMALL + i];
    if (tid < 128) s.P[P_AMBBIAS + tid] = a.amb_bias[tid];
    if (tid < 32) {
        const uint32_t g = tid >> 4, l = tid & 15;
        gf::LevelMeta* m = reinterpret_cast<gf::LevelMeta*>(s.P + P_META) + tid;
        *m = g ? gf::make_level_meta<2>(a.lv2.scale[l], a.lv2.resolution[l], a.amb_offsets, l, a.gridtype)
               : gf::make_level_meta<3>(a.lv3.scale[l], a.lv3.resolution[l], a.pos_offsets, l, a.gridtype);
    }
    if (tid < kHistBins) s.hist[tid] = 0;
    if (tid == 0) s.misc[9] = 0;     // samples composited by this workgroup (statistics)

    // pooled ray of this lane (owners only): everything the marcher and the compositor carry between rounds
    int ray = -1;
    // (the direction lives in LDS by pool slot, p_dx/p_dy/p_dz, where the SH evaluation reads it too; the origin is the camera's in pose mode
    // and is re-read per round otherwise: six registers less across the MFMA segments, where all three kernels sit at the 256-register limit)
    float r_t = 0, r_far = 0;
    gf::RayAcc acc = {0, 0, 0, 0, 0, 0};
    uint32_t r_done = 0;
    // statistics: s.misc[10..12] = samples, rounds, tiles of this workgroup (thread 0 adds per round; LDS, not three registers)
    if (tid == 0) { s.misc[10] = 0; s.misc[11] = 0; s.misc[12] = 0; }
    bool queue_open = true;         // uniform
    // (Round 3 tried a STATIC block of the queue as every workgroup's first refill -- at launch all 1 024 owner waves hit one queue-head word,
    // and L2 retires same-address atomics at ~10 ns each.  Alone on the GPU the launch got 1-2 % shorter; with four frames in flight the
    // frame rate DROPPED 1 % (fp32: 727 vs 734 fps, profiles/round3/r3w_static_fill_ab.txt): a workgroup that is scheduled late -- its CU
    // still busy with another frame -- then sits on 128 rays nobody else can take.  The dynamic queue stays.)
#ifdef GF_TRACE
    uint32_t tr_round = 0;
    unsigned long long span_t1 = span_t0, span_r1 = span_r0;   // end of the last round
#endif

#ifndef GF_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);   // everything outside the MFMA segments runs at high priority (see field_round)
#endif
    for (;;) {
        __syncthreads();  // previous round fully retired (staging, H)
        GF_STAMP(0);
#ifdef GF_DIAG
        if (a.poison_round) {   // nothing in the activation buffer is live between rounds
            uint32_t* w = reinterpret_cast<uint32_t*>(s.H);
            for (int i = tid; i < kPass * kHS; i += kThreads) w[i] = a.poison_round;
            __syncthreads();
        }
        dg_round++;
#endif
        // ------------------------------------------------------------------ refill empty pool slots from the queue
        if (queue_open && wave < 2) {  // wave-uniform branch
            const bool want = ray < 0 && (uint32_t)tid < s.misc[15];
            const unsigned long long m = __ballot(want);
            const uint32_t nw = (uint32_t)__popcll(m);
            uint32_t base = 0;
            if (lane == 0 && nw) base = atomicAdd(&a.ctrl[qhead], nw);
            base = __shfl(base, 0);
            if (want) {
                const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (idx < s.misc[14]) {
                    ray = a.queue[idx];
                    const float* d = a.rays_d + (size_t)ray * 3;
                    r_t = a.rays_t[ray];
                    r_far = a.far_occ[ray];
                    if (a.phase == 0) {
                        acc.weight_sum = 0.0f; acc.depth = 0.0f; acc.r = 0.0f; acc.g = 0.0f; acc.b = 0.0f;
                    } else {
                        acc.weight_sum = a.weights_sum[ray]; acc.depth = a.depth[ray];
                        acc.r = a.image[(size_t)ray * 3]; acc.g = a.image[(size_t)ray * 3 + 1]; acc.b = a.image[(size_t)ray * 3 + 2];
                    }
                    r_done = 0;
                    s.p_dx[tid] = d[0]; s.p_dy[tid] = d[1]; s.p_dz[tid] = d[2];
                }
            }
            if (lane == 0) s.misc[4 + wave] = (nw && base + nw >= s.misc[14]) ? 1u : 0u;  // this wave saw the end of the queue
        }
        GF_STAMP(1);
        // ------------------------------------------------------------------ pool census
        const bool alive = ray >= 0;
        unsigned long long amask = 0;
        if (wave < 2) {
            amask = __ballot(alive);
            if (lane == 0) s.misc[wave] = (uint32_t)__popcll(amask);
        }
        __syncthreads();
        GF_STAMP(2);
        const uint32_t n_pool = s.misc[0] + s.misc[1];
        if (queue_open && (s.misc[4] | s.misc[5])) queue_open = false;
        if (n_pool == 0) {
            if (!queue_open) break;   // nothing alive, nothing left to fetch
            continue;                 // the queue still has entries: fetch again
        }
        // Samples per ray this round: n = floor(128 / n_pool) for everyone, one more for the first `extra` rays, so the 128
        // slots are all used whenever the pool is not full (how a ray's budget is cut into rounds does not change its result).
        uint32_t n = kPass / n_pool;
        n = n > 8u ? 8u : n;          // >= 1 since n_pool <= 128
#ifndef GF_UNIFORM_N
        const uint32_t extra = n < 8u ? (uint32_t)kPass - n * n_pool : 0u;   // < n_pool
#else
        const uint32_t extra = 0u;
#endif
        // ------------------------------------------------------------------ A. march
        uint32_t mcnt = 0, req = 0, rank = 0;
        if (alive) {
            rank = (wave ? s.misc[0] : 0u) + (uint32_t)__popcll(amask & ((1ull << lane) - 1ull));
            const uint32_t left = s.misc[13] - r_done;
            const uint32_t mine = n + (rank < extra ? 1u : 0u);
            req = mine < left ? mine : left;
            const uint32_t base = rank * n + (rank < extra ? rank : extra);
            // One sample of look-ahead: the traversal continues to the START of the sample after the requested ones (bounded like the rest).
            // If there is none, the ray is known to be exhausted in THIS round and retires with the same terminal index (samples + 1) it
            // would have discovered in the next one -- where it held a pool slot and produced nothing: one empty slot per exiting ray,
            // 10 % of all slots.  If there is one, the ray clock is rewound to its start, where the next round's march begins anyway.
            float t_next = 0.0f;
            float r_ox = a.cam_o[0], r_oy = a.cam_o[1], r_oz = a.cam_o[2];
            if (!a.pose_mode) {
                const float* o = a.rays_o + (size_t)ray * 3;
                r_ox = o[0]; r_oy = o[1]; r_oz = o[2];
            }
            const float r_dx = s.p_dx[tid], r_dy = s.p_dy[tid], r_dz = s.p_dz[tid];
            mcnt = gf::march_ray(a.mp, r_ox, r_oy, r_oz, r_dx, r_dy, r_dz, r_far, 0.0f, req + 1u, r_t,
                                [&](uint32_t q, float x, float y, float z, float dt, float t_after, float t_at) {
                                    if (q >= req) { t_next = t_at; return; }
                                    s.sx[base + q] = x; s.sy[base + q] = y; s.sz[base + q] = z;
                                    s.sdt[base + q] = dt; s.st[base + q] = t_after;
#ifdef GF_DIAG
                                    const uint32_t k = (a.phase ? a.max_steps : 0u) + r_done + q;
                                    uint32_t key = 0xFFFFFFFFu;
                                    if (a.diag && k < a.diag_stride) {
                                        key = (uint32_t)ray * a.diag_stride + k;
                                        float* rec = a.diag + (size_t)key * kDiagWords;
                                        rec[15] = x; rec[4] = dt; rec[5] = t_after;
                                        reinterpret_cast<uint32_t*>(rec)[18] = a.phase; reinterpret_cast<uint32_t*>(rec)[19] = base + q;
                                    }
                                    s.dkey[base + q] = key;
#endif
                                }, req + 1u + kMarchSlack);
            if (mcnt > req) { mcnt = req; r_t = t_next; }
        }
        GF_STAMP(3);
        if (owner) s.rcnt[tid] = (uint8_t)mcnt;
        __syncthreads();
        GF_STAMP(4);
        if (wave == 0) {  // exclusive scan of 128 counts, two per lane
            const uint32_t c0 = s.rcnt[2 * lane], c1 = s.rcnt[2 * lane + 1];
            uint32_t incl = c0 + c1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const uint32_t excl = incl - (c0 + c1);
            s.rbase[2 * lane] = (uint8_t)excl;
            s.rbase[2 * lane + 1] = (uint8_t)(excl + c0);
            if (lane == 63) s.misc[2] = incl;
        }
        __syncthreads();
        const uint32_t Mv = s.misc[2];
        if (alive) {
            const uint32_t b = s.rbase[tid];
            const uint32_t base = rank * n + (rank < extra ? rank : extra);
            for (uint32_t q = 0; q < mcnt; q++) { s.d2r[b + q] = (uint8_t)(base + q); s.rrank[b + q] = (uint8_t)tid; }
        }
        if (tid == 0) { s.misc[10] += Mv; s.misc[11] += 1u; s.misc[12] += (Mv + 31) / 32; }
        GF_STAMP(5);

        // ------------------------------------------------------------------ B. field
        if (Mv > 0) {  // uniform over the workgroup
            __syncthreads();   // dense map published
            GF_STAMP(6);
            const uint32_t nt = (Mv + 31) / 32;
            if constexpr (MODE == 1) {
                field_round16<false>(a, s, Mv, __builtin_amdgcn_readfirstlane((int)nt), wave, lane);
            } else if constexpr (MODE == 2) {
                // two instantiations, ONE arithmetic: every floating-point operation of the round is an explicit builtin (MFMA, fmaf, med3,
                // round-to-nearest conversions), so which of them evaluates a sample cannot change its value (the fast tier once differed
                // between instantiations through its conversions' instruction selection; test_full_size_frames_are_reproducible[split])
                if (nt <= 2) field_round_split<2>(a, s, Mv, __builtin_amdgcn_readfirstlane((int)nt), wave, lane);
                else field_round_split<4>(a, s, Mv, __builtin_amdgcn_readfirstlane((int)nt), wave, lane);
            } else {
                if (nt == 4) field_round<4>(a, s, Mv, wave, lane);
                else if (nt == 3) field_round<3>(a, s, Mv, wave, lane);
                else if (nt == 2) field_round<2>(a, s, Mv, wave, lane);
                else field_round<1>(a, s, Mv, wave, lane);
            }
        }

        // ------------------------------------------------------------------ C. composite, retire
        bool survivor = false;
        if (alive) {
            // this ray's sample count and first raw slot come back from LDS (the scan left them there): two registers less across the field
            const uint32_t cnt = s.rcnt[tid];
            const uint32_t base = cnt ? (uint32_t)s.d2r[s.rbase[tid]] : 0u;
            bool died = false;
            uint32_t d = 0;
            for (uint32_t q = 0; q < cnt; q++) {
                r_done++;
#ifdef GF_DIAG
                if (a.diag && s.dkey[base + q] != 0xFFFFFFFFu) {
                    float* rec = a.diag + (size_t)s.dkey[base + q] * kDiagWords;
                    uint32_t* ru = reinterpret_cast<uint32_t*>(rec);
                    rec[0] = s.sx[base + q]; rec[1] = s.sy[base + q]; rec[2] = s.sz[base + q]; rec[3] = s.ob[base + q];
                    ru[6] = a.diag_tag; ru[7] = blockIdx.x; ru[8] = dg_round; ru[9] = (uint32_t)s.rbase[tid] + q; ru[10] = Mv;
                    ru[11] = (n_pool << 8) | n;
                }
#endif
                if (!gf::composite_sample(acc, s.sx[base + q], s.sy[base + q], s.sz[base + q], s.ob[base + q], s.sdt[base + q], s.st[base + q], a.T_thresh)) {
                    died = true;  // T < T_thresh: terminates at this sample (raymarching.cu:1004)
                    d = r_done;
                    break;
                }
            }
            if (!died && !(r_t < r_far)) {  // the marcher ran out (look-ahead included): the next request finds nothing (raymarching.cu:977)
                died = true;
                d = r_done + 1;
            }
            const bool finished = !died && r_done == s.misc[13];
            if (died || finished) {
                atomicAdd(&s.misc[9], r_done);   // statistics: what this ray's compositor consumed in this phase
                a.weights_sum[ray] = acc.weight_sum;
                a.depth[ray] = acc.depth;
                a.image[(size_t)ray * 3 + 0] = acc.r; a.image[(size_t)ray * 3 + 1] = acc.g; a.image[(size_t)ray * 3 + 2] = acc.b;
                if (finished) a.rays_t[ray] = r_t;
                if (died && a.phase == 0) {
                    if (d < (uint32_t)kHistBins) atomicAdd(&s.hist[d], 1u);
                    else atomicAdd(&a.ctrl[gf::kCtrlHist + d], 1u);     // max_steps > 64 only: a ray that outlives 64 samples before it ends
                }
                survivor = finished && a.phase == 0;
            }
            if (died || finished) {
                if (!survivor) ray = -1;   // survivors keep the index until the list below has taken it
            }
        }
        if (a.phase == 0 && wave < 2) {
            const unsigned long long m = __ballot(survivor);
            const uint32_t ns = (uint32_t)__popcll(m);
            uint32_t base = 0;
            if (lane == 0 && ns) base = atomicAdd(&a.ctrl[gf::kCtrlNSurv], ns);
            base = __shfl(base, 0);
            if (survivor) { a.survivors[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ray; ray = -1; }
        }
        GF_STAMP(37);
#ifdef GF_TRACE
        if (tid == 0) { s.tr[38] = Mv; s.tr[39] = n_pool; s.tr[40] = n; s.tr[41] = __builtin_amdgcn_s_getreg(63492 /* HW_REG_HW_ID, 32 bits */); }
        __syncthreads();
        if (a.trace && blockIdx.x < kTraceWGs && tr_round < kTraceRounds && tid < kTraceSlots)
            a.trace[((a.phase * kTraceWGs + blockIdx.x) * kTraceRounds + tr_round) * kTraceSlots + tid] = s.tr[tid];
        tr_round++;
        span_t1 = __builtin_amdgcn_s_memtime(); span_r1 = __builtin_amdgcn_s_memrealtime();
#endif
    }

    __syncthreads();
    {   // the lane index is re-derived here (mbcnt) rather than kept: the allocator had parked `4 * tid` in scratch from the first line to this one
        const int tid_e = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (a.phase == 0 && tid_e >= 1 && tid_e <= (int)a.max_steps && tid_e < kHistBins) {
            const uint32_t v = s.hist[tid_e];
            if (v) atomicAdd(&a.ctrl[gf::kCtrlHist + tid_e], v);
        }
    }
#ifdef GF_TRACE
    if (tid == 0 && a.spans) {   // per-workgroup lifetime (s_memtime offsets differ between CUs: only differences are meaningful)
        unsigned long long* sp = a.spans + ((size_t)a.phase * 512 + blockIdx.x) * 8;
        sp[0] = span_t0; sp[1] = span_t1; sp[2] = (unsigned long long)s.misc[11] | ((unsigned long long)s.misc[10] << 32);
        sp[3] = __builtin_amdgcn_s_getreg(63508 /* HW_REG_XCC_ID */); sp[4] = span_r0; sp[5] = span_r1;
        sp[6] = __builtin_amdgcn_s_memtime(); sp[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (tid == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(a.ctrl);
        atomicAdd(&st[gf::kCtrlStatA / 2 + a.phase], (unsigned long long)s.misc[10] | ((unsigned long long)s.misc[12] << 32));
        atomicAdd(&st[gf::kCtrlStatB / 2 + a.phase], (unsigned long long)s.misc[11] | ((unsigned long long)s.misc[9] << 32));
    }
}

// ---------------------------------------------------------------------------------------------------- occupancy-grid refresh
// NeRFRenderer.update_extra_state's field queries (renderer.py:232-246) for ALL cascades and cells in one launch: cell -> jittered
// centre -> density head on the matrix pipe (the same field_round as the renderer, cut after the density row) -> sigma * density_scale
// into tmp_grid at the cell's Morton index.  Cells are walked in the reference's meshgrid order (x slowest), which is also the order of
// the jitter array, 128 per round.
struct GridArgs {
    const float* noise;     // [C][G^3][3] U[0,1) in meshgrid order, or NULL (cell centres)
    float* tmp_grid;        // [C][G^3] Morton order
    uint32_t C, G;
    float density_scale;
};

__global__ void __launch_bounds__(kThreads, 2) k_grid_density(const HeadArgs a, const GridArgs u) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)gf::HS_TOTAL; i += kThreads) s.P[P_SMALL + i] = a.head_pack[gf::HP_SMALL + i];
    if (tid < 128) s.P[P_AMBBIAS + tid] = a.amb_bias[tid];
    if (tid < 32) {
        const uint32_t g = tid >> 4, l = tid & 15;
        gf::LevelMeta* m = reinterpret_cast<gf::LevelMeta*>(s.P + P_META) + tid;
        *m = g ? gf::make_level_meta<2>(a.lv2.scale[l], a.lv2.resolution[l], a.amb_offsets, l, a.gridtype)
               : gf::make_level_meta<3>(a.lv3.scale[l], a.lv3.resolution[l], a.pos_offsets, l, a.gridtype);
    }
    const uint32_t G = u.G, G3 = G * G * G, total = u.C * G3;
    const uint32_t chunks = (total + kPass - 1) / kPass;
#ifndef GF_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        __syncthreads();   // previous round retired
        const uint32_t i = chunk * kPass + (uint32_t)tid;
        uint32_t cas = 0, cx = 0, cy = 0, cz = 0;
        if (tid < kPass && i < total) {
#pragma clang fp contract(off)
            cas = i / G3;
            const uint32_t lin = i - cas * G3;
            cx = lin / (G * G); cy = (lin / G) % G; cz = lin % G;
            // xyzs = 2 * coords / (G - 1) - 1;  cas_xyzs = xyzs * (bound_c - half) + (noise * 2 - 1) * half    (renderer.py:236-243)
            const float bound_c = fminf(scalbnf(1.0f, (int)cas), a.bound);
            const float hgs = (float)((double)bound_c / (double)G);
            const float span = (float)((double)bound_c - (double)bound_c / (double)G);
            const float gm1 = (float)(G - 1);
            const float c3[3] = {(float)cx, (float)cy, (float)cz};
            float p[3];
            for (int d = 0; d < 3; d++) {
                float v = (2.0f * c3[d] / gm1 - 1.0f) * span;
                if (u.noise) v = v + (u.noise[(size_t)i * 3 + d] * 2.0f - 1.0f) * hgs;
                p[d] = v;
            }
            s.sx[tid] = p[0]; s.sy[tid] = p[1]; s.sz[tid] = p[2];
            s.d2r[tid] = (uint8_t)tid;
        }
        const uint32_t left = total - chunk * kPass;
        const uint32_t Mv = left < (uint32_t)kPass ? left : (uint32_t)kPass;
        __syncthreads();
        const uint32_t nt = (Mv + 31) / 32;
        if (nt == 4) field_round<4, true>(a, s, Mv, wave, lane);
        else if (nt == 3) field_round<3, true>(a, s, Mv, wave, lane);
        else if (nt == 2) field_round<2, true>(a, s, Mv, wave, lane);
        else field_round<1, true>(a, s, Mv, wave, lane);
        if (tid < kPass && i < total) u.tmp_grid[(size_t)cas * G3 + gf::morton3d(cx, cy, cz)] = s.sx[tid] * u.density_scale;
    }
}

// ---------------------------------------------------------------------------------------------------- field on a dense point list
// RADNeRF.forward (radnerf.py:73-105) for M points in ONE launch: what the training marcher, the viewer and update_extra_state hand to the
// field -- (xyz, dir) lists, not rays.  The same field_round as the renderer, 128 points per round; no autograd (the caller is in
// inference mode, or renders a frozen head under no_grad: radnerf_torso.py:97-150).
struct PointArgs {
    const float* xyz; const float* dirs;      // [M,3] each
    const float* col_bias;                    // [128] accumulator order: W_color0[:, 144:148] @ individual_code, or NULL = the packed one
    float* sigma; float* rgb; float* ambient; // [M], [M,3], [M,2] (ambient may be NULL)
    uint32_t M;
    SaveBufs sv;                              // SAVE launches only
};

template <bool SAVE>
__global__ void __launch_bounds__(kThreads, 2) k_field_points(const HeadArgs a, const PointArgs u) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)gf::HS_TOTAL; i += kThreads) s.P[P_SMALL + i] = a.head_pack[gf::HP_SMALL + i];
    __syncthreads();
    if (u.col_bias && tid < 128) s.P[P_SMALL + gf::HS_COLBIAS + tid] = u.col_bias[tid];
    if (tid < 128) s.P[P_AMBBIAS + tid] = a.amb_bias[tid];
    if (tid < 32) {
        const uint32_t g = tid >> 4, l = tid & 15;
        gf::LevelMeta* m = reinterpret_cast<gf::LevelMeta*>(s.P + P_META) + tid;
        *m = g ? gf::make_level_meta<2>(a.lv2.scale[l], a.lv2.resolution[l], a.amb_offsets, l, a.gridtype)
               : gf::make_level_meta<3>(a.lv3.scale[l], a.lv3.resolution[l], a.pos_offsets, l, a.gridtype);
    }
    const uint32_t chunks = (u.M + kPass - 1) / kPass;
#ifndef GF_NO_SETPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        __syncthreads();   // previous round retired
        const uint32_t i = chunk * kPass + (uint32_t)tid;
        if (tid < kPass && i < u.M) {
            s.sx[tid] = u.xyz[(size_t)i * 3]; s.sy[tid] = u.xyz[(size_t)i * 3 + 1]; s.sz[tid] = u.xyz[(size_t)i * 3 + 2];
            s.p_dx[tid] = u.dirs[(size_t)i * 3]; s.p_dy[tid] = u.dirs[(size_t)i * 3 + 1]; s.p_dz[tid] = u.dirs[(size_t)i * 3 + 2];
            s.d2r[tid] = (uint8_t)tid;
            s.rrank[tid] = (uint8_t)tid;
        }
        const uint32_t left = u.M - chunk * kPass;
        const uint32_t Mv = left < (uint32_t)kPass ? left : (uint32_t)kPass;
        __syncthreads();
        const uint32_t nt = (Mv + 31) / 32;
        const uint32_t gbase = chunk * kPass;
        if (nt == 4) field_round<4, false, true, SAVE>(a, s, Mv, wave, lane, &u.sv, gbase);
        else if (nt == 3) field_round<3, false, true, SAVE>(a, s, Mv, wave, lane, &u.sv, gbase);
        else if (nt == 2) field_round<2, false, true, SAVE>(a, s, Mv, wave, lane, &u.sv, gbase);
        else field_round<1, false, true, SAVE>(a, s, Mv, wave, lane, &u.sv, gbase);
        if (tid < kPass && i < u.M) {
            u.sigma[i] = s.sx[tid];
            u.rgb[(size_t)i * 3] = s.sy[tid]; u.rgb[(size_t)i * 3 + 1] = s.sz[tid]; u.rgb[(size_t)i * 3 + 2] = s.ob[tid];
            if (u.ambient) { u.ambient[(size_t)i * 2] = s.sdt[tid]; u.ambient[(size_t)i * 2 + 1] = s.st[tid]; }
        }
    }
}


// The training forward on the f16 tier (gf_field_forward_train16): f16 MFMA operands, fp32 accumulation -- the arithmetic the reference's
// autocast run has (cond_encoder.py:106-111 under utils/commons/trainer.py:307-382) -- with every layer's activations saved as binary16.
struct PointArgs16 {
    const float* xyz; const float* dirs; const float* col_bias;
    float* sigma; float* rgb; float* ambient;
    uint32_t M;
    SaveBufs16 sv;
};

__global__ void __launch_bounds__(kThreads, 2) k_field_points16(const HeadArgs a, const PointArgs16 u) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)gf::HS_TOTAL; i += kThreads) s.P[P_SMALL + i] = a.head_pack[gf::HP_SMALL + i];
    __syncthreads();
    if (u.col_bias && tid < 128) s.P[P_SMALL + gf::HS_COLBIAS + tid] = u.col_bias[tid];
    if (tid < 128) s.P[P_AMBBIAS + tid] = a.amb_bias[tid];
    if (tid < 32) {
        const uint32_t g = tid >> 4, l = tid & 15;
        gf::LevelMeta* m = reinterpret_cast<gf::LevelMeta*>(s.P + P_META) + tid;
        *m = g ? gf::make_level_meta<2>(a.lv2.scale[l], a.lv2.resolution[l], a.amb_offsets, l, a.gridtype)
               : gf::make_level_meta<3>(a.lv3.scale[l], a.lv3.resolution[l], a.pos_offsets, l, a.gridtype);
    }
    const uint32_t chunks = (u.M + kPass - 1) / kPass;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        __syncthreads();   // previous round retired
        const uint32_t i = chunk * kPass + (uint32_t)tid;
        if (tid < kPass) {
            const bool in = i < u.M;
            const size_t p = in ? i : (size_t)chunk * kPass;      // slots beyond the list: a valid stand-in (their SH rows are read by no live column)
            s.sx[tid] = u.xyz[p * 3]; s.sy[tid] = u.xyz[p * 3 + 1]; s.sz[tid] = u.xyz[p * 3 + 2];
            s.p_dx[tid] = u.dirs[p * 3]; s.p_dy[tid] = u.dirs[p * 3 + 1]; s.p_dz[tid] = u.dirs[p * 3 + 2];
            s.d2r[tid] = (uint8_t)tid;
            s.rrank[tid] = (uint8_t)tid;
        }
        const uint32_t left = u.M - chunk * kPass;
        const uint32_t Mv = left < (uint32_t)kPass ? left : (uint32_t)kPass;
        __syncthreads();
        const uint32_t nt = (Mv + 31) / 32;
        field_round16<true>(a, s, Mv, __builtin_amdgcn_readfirstlane((int)nt), wave, lane, &u.sv, chunk * kPass);
        if (tid < kPass && i < u.M) {
            u.sigma[i] = s.sx[tid];
            u.rgb[(size_t)i * 3] = s.sy[tid]; u.rgb[(size_t)i * 3 + 1] = s.sz[tid]; u.rgb[(size_t)i * 3 + 2] = s.ob[tid];
            u.ambient[(size_t)i * 2] = s.sdt[tid]; u.ambient[(size_t)i * 2 + 1] = s.st[tid];
        }
    }
}

// ---------------------------------------------------------------------------------------------------- field backward (dX chain)
// The input-gradient chain of RADNeRF.forward for a dense point list in ONE launch (geneface_amd/train_field.py): from (d sigma, d rgb,
// d ambient) back through colour net -> geometry feature -> sigma net -> 2-D lookup (input gradient, re-gathered) -> ambient net, writing
// the pre-activation gradient of every layer as an [M, 128] matrix (the weight gradients are tall GEMMs over those and the saved
// activations) and the gradients of both grid feature sets (-> the table scatter kernels).  Same machinery as the forward: 128 points per
// round, activations (here: gradients) in the LDS buffer H, wave w owns 32 output features, A operands = TRANSPOSED weight blocks streamed
// L2 -> registers (gf_field_bwd stream: six 128 x 128 layers, the two narrow ones zero-padded), ReLU derivatives from the forward's mask bits.
constexpr int BG_C1 = 0, BG_S3 = 16, BG_S2 = 32, BG_S1 = 48, BG_A2 = 64, BG_A1 = 80, BG_TOTAL = 96;

struct BwdArgs {
    const float* stream;                                   // [4 waves][BG_TOTAL][64][4]
    const float *g_sigma, *g_rgb, *g_amb;                  // [M], [M,3], [M,2] incoming
    const float *sigma, *rgb, *amb;                        // forward outputs
    const uint16_t *m_hc1, *m_hs2, *m_hs1, *m_ha2, *m_ha1; // forward ReLU masks
    float *g_zc, *g_h0, *g_za;                             // [M,3], [M], [M,2]   pre-activation gradients of the three skinny layers
    float *g_hc1, *g_geo, *g_hs2, *g_hs1, *g_ha2, *g_ha1;  // [M,128] each
    float *g_f3, *g_f2;                                    // [16 levels][M][2] each: the layout the table scatter kernel reads
    float *s_hc1, *s_ha1;                                  // [128] each, ZEROED by the caller: column sums of g_hc1 / g_ha1 over the points
    uint32_t* lvl_max;                                     // or NULL; [2][16] ZEROED by the caller: max |g_f3| / |g_f2| per level (float bit patterns)
    uint32_t M;
};

// accumulators -> LDS (and row gbase + sample of G), optionally through the ReLU mask of the layer whose pre-activation gradient this is
// O16: the rows of G are binary16 (the AMP tier: the weight-gradient products then run on half operands, as the reference's autocast step does)
template <int NT, bool MASK, bool O16 = false>
__device__ __forceinline__ void bwd_store(float* Hw, float* __restrict__ G, const uint16_t* __restrict__ mask, uint32_t gbase, uint32_t Mv, int wave,
                                          int lane, const floatx16 (&acc)[4], float* colsum = nullptr /* LDS [128]: running column sums of G, or NULL */) {
    const int half = lane >> 5, j = lane & 31;
    float part[16];   // this lane's share of the column sums of this call (only with colsum)
#pragma unroll
    for (int r = 0; r < 16; r++) part[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        uint32_t bits = 0xFFFFu;
        if (MASK) bits = mask[(((size_t)(gbase / kPass) * 4 + t) * 4 + wave) * 64 + lane];
        const bool ok = (uint32_t)(t * 32 + j) < Mv;
        const size_t roff = (size_t)(gbase + (uint32_t)(t * 32 + j)) * 128 + 32 * wave + 4 * half;
        float* row = (G && !O16) ? G + roff : nullptr;
        _Float16* row16 = (G && O16) ? reinterpret_cast<_Float16*>(G) + roff : nullptr;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 v = {acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
            if (MASK) {
                v.x = (bits >> (4 * q + 0)) & 1u ? v.x : 0.0f; v.y = (bits >> (4 * q + 1)) & 1u ? v.y : 0.0f;
                v.z = (bits >> (4 * q + 2)) & 1u ? v.z : 0.0f; v.w = (bits >> (4 * q + 3)) & 1u ? v.w : 0.0f;
            }
            *reinterpret_cast<float4*>(Hw + t * 32 * kHS + 8 * q) = v;
            if (row && ok) *reinterpret_cast<float4*>(row + 8 * q) = v;
            if (O16 && row16 && ok) *reinterpret_cast<half4*>(row16 + 8 * q) = half4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            if (colsum && ok) { part[4 * q] += v.x; part[4 * q + 1] += v.y; part[4 * q + 2] += v.z; part[4 * q + 3] += v.w; }
        }
    }
    if (colsum) {
        // The 32 lanes of a half hold the same 16 features over different samples: one butterfly per call, then the feature's one owner adds it
        // to the running sum in LDS.  (The sums used to live in 32 registers per lane for the whole kernel, most of its 488 bytes of scratch.)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = part[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (j == 0) colsum[32 * wave + 8 * (r >> 2) + 4 * half + (r & 3)] += v;
        }
    }
}

// a layer whose result was written to H in ROW layout (the two rank-k products): pull this lane's accumulator-layout share back,
// mask it, write it to H and to G
template <int NT, bool O16 = false>
__device__ __forceinline__ void bwd_mask_pass(float* Hw, float* __restrict__ G, const uint16_t* __restrict__ mask, uint32_t gbase, uint32_t Mv, int wave,
                                              int lane, floatx16 (&acc)[4], float* colsum = nullptr) {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(Hw + t * 32 * kHS + 8 * q);
            acc[t][4 * q + 0] = v.x; acc[t][4 * q + 1] = v.y; acc[t][4 * q + 2] = v.z; acc[t][4 * q + 3] = v.w;
        }
    bwd_store<NT, true, O16>(Hw, G, mask, gbase, Mv, wave, lane, acc, colsum);
}

template <int NT, bool O16 = false>
__device__ __forceinline__ void bwd_round(const HeadArgs& a, const BwdArgs& u, const Smem& s, uint32_t Mv, uint32_t gbase, int wave, int lane,
                                          float* cs_hc1, float* cs_ha1 /* LDS [128] each */) {
    const int half = lane >> 5, j = lane & 31;
    const uint32_t sI = (uint32_t)(wave * 32 + j);
    const bool tile_on = wave < NT;
    const bool valid = sI < Mv;
    const size_t pt = (size_t)gbase + sI;
    float* Hrow = s.H + sI * kHS;
    const float* Hb = s.H + j * kHS + 4 * half;
    float* Hw = s.H + j * kHS + 32 * wave + 4 * half;
    const char* Ws = reinterpret_cast<const char*>(u.stream) + (size_t)wave * BG_TOTAL * 1024;
    uint32_t lane16 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(lane16));
    const gf::LevelMeta* meta = reinterpret_cast<const gf::LevelMeta*>(s.P + P_META);
    floatx16 A[4];
    WPipeT<kWAheadTrain> wp;
#pragma unroll
    for (int g = 0; g < kWAheadTrain; g++) wp.q[g] = load_group(Ws, g, lane16);

    // ---- colour tail: d z_c = d rgb * rgb (1 - rgb);  H row <- W_c2^T d z_c  (lane half h writes features 64h .. 64h+63)
    float gh0 = 0.0f;
    if (tile_on) {
        float gz[3] = {0.0f, 0.0f, 0.0f};
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; c++) { const float r = u.rgb[pt * 3 + c]; gz[c] = u.g_rgb[pt * 3 + c] * r * (1.0f - r); }
            const float sg = u.sigma[pt];
            gh0 = u.g_sigma[pt] * fminf(fmaxf(sg, 3.0590232e-7f), 3269017.37f);   // trunc_exp backward: g * exp(clamp(x, -15, 15))
            if (half == 0) { u.g_zc[pt * 3] = gz[0]; u.g_zc[pt * 3 + 1] = gz[1]; u.g_zc[pt * 3 + 2] = gz[2]; u.g_h0[pt] = gh0; }
        }
        if (half == 0) s.sdt[sI] = gh0;        // per-sample scalar the rank-1 term of the sigma layer reads by column
        const float* W = s.P + P_SMALL + gf::HS_COL2 + 64 * half;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float4 w0 = *reinterpret_cast<const float4*>(W + 4 * i), w1 = *reinterpret_cast<const float4*>(W + 128 + 4 * i),
                         w2 = *reinterpret_cast<const float4*>(W + 256 + 4 * i);
            float4 v;
            v.x = gz[0] * w0.x + gz[1] * w1.x + gz[2] * w2.x; v.y = gz[0] * w0.y + gz[1] * w1.y + gz[2] * w2.y;
            v.z = gz[0] * w0.z + gz[1] * w1.z + gz[2] * w2.z; v.w = gz[0] * w0.w + gz[1] * w1.w + gz[2] * w2.w;
            *reinterpret_cast<float4*>(Hrow + 64 * half + 4 * i) = v;
        }
    }
    __syncthreads();
    bwd_mask_pass<NT, O16>(Hw, O16 ? u.g_hc1 : nullptr, u.m_hc1, gbase, Mv, wave, lane, A, cs_hc1);      // d h_c1 (pre-activation)
    __syncthreads();
    if constexpr (!O16) rows32_to_global(s.H, u.g_hc1, gbase, Mv, wave * 64 + lane);
    // ---- d geo = W_c1[:, 16:144]^T d h_c1
    obw_zero<NT>(A);
    obw_mfma<NT, BG_C1, 16, BG_TOTAL>(wp, Ws, lane16, Hb, A);
    __syncthreads();
    bwd_store<NT, false, O16>(Hw, O16 ? u.g_geo : nullptr, nullptr, gbase, Mv, wave, lane, A);
    __syncthreads();
    if constexpr (!O16) rows32_to_global(s.H, u.g_geo, gbase, Mv, wave * 64 + lane);
    // ---- d h_s2 = W_s3[1:]^T d geo + d h0 (x) W_s3[0], masked
    {
        const float* w0 = s.P + P_SMALL + gf::HS_SIGROW + 32 * wave + 4 * half;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float g0 = s.sdt[t * 32 + j];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 w = *reinterpret_cast<const float4*>(w0 + 8 * q);
                A[t][4 * q + 0] = g0 * w.x; A[t][4 * q + 1] = g0 * w.y; A[t][4 * q + 2] = g0 * w.z; A[t][4 * q + 3] = g0 * w.w;
            }
        }
    }
    obw_mfma<NT, BG_S3, 16, BG_TOTAL>(wp, Ws, lane16, Hb, A);
    __syncthreads();
    bwd_store<NT, true, O16>(Hw, O16 ? u.g_hs2 : nullptr, u.m_hs2, gbase, Mv, wave, lane, A);
    __syncthreads();
    if constexpr (!O16) rows32_to_global(s.H, u.g_hs2, gbase, Mv, wave * 64 + lane);
    // ---- d h_s1 = W_s2^T d h_s2, masked
    obw_zero<NT>(A);
    obw_mfma<NT, BG_S2, 16, BG_TOTAL>(wp, Ws, lane16, Hb, A);
    __syncthreads();
    bwd_store<NT, true, O16>(Hw, O16 ? u.g_hs1 : nullptr, u.m_hs1, gbase, Mv, wave, lane, A);
    __syncthreads();
    if constexpr (!O16) rows32_to_global(s.H, u.g_hs1, gbase, Mv, wave * 64 + lane);
    // ---- [d f3 (sigma branch) | d f2] = W_s1^T d h_s1   (64 real outputs, zero-padded to 128)
    obw_zero<NT>(A);
    obw_mfma<NT, BG_S1, 16, BG_TOTAL>(wp, Ws, lane16, Hb, A);
    __syncthreads();
    bwd_store<NT, false>(Hw, nullptr, nullptr, gbase, Mv, wave, lane, A);
    __syncthreads();
    // ---- 2-D lookup: d f2 out, input gradient in; ambient tail: d z_a = (d amb + 0.5 d x2) (1 - amb^2);  H row <- W_a3^T d z_a
    if (tile_on) {
        float gf3[16];   // d f3 of the sigma branch: parked in its final place in g_f3 until the ambient branch's share arrives (two layers on)
        float gamb[2] = {0.0f, 0.0f}, ambv[2] = {0.0f, 0.0f};   // loaded here, not with the other per-point scalars: four registers less across four layers
        if (valid) {
            gamb[0] = u.g_amb[pt * 2]; gamb[1] = u.g_amb[pt * 2 + 1];
            ambv[0] = u.amb[pt * 2]; ambv[1] = u.amb[pt * 2 + 1];
        }
        float gf2[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v3 = *reinterpret_cast<const float4*>(Hrow + 16 * half + 4 * q), v2 = *reinterpret_cast<const float4*>(Hrow + 32 + 16 * half + 4 * q);
            gf3[4 * q] = v3.x; gf3[4 * q + 1] = v3.y; gf3[4 * q + 2] = v3.z; gf3[4 * q + 3] = v3.w;
            gf2[4 * q] = v2.x; gf2[4 * q + 1] = v2.y; gf2[4 * q + 2] = v2.z; gf2[4 * q + 3] = v2.w;
        }
        if (valid) {   // [level][point][channel]: the layout gf_grid_encode_backward indexes (gridencoder.cu:275), no transpose on the host
#pragma unroll
            for (int l = 0; l < 8; l++) *reinterpret_cast<float2*>(u.g_f3 + ((size_t)(8 * half + l) * u.M + pt) * 2) = float2{gf3[2 * l], gf3[2 * l + 1]};
#pragma unroll
            for (int l = 0; l < 8; l++) *reinterpret_cast<float2*>(u.g_f2 + ((size_t)(8 * half + l) * u.M + pt) * 2) = float2{gf2[2 * l], gf2[2 * l + 1]};
            if (u.lvl_max) {   // running per-level maxima of this workgroup (bit patterns of |g| order like the values; a NaN ends up on top)
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const uint32_t m0 = __float_as_uint(gf2[2 * l]) & 0x7fffffffu, m1 = __float_as_uint(gf2[2 * l + 1]) & 0x7fffffffu;
                    atomicMax(&s.hist[16 + 8 * half + l], m0 > m1 ? m0 : m1);
                }
            }
        }
        const float x2[2] = {(ambv[0] + 1.0f) / 2.0f, (ambv[1] + 1.0f) / 2.0f};
        float dx[2];
        gf::encode8_grad2(a.amb_table, meta + 16 + half * 8, a.gridtype, a.interp, x2, gf2, dx);
        dx[0] += __shfl_xor(dx[0], 32);
        dx[1] += __shfl_xor(dx[1], 32);
        const float gza[2] = {valid ? (gamb[0] + 0.5f * dx[0]) * (1.0f - ambv[0] * ambv[0]) : 0.0f,
                              valid ? (gamb[1] + 0.5f * dx[1]) * (1.0f - ambv[1] * ambv[1]) : 0.0f};
        if (valid && half == 0) { u.g_za[pt * 2] = gza[0]; u.g_za[pt * 2 + 1] = gza[1]; }
        const float* W = s.P + P_SMALL + gf::HS_AMB3 + 64 * half;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float4 w0 = *reinterpret_cast<const float4*>(W + 4 * i), w1 = *reinterpret_cast<const float4*>(W + 128 + 4 * i);
            const float4 v = {gza[0] * w0.x + gza[1] * w1.x, gza[0] * w0.y + gza[1] * w1.y, gza[0] * w0.z + gza[1] * w1.z, gza[0] * w0.w + gza[1] * w1.w};
            *reinterpret_cast<float4*>(Hrow + 64 * half + 4 * i) = v;
        }
    }
    __syncthreads();
    bwd_mask_pass<NT, O16>(Hw, O16 ? u.g_ha2 : nullptr, u.m_ha2, gbase, Mv, wave, lane, A);      // d h_a2
    __syncthreads();
    if constexpr (!O16) rows32_to_global(s.H, u.g_ha2, gbase, Mv, wave * 64 + lane);
    // ---- d h_a1 = W_a2^T d h_a2, masked
    obw_zero<NT>(A);
    obw_mfma<NT, BG_A2, 16, BG_TOTAL>(wp, Ws, lane16, Hb, A);
    __syncthreads();
    bwd_store<NT, true, O16>(Hw, O16 ? u.g_ha1 : nullptr, u.m_ha1, gbase, Mv, wave, lane, A, cs_ha1);
    __syncthreads();
    if constexpr (!O16) rows32_to_global(s.H, u.g_ha1, gbase, Mv, wave * 64 + lane);
    // ---- d f3 (ambient branch) = W_a1[:, :32]^T d h_a1   (32 real outputs, zero-padded)
    obw_zero<NT>(A);
    obw_mfma<NT, BG_A1, 16, BG_TOTAL>(wp, Ws, lane16, Hb, A);
    __syncthreads();
    bwd_store<NT, false>(Hw, nullptr, nullptr, gbase, Mv, wave, lane, A);
    __syncthreads();
    if (tile_on && valid) {
        float gf3[16];
#pragma unroll
        for (int l = 0; l < 8; l++) {   // this lane's own store of two layers ago (same thread, same address: ordered)
            const float2 p = *reinterpret_cast<const float2*>(u.g_f3 + ((size_t)(8 * half + l) * u.M + pt) * 2);
            gf3[2 * l] = p.x; gf3[2 * l + 1] = p.y;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(Hrow + 16 * half + 4 * q);
            gf3[4 * q] += v.x; gf3[4 * q + 1] += v.y; gf3[4 * q + 2] += v.z; gf3[4 * q + 3] += v.w;
        }
#pragma unroll
        for (int l = 0; l < 8; l++) *reinterpret_cast<float2*>(u.g_f3 + ((size_t)(8 * half + l) * u.M + pt) * 2) = float2{gf3[2 * l], gf3[2 * l + 1]};
        if (u.lvl_max) {
#pragma unroll
            for (int l = 0; l < 8; l++) {
                const uint32_t m0 = __float_as_uint(gf3[2 * l]) & 0x7fffffffu, m1 = __float_as_uint(gf3[2 * l + 1]) & 0x7fffffffu;
                atomicMax(&s.hist[8 * half + l], m0 > m1 ? m0 : m1);
            }
        }
    }
    __syncthreads();
}

template <bool O16>
__global__ void __launch_bounds__(kThreads, 2) k_field_backward(const HeadArgs a, const BwdArgs u) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)gf::HS_TOTAL; i += kThreads) s.P[P_SMALL + i] = a.head_pack[gf::HP_SMALL + i];
    if (tid < 32) {
        const uint32_t g = tid >> 4, l = tid & 15;
        gf::LevelMeta* m = reinterpret_cast<gf::LevelMeta*>(s.P + P_META) + tid;
        *m = g ? gf::make_level_meta<2>(a.lv2.scale[l], a.lv2.resolution[l], a.amb_offsets, l, a.gridtype)
               : gf::make_level_meta<3>(a.lv3.scale[l], a.lv3.resolution[l], a.pos_offsets, l, a.gridtype);
    }
    if (tid < 32) s.hist[tid] = 0;   // [0..15] max |d f3| per level, [16..31] max |d f2| (visible behind the first round's barriers)
    const uint32_t chunks = (u.M + kPass - 1) / kPass;
    float* cs_hc1 = s.sx;   // [128] column sums of d h_c1 / d h_a1 over this workgroup's points (-> gradients of the identity code and of
    float* cs_ha1 = s.sy;   // cond_feat): the sample staging arrays are free in this kernel
    if (tid < 128) { cs_hc1[tid] = 0.0f; cs_ha1[tid] = 0.0f; }
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        __syncthreads();
        const uint32_t gbase = chunk * kPass;
        const uint32_t left = u.M - gbase;
        const uint32_t Mv = left < (uint32_t)kPass ? left : (uint32_t)kPass;
        const uint32_t nt = (Mv + 31) / 32;
        if (nt == 4) bwd_round<4, O16>(a, u, s, Mv, gbase, wave, lane, cs_hc1, cs_ha1);
        else if (nt == 3) bwd_round<3, O16>(a, u, s, Mv, gbase, wave, lane, cs_hc1, cs_ha1);
        else if (nt == 2) bwd_round<2, O16>(a, u, s, Mv, gbase, wave, lane, cs_hc1, cs_ha1);
        else bwd_round<1, O16>(a, u, s, Mv, gbase, wave, lane, cs_hc1, cs_ha1);
    }
    __syncthreads();
    if (tid < 128) {   // one atomic per feature and workgroup
        atomicAdd(&u.s_hc1[tid], cs_hc1[tid]);
        atomicAdd(&u.s_ha1[tid], cs_ha1[tid]);
    }
    __syncthreads();
    if (u.lvl_max && tid < 32 && s.hist[tid]) atomicMax(&u.lvl_max[tid], s.hist[tid]);
}

// ---------------------------------------------------------------------------------------------------- field backward on the f16 tier (round 6)
// The same input-gradient chain with f16 MFMA operands (the AMP training tier: the reference's autocast step back-propagates its Linear layers
// in half, fp32 accumulation).  Six transposed 128 x 128 blocks as f16 A-operand streams [wave][layer 6][group 8][lane][8 halves]
// (element = Wt[32 wave + (lane & 31)][16 group + 8 (lane >> 5) + i]), the round's gradients as binary16 rows [128][136] in LDS, accumulators,
// masks, column sums, the three skinny transposes, the lookup's input gradient and both grid-feature gradients in fp32.  Differences from
// bwd_round beyond the operand type: the two rank-k products (W_c2^T d z_c, W_a3^T d z_a) are computed directly in ACCUMULATOR layout from
// per-sample scalars in LDS (no row pass + re-read), and the two narrow layers' real outputs (64 / 32 features) go to an fp32 scratch
// [128][64] behind the f16 rows instead of through them: the grid-feature gradients never see f16.
constexpr int BH_C1 = 0, BH_S3 = 8, BH_S2 = 16, BH_S1 = 24, BH_A2 = 32, BH_A1 = 40, BH_TOTAL = 48;
static_assert(kPass * kHS16 * 2 + kPass * 64 * 4 <= kPass * kHS * 4, "f16 gradient rows + the fp32 scratch fit the activation buffer");

// accumulators -> f16 LDS rows (and binary16 row gbase + sample of G), optionally through a ReLU mask; running column sums in fp32
template <bool MASK, bool CS = false>
__device__ __forceinline__ void bwd16_store(_Float16* Hw, _Float16* __restrict__ G, const uint16_t* __restrict__ mask, uint32_t gbase, uint32_t Mv, int wave,
                                            int lane, const floatx16 (&acc)[4], int nt, float* colsum = nullptr) {
    const int half = lane >> 5, j = lane & 31;
    float part[CS ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (CS ? 16 : 1); r++) part[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < nt) {
            uint32_t bits = 0xFFFFu;
            if (MASK) bits = mask[(((size_t)(gbase / kPass) * 4 + t) * 4 + wave) * 64 + lane];
            const bool ok = (uint32_t)(t * 32 + j) < Mv;
            _Float16* row = nullptr;      // (round 6: the rows leave LDS through rows16_to_global after the barrier; G is kept for a caller that has no such window)
            if (G) row = G + (size_t)(gbase + (uint32_t)(t * 32 + j)) * 128 + 32 * wave + 4 * half;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 v = {acc[t][4 * q + 0], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
                if (MASK) {
                    v.x = (bits >> (4 * q + 0)) & 1u ? v.x : 0.0f; v.y = (bits >> (4 * q + 1)) & 1u ? v.y : 0.0f;
                    v.z = (bits >> (4 * q + 2)) & 1u ? v.z : 0.0f; v.w = (bits >> (4 * q + 3)) & 1u ? v.w : 0.0f;
                }
                const half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                *reinterpret_cast<half4*>(Hw + t * 32 * kHS16 + 8 * q) = h;
#ifndef GF_PROBE_NO_SAVES
                if (row && ok) *reinterpret_cast<half4*>(row + 8 * q) = h;
#endif
                if constexpr (CS) {
                    if (ok) { part[4 * q] += v.x; part[4 * q + 1] += v.y; part[4 * q + 2] += v.z; part[4 * q + 3] += v.w; }
                }
            }
        }
    }
    if constexpr (CS) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = part[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (j == 0) colsum[32 * wave + 8 * (r >> 2) + 4 * half + (r & 3)] += v;
        }
    }
}

// acc[t][r] = sum_c z_c[sample t * 32 + j] * rows[c][32 wave + 8 q + 4 half + i]: a rank-NC product straight into accumulator layout
template <int NC>
__device__ __forceinline__ void rank_k_acc(const float* const (&z)[NC], const float* rows, int wave, int half, int j, floatx16 (&acc)[4], int nt) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (t < nt) {
            float zc[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) zc[c] = z[c][t * 32 + j];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float4 v = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const float4 w = *reinterpret_cast<const float4*>(rows + c * 128 + 32 * wave + 8 * q + 4 * half);
                    v.x = __builtin_fmaf(zc[c], w.x, v.x); v.y = __builtin_fmaf(zc[c], w.y, v.y);
                    v.z = __builtin_fmaf(zc[c], w.z, v.z); v.w = __builtin_fmaf(zc[c], w.w, v.w);
                }
                acc[t][4 * q] = v.x; acc[t][4 * q + 1] = v.y; acc[t][4 * q + 2] = v.z; acc[t][4 * q + 3] = v.w;
            }
        }
    }
}

// the real outputs of a narrow layer (features < NREAL) from accumulator layout into the fp32 scratch [128][64]
template <int NREAL>
__device__ __forceinline__ void scratch_store(float* scr, int wave, int lane, const floatx16 (&acc)[4], int nt) {
    const int half = lane >> 5, j = lane & 31;
    if (32 * wave >= NREAL) return;       // wave-uniform
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (t < nt) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4*>(scr + (t * 32 + j) * 64 + 32 * wave + 8 * q + 4 * half) = float4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        }
}

__device__ __forceinline__ void bwd_round16(const HeadArgs& a, const BwdArgs& u, const Smem& s, uint32_t Mv, uint32_t gbase, int nt, int wave, int lane_in,
                                            float* cs_hc1, float* cs_ha1 /* LDS [128] each */) {
    // The lane index is made opaque once per round: left visible, every lane-dependent part of the ~60 global addresses of a round (six
    // [M,128] outputs x 4 tiles, the level-major grid gradients, the masks) is loop-invariant over the chunk loop, gets hoisted in front of
    // it as 64-bit values and spilled (288 bytes of scratch per lane; a launch that touches scratch at all pays for it: NOTES 4.7).
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5, j = lane & 31;
    const uint32_t sI = (uint32_t)(wave * 32 + j);
    const bool tile_on = wave < nt;
    const bool valid = sI < Mv;
    const size_t pt = (size_t)gbase + (valid ? sI : 0u);
    _Float16* H16 = reinterpret_cast<_Float16*>(s.H);
    const _Float16* Hb = H16 + j * kHS16 + 8 * half;
    _Float16* Hw = H16 + j * kHS16 + 32 * wave + 4 * half;
    float* scr = reinterpret_cast<float*>(H16 + kPass * kHS16);            // [128][64] fp32
    const char* Ws = reinterpret_cast<const char*>(u.stream) + (size_t)wave * BH_TOTAL * 1024;
    uint32_t lane16 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(lane16));
    const gf::LevelMeta* meta = reinterpret_cast<const gf::LevelMeta*>(s.P + P_META);
    auto G16 = [](float* p) { return reinterpret_cast<_Float16*>(p); };
    floatx16 A[4];
    WPipe16 wp;
#pragma unroll
    for (int g = 0; g < kWAhead16; g++) wp.q[g] = load_group(Ws, g, lane16);

    // ---- per-sample scalars: d z_c = d rgb * rgb (1 - rgb), d h0 = d sigma * exp(clamp(h0)) -> LDS by sample (zeros beyond the list)
    if (tile_on && half == 0) {
        float gz[3] = {0.0f, 0.0f, 0.0f}, gh0 = 0.0f;
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; c++) { const float r = u.rgb[pt * 3 + c]; gz[c] = u.g_rgb[pt * 3 + c] * r * (1.0f - r); }
            gh0 = u.g_sigma[pt] * fminf(fmaxf(u.sigma[pt], 3.0590232e-7f), 3269017.37f);
            u.g_zc[pt * 3] = gz[0]; u.g_zc[pt * 3 + 1] = gz[1]; u.g_zc[pt * 3 + 2] = gz[2]; u.g_h0[pt] = gh0;
        }
        s.sx[sI] = gz[0]; s.sy[sI] = gz[1]; s.sz[sI] = gz[2]; s.sdt[sI] = gh0;
    }
    __syncthreads();
    // ---- d h_c1 = W_c2^T d z_c, masked (accumulator layout directly)
    {
        const float* const z[3] = {s.sx, s.sy, s.sz};
        rank_k_acc<3>(z, s.P + P_SMALL + gf::HS_COL2, wave, half, j, A, nt);
    }
    bwd16_store<true, true>(Hw, nullptr, u.m_hc1, gbase, Mv, wave, lane, A, nt, cs_hc1);
    __syncthreads();
    rows16_to_global<true>(H16, G16(u.g_hc1), gbase, Mv, wave * 64 + lane);
    // ---- d geo = W_c1[:, 16:144]^T d h_c1
    obw_zero<4>(A);
    obw16_mfma<BH_C1, 8, kHS16, BH_TOTAL>(wp, Ws, lane16, Hb, A, nt);
    __syncthreads();
    bwd16_store<false>(Hw, nullptr, nullptr, gbase, Mv, wave, lane, A, nt);
    __syncthreads();
    rows16_to_global<true>(H16, G16(u.g_geo), gbase, Mv, wave * 64 + lane);
    // ---- d h_s2 = W_s3[1:]^T d geo + d h0 (x) W_s3[0], masked
    {
        const float* const z[1] = {s.sdt};
        rank_k_acc<1>(z, s.P + P_SMALL + gf::HS_SIGROW, wave, half, j, A, nt);
    }
    obw16_mfma<BH_S3, 8, kHS16, BH_TOTAL>(wp, Ws, lane16, Hb, A, nt);
    __syncthreads();
    bwd16_store<true>(Hw, nullptr, u.m_hs2, gbase, Mv, wave, lane, A, nt);
    __syncthreads();
    rows16_to_global<true>(H16, G16(u.g_hs2), gbase, Mv, wave * 64 + lane);
    // ---- d h_s1 = W_s2^T d h_s2, masked
    obw_zero<4>(A);
    obw16_mfma<BH_S2, 8, kHS16, BH_TOTAL>(wp, Ws, lane16, Hb, A, nt);
    __syncthreads();
    bwd16_store<true>(Hw, nullptr, u.m_hs1, gbase, Mv, wave, lane, A, nt);
    __syncthreads();
    rows16_to_global<true>(H16, G16(u.g_hs1), gbase, Mv, wave * 64 + lane);
    // ---- [d f3 (sigma branch) | d f2] = W_s1^T d h_s1  (64 real outputs) -> fp32 scratch
    obw_zero<4>(A);
    obw16_mfma<BH_S1, 8, kHS16, BH_TOTAL>(wp, Ws, lane16, Hb, A, nt);
    scratch_store<64>(scr, wave, lane, A, nt);
    __syncthreads();
    // ---- 2-D lookup: d f2 out, input gradient in; ambient tail d z_a = (d amb + 0.5 d x2) (1 - amb^2) -> LDS by sample
    if (tile_on) {
        if (valid) {      // d f3 of the sigma branch: parked in its final place until the ambient branch's share arrives (registers: bwd_round does the same)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 v3 = *reinterpret_cast<const float4*>(scr + sI * 64 + 16 * half + 4 * q);
                *reinterpret_cast<float2*>(u.g_f3 + ((size_t)(8 * half + 2 * q) * u.M + pt) * 2) = float2{v3.x, v3.y};
                *reinterpret_cast<float2*>(u.g_f3 + ((size_t)(8 * half + 2 * q + 1) * u.M + pt) * 2) = float2{v3.z, v3.w};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float gf2[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v2 = *reinterpret_cast<const float4*>(scr + sI * 64 + 32 + 16 * half + 4 * q);
            gf2[4 * q] = v2.x; gf2[4 * q + 1] = v2.y; gf2[4 * q + 2] = v2.z; gf2[4 * q + 3] = v2.w;
        }
        float gamb[2] = {0.0f, 0.0f}, ambv[2] = {0.0f, 0.0f};
        if (valid) {
            gamb[0] = u.g_amb[pt * 2]; gamb[1] = u.g_amb[pt * 2 + 1];
            ambv[0] = u.amb[pt * 2]; ambv[1] = u.amb[pt * 2 + 1];
#pragma unroll
            for (int l = 0; l < 8; l++) *reinterpret_cast<float2*>(u.g_f2 + ((size_t)(8 * half + l) * u.M + pt) * 2) = float2{gf2[2 * l], gf2[2 * l + 1]};
            if (u.lvl_max) {
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const uint32_t m0 = __float_as_uint(gf2[2 * l]) & 0x7fffffffu, m1 = __float_as_uint(gf2[2 * l + 1]) & 0x7fffffffu;
                    atomicMax(&s.hist[16 + 8 * half + l], m0 > m1 ? m0 : m1);
                }
            }
        }
        const float x2[2] = {(ambv[0] + 1.0f) / 2.0f, (ambv[1] + 1.0f) / 2.0f};
        float dx[2];
        gf::encode8_grad2(a.amb_table, meta + 16 + half * 8, a.gridtype, a.interp, x2, gf2, dx);
        dx[0] += __shfl_xor(dx[0], 32);
        dx[1] += __shfl_xor(dx[1], 32);
        const float gza[2] = {valid ? (gamb[0] + 0.5f * dx[0]) * (1.0f - ambv[0] * ambv[0]) : 0.0f,
                              valid ? (gamb[1] + 0.5f * dx[1]) * (1.0f - ambv[1] * ambv[1]) : 0.0f};
        if (half == 0) {
            if (valid) { u.g_za[pt * 2] = gza[0]; u.g_za[pt * 2 + 1] = gza[1]; }
            s.st[sI] = gza[0]; s.ob[sI] = gza[1];
        }
    }
    __syncthreads();
    // ---- d h_a2 = W_a3^T d z_a, masked
    {
        const float* const z[2] = {s.st, s.ob};
        rank_k_acc<2>(z, s.P + P_SMALL + gf::HS_AMB3, wave, half, j, A, nt);
    }
    bwd16_store<true>(Hw, nullptr, u.m_ha2, gbase, Mv, wave, lane, A, nt);
    __syncthreads();
    rows16_to_global<true>(H16, G16(u.g_ha2), gbase, Mv, wave * 64 + lane);
    // ---- d h_a1 = W_a2^T d h_a2, masked
    obw_zero<4>(A);
    obw16_mfma<BH_A2, 8, kHS16, BH_TOTAL>(wp, Ws, lane16, Hb, A, nt);
    __syncthreads();
    bwd16_store<true, true>(Hw, nullptr, u.m_ha1, gbase, Mv, wave, lane, A, nt, cs_ha1);
    __syncthreads();
    rows16_to_global<true>(H16, G16(u.g_ha1), gbase, Mv, wave * 64 + lane);
    // ---- d f3 (ambient branch) = W_a1[:, :32]^T d h_a1  (32 real outputs) -> fp32 scratch, added to the sigma branch's share
    obw_zero<4>(A);
    obw16_mfma<BH_A1, 8, kHS16, BH_TOTAL>(wp, Ws, lane16, Hb, A, nt);
    scratch_store<32>(scr, wave, lane, A, nt);
    __syncthreads();
    if (tile_on && valid) {
        float gf3[16];
#pragma unroll
        for (int l = 0; l < 8; l++) {   // this lane's own store of two layers ago (same thread, same address: ordered)
            const float2 p = *reinterpret_cast<const float2*>(u.g_f3 + ((size_t)(8 * half + l) * u.M + pt) * 2);
            gf3[2 * l] = p.x; gf3[2 * l + 1] = p.y;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 v = *reinterpret_cast<const float4*>(scr + sI * 64 + 16 * half + 4 * q);
            gf3[4 * q] += v.x; gf3[4 * q + 1] += v.y; gf3[4 * q + 2] += v.z; gf3[4 * q + 3] += v.w;
        }
#pragma unroll
        for (int l = 0; l < 8; l++) *reinterpret_cast<float2*>(u.g_f3 + ((size_t)(8 * half + l) * u.M + pt) * 2) = float2{gf3[2 * l], gf3[2 * l + 1]};
        if (u.lvl_max) {
#pragma unroll
            for (int l = 0; l < 8; l++) {
                const uint32_t m0 = __float_as_uint(gf3[2 * l]) & 0x7fffffffu, m1 = __float_as_uint(gf3[2 * l + 1]) & 0x7fffffffu;
                atomicMax(&s.hist[8 * half + l], m0 > m1 ? m0 : m1);
            }
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(kThreads, 2) k_field_backward16(const HeadArgs a, const BwdArgs u) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)gf::HS_TOTAL; i += kThreads) s.P[P_SMALL + i] = a.head_pack[gf::HP_SMALL + i];
    if (tid < 32) {
        const uint32_t g = tid >> 4, l = tid & 15;
        gf::LevelMeta* m = reinterpret_cast<gf::LevelMeta*>(s.P + P_META) + tid;
        *m = g ? gf::make_level_meta<2>(a.lv2.scale[l], a.lv2.resolution[l], a.amb_offsets, l, a.gridtype)
               : gf::make_level_meta<3>(a.lv3.scale[l], a.lv3.resolution[l], a.pos_offsets, l, a.gridtype);
    }
    if (tid < 32) s.hist[tid] = 0;
    const uint32_t chunks = (u.M + kPass - 1) / kPass;
    float* cs_hc1 = s.p_dx;   // [128] each: the pool-direction arrays are free in this kernel (the sample staging arrays carry the per-sample scalars)
    float* cs_ha1 = s.p_dy;
    if (tid < 128) { cs_hc1[tid] = 0.0f; cs_ha1[tid] = 0.0f; }
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        __syncthreads();
        const uint32_t gbase = chunk * kPass;
        const uint32_t left = u.M - gbase;
        const uint32_t Mv = left < (uint32_t)kPass ? left : (uint32_t)kPass;
        bwd_round16(a, u, s, Mv, gbase, __builtin_amdgcn_readfirstlane((int)((Mv + 31) / 32)), wave, lane, cs_hc1, cs_ha1);
    }
    __syncthreads();
    if (tid < 128) {
        atomicAdd(&u.s_hc1[tid], cs_hc1[tid]);
        atomicAdd(&u.s_ha1[tid], cs_ha1[tid]);
    }
    __syncthreads();
    if (u.lvl_max && tid < 32 && s.hist[tid]) atomicMax(&u.lvl_max[tid], s.hist[tid]);
}

// ---------------------------------------------------------------------------------------------------- frame setup
// Pinhole ray of pixel n: pixel centres at +0.5, row-major pixels (utils.py:296-363), same operation order as the torch code, no
// contraction.  ONE definition for k_frame_init (pose mode) and k_pinhole_rays (gf_pinhole_rays: the same rays as tensors), so a frame
// rendered from explicit gf_pinhole_rays rays sees the bits a pose-mode frame generates for itself.
__device__ __forceinline__ void pinhole_ray(uint32_t n, uint32_t img_w, float fx, float fy, float cx, float cy, const float* pose,
                                            float& dx, float& dy, float& dz) {
#pragma clang fp contract(off)
    const uint32_t row = n / img_w, col = n - row * img_w;
    const float xs = ((float)col + 0.5f - cx) / fx, ys = ((float)row + 0.5f - cy) / fy, zs = 1.0f;
    const float nrm = sqrtf(xs * xs + ys * ys + zs * zs);
    const float ux = xs / nrm, uy = ys / nrm, uz = zs / nrm;
    dx = ux * pose[0] + uy * pose[1] + uz * pose[2];
    dy = ux * pose[4] + uy * pose[5] + uz * pose[6];
    dz = ux * pose[8] + uy * pose[9] + uz * pose[10];
}

struct PinholeArgs { float pose[12]; float fx, fy, cx, cy; uint32_t img_w, N; float *rays_o, *rays_d; };
__global__ void __launch_bounds__(256) k_pinhole_rays(const PinholeArgs a) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= a.N) return;
    float dx, dy, dz;
    pinhole_ray(n, a.img_w, a.fx, a.fy, a.cx, a.cy, a.pose, dx, dy, dz);
    a.rays_d[(size_t)n * 3] = dx; a.rays_d[(size_t)n * 3 + 1] = dy; a.rays_d[(size_t)n * 3 + 2] = dz;
    a.rays_o[(size_t)n * 3] = a.pose[3]; a.rays_o[(size_t)n * 3 + 1] = a.pose[7]; a.rays_o[(size_t)n * 3 + 2] = a.pose[11];
}

struct InitArgs {
    gf::MarchParams mp;
    const float* rays_o_in; const float* rays_d_in;  // explicit rays, or NULL
    float pose[12]; float fx, fy, cx, cy; uint32_t img_w;
    const float* aabb; float min_near;
    float occ[6]; uint32_t has_occ;
    const float* noise;   // [N] or NULL: first-iteration jitter of perturb=True (renderer.py:338-342, raymarching.cu:851)
    float *rays_o, *rays_d, *nears, *fars, *far_occ, *rays_t, *weights_sum, *depth, *image;
    int* hit_list; uint32_t* ctrl; uint32_t N;
};

// 1024-lane workgroups, ONE pair of control-block atomics per workgroup.  With one pair per wave (8 192 atomics on two addresses per frame)
// the launch took 45 us whatever else it did -- 52 us with the walk AND the stores compiled out (profiles/round3/r3l_init_ab.txt): L2
// retires same-address atomics at about one per 10 ns, and that serial chain was the kernel.
#ifndef GF_INIT_THREADS
#define GF_INIT_THREADS 1024
#endif
constexpr int kInitThreads = GF_INIT_THREADS;
__global__ void __launch_bounds__(kInitThreads) k_frame_init(const InitArgs a) {
    __shared__ uint32_t wave_hits[kInitThreads / 64], wave_miss[kInitThreads / 64], wg_base;
    const uint32_t n = blockIdx.x * kInitThreads + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool hit = false, miss = false;
    if (n < a.N) {
        float ox, oy, oz, dx, dy, dz;
        if (a.rays_o_in) {
            ox = a.rays_o_in[(size_t)n * 3]; oy = a.rays_o_in[(size_t)n * 3 + 1]; oz = a.rays_o_in[(size_t)n * 3 + 2];
            dx = a.rays_d_in[(size_t)n * 3]; dy = a.rays_d_in[(size_t)n * 3 + 1]; dz = a.rays_d_in[(size_t)n * 3 + 2];
        } else {
            pinhole_ray(n, a.img_w, a.fx, a.fy, a.cx, a.cy, a.pose, dx, dy, dz);
            ox = a.pose[3]; oy = a.pose[7]; oz = a.pose[11];
        }
        float near, far;
        gf::near_far_from_aabb_1(ox, oy, oz, dx, dy, dz, a.aabb, a.min_near, near, far);
        a.nears[n] = near;
        a.fars[n] = far;
        a.weights_sum[n] = 0.0f;
        a.depth[n] = 0.0f;
        a.image[(size_t)n * 3] = 0.0f; a.image[(size_t)n * 3 + 1] = 0.0f; a.image[(size_t)n * 3 + 2] = 0.0f;
        // Marching may stop where the ray leaves the (slightly padded) box around the occupied cells: every position beyond
        // it lies in an unoccupied cell, so the reference's marcher would only skip from there to `far`.
        float far_m = far;
        if (a.has_occ) {
            float on, of;
            gf::near_far_from_aabb_1(ox, oy, oz, dx, dy, dz, a.occ, 0.0f, on, of);
            if (of == FLT_MAX) far_m = near;              // misses the occupied region: no sample on this ray
            else far_m = fminf(far, of);
        }
        // march through empty space to the first occupied sample; the field kernel restarts the marcher exactly there.  (Round 3 tried taking
        // the occupancy-bit loads off the dependency chain -- 4 or 8 visits computed ahead assuming "empty", their bytes loaded together,
        // answers checked in order: same result bit for bit, but 36-38 us against 33-34: the kernel is bound by instruction issue, and the
        // visits computed beyond the hit are pure cost.  profiles/round3/r3l_init_ab.txt.)
        float t = near, t_first = near;
        const uint32_t got = gf::march_ray(a.mp, ox, oy, oz, dx, dy, dz, far_m, a.noise ? a.noise[n] : 0.0f, 1u, t,
                                           [&](uint32_t, float, float, float, float, float, float t_at) { t_first = t_at; });
        hit = got > 0;
        miss = !hit;
        if (hit) {   // marcher state only for the rays the field kernel will pick up (~35 % of a head frame): 20-32 B less per missed ray
            a.rays_d[(size_t)n * 3] = dx; a.rays_d[(size_t)n * 3 + 1] = dy; a.rays_d[(size_t)n * 3 + 2] = dz;
            if (a.rays_o_in) { a.rays_o[(size_t)n * 3] = ox; a.rays_o[(size_t)n * 3 + 1] = oy; a.rays_o[(size_t)n * 3 + 2] = oz; }
            a.far_occ[n] = far_m;
            a.rays_t[n] = t_first;
        }
    }
    // rays with a sample -> hit list (order is irrelevant: rays are independent); rays without one terminate at index 1
    const unsigned long long hm = __ballot(hit), mm = __ballot(miss);
    const int wave = threadIdx.x >> 6;
    if (lane == 0) { wave_hits[wave] = (uint32_t)__popcll(hm); wave_miss[wave] = (uint32_t)__popcll(mm); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t nh = 0, nm = 0;
#pragma unroll
        for (int w = 0; w < kInitThreads / 64; w++) { nh += wave_hits[w]; nm += wave_miss[w]; }
        wg_base = nh ? atomicAdd(&a.ctrl[gf::kCtrlNHit], nh) : 0u;
        if (nm) atomicAdd(&a.ctrl[gf::kCtrlHist + 1], nm);
    }
    __syncthreads();
    if (hit) {
        uint32_t base = wg_base;
        for (int w = 0; w < wave; w++) base += wave_hits[w];
        a.hit_list[base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (int)n;
    }
}

// head-only tail of NeRFRenderer.render (renderer.py:354-364): background blend, clamp, depth normalisation
__global__ void __launch_bounds__(256) k_head_finish(uint32_t N, const float* __restrict__ image, const float* __restrict__ weights_sum,
                                                     const float* __restrict__ depth, const float* __restrict__ nears,
                                                     const float* __restrict__ fars, const float* __restrict__ bg,
                                                     float* __restrict__ out_rgb, float* __restrict__ out_depth,
                                                     uint8_t* __restrict__ out_rgb8) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float ws = weights_sum[n];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v = image[(size_t)n * 3 + c] + (1 - ws) * bg[(size_t)n * 3 + c];
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        out_rgb[(size_t)n * 3 + c] = v;
        if (out_rgb8) out_rgb8[(size_t)n * 3 + c] = (uint8_t)(v * 255.0f);
    }
    out_depth[n] = fmaxf(depth[n] - nears[n], 0.0f) / (fars[n] - nears[n]);
}

int check_frame(const gf_frame_t* f) {
    if (!f) return gf_set_error(GF_ERR_INVALID, "frame: null descriptor");
    if (f->n_rays == 0) return gf_set_error(GF_ERR_INVALID, "frame: n_rays == 0");
    if (!f->workspace || !f->aabb || !f->bitfield || !f->pos_table || !f->pos_offsets || !f->amb_table || !f->amb_offsets ||
        !f->head_pack || !f->amb_bias)
        return gf_set_error(GF_ERR_INVALID, "frame: null pointer in the head description");
    if ((f->rays_o == nullptr) != (f->rays_d == nullptr)) return gf_set_error(GF_ERR_INVALID, "frame: rays_o and rays_d must both be given or both NULL");
    if (!f->rays_o && (uint64_t)f->img_h * f->img_w != f->n_rays) return gf_set_error(GF_ERR_INVALID, "frame: img_h*img_w != n_rays");
    if (f->max_steps == 0 || f->cascade == 0 || f->grid_size == 0 || f->grid_size > 1024) return gf_set_error(GF_ERR_INVALID, "frame: bad marcher configuration");
    if (f->max_steps > gf::kMaxSteps) return gf_set_error(GF_ERR_UNSUPPORTED, "frame: max_steps > %u (the reference's own default and viewer maximum)", gf::kMaxSteps);
    if (f->gridtype > 1 || f->interp > 1) return gf_set_error(GF_ERR_INVALID, "frame: gridtype/interp must be 0 or 1");
    if (f->precision > 2) return gf_set_error(GF_ERR_INVALID, "frame: precision must be 0 (fp32), 1 (fast) or 2 (split)");
    if (f->precision == 1 && !f->head_pack16) return gf_set_error(GF_ERR_INVALID, "frame: precision = 1 needs head_pack16");
    if (f->precision == 2 && !f->head_pack_split) return gf_set_error(GF_ERR_INVALID, "frame: precision = 2 needs head_pack_split");
    return GF_OK;
}

// The control block is cleared by a one-workgroup kernel of our own (round 5): hipMemsetAsync of these few hundred bytes reached the GPU as TWO
// runtime fill kernels of 256 workgroups each, 4.6 + 4.9 us of every frame's stream (rocprofv3 kernel trace, profiles/round5/).
__global__ void __launch_bounds__(256) k_ctrl_clear(uint32_t* __restrict__ ctrl, uint32_t words) {
    for (uint32_t i = threadIdx.x; i < words; i += 256) ctrl[i] = 0u;
}

#ifdef GF_CUMASK
// A/B build only (tools/cumask_ab.py): GF_CUMASK_RESERVE = R keeps the persistent head grids OFF R compute units, so the
// small kernels of the frames in flight (k_frame_init, the condition encoder, the torso launches -- all still on the caller's unmasked stream)
// always find R CUs with free registers.  The two phase kernels go to a stream created with a CU mask (one per caller stream), ordered with
// the caller's by events; their grid is 2 workgroups per CU they may use.  GF_CUMASK_HIGH=1 takes the R HIGHEST mask bits instead of the lowest.
struct MaskedStream { hipStream_t caller, head; hipEvent_t e_init, e_head; };
static MaskedStream g_masked[16];
static int g_n_masked = 0, g_reserve = -1;
static MaskedStream* masked_stream_for(hipStream_t s) {
    if (g_reserve < 0) { const char* e = getenv("GF_CUMASK_RESERVE"); g_reserve = e ? atoi(e) : 0; if (g_reserve < 0 || g_reserve > 128) g_reserve = 0; }
    if (g_reserve == 0) return nullptr;
    for (int i = 0; i < g_n_masked; i++) if (g_masked[i].caller == s) return &g_masked[i];
    if (g_n_masked == 16) return nullptr;
    MaskedStream& m = g_masked[g_n_masked];
    uint32_t mask[8];
    for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
    const char* hi = getenv("GF_CUMASK_HIGH");
    for (int b = 0; b < g_reserve; b++) { const int bit = (hi && atoi(hi)) ? 255 - b : b; mask[bit >> 5] &= ~(1u << (bit & 31)); }
    if (hipExtStreamCreateWithCUMask(&m.head, 8, mask) != hipSuccess) return nullptr;
    (void)hipEventCreateWithFlags(&m.e_init, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&m.e_head, hipEventDisableTiming);
    m.caller = s;
    g_n_masked++;
    return &m;
}
#endif

int launch_head(const gf_frame_t* f, hipStream_t s, hipEvent_t* ev /* nullable: 6 events around the two phase kernels (0..3) and k_frame_init (4, 5) */) {
    const gf::FrameWs w = gf::carve_workspace(f->workspace, f->n_rays);
    const uint32_t N = f->n_rays;
    hipLaunchKernelGGL(k_ctrl_clear, dim3(1), dim3(256), 0, s, w.ctrl, gf::ctrl_words_used(f->max_steps));

    InitArgs ia;
    gf::fill_march_params(ia.mp, f->bitfield, f->bound, f->dt_gamma, f->max_steps, f->cascade, f->grid_size);
    ia.rays_o_in = f->rays_o; ia.rays_d_in = f->rays_d;
    for (int i = 0; i < 12; i++) ia.pose[i] = f->pose[i];
    ia.fx = f->intrinsics[0]; ia.fy = f->intrinsics[1]; ia.cx = f->intrinsics[2]; ia.cy = f->intrinsics[3];
    ia.img_w = f->img_w ? f->img_w : 1;
    ia.aabb = f->aabb; ia.min_near = f->min_near;
    ia.has_occ = f->has_occ_aabb;
    ia.noise = f->perturb_noise;
    // pad the box by a hundredth of a cell: the slab test and the marcher's o + t*d round differently
    const float pad = 0.01f * 2.0f * f->bound / (float)f->grid_size;
    for (int i = 0; i < 3; i++) {
        // positions are clamped to +-bound before the cell lookup (raymarching.cu:883-885): a side of the box that reaches the
        // bound is open-ended, because everything beyond it maps onto its boundary cells
        const float cell = 2.0f * f->bound / (float)f->grid_size;
        ia.occ[i] = f->occ_aabb[i] <= -f->bound + cell ? -1e30f : f->occ_aabb[i] - pad;
        ia.occ[3 + i] = f->occ_aabb[3 + i] >= f->bound - cell ? 1e30f : f->occ_aabb[3 + i] + pad;
    }
    ia.rays_o = w.rays_o; ia.rays_d = w.rays_d; ia.nears = w.nears; ia.fars = w.fars; ia.far_occ = w.far_occ; ia.rays_t = w.rays_t;
    ia.weights_sum = w.weights_sum; ia.depth = w.depth; ia.image = w.image; ia.hit_list = w.alive_b; ia.ctrl = w.ctrl; ia.N = N;
    if (ev) (void)hipEventRecord(ev[4], s);
    hipLaunchKernelGGL(k_frame_init, dim3(gf_div_up(N, (uint32_t)kInitThreads)), dim3(kInitThreads), 0, s, ia);
    if (ev) (void)hipEventRecord(ev[5], s);

    HeadArgs ha;
    ha.mp = ia.mp;
    if (gf::fill_grid_levels(ha.lv3, 16, f->pos_S, f->base_res) || gf::fill_grid_levels(ha.lv2, 16, f->amb_S, f->base_res))
        return gf_set_error(GF_ERR_INVALID, "frame: bad grid levels");
    ha.pos_table = f->pos_table; ha.pos_offsets = f->pos_offsets; ha.amb_table = f->amb_table; ha.amb_offsets = f->amb_offsets;
    ha.head_pack = f->head_pack; ha.amb_bias = f->amb_bias;
    ha.rays_o = w.rays_o; ha.rays_d = w.rays_d; ha.far_occ = w.far_occ;
    ha.rays_t = w.rays_t; ha.weights_sum = w.weights_sum; ha.depth = w.depth; ha.image = w.image;
    ha.survivors = w.alive_a;
    ha.ctrl = w.ctrl; ha.N = N; ha.max_steps = f->max_steps; ha.gridtype = f->gridtype; ha.interp = f->interp;
    ha.T_thresh = f->T_thresh; ha.bound = f->bound;
    ha.pose_mode = f->rays_o == nullptr;
    ha.cam_o[0] = f->pose[3]; ha.cam_o[1] = f->pose[7]; ha.cam_o[2] = f->pose[11];
#ifdef GF_TRACE
    ha.trace = g_trace_buf;
    ha.spans = g_span_buf;
#endif
#ifdef GF_DIAG
    ha.diag = nullptr; ha.diag_stride = 0; ha.diag_tag = ++g_diag_tag;
    for (auto& e : g_diag_reg)
        if (e.ws == f->workspace) { ha.diag = e.buf; ha.diag_stride = e.stride; e.last_tag = ha.diag_tag; }
    ha.poison = g_diag_cfg[0]; ha.poison_round = g_diag_cfg[1]; ha.pool_cap_override = g_diag_cfg[3];
#endif

    const uint32_t mode = f->precision;
    ha.head_pack16 = f->head_pack16;
    ha.head_pack_split = f->head_pack_split;
    static GfLdsAttr lds[3];
    {
        const void* fn = mode == 1 ? reinterpret_cast<const void*>(k_head_phase<1>)
                       : mode == 2 ? reinterpret_cast<const void*>(k_head_phase<2>) : reinterpret_cast<const void*>(k_head_phase<0>);
        if (const int e = gf_raise_lds_limit(lds[mode], fn, kSmemBytes, "frame")) return e;
    }
    // persistent grid: 2 workgroups per CU x 256 CUs, never more workgroups than pools' worth of rays
    const uint32_t pools = gf_div_up(N, (uint32_t)kPool);
    uint32_t grid = pools < 512u ? pools : 512u;
#ifdef GF_TRACE
    if (const char* e = getenv("GF_HEAD_GRID")) { const uint32_t g = (uint32_t)atoi(e); if (g >= 1 && g <= 512) grid = g; }   // timeline experiments
#endif
#ifdef GF_DIAG
    if (g_diag_cfg[2] >= 1 && g_diag_cfg[2] <= 512) grid = g_diag_cfg[2];
#endif
    hipStream_t hs = s;
#ifdef GF_CUMASK
    MaskedStream* ms = ev ? nullptr : masked_stream_for(s);
    if (ms) {
        hs = ms->head;
        const uint32_t g = 2u * (256u - (uint32_t)g_reserve);
        if (grid > g) grid = g;
        (void)hipEventRecord(ms->e_init, s);
        (void)hipStreamWaitEvent(hs, ms->e_init, 0);
    }
#endif
    for (uint32_t phase = 0; phase < 2; phase++) {
        ha.phase = phase;
        ha.queue = phase ? w.alive_a : w.alive_b;
        if (ev) (void)hipEventRecord(ev[2 * phase], s);
        if (mode == 1) hipLaunchKernelGGL(k_head_phase<1>, dim3(grid), dim3(kThreads), kSmemBytes, hs, ha);
        else if (mode == 2) hipLaunchKernelGGL(k_head_phase<2>, dim3(grid), dim3(kThreads), kSmemBytes, hs, ha);
        else hipLaunchKernelGGL(k_head_phase<0>, dim3(grid), dim3(kThreads), kSmemBytes, hs, ha);
        if (ev) (void)hipEventRecord(ev[2 * phase + 1], s);
    }
#ifdef GF_CUMASK
    if (ms) {
        (void)hipEventRecord(ms->e_head, hs);
        (void)hipStreamWaitEvent(s, ms->e_head, 0);
    }
#endif
    return gf_check_launch("render_head");
}

}  // namespace

#ifdef GF_TRACE
// trace build only: device buffer of 2 * kTraceWGs * kTraceRounds * kTraceSlots uint32 (phase, workgroup, round, slot)
GF_EXPORT void gf_trace_set(void* dev_buf) { g_trace_buf = reinterpret_cast<uint32_t*>(dev_buf); }
// device buffer of 2 * 512 * 4 uint64: per phase and workgroup {start tick, end tick, rounds, XCC id}
GF_EXPORT void gf_trace_set_spans(void* dev_buf) { g_span_buf = reinterpret_cast<unsigned long long*>(dev_buf); }
GF_EXPORT uint32_t gf_trace_dims(uint32_t which) { return which == 0 ? kTraceWGs : which == 1 ? kTraceRounds : kTraceSlots; }
#endif

#ifdef GF_DIAG
// diag build only: per-sample records of the frames rendered into `workspace` go to `dev_buf` ([n_rays][stride][gf_diag_words()] words)
GF_EXPORT int gf_diag_register(uint32_t slot, const void* workspace, void* dev_buf, uint32_t stride) {
    if (slot >= 8) return -1;
    g_diag_reg[slot] = {workspace, reinterpret_cast<float*>(dev_buf), stride, 0};
    return 0;
}
GF_EXPORT uint32_t gf_diag_last_tag(uint32_t slot) { return slot < 8 ? g_diag_reg[slot].last_tag : 0; }
GF_EXPORT uint32_t gf_diag_words(void) { return kDiagWords; }
GF_EXPORT void gf_diag_config(uint32_t poison, uint32_t poison_round, uint32_t grid, uint32_t pool_cap) {
    g_diag_cfg[0] = poison; g_diag_cfg[1] = poison_round; g_diag_cfg[2] = grid; g_diag_cfg[3] = pool_cap;
}
#endif

GF_EXPORT uint64_t gf_frame_workspace_bytes(uint32_t n_rays) { return gf::carve_workspace(reinterpret_cast<void*>(uintptr_t(1) << 20), n_rays).bytes; }

GF_EXPORT uint32_t gf_frame_ctrl_words(void) { return gf::kCtrlWords; }

// Byte offset of the control block inside the workspace (see frame.hpp for the word layout).
GF_EXPORT uint64_t gf_frame_ctrl_offset(uint32_t n_rays) {
    char* const base = reinterpret_cast<char*>(uintptr_t(1) << 20);
    const gf::FrameWs w = gf::carve_workspace(base, n_rays);
    return (uint64_t)((char*)w.ctrl - base);
}

// NeRFRenderer.update_extra_state, field queries (renderer.py:232-246): sigma * density_scale of every cell of every cascade into
// tmp_grid [cascade][grid_size^3] (Morton order).  Uses f's tables, packed head weights, amb_bias, bound / cascade / grid_size / gridtype /
// interp / level scales; noise_or_null = [cascade][grid_size^3][3] U[0,1) jitter in meshgrid order (x slowest), NULL = cell centres.
GF_EXPORT int gf_grid_density(const gf_frame_t* f, const float* noise_or_null, float density_scale, float* tmp_grid, void* stream) {
    if (!f || !tmp_grid) return gf_set_error(GF_ERR_INVALID, "grid_density: null pointer");
    if (!f->pos_table || !f->pos_offsets || !f->amb_table || !f->amb_offsets || !f->head_pack || !f->amb_bias)
        return gf_set_error(GF_ERR_INVALID, "grid_density: null pointer in the field description");
    if (f->cascade == 0 || f->grid_size == 0 || f->grid_size > 1024 || f->gridtype > 1 || f->interp > 1)
        return gf_set_error(GF_ERR_INVALID, "grid_density: bad grid configuration");
    HeadArgs ha = {};
    if (gf::fill_grid_levels(ha.lv3, 16, f->pos_S, f->base_res) || gf::fill_grid_levels(ha.lv2, 16, f->amb_S, f->base_res))
        return gf_set_error(GF_ERR_INVALID, "grid_density: bad grid levels");
    ha.pos_table = f->pos_table; ha.pos_offsets = f->pos_offsets; ha.amb_table = f->amb_table; ha.amb_offsets = f->amb_offsets;
    ha.head_pack = f->head_pack; ha.amb_bias = f->amb_bias;
    ha.gridtype = f->gridtype; ha.interp = f->interp; ha.bound = f->bound;
    GridArgs ga = {noise_or_null, tmp_grid, f->cascade, f->grid_size, density_scale};
    static GfLdsAttr lds;
    if (const int e = gf_raise_lds_limit(lds, reinterpret_cast<const void*>(k_grid_density), kSmemBytes, "grid_density")) return e;
    const uint64_t total = (uint64_t)f->cascade * f->grid_size * f->grid_size * f->grid_size;
    if (total >= (1ull << 32)) return gf_set_error(GF_ERR_UNSUPPORTED, "grid_density: more than 2^32 cells");
    const uint32_t chunks = (uint32_t)((total + kPass - 1) / kPass);
    hipLaunchKernelGGL(k_grid_density, dim3(chunks < 512u ? chunks : 512u), dim3(kThreads), kSmemBytes, gf_stream(stream), ha, ga);
    return gf_check_launch("grid_density");
}

// RADNeRF.forward (radnerf.py:73-105) on a dense point list, inference arithmetic (no autograd): sigma [M], rgb [M,3], ambient [M,2]
// (tanh output; may be NULL).  Uses f's tables, level scales, packed head weights, amb_bias, bound, gridtype, interp.
// col_bias_or_null: [128] W_color0[:, 144:148] @ individual_code in accumulator order (gf_clayout_perm), NULL = the code packed into head_pack.
static int field_forward_impl(const gf_frame_t* f, const float* xyz, const float* dirs, uint32_t M, const float* col_bias_or_null,
                              float* sigma, float* rgb, float* ambient_or_null, const gf_field_saves_t* saves, void* stream) {
    if (M == 0) return GF_OK;
    if (!f || !xyz || !dirs || !sigma || !rgb) return gf_set_error(GF_ERR_INVALID, "field_forward: null pointer");
    if (!f->pos_table || !f->pos_offsets || !f->amb_table || !f->amb_offsets || !f->head_pack || !f->amb_bias)
        return gf_set_error(GF_ERR_INVALID, "field_forward: null pointer in the field description");
    if (f->gridtype > 1 || f->interp > 1) return gf_set_error(GF_ERR_INVALID, "field_forward: gridtype/interp must be 0 or 1");
    HeadArgs ha = {};
    if (gf::fill_grid_levels(ha.lv3, 16, f->pos_S, f->base_res) || gf::fill_grid_levels(ha.lv2, 16, f->amb_S, f->base_res))
        return gf_set_error(GF_ERR_INVALID, "field_forward: bad grid levels");
    ha.pos_table = f->pos_table; ha.pos_offsets = f->pos_offsets; ha.amb_table = f->amb_table; ha.amb_offsets = f->amb_offsets;
    ha.head_pack = f->head_pack; ha.amb_bias = f->amb_bias;
    ha.gridtype = f->gridtype; ha.interp = f->interp; ha.bound = f->bound;
    PointArgs pa = {xyz, dirs, col_bias_or_null, sigma, rgb, ambient_or_null, M, {}};
    if (saves) {
        if (M > (1u << 23) - 128u) return gf_set_error(GF_ERR_UNSUPPORTED, "field_forward_train: the saves are addressed with 32-bit byte offsets: M < 2^23");
        if (!saves->f3 || !saves->ha1 || !saves->ha2 || !saves->f2 || !saves->hs1 || !saves->hs2 || !saves->geo || !saves->hc1)
            return gf_set_error(GF_ERR_INVALID, "field_forward_train: null save buffer");
        pa.sv = {saves->f3, saves->ha1, saves->ha2, saves->f2, saves->hs1, saves->hs2, saves->geo, saves->hc1,
                 saves->m_ha1, saves->m_ha2, saves->m_hs1, saves->m_hs2, saves->m_hc1, saves->sh};
    }
    static GfLdsAttr lds[2];
    if (const int e = gf_raise_lds_limit(lds[0], reinterpret_cast<const void*>(k_field_points<false>), kSmemBytes, "field_forward")) return e;
    if (const int e = gf_raise_lds_limit(lds[1], reinterpret_cast<const void*>(k_field_points<true>), kSmemBytes, "field_forward")) return e;
    const uint32_t chunks = gf_div_up(M, (uint32_t)kPass);
    const dim3 grid(chunks < 512u ? chunks : 512u);
    if (saves) hipLaunchKernelGGL(k_field_points<true>, grid, dim3(kThreads), kSmemBytes, gf_stream(stream), ha, pa);
    else hipLaunchKernelGGL(k_field_points<false>, grid, dim3(kThreads), kSmemBytes, gf_stream(stream), ha, pa);
    return gf_check_launch("field_forward");
}

// The training forward on the f16 tier: f->head_pack16 (gf_head_pack16) beside head_pack (its fp32 skinny rows and constant bias are read too);
// the nine matrices of `saves` are binary16, the masks as in gf_field_forward_train.
GF_EXPORT int gf_field_forward_train16(const gf_frame_t* f, const float* xyz, const float* dirs, uint32_t M, const float* col_bias_or_null, float* sigma,
                                       float* rgb, float* ambient, const gf_field_saves_t* saves, void* stream) {
    if (M == 0) return GF_OK;
    if (!f || !xyz || !dirs || !sigma || !rgb || !ambient || !saves) return gf_set_error(GF_ERR_INVALID, "field_forward_train16: null pointer");
    if (!f->pos_table || !f->pos_offsets || !f->amb_table || !f->amb_offsets || !f->head_pack || !f->head_pack16 || !f->amb_bias)
        return gf_set_error(GF_ERR_INVALID, "field_forward_train16: null pointer in the field description (head_pack16 is needed)");
    if (f->gridtype > 1 || f->interp > 1) return gf_set_error(GF_ERR_INVALID, "field_forward_train16: gridtype/interp must be 0 or 1");
    if (!saves->f3 || !saves->ha1 || !saves->ha2 || !saves->f2 || !saves->hs1 || !saves->hs2 || !saves->geo || !saves->hc1 || !saves->sh ||
        !saves->m_ha1 || !saves->m_ha2 || !saves->m_hs1 || !saves->m_hs2 || !saves->m_hc1)
        return gf_set_error(GF_ERR_INVALID, "field_forward_train16: null save buffer");
    if (M > (1u << 24) - 128u) return gf_set_error(GF_ERR_UNSUPPORTED, "field_forward_train16: the saves are addressed with 32-bit byte offsets: M < 2^24");
    HeadArgs ha = {};
    if (gf::fill_grid_levels(ha.lv3, 16, f->pos_S, f->base_res) || gf::fill_grid_levels(ha.lv2, 16, f->amb_S, f->base_res))
        return gf_set_error(GF_ERR_INVALID, "field_forward_train16: bad grid levels");
    ha.pos_table = f->pos_table; ha.pos_offsets = f->pos_offsets; ha.amb_table = f->amb_table; ha.amb_offsets = f->amb_offsets;
    ha.head_pack = f->head_pack; ha.head_pack16 = f->head_pack16; ha.amb_bias = f->amb_bias;
    ha.gridtype = f->gridtype; ha.interp = f->interp; ha.bound = f->bound;
    auto h = [](float* p) { return reinterpret_cast<_Float16*>(p); };
    PointArgs16 pa = {xyz, dirs, col_bias_or_null, sigma, rgb, ambient, M,
                      {h(saves->f3), h(saves->ha1), h(saves->ha2), h(saves->f2), h(saves->hs1), h(saves->hs2), h(saves->geo), h(saves->hc1), h(saves->sh),
                       saves->m_ha1, saves->m_ha2, saves->m_hs1, saves->m_hs2, saves->m_hc1}};
    static GfLdsAttr lds;
    if (const int e = gf_raise_lds_limit(lds, reinterpret_cast<const void*>(k_field_points16), kSmemBytes, "field_forward_train16")) return e;
    const uint32_t chunks = gf_div_up(M, (uint32_t)kPass);
    hipLaunchKernelGGL(k_field_points16, dim3(chunks < 512u ? chunks : 512u), dim3(kThreads), kSmemBytes, gf_stream(stream), ha, pa);
    return gf_check_launch("field_forward_train16");
}

GF_EXPORT int gf_field_forward(const gf_frame_t* f, const float* xyz, const float* dirs, uint32_t M, const float* col_bias_or_null,
                               float* sigma, float* rgb, float* ambient_or_null, void* stream) {
    return field_forward_impl(f, xyz, dirs, M, col_bias_or_null, sigma, rgb, ambient_or_null, nullptr, stream);
}

// The same launch for the training forward: additionally every layer's activations, which the backward pass (geneface_amd/train_field.py)
// turns into weight / table / input gradients.
GF_EXPORT int gf_field_forward_train(const gf_frame_t* f, const float* xyz, const float* dirs, uint32_t M, const float* col_bias_or_null,
                                     float* sigma, float* rgb, float* ambient, const gf_field_saves_t* saves, void* stream) {
    if (!saves || !ambient) return gf_set_error(GF_ERR_INVALID, "field_forward_train: null pointer");
    return field_forward_impl(f, xyz, dirs, M, col_bias_or_null, sigma, rgb, ambient, saves, stream);
}


// The dX chain of the training backward (see k_field_backward).  `bwd_stream`: gf_field_bwd_stream_floats() floats, the six transposed
// 128 x 128 weight blocks as A-operand streams; f supplies the ambient table / offsets / level scales, head_pack (its skinny rows) and
// gridtype / interp.  All [M, *] outputs are fully written for the M points.
GF_EXPORT uint32_t gf_field_bwd_stream_floats(void) { return 4u * BG_TOTAL * 256u; }
GF_EXPORT uint32_t gf_field_bwd16_stream_halves(void) { return 4u * BH_TOTAL * 512u; }

GF_EXPORT int gf_field_backward(const gf_frame_t* f, const float* bwd_stream, uint32_t M, const gf_field_grads_t* g, void* stream) {
    if (M == 0) return GF_OK;
    if (!f || !bwd_stream || !g) return gf_set_error(GF_ERR_INVALID, "field_backward: null pointer");
    if (!f->amb_table || !f->amb_offsets || !f->pos_offsets || !f->head_pack) return gf_set_error(GF_ERR_INVALID, "field_backward: null pointer in the field description");
    const void* need[] = {g->g_sigma, g->g_rgb, g->g_amb, g->sigma, g->rgb, g->amb, g->m_hc1, g->m_hs2, g->m_hs1, g->m_ha2, g->m_ha1, g->g_zc, g->g_h0, g->g_za,
                          g->g_hc1, g->g_geo, g->g_hs2, g->g_hs1, g->g_ha2, g->g_ha1, g->g_f3, g->g_f2, g->s_hc1, g->s_ha1};
    for (const void* p : need) if (!p) return gf_set_error(GF_ERR_INVALID, "field_backward: null buffer");
    if (f->gridtype > 1 || f->interp > 1) return gf_set_error(GF_ERR_INVALID, "field_backward: gridtype/interp must be 0 or 1");
    HeadArgs ha = {};
    if (gf::fill_grid_levels(ha.lv3, 16, f->pos_S, f->base_res) || gf::fill_grid_levels(ha.lv2, 16, f->amb_S, f->base_res))
        return gf_set_error(GF_ERR_INVALID, "field_backward: bad grid levels");
    ha.pos_offsets = f->pos_offsets; ha.amb_table = f->amb_table; ha.amb_offsets = f->amb_offsets; ha.head_pack = f->head_pack;
    ha.gridtype = f->gridtype; ha.interp = f->interp; ha.bound = f->bound;
    BwdArgs ba = {bwd_stream, g->g_sigma, g->g_rgb, g->g_amb, g->sigma, g->rgb, g->amb, g->m_hc1, g->m_hs2, g->m_hs1, g->m_ha2, g->m_ha1,
                  g->g_zc, g->g_h0, g->g_za, g->g_hc1, g->g_geo, g->g_hs2, g->g_hs1, g->g_ha2, g->g_ha1, g->g_f3, g->g_f2, g->s_hc1, g->s_ha1, g->level_max, M};
    static GfLdsAttr lds[2];
    if (const int e = gf_raise_lds_limit(lds[0], reinterpret_cast<const void*>(k_field_backward<false>), kSmemBytes, "field_backward")) return e;
    if (const int e = gf_raise_lds_limit(lds[1], reinterpret_cast<const void*>(k_field_backward<true>), kSmemBytes, "field_backward")) return e;
    const uint32_t chunks = gf_div_up(M, (uint32_t)kPass);
    if (g->out16 == 0 && M > (1u << 23) - 128u) return gf_set_error(GF_ERR_UNSUPPORTED, "field_backward: the fp32 chain addresses its [M,128] rows with 32-bit byte offsets: M < 2^23");
    if (g->out16 == 2 && M > (1u << 24) - 128u) return gf_set_error(GF_ERR_UNSUPPORTED, "field_backward: the f16 chain addresses its [M,128] rows with 32-bit byte offsets: M < 2^24");
    if (g->out16 == 2) {      // the whole chain on the f16 matrix pipe: bwd_stream holds gf_field_bwd16_stream_halves() binary16 values
        static GfLdsAttr lds16;
        if (const int e = gf_raise_lds_limit(lds16, reinterpret_cast<const void*>(k_field_backward16), kSmemBytes, "field_backward")) return e;
        hipLaunchKernelGGL(k_field_backward16, dim3(chunks < 512u ? chunks : 512u), dim3(kThreads), kSmemBytes, gf_stream(stream), ha, ba);
    } else if (g->out16) hipLaunchKernelGGL(k_field_backward<true>, dim3(chunks < 512u ? chunks : 512u), dim3(kThreads), kSmemBytes, gf_stream(stream), ha, ba);
    else hipLaunchKernelGGL(k_field_backward<false>, dim3(chunks < 512u ? chunks : 512u), dim3(kThreads), kSmemBytes, gf_stream(stream), ha, ba);
    return gf_check_launch("field_backward");
}

GF_EXPORT uint64_t gf_frame_field_offset(uint32_t n_rays, uint32_t field) {
    char* const base = reinterpret_cast<char*>(uintptr_t(1) << 20);
    const gf::FrameWs w = gf::carve_workspace(base, n_rays);
    const void* const f[10] = {w.nears, w.fars, w.rays_t, w.weights_sum, w.depth, w.image, w.rays_o, w.rays_d, w.far_occ, w.alive_b};
    return field < 10 ? (uint64_t)((const char*)f[field] - base) : ~uint64_t(0);
}

GF_EXPORT uint64_t gf_frame_sizeof(void) { return sizeof(gf_frame_t); }

// get_rays, N = -1 branch (utils.py:282-363) as ONE launch: the rays a pose-mode frame generates inside k_frame_init, as tensors.
GF_EXPORT int gf_pinhole_rays(const float* pose12_host, const float* intrinsics4_host, uint32_t img_h, uint32_t img_w, float* rays_o,
                              float* rays_d, void* stream) {
    if (!pose12_host || !intrinsics4_host || !rays_o || !rays_d) return gf_set_error(GF_ERR_INVALID, "pinhole_rays: null pointer");
    const uint64_t N = (uint64_t)img_h * img_w;
    if (N == 0 || N >= (1ull << 32)) return gf_set_error(GF_ERR_INVALID, "pinhole_rays: img_h*img_w must be in [1, 2^32)");
    PinholeArgs pa;
    for (int i = 0; i < 12; i++) pa.pose[i] = pose12_host[i];
    pa.fx = intrinsics4_host[0]; pa.fy = intrinsics4_host[1]; pa.cx = intrinsics4_host[2]; pa.cy = intrinsics4_host[3];
    pa.img_w = img_w; pa.N = (uint32_t)N; pa.rays_o = rays_o; pa.rays_d = rays_d;
    hipLaunchKernelGGL(k_pinhole_rays, dim3(gf_div_up((uint32_t)N, 256u)), dim3(256), 0, gf_stream(stream), pa);
    return gf_check_launch("pinhole_rays");
}

// HOST: do the fused kernels support these grid tables (see grid_core.hpp, LevelMeta)?  offsets_host = GridEncoder.offsets [L+1].
GF_EXPORT int gf_grid_levels_fusable(const int32_t* offsets_host, uint32_t L, uint32_t D, float S, uint32_t H) {
    if (!offsets_host || L != 16 || (D != 2 && D != 3)) return gf_set_error(GF_ERR_UNSUPPORTED, "fused path: grids must have 16 levels and 2 or 3 dimensions");
    if (!gf::grid_levels_fusable(offsets_host, L, D, S, H))
        return gf_set_error(GF_ERR_UNSUPPORTED, "fused path: a wrapped grid level has a table size that is not a power of two");
    return GF_OK;
}

// Head pass (NeRFRenderer.render inference branch up to the background blend): fills the workspace accumulators.
// When f->torso_pack == NULL the head-only tail (bg blend, clamp, depth) is also enqueued and the outputs are final.
GF_EXPORT int gf_render_head(const gf_frame_t* f, void* stream) {
    int rc = check_frame(f);
    if (rc) return rc;
    hipStream_t s = gf_stream(stream);
    rc = launch_head(f, s, nullptr);
    if (rc) return rc;
    if (!f->torso_pack) {
        if (!f->bg_color || !f->out_rgb || !f->out_depth) return gf_set_error(GF_ERR_INVALID, "frame: null output / background pointer");
        const gf::FrameWs w = gf::carve_workspace(f->workspace, f->n_rays);
        hipLaunchKernelGGL(k_head_finish, dim3(gf_div_up(f->n_rays, 256u)), dim3(256), 0, s, f->n_rays, w.image, w.weights_sum, w.depth, w.nears,
                           w.fars, f->bg_color, f->out_rgb, f->out_depth, f->out_rgb8);
        return gf_check_launch("head_finish");
    }
    return GF_OK;
}

// Same work as gf_render_head's two field kernels, bracketed by HIP events on `stream`; synchronises, then reports
// their durations (ms): phase_ms_host[0] = first max_steps samples, [1] = the remaining B - max_steps, [2] = k_frame_init (the marcher's
// empty-space walk; north_star asks for its rate).  phase_ms_host has room for 4 floats.  Measurement only.
GF_EXPORT int gf_render_head_timed(const gf_frame_t* f, void* stream, float* phase_ms_host, uint32_t* n_phases_host) {
    int rc = check_frame(f);
    if (rc) return rc;
    static hipEvent_t ev[6];
    static bool made = false;
    if (!made) {
        for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return gf_set_error(GF_ERR_HIP, "hipEventCreate failed");
        made = true;
    }
    hipStream_t s = gf_stream(stream);
    rc = launch_head(f, s, ev);
    if (rc) return rc;
    if (hipStreamSynchronize(s) != hipSuccess) return gf_set_error(GF_ERR_HIP, "hipStreamSynchronize failed");
    for (uint32_t i = 0; i < 3; i++) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        phase_ms_host[i] = ms;
    }
    *n_phases_host = 3;
    return GF_OK;
}
