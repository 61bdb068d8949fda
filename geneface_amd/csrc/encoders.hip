// Stand-alone encoder operators behind the `_gridencoder`, `_shencoder`, `_freqencoder` seams.
//   grid : /root/reference/modules/radnerfs/encoders/gridencoder/src/gridencoder.cu:88-244, launch :370-400
//   SH   : .../encoders/shencoder/src/shencoder.cu:28-121 (degree <= 8; bands 4..7 table driven, sh_core.hpp::sh_high), launch :385-391
//   freq : .../encoders/freqencoder/src/freqencoder.cu:30-58, launch :96-110
#include <atomic>

#include "common.hpp"
#include <stdlib.h>
#include "grid_core.hpp"
#include "sh_core.hpp"

namespace {

constexpr int kBlock = 256;

// LAYOUT_BLC == false : outputs [L,B,C] (what the pybind seam promises, grid.py:47)
// LAYOUT_BLC == true  : outputs [B,L*C] directly (what GridEncoder.forward returns after its permute)
template <uint32_t D, uint32_t C, bool LAYOUT_BLC, class T = float>
__global__ void __launch_bounds__(kBlock) k_grid_encode(const float* __restrict__ inputs, const T* __restrict__ embeddings,
                                                        const int* __restrict__ offsets, float* __restrict__ outputs, uint32_t B,
                                                        gf::GridLevels lv, float* __restrict__ dy_dx, uint32_t gridtype,
                                                        bool align_corners, uint32_t interp) {
    uint32_t b, level;
    if (LAYOUT_BLC) {  // consecutive lanes -> consecutive levels of one point: coalesced [B, L*C] stores
        const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
        b = tid / lv.L;
        level = tid - b * lv.L;
    } else {           // level-major like the reference: one level's table per blockIdx.y
        b = blockIdx.x * kBlock + threadIdx.x;
        level = blockIdx.y;
    }
    if (b >= B) return;
    const uint32_t L = lv.L;

    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        oob |= !(x[d] >= 0 && x[d] <= 1);   // NaN counts as out of range: no address is ever formed from it (the fused lookups do the same)
    }
    float* out = LAYOUT_BLC ? outputs + (size_t)b * L * C + level * C : outputs + ((size_t)level * B + b) * C;
    float* dd = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : nullptr;
    float res[C];
    if (oob) {
#pragma unroll
        for (uint32_t c = 0; c < C; c++) res[c] = 0.0f;
        if (dd) for (uint32_t i = 0; i < D * C; i++) dd[i] = 0.0f;
    } else {
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        gf::grid_level_lookup<D, C, T>(embeddings + (size_t)off * C, hashmap_size, lv.scale[level], lv.resolution[level], gridtype,
                                       align_corners, interp, x, res, dd);
    }
#pragma unroll
    for (uint32_t c = 0; c < C; c++) out[c] = res[c];
}

template <uint32_t D, uint32_t C>
int launch_grid(bool blc, const float* inputs, const float* embeddings, const int* offsets, float* outputs, uint32_t B,
                const gf::GridLevels& lv, float* dy_dx, uint32_t gridtype, bool align_corners, uint32_t interp, hipStream_t s) {
    if (blc) {
        const uint64_t total = (uint64_t)B * lv.L;
        hipLaunchKernelGGL((k_grid_encode<D, C, true>), dim3((uint32_t)gf_div_up<uint64_t>(total, kBlock)), dim3(kBlock), 0, s, inputs,
                           embeddings, offsets, outputs, B, lv, dy_dx, gridtype, align_corners, interp);
    } else {
        hipLaunchKernelGGL((k_grid_encode<D, C, false>), dim3(gf_div_up(B, (uint32_t)kBlock), lv.L), dim3(kBlock), 0, s, inputs, embeddings,
                           offsets, outputs, B, lv, dy_dx, gridtype, align_corners, interp);
    }
    return gf_check_launch("grid_encode_forward");
}

template <uint32_t D>
int dispatch_c(uint32_t C, bool blc, const float* inputs, const float* embeddings, const int* offsets, float* outputs, uint32_t B,
               const gf::GridLevels& lv, float* dy_dx, uint32_t gridtype, bool ac, uint32_t interp, hipStream_t s) {
    switch (C) {
        case 1: return launch_grid<D, 1>(blc, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 2: return launch_grid<D, 2>(blc, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 4: return launch_grid<D, 4>(blc, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 8: return launch_grid<D, 8>(blc, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        default: return gf_set_error(GF_ERR_INVALID, "GridEncoding: C must be 1, 2, 4, or 8.");
    }
}

// Half TABLES (binary16 bit patterns), fp32 everything else: C in {2, 4, 8} as the reference's wrapper guarantees (grid.py:43: C % 2 == 0).
template <uint32_t D>
int dispatch_c_f16(uint32_t C, const float* inputs, const _Float16* embeddings, const int* offsets, float* outputs, uint32_t B,
                   const gf::GridLevels& lv, float* dy_dx, uint32_t gridtype, bool ac, uint32_t interp, hipStream_t s) {
    const dim3 grid(gf_div_up(B, (uint32_t)kBlock), lv.L), block(kBlock);
    switch (C) {
        case 2: hipLaunchKernelGGL((k_grid_encode<D, 2, false, _Float16>), grid, block, 0, s, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp); break;
        case 4: hipLaunchKernelGGL((k_grid_encode<D, 4, false, _Float16>), grid, block, 0, s, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp); break;
        case 8: hipLaunchKernelGGL((k_grid_encode<D, 8, false, _Float16>), grid, block, 0, s, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp); break;
        default: return gf_set_error(GF_ERR_INVALID, "GridEncoding (half tables): C must be 2, 4, or 8.");
    }
    return gf_check_launch("grid_encode_forward_f16");
}

int grid_encode_any(bool blc, const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                    uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners,
                    uint32_t interp, void* stream) {
    if (B == 0) return GF_OK;
    if (!inputs || !embeddings || !offsets || !outputs) return gf_set_error(GF_ERR_INVALID, "grid_encode_forward: null pointer");
    if (gridtype > 1 || interp > 1) return gf_set_error(GF_ERR_INVALID, "grid_encode_forward: gridtype/interp must be 0 or 1");
    if (C >= 2 && ((uintptr_t)embeddings & (C >= 4 ? 15u : 7u))) return gf_set_error(GF_ERR_INVALID, "grid_encode_forward: embeddings misaligned");
    gf::GridLevels lv;
    if (gf::fill_grid_levels(lv, L, S, H) != 0) return gf_set_error(GF_ERR_INVALID, "grid_encode_forward: L must be in [1,32]");
    hipStream_t s = gf_stream(stream);
    const bool ac = align_corners != 0;
    switch (D) {
        case 2: return dispatch_c<2>(C, blc, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 3: return dispatch_c<3>(C, blc, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 4: return dispatch_c<4>(C, blc, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 5: return dispatch_c<5>(C, blc, inputs, embeddings, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        default: return gf_set_error(GF_ERR_INVALID, "GridEncoding: D must be 2, 3, 4, or 5.");
    }
}

__global__ void __launch_bounds__(kBlock) k_sh(const float* __restrict__ inputs, float* __restrict__ outputs, uint32_t B, uint32_t degree,
                                               float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const float x = inputs[(size_t)b * 3], y = inputs[(size_t)b * 3 + 1], z = inputs[(size_t)b * 3 + 2];
    float sh[16];
    gf::sh4(x, y, z, sh);
    const uint32_t n = degree * degree, n4 = n < 16u ? n : 16u;
    float* o = outputs + (size_t)b * n;
    for (uint32_t i = 0; i < n4; i++) o[i] = sh[i];
    float* d = dy_dx ? dy_dx + (size_t)b * 3 * n : nullptr;   // [B, 3, degree^2]
    if (d) {
        float gx[16], gy[16], gz[16];
        gf::sh4_grad(x, y, z, gx, gy, gz);
        for (uint32_t i = 0; i < n4; i++) { d[i] = gx[i]; d[n + i] = gy[i]; d[2 * n + i] = gz[i]; }
    }
    if (degree > 4) gf::sh_high(x, y, z, degree, o, d, d ? d + n : nullptr, d ? d + 2 * n : nullptr);   // bands 4..7, shencoder.cu:69-121,150-356
}

// shencoder.cu:359-383: grad_inputs[b][d] += sum_k grad[b][k] * dy_dx[b][d][k]
__global__ void __launch_bounds__(kBlock) k_sh_backward(const float* __restrict__ grad, const float* __restrict__ dy_dx, uint32_t B, uint32_t n,
                                                        float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= B * 3) return;
    const uint32_t b = t / 3, d = t - b * 3;
    const float* g = grad + (size_t)b * n;
    const float* dd = dy_dx + (size_t)b * 3 * n + (size_t)d * n;
    float acc = grad_inputs[t];
    for (uint32_t k = 0; k < n; k++) acc += g[k] * dd[k];
    grad_inputs[t] = acc;
}

// freqencoder.cu:63-94
__global__ void __launch_bounds__(kBlock) k_freq_backward(const float* __restrict__ grad, const float* __restrict__ outputs, uint32_t B, uint32_t D,
                                                          uint32_t deg, uint32_t C, float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float* g = grad + (size_t)b * C;
    const float* o = outputs + (size_t)b * C;
    float result = g[d];
    g += D; o += D;
    for (uint32_t f = 0; f < deg; f++) {
        result += scalbnf(1.0f, (int)f) * (g[d] * o[D + d] - g[D + d] * o[d]);
        g += 2 * D; o += 2 * D;
    }
    grad_inputs[t] = result;
}

// gridencoder.cu:248-339: every corner's share of the output gradient is scatter-added into the table (f32 adds: the accumulation
// order, hence the last bits, vary from run to run exactly as in the reference).
// The reference issues one global atomic per (point, level, corner, channel): 268 M for a training batch of 1 M points, and this
// chip retires scattered f32 atomics at ~21 G/s however they are spread (tools/atomic_probe.hip), on top of the serialisation on the
// coarse levels (a million points over a 17^3 table).  Here the TABLE is partitioned instead of the points: a workgroup owns
// kGbLdsFloats / C consecutive rows of one level, keeps them in 128 KiB of LDS (one workgroup per CU, 1024 lanes), streams a slice
// of the points, evaluates every corner and accumulates the ones that land in its rows with ds_add_f32; at the end each touched
// entry costs ONE global atomic.  Points are re-read once per partition (<= kGbMaxParts, out of L2), index arithmetic is repeated
// -- both cheap next to the atomics they replace: <= slices x table entries instead of points x corners x channels, and the flush
// walks the table in address order (coalesced atomics).  Measured on 1 M ray-ordered points, 16 levels, C = 2 (tools/bench_grid_backward.py):
// 3-D 19.3 ms (direct atomics; the reference's scheme) -> 1.6 ms, 2-D 8.4 -> 0.85 ms.  Round-2 split of the 1.6 ms: index arithmetic + a
// flush of EVERY entry 0.28 ms, so the ds_add_f32 stream itself is 1.3 ms (268 M adds = 0.4 per clock and CU); point order (ray order,
// random, ray-interleaved) moves it by < 6 %: it is the LDS atomic rate, not same-address serialisation.
// A level too large for kGbMaxParts partitions (log2_hashmap_size > 17 at C = 2) falls back to direct global atomics.
// Round 2: the LDS accumulators are 64-bit FIXED POINT.  tools/lds_atomic_probe.hip: ds_add_f32 retires 0.38 adds per clock and CU on gfx950
// (exactly the rate the float version of this kernel ran at), ds_add_u32 11.6 and ds_add_u64 6.7 -- the float atomic is ~20x slower than the
// integer ones.  Every contribution w * g is scaled by 2^40 / max|g| of its level (k_grid_absmax, one pass over the gradient), rounded and
// added as int64: no overflow below 2^23 full-size contributions per entry and slice (dispatch_backward_c keeps a slice at <= 2^20 points), a
// quantum of 9e-13 of the level's largest gradient, and -- integer adds commute -- a fixed-point partial that does not depend on the order the
// points arrive in (the float-atomic path below for contributions under 2^14 quanta, and the float flush into the table, are order dependent
// at the last-ulp level like any atomic scatter).  Round 3: the scale was 2^30 with int32 rounding, which
// turned every contribution below 5e-10 of the level's maximum into an exact zero; the reference trains these tables with Adam eps = 1e-15
// (tasks/radnerfs/radnerf.py:63) precisely so that rarely-hit entries with tiny gradients still take full-size steps.  Now a contribution
// below 2^14 quanta (1.5e-8 of the maximum) goes to the table as a float atomic instead, so every entry keeps fp32 RELATIVE accuracy over
// the whole dynamic range (tests/test_gpu_train.py::test_grid_backward_keeps_tiny_gradients: 14 decades against an fp64 scatter).
constexpr uint32_t kGbThreads = 1024;
#ifndef GF_GB_ENTRIES
#define GF_GB_ENTRIES 16384
#endif
constexpr uint32_t kGbLdsEntries = GF_GB_ENTRIES;   // int64 accumulators: 128 KiB, one workgroup per CU
constexpr uint32_t kGbMaxParts = 8;
#ifndef GF_GB_BATCH
#define GF_GB_BATCH 4
#endif
constexpr uint32_t kGbBatch = GF_GB_BATCH;                  // points per lane whose loads are in flight together (1 = the loop before round 6)
constexpr float kGbFixedOne = 1099511627776.0f;   // 2^40: one quantum = 9e-13 of the level's largest gradient; 2^23 contributions of full size fit an int64
#ifndef GF_GB_SMALL
#define GF_GB_SMALL 16384.0f
#endif
constexpr float kGbSmall = GF_GB_SMALL;              // contributions below 2^14 quanta (1.5e-8 of the largest gradient) would lose more than 2^-15 of their
                                                  // value to the rounding: they take a float atomic into the table instead (rare; keeps every entry's
                                                  // RELATIVE accuracy, which is what Adam with eps = 1e-15 -- tasks/radnerfs/radnerf.py:63 -- steps by)

// per-level max |grad| (bit pattern of a non-negative float orders like the uint): lvl_max[L] zeroed by the caller
template <uint32_t C>
__global__ void __launch_bounds__(256) k_grid_absmax(const float* __restrict__ grad, uint32_t B, uint32_t* __restrict__ lvl_max) {
    const uint32_t level = blockIdx.y;
    const float* g = grad + (size_t)level * B * C;
    const size_t n = (size_t)B * C;
    float m = 0.0f;
    auto take = [&](float v) { v = fabsf(v); m = (v == v) ? fmaxf(m, v) : INFINITY; };   // a NaN gradient counts as an overflow (below)
    const size_t head = ((16 - ((uintptr_t)g & 15)) & 15) / 4 < n ? ((16 - ((uintptr_t)g & 15)) & 15) / 4 : n;   // floats before 16-byte alignment
    const size_t n4 = (n - head) / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g + head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = g4[i];
        take(v.x); take(v.y); take(v.z); take(v.w);
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) take(g[threadIdx.x]);
        const size_t tail0 = head + n4 * 4;
        if (tail0 + threadIdx.x < n) take(g[tail0 + threadIdx.x]);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(&lvl_max[level], __float_as_uint(m));
}

// The point loop of k_grid_backward, specialised on the row rule so that the unrolled corner code carries no workgroup-uniform branches:
// MODE 1 = power-of-two hash, MODE 2 = dense strides (both through LevelMeta, non-aligned corners), MODE 0 = the generic rule (grid_row).
// DIRECT = the level does not fit kGbMaxParts partitions: float atomics straight into the table.
template <uint32_t D, uint32_t C, int MODE, bool DIRECT>
__device__ __forceinline__ void gb_points(long long* tab, const float* __restrict__ grad, const float* __restrict__ inputs, float* __restrict__ table,
                                          uint32_t level, uint32_t B, uint32_t b0, uint32_t b1, float scale, uint32_t resolution, uint32_t hashmap_size,
                                          uint32_t gridtype, bool align_corners, uint32_t interp, const gf::LevelMeta& lm, uint32_t row0, uint32_t nrows,
                                          float to_fixed, const uint32_t* __restrict__ idx = nullptr /* a bin of point indices: [b0, b1) index IT */) {
    // kGbBatch points per lane and trip, their inputs and gradients loaded TOGETHER before any is processed (round 6): one workgroup per CU
    // (the 128 KiB table) is 4 waves per SIMD, and a trip used to wait for two dependent loads (the point, then -- if in range -- its gradient)
    // before ~100-400 instructions of work: the loop ran at the latency of those loads (tools/grid_backward_levels.py).
    for (uint32_t bb = b0 + threadIdx.x; bb < b1; bb += kGbThreads * kGbBatch) {
      float xs[kGbBatch][D], gsv[kGbBatch][C];
#pragma unroll
      for (uint32_t k = 0; k < kGbBatch; k++) {
          const uint32_t bi = bb + k * kGbThreads;
          const bool in = bi < b1;
          const uint32_t b = (idx && in) ? idx[bi] : bi;
#pragma unroll
          for (uint32_t d = 0; d < D; d++) xs[k][d] = in ? inputs[(size_t)b * D + d] : -1.0f;      // beyond the slice: out of range, skipped below
#pragma unroll
          for (uint32_t c = 0; c < C; c++) gsv[k][c] = in ? grad[((size_t)level * B + b) * C + c] : 0.0f;
      }
#pragma unroll
      for (uint32_t k = 0; k < kGbBatch; k++) {
        float x[D];
        bool oob = false;
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            x[d] = xs[k][d];
            oob |= !(x[d] >= 0 && x[d] <= 1);   // NaN counts as out of range: no address is ever formed from it (the fused lookups do the same)
        }
        if (oob) continue;
        float g[C];
        bool any = false;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            g[c] = gsv[k][c];
            any |= g[c] != 0.0f;
        }
        if (!any) continue;                                        // samples behind a ray's termination point carry exact zeros
        float gs[C];                                               // in fixed-point units (|gs| <= 2^40)
#pragma unroll
        for (uint32_t c = 0; c < C; c++) gs[c] = DIRECT ? g[c] : g[c] * to_fixed;
        float pos[D];
        uint32_t pos_grid[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            pos[d] = __builtin_fmaf(x[d], scale, (MODE == 0 && align_corners) ? 0.0f : 0.5f);
            const float fl = floorf(pos[d]);
            pos_grid[d] = (uint32_t)fl;
            pos[d] -= fl;
            if (interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
        }
        // per-axis terms of the row index, both sides of the cell (the corner loop only combines them)
        uint32_t term[D][2];
        if constexpr (MODE != 0) {
            constexpr uint32_t P1 = 2654435761u, P2 = 805459861u;
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t m = MODE == 1 ? (d == 0 ? 1u : (d == 1 ? P1 : P2)) : (d == 0 ? 1u : (d == 1 ? lm.s1 : lm.s2));
                term[d][0] = pos_grid[d] * m;
                term[d][1] = term[d][0] + m;
            }
        }
#pragma unroll
        for (uint32_t corner = 0; corner < (1u << D); corner++) {
            uint32_t row;
            if constexpr (MODE == 1) {
                row = term[0][corner & 1u];
#pragma unroll
                for (uint32_t d = 1; d < D; d++) row ^= term[d][(corner >> d) & 1u];
                row &= lm.mask;
            } else if constexpr (MODE == 2) {
                row = term[0][corner & 1u];
#pragma unroll
                for (uint32_t d = 1; d < D; d++) row += term[d][(corner >> d) & 1u];
                row &= lm.mask;
            } else {
                uint32_t pl[D];
#pragma unroll
                for (uint32_t d = 0; d < D; d++) pl[d] = pos_grid[d] + ((corner >> d) & 1u);
                row = gf::grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pl);
            }
            const uint32_t r = row - row0;                         // rows below row0 wrap to huge values
            if (DIRECT || r < nrows) {
                float w = 1.0f;
#pragma unroll
                for (uint32_t d = 0; d < D; d++) w *= (corner & (1u << d)) ? pos[d] : 1 - pos[d];
                if constexpr (DIRECT) {
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) unsafeAtomicAdd(table + (size_t)row * C + c, w * gs[c]);
                } else {
#pragma unroll
                    for (uint32_t c = 0; c < C; c++) {
                        const float v = w * gs[c];
                        if (fabsf(v) >= kGbSmall) atomicAdd(reinterpret_cast<unsigned long long*>(&tab[r * C + c]), (unsigned long long)__float2ll_rn(v));
                        else if (g[c] != 0.0f) unsafeAtomicAdd(table + (size_t)row * C + c, w * g[c]);
                    }
                }
            }
        }
      }
    }
}

// What k_grid_backward decides per level, in one place (k_grid_bin must decide the same): partitions, the direct fall-back, the row rule.
struct GbLevelPlan { uint32_t off, hashmap_size, nparts; bool direct; int mode; gf::LevelMeta lm; };
template <uint32_t D, uint32_t C>
__device__ __forceinline__ GbLevelPlan gb_level_plan(const int* __restrict__ offsets, const gf::GridLevels& lv, uint32_t level, uint32_t gridtype,
                                                     bool align_corners) {
    GbLevelPlan p;
    p.off = (uint32_t)offsets[level];
    p.hashmap_size = (uint32_t)offsets[level + 1] - p.off;
    constexpr uint32_t rows_per_part = kGbLdsEntries / C;
    p.nparts = (p.hashmap_size + rows_per_part - 1) / rows_per_part;
    p.direct = p.nparts > kGbMaxParts;
    if (p.direct) p.nparts = 1;
    // Row index without the generic rule's integer modulo (grid_row): dense levels never reach the table size, wrapped levels have a
    // power-of-two size (grid.py:118-134 caps them at 2^log2_hashmap_size) -- the same reduction the fused lookup uses (LevelMeta).
    p.lm = gf::LevelMeta{};
    p.mode = 0;
    if constexpr (D <= 3) {
        p.lm = gf::make_level_meta<D>(lv.scale[level], lv.resolution[level], offsets, level, gridtype);
        if (!align_corners && (p.lm.mask == 0xFFFFFFFFu || (p.hashmap_size & (p.hashmap_size - 1u)) == 0u)) p.mode = p.lm.use_hash ? 1 : 2;
    }
    return p;
}
__device__ __forceinline__ bool gb_level_binned(const GbLevelPlan& p) { return !p.direct && p.nparts >= 2 && p.mode != 0; }

// Binning pass (round 6): k_grid_backward's workgroup (level, partition, slice) used to examine EVERY point of its slice -- ~250 instructions to
// find that, on a level of eight partitions, one corner in eight is its own (tools/grid_backward_levels.py: 111 M examinations of 16 M (point,
// level) pairs per 1 M-point call).  Here one lane per point walks the levels ONCE, finds which partitions its corners touch (a cell's corners
// touch ~2.4 of 8 on the tiled 3-D grid, ~2 on the 2-D one) and appends the point's index to those partitions' lists; the scatter then reads each
// partition's own list.  Appending is aggregated twice -- ballot per wave, LDS counters per 1 024-lane workgroup -- so a workgroup spends ONE
// global atomic per (level, partition) and 1 024 points: the first version took one per wave and ran 9.5 ms on 120 contended addresses.
// The order inside a list depends on the order in which waves and workgroups arrive; a slice's integer partial depends only on WHICH points
// it holds, and the float flush is order-dependent in the last ulp with or without lists.  counts [32][kGbMaxParts] ZEROED; bins [L][kGbMaxParts][B].  Points with an all-zero gradient at a level
// are not listed for it.
constexpr uint32_t kBinThreads = 1024, kBinSlots = gf::kMaxLevels * kGbMaxParts;
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kBinThreads) k_grid_bin(const float* __restrict__ grad, const float* __restrict__ inputs, const int* __restrict__ offsets,
                                                          uint32_t B, gf::GridLevels lv, uint32_t gridtype, bool align_corners,
                                                          const uint32_t* __restrict__ lvl_max, uint32_t* __restrict__ counts, uint32_t* __restrict__ bins) {
    __shared__ uint32_t cnt[kBinSlots], base[kBinSlots], woff[kBinThreads / 64][kBinSlots];
    __shared__ uint8_t msk[gf::kMaxLevels][kBinThreads];                      // the partitions of every level a lane's point goes to
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t rows_per_part = kGbLdsEntries / C;
    for (uint32_t chunk = blockIdx.x * kBinThreads; chunk < B; chunk += gridDim.x * kBinThreads) {       // workgroup-uniform
        if (threadIdx.x < kBinSlots) cnt[threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t b = chunk + threadIdx.x;
        bool in = b < B;
        float x[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            x[d] = in ? inputs[(size_t)b * D + d] : 0.0f;
            in = in && (x[d] >= 0 && x[d] <= 1);
        }
        for (uint32_t level = 0; level < lv.L; level++) {
            {
                const float gmax = __uint_as_float(lvl_max[level]);
                const GbLevelPlan p = gb_level_plan<D, C>(offsets, lv, level, gridtype, align_corners);
                if ((gmax < INFINITY) && (gmax > 0.0f) && gb_level_binned(p)) {   // (k_grid_backward returns at once for the other gradients)
                    bool any = false;
                    if (in) {
#pragma unroll
                        for (uint32_t c = 0; c < C; c++) any |= grad[((size_t)level * B + b) * C + c] != 0.0f;
                    }
                    uint32_t term[D][2];
                    constexpr uint32_t P1 = 2654435761u, P2 = 805459861u;
#pragma unroll
                    for (uint32_t d = 0; d < D; d++) {
                        const uint32_t pg = (uint32_t)floorf(__builtin_fmaf(x[d], lv.scale[level], 0.5f));
                        const uint32_t m = p.mode == 1 ? (d == 0 ? 1u : (d == 1 ? P1 : P2)) : (d == 0 ? 1u : (d == 1 ? p.lm.s1 : p.lm.s2));
                        term[d][0] = pg * m;
                        term[d][1] = term[d][0] + m;
                    }
                    uint32_t mask = 0;
#pragma unroll
                    for (uint32_t corner = 0; corner < (1u << D); corner++) {
                        uint32_t row = term[0][corner & 1u];
#pragma unroll
                        for (uint32_t d = 1; d < D; d++) row = p.mode == 1 ? (row ^ term[d][(corner >> d) & 1u]) : (row + term[d][(corner >> d) & 1u]);
                        row &= p.lm.mask;
                        const uint32_t part = row / rows_per_part;
                        if (part < p.nparts) mask |= 1u << part;               // (a row beyond the level cannot happen for x in [0, 1]; k_grid_backward skips it too)
                    }
                    if (!any) mask = 0;
                    msk[level][threadIdx.x] = (uint8_t)mask;
                    for (uint32_t part = 0; part < p.nparts; part++) {         // this wave's share of the workgroup's counters
                        const unsigned long long m = __ballot((mask >> part) & 1u);
                        if (m != 0 && lane == 0) woff[wave][level * kGbMaxParts + part] = atomicAdd(&cnt[level * kGbMaxParts + part], (uint32_t)__popcll(m));
                    }
                } else {
                    msk[level][threadIdx.x] = 0;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < kBinSlots) {                                         // one global atomic per (level, partition) and workgroup trip
            const uint32_t c = cnt[threadIdx.x];
            base[threadIdx.x] = c ? atomicAdd(&counts[threadIdx.x], c) : 0u;
        }
        __syncthreads();
        for (uint32_t level = 0; level < lv.L; level++) {
            {
                const uint32_t mask = msk[level][threadIdx.x];
                if (__ballot(mask != 0) != 0) {                                // uniform per wave
                    for (uint32_t part = 0; part < kGbMaxParts; part++) {
                        const bool bit = (mask >> part) & 1u;
                        const unsigned long long m = __ballot(bit);
                        if (bit) {
                            const uint32_t slot = level * kGbMaxParts + part;
                            bins[(size_t)slot * B + base[slot] + woff[wave][slot] +
                                 __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = b;
                        }
                    }
                }
            }
        }
        __syncthreads();                                                       // woff / base are rewritten by the next trip
    }
}

template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kGbThreads) k_grid_backward(const float* __restrict__ grad, const float* __restrict__ inputs,
                                                              const int* __restrict__ offsets, float* __restrict__ grad_grid, uint32_t B,
                                                              gf::GridLevels lv, uint32_t gridtype, bool align_corners, uint32_t interp,
                                                              uint32_t flush_budget, uint32_t min_slices, const uint32_t* __restrict__ lvl_max,
                                                              const uint32_t* __restrict__ counts, const uint32_t* __restrict__ bins /* both NULL: no lists */) {
    __shared__ long long tab[kGbLdsEntries];
    const uint32_t level = blockIdx.y;
    const float gmax = __uint_as_float(lvl_max[level]);
    if (!(gmax < INFINITY)) {                                      // inf / NaN in the gradient (fp16 loss scaling overflowed): the float scatter would
        if (blockIdx.x == 0 && threadIdx.x == 0)                   // have put a non-finite value into the table, which is what GradScaler looks for
            unsafeAtomicAdd(grad_grid + (size_t)offsets[level] * C, __uint_as_float(0x7fc00000u));
        return;
    }
    if (!(gmax > 0.0f)) return;                                    // an all-zero gradient level adds nothing
    // 2^40 / gmax must stay finite: a level whose largest gradient is below 1e-25 scatters through the float path alone (to_fixed = 0)
    const float to_fixed = gmax > 1e-25f ? kGbFixedOne / gmax : 0.0f;
    const double from_fixed = (double)gmax / (double)kGbFixedOne;
    const GbLevelPlan plan = gb_level_plan<D, C>(offsets, lv, level, gridtype, align_corners);
    const uint32_t off = plan.off, hashmap_size = plan.hashmap_size, nparts = plan.nparts;
    constexpr uint32_t rows_per_part = kGbLdsEntries / C;
    const bool direct = plan.direct;                               // workgroup-uniform
    // Slices of the point list per level: a coarse level has few rows, so many slices cost little at the flush and cut the time its
    // workgroups spend serialising same-address LDS adds (neighbouring samples of a ray share the coarse cells); a fine level has
    // many rows (every one touched, every one an atomic per slice) and few conflicts, so it gets few slices.
    uint32_t slices = flush_budget / (hashmap_size * C);
    slices = slices < min_slices ? min_slices : slices;
    slices = slices * nparts > gridDim.x ? gridDim.x / nparts : slices;
    const uint32_t part = blockIdx.x % nparts, slice = blockIdx.x / nparts;
    if (slice >= slices) return;
    const uint32_t row0 = direct ? 0u : part * rows_per_part;
    const uint32_t nrows = direct ? 0u : (hashmap_size - row0 < rows_per_part ? hashmap_size - row0 : rows_per_part);
    for (uint32_t i = threadIdx.x; i < nrows * C; i += kGbThreads) tab[i] = 0;
    __syncthreads();
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    float* table = grad_grid + (size_t)off * C;
    const gf::LevelMeta lm = plan.lm;
    const int mode = plan.mode;                                    // workgroup-uniform
    // the points this workgroup examines: its slice of the whole list, or -- k_grid_bin ran -- of its partition's own list
    const uint32_t* idx = (bins && gb_level_binned(plan)) ? bins + ((size_t)level * kGbMaxParts + part) * B : nullptr;
    const uint32_t total = idx ? counts[level * kGbMaxParts + part] : B;
    const uint32_t per = (total + slices - 1) / slices;
    const uint32_t b0 = slice * per < total ? slice * per : total, b1 = b0 + per < total ? b0 + per : total;
#define GF_GB_POINTS(MODE, DIRECT) gb_points<D, C, MODE, DIRECT>(tab, grad, inputs, table, level, B, b0, b1, scale, resolution, hashmap_size, gridtype, \
                                                               align_corners, interp, lm, row0, nrows, to_fixed, idx)
    if (direct) GF_GB_POINTS(0, true);
    else if (mode == 1) GF_GB_POINTS(1, false);
    else if (mode == 2) GF_GB_POINTS(2, false);
    else GF_GB_POINTS(0, false);
#undef GF_GB_POINTS
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nrows * C; i += kGbThreads) {
        const long long q = tab[i];
        if (q != 0) unsafeAtomicAdd(table + (size_t)row0 * C + i, (float)((double)q * from_fixed));
    }
}

// gridencoder.cu:505-596: normalised total-variation gradient around the lattice node each input falls on, accumulated into grad.
// One lane per (point, level); the table differences are a handful of gathers, the accumulation one f32 atomic per channel.
template <uint32_t D, uint32_t C>
__global__ void __launch_bounds__(kBlock) k_grad_tv(const float* __restrict__ inputs, const float* __restrict__ embeddings, float* __restrict__ grad,
                                                    const int* __restrict__ offsets, float weight, uint32_t B, gf::GridLevels lv, uint32_t gridtype,
                                                    bool align_corners) {
    const uint32_t b = blockIdx.x * kBlock + threadIdx.x, level = blockIdx.y;
    if (b >= B) return;
    float x[D];
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        oob |= !(x[d] >= 0 && x[d] <= 1);   // NaN counts as out of range: no address is ever formed from it (the fused lookups do the same)
    }
    if (oob) return;
    const uint32_t off = (uint32_t)offsets[level], hashmap_size = (uint32_t)offsets[level + 1] - off;
    const float* table = embeddings + (size_t)off * C;
    const float scale = lv.scale[level];
    const uint32_t resolution = lv.resolution[level];
    uint32_t pg[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) pg[d] = (uint32_t)floorf(__builtin_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f));
    float centre[C], results[C], idelta[C];
    const uint32_t row = gf::grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pg);
#pragma unroll
    for (uint32_t c = 0; c < C; c++) { centre[c] = table[(size_t)row * C + c]; results[c] = 0.0f; idelta[c] = 0.0f; }
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        const uint32_t cur = pg[d];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            if (side == 0 ? cur < resolution : cur > 0) {
                pg[d] = side == 0 ? cur + 1 : cur - 1;
                const uint32_t r2 = gf::grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pg);
#pragma unroll
                for (uint32_t c = 0; c < C; c++) {
                    const float g = centre[c] - table[(size_t)r2 * C + c];
                    results[c] += g;
                    idelta[c] += g * g;
                }
            }
        }
        pg[d] = cur;
    }
    const float w = weight / (float)(2 * D);
#pragma unroll
    for (uint32_t c = 0; c < C; c++) unsafeAtomicAdd(grad + ((size_t)off + row) * C + c, w * results[c] * (1.0f / sqrtf(idelta[c] + 1e-9f)));
}

template <uint32_t D>
int dispatch_tv_c(uint32_t C, const float* inputs, const float* embeddings, float* grad, const int* offsets, float weight, uint32_t B,
                  const gf::GridLevels& lv, uint32_t gridtype, bool ac, hipStream_t s) {
    const dim3 grid(gf_div_up(B, (uint32_t)kBlock), lv.L), block(kBlock);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grad_tv<D, 1>), grid, block, 0, s, inputs, embeddings, grad, offsets, weight, B, lv, gridtype, ac); break;
        case 2: hipLaunchKernelGGL((k_grad_tv<D, 2>), grid, block, 0, s, inputs, embeddings, grad, offsets, weight, B, lv, gridtype, ac); break;
        case 4: hipLaunchKernelGGL((k_grad_tv<D, 4>), grid, block, 0, s, inputs, embeddings, grad, offsets, weight, B, lv, gridtype, ac); break;
        case 8: hipLaunchKernelGGL((k_grad_tv<D, 8>), grid, block, 0, s, inputs, embeddings, grad, offsets, weight, B, lv, gridtype, ac); break;
        default: return gf_set_error(GF_ERR_INVALID, "GridEncoding: C must be 1, 2, 4, or 8.");
    }
    return gf_check_launch("grad_total_variation");
}

// gridencoder.cu:343-368
__global__ void __launch_bounds__(kBlock) k_grid_input_backward(const float* __restrict__ grad, const float* __restrict__ dy_dx,
                                                                float* __restrict__ grad_inputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D, d = t - b * D;
    const float* dd = dy_dx + (size_t)b * L * D * C;
    float result = 0.0f;
    for (uint32_t l = 0; l < L; l++)
        for (uint32_t ch = 0; ch < C; ch++) result += grad[((size_t)l * B + b) * C + ch] * dd[l * D * C + d * C + ch];
    grad_inputs[t] = result;
}

template <uint32_t D>
int dispatch_backward_c(uint32_t C, const float* grad, const float* inputs, const int* offsets, float* grad_grid, uint32_t B,
                        const gf::GridLevels& lv, uint32_t gridtype, bool ac, uint32_t interp, hipStream_t s, const uint32_t* level_max = nullptr,
                        uint32_t* counts = nullptr, uint32_t* bins = nullptr /* gf_grid_encode_backward_binned's workspace: the binning pass runs first */) {
    // workgroups per level (grid.x): partitions x slices, the surplus exits at once.  flush_budget = table floats x slices a level may
    // spend on flush atomics; min_slices keeps the fine levels' point lists short enough to balance the chip.
    uint32_t flush_budget = 1u << 22, min_slices = B >= (1u << 19) ? 8u : (B >= (1u << 16) ? 2u : 1u), wgs = 128;
    if (B < (1u << 16)) wgs = 32;
#ifdef GF_GB_TUNE      // A/B builds only: GF_GB_FLUSH_LOG2 / GF_GB_MIN_SLICES / GF_GB_WGS from the environment
    if (const char* e = getenv("GF_GB_FLUSH_LOG2")) flush_budget = 1u << atoi(e);
    if (const char* e = getenv("GF_GB_MIN_SLICES")) min_slices = (uint32_t)atoi(e);
    if (const char* e = getenv("GF_GB_WGS")) wgs = (uint32_t)atoi(e);
#endif
    // int64 headroom (kGbFixedOne = 2^40): an entry takes at most 2^D <= 8 full-size contributions per point of a slice, so a slice holds at most
    // 2^20 points (8 x 2^20 x 2^40 = 2^63); a point list longer than min_slices x 2^20 gets more slices (and the workgroups for them)
    const uint32_t need = (uint32_t)(((uint64_t)B + (1u << 20) - 1) >> 20);
    if (need > min_slices) min_slices = need;
    if (wgs < min_slices * kGbMaxParts) wgs = min_slices * kGbMaxParts;
    // per-level scale of the fixed-point accumulators: handed in by the caller (the kernel that produced the gradient knows its maxima), or one
    // max pass over the gradient into a slot of a small device ring (calls on different streams take different slots)
    const uint32_t* lvl_max = level_max;
    uint32_t* own_max = nullptr;
    if (!lvl_max) {
        static uint32_t* rings[64] = {};   // one per device of the process (allocated on first use, never freed)
        static std::atomic<unsigned> ring_pos{0};
        constexpr unsigned kSlots = 64;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return gf_set_error(GF_ERR_HIP, "grid_encode_backward: no current device");
        uint32_t*& ring = rings[dev];
        if (!ring && hipMalloc(&ring, kSlots * gf::kMaxLevels * sizeof(uint32_t)) != hipSuccess) return gf_set_error(GF_ERR_HIP, "grid_encode_backward: hipMalloc failed");
        own_max = ring + (size_t)(ring_pos.fetch_add(1u) % kSlots) * gf::kMaxLevels;
        if (hipMemsetAsync(own_max, 0, gf::kMaxLevels * sizeof(uint32_t), s) != hipSuccess) return gf_set_error(GF_ERR_HIP, "grid_encode_backward: hipMemsetAsync failed");
        lvl_max = own_max;
    }
    const dim3 mgrid(B >= (1u << 16) ? 128u : 8u, lv.L);
    const dim3 grid(wgs, lv.L), block(kGbThreads);
    const uint32_t btrips = gf_div_up(B, kBinThreads);
    const dim3 bgrid(btrips < 1024u ? btrips : 1024u);
    if (bins && hipMemsetAsync(counts, 0, gf::kMaxLevels * kGbMaxParts * sizeof(uint32_t), s) != hipSuccess)
        return gf_set_error(GF_ERR_HIP, "grid_encode_backward: hipMemsetAsync failed");
#define GF_GB_CASE(CC)                                                                                                                          \
    case CC:                                                                                                                                    \
        if (own_max) hipLaunchKernelGGL((k_grid_absmax<CC>), mgrid, dim3(256), 0, s, grad, B, own_max);                                         \
        if constexpr (D <= 3) {                                                                                                                 \
            if (bins) hipLaunchKernelGGL((k_grid_bin<D, CC>), bgrid, dim3(kBinThreads), 0, s, grad, inputs, offsets, B, lv, gridtype, ac, lvl_max, counts, bins); \
        }                                                                                                                                       \
        hipLaunchKernelGGL((k_grid_backward<D, CC>), grid, block, 0, s, grad, inputs, offsets, grad_grid, B, lv, gridtype, ac, interp, flush_budget, \
                           min_slices, lvl_max, (const uint32_t*)(D <= 3 ? counts : nullptr), (const uint32_t*)(D <= 3 ? bins : nullptr));      \
        break;
    switch (C) {
        GF_GB_CASE(1) GF_GB_CASE(2) GF_GB_CASE(4) GF_GB_CASE(8)
        default: return gf_set_error(GF_ERR_INVALID, "GridEncoding: C must be 1, 2, 4, or 8.");
    }
#undef GF_GB_CASE
    return gf_check_launch("grid_encode_backward");
}

// one lane per output element, like the reference (the output row is what bounds this kernel)
__global__ void __launch_bounds__(kBlock) k_freq(const float* __restrict__ inputs, uint32_t B, uint32_t D, uint32_t C, float* __restrict__ outputs) {
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (uint64_t)B * C) return;
    const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (uint64_t)b * C);
    const float* in = inputs + (size_t)b * D;
    outputs[t] = gf::freq_element(in, D, c);
}

}  // namespace

GF_EXPORT int gf_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                                     uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype,
                                     int align_corners, uint32_t interp, void* stream) {
    return grid_encode_any(false, inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, stream);
}

GF_EXPORT int gf_grid_encode_forward_f16(const float* inputs, const uint16_t* embeddings_f16, const int32_t* offsets, float* outputs, uint32_t B,
                                         uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners,
                                         uint32_t interp, void* stream) {
    if (B == 0) return GF_OK;
    if (!inputs || !embeddings_f16 || !offsets || !outputs) return gf_set_error(GF_ERR_INVALID, "grid_encode_forward_f16: null pointer");
    if (gridtype > 1 || interp > 1) return gf_set_error(GF_ERR_INVALID, "grid_encode_forward_f16: gridtype/interp must be 0 or 1");
    if ((uintptr_t)embeddings_f16 & (2u * C - 1u) & 15u) return gf_set_error(GF_ERR_INVALID, "grid_encode_forward_f16: embeddings misaligned");
    gf::GridLevels lv;
    if (gf::fill_grid_levels(lv, L, S, H) != 0) return gf_set_error(GF_ERR_INVALID, "grid_encode_forward_f16: L must be in [1,32]");
    hipStream_t s = gf_stream(stream);
    const bool ac = align_corners != 0;
    const _Float16* e = reinterpret_cast<const _Float16*>(embeddings_f16);
    switch (D) {
        case 2: return dispatch_c_f16<2>(C, inputs, e, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 3: return dispatch_c_f16<3>(C, inputs, e, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 4: return dispatch_c_f16<4>(C, inputs, e, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        case 5: return dispatch_c_f16<5>(C, inputs, e, offsets, outputs, B, lv, dy_dx, gridtype, ac, interp, s);
        default: return gf_set_error(GF_ERR_INVALID, "GridEncoding: D must be 2, 3, 4, or 5.");
    }
}

GF_EXPORT int gf_grid_encode_forward_blc(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                                         uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype,
                                         int align_corners, uint32_t interp, void* stream) {
    return grid_encode_any(true, inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp, stream);
}

// ---- the fused head kernel's specialised lookup (grid_core.hpp::encode8), exposed stand-alone so that it can be checked point
// by point against the generic operator: one lane pair per point (half h evaluates levels 8h .. 8h+7), output [B, 32].
template <uint32_t D>
__global__ void __launch_bounds__(256) k_encode8(const float* __restrict__ inputs, const float* __restrict__ table, const int* __restrict__ offsets,
                                                 gf::GridLevels lv, uint32_t B, uint32_t gridtype, uint32_t interp, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) gf::LevelMeta meta[16];
    if (threadIdx.x < 16) meta[threadIdx.x] = gf::make_level_meta<D>(lv.scale[threadIdx.x], lv.resolution[threadIdx.x], offsets, threadIdx.x, gridtype);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, half = lane >> 5;
    const uint32_t b = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 32u + (lane & 31u);
    if (b >= B) return;
    float x[D], f[16];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) x[d] = inputs[(size_t)b * D + d];
    gf::encode8<D>(table, meta + half * 8, gridtype, interp, x, f);
#pragma unroll
    for (int i = 0; i < 16; i++) out[(size_t)b * 32 + half * 16 + i] = f[i];
}

GF_EXPORT int gf_grid_encode_fused_lookup(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                                          uint32_t D, float S, uint32_t H, uint32_t gridtype, uint32_t interp, void* stream) {
    if (B == 0) return GF_OK;
    if (!inputs || !embeddings || !offsets || !outputs) return gf_set_error(GF_ERR_INVALID, "grid_encode_fused_lookup: null pointer");
    if (D != 2 && D != 3) return gf_set_error(GF_ERR_UNSUPPORTED, "grid_encode_fused_lookup: D must be 2 or 3");
    gf::GridLevels lv;
    if (gf::fill_grid_levels(lv, 16, S, H) != 0) return gf_set_error(GF_ERR_INVALID, "grid_encode_fused_lookup: bad levels");
    const dim3 grid(gf_div_up(B, 128u)), block(256);
    if (D == 3) hipLaunchKernelGGL(k_encode8<3>, grid, block, 0, gf_stream(stream), inputs, embeddings, offsets, lv, B, gridtype, interp, outputs);
    else hipLaunchKernelGGL(k_encode8<2>, grid, block, 0, gf_stream(stream), inputs, embeddings, offsets, lv, B, gridtype, interp, outputs);
    return gf_check_launch("grid_encode_fused_lookup");
}

GF_EXPORT int gf_grid_level_meta(uint32_t L, float S, uint32_t H, float* scale_out, uint32_t* resolution_out) {
    gf::GridLevels lv;
    if (gf::fill_grid_levels(lv, L, S, H) != 0) return gf_set_error(GF_ERR_INVALID, "grid_level_meta: L must be in [1,32]");
    for (uint32_t l = 0; l < L; l++) { scale_out[l] = lv.scale[l]; resolution_out[l] = lv.resolution[l]; }
    return GF_OK;
}

GF_EXPORT int gf_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree, float* dy_dx, void* stream) {
    if (B == 0) return GF_OK;
    if (D != 3) return gf_set_error(GF_ERR_INVALID, "SH encoder only support input dim == 3");
    if (degree < 1 || degree > 8) return gf_set_error(GF_ERR_INVALID, "SH encoder only supports degree in [1, 8]");   // sphere_harmonics.py:70
    if (!inputs || !outputs) return gf_set_error(GF_ERR_INVALID, "sh_encode_forward: null pointer");
    hipLaunchKernelGGL(k_sh, dim3(gf_div_up(B, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), inputs, outputs, B, degree, dy_dx);
    return gf_check_launch("sh_encode_forward");
}

// sh_encode_backward (shencoder.h:10, kernel shencoder.cu:359-383): grad [B, degree^2], dy_dx [B, 3, degree^2], grad_inputs [B,3] accumulates.
GF_EXPORT int gf_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree, const float* dy_dx,
                                    float* grad_inputs, void* stream) {
    (void)inputs;
    if (B == 0) return GF_OK;
    if (D != 3 || degree < 1 || degree > 8) return gf_set_error(GF_ERR_INVALID, "sh_encode_backward: D must be 3, degree 1..8");
    if (!grad || !dy_dx || !grad_inputs) return gf_set_error(GF_ERR_INVALID, "sh_encode_backward: null pointer");
    hipLaunchKernelGGL(k_sh_backward, dim3(gf_div_up(B * 3, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), grad, dy_dx, B, degree * degree, grad_inputs);
    return gf_check_launch("sh_encode_backward");
}

// freq_encode_backward (freqencoder.h:10, kernel freqencoder.cu:63-94): grad / outputs [B,C], grad_inputs [B,D].
GF_EXPORT int gf_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* grad_inputs,
                                      void* stream) {
    if (B == 0) return GF_OK;
    if (C != D + 2 * D * deg) return gf_set_error(GF_ERR_INVALID, "freq_encode_backward: C must equal D + 2*D*deg");
    if (!grad || !outputs || !grad_inputs) return gf_set_error(GF_ERR_INVALID, "freq_encode_backward: null pointer");
    hipLaunchKernelGGL(k_freq_backward, dim3(gf_div_up(B * D, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), grad, outputs, B, D, deg, C, grad_inputs);
    return gf_check_launch("freq_encode_backward");
}

// grid_encode_backward (gridencoder.h:12, kernels gridencoder.cu:248-368).  grad [L,B,C]; grad_embeddings [sO,C] ZERO-FILLED by the
// caller (accumulated with f32 atomics); dy_dx [B, L*D*C] from the forward and grad_inputs [B,D] are both NULL or both given.
GF_EXPORT int gf_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets, float* grad_embeddings,
                                      uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                                      uint32_t gridtype, int align_corners, uint32_t interp, void* stream) {
    (void)embeddings;
    if (B == 0) return GF_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward: null pointer");
    if ((dy_dx == nullptr) != (grad_inputs == nullptr)) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward: dy_dx and grad_inputs go together");
    if (gridtype > 1 || interp > 1) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward: gridtype/interp must be 0 or 1");
    gf::GridLevels lv;
    if (gf::fill_grid_levels(lv, L, S, H) != 0) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward: L must be in [1,32]");
    hipStream_t s = gf_stream(stream);
    const bool ac = align_corners != 0;
    int rc;
    switch (D) {
        case 2: rc = dispatch_backward_c<2>(C, grad, inputs, offsets, grad_embeddings, B, lv, gridtype, ac, interp, s); break;
        case 3: rc = dispatch_backward_c<3>(C, grad, inputs, offsets, grad_embeddings, B, lv, gridtype, ac, interp, s); break;
        case 4: rc = dispatch_backward_c<4>(C, grad, inputs, offsets, grad_embeddings, B, lv, gridtype, ac, interp, s); break;
        case 5: rc = dispatch_backward_c<5>(C, grad, inputs, offsets, grad_embeddings, B, lv, gridtype, ac, interp, s); break;
        default: return gf_set_error(GF_ERR_INVALID, "GridEncoding: D must be 2, 3, 4, or 5.");
    }
    if (rc) return rc;
    if (dy_dx) {
        hipLaunchKernelGGL(k_grid_input_backward, dim3(gf_div_up(B * D, (uint32_t)kBlock)), dim3(kBlock), 0, s, grad, dy_dx, grad_inputs, B, D, C, L);
        return gf_check_launch("grid_encode_backward(inputs)");
    }
    return GF_OK;
}

// gf_grid_encode_backward for a gradient already in [L, B, C] order whose per-level max |g| the caller knows (level_max[L], bit patterns of
// non-negative floats; gf_field_backward writes them next to the gradient): the table scatter without the max pass, no input gradient.
GF_EXPORT int gf_grid_encode_backward_scaled(const float* grad, const float* inputs, const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                             uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                             const uint32_t* level_max, void* stream) {
    if (B == 0) return GF_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings || !level_max) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward_scaled: null pointer");
    if (gridtype > 1 || interp > 1) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward_scaled: gridtype/interp must be 0 or 1");
    gf::GridLevels lv;
    if (gf::fill_grid_levels(lv, L, S, H) != 0) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward_scaled: L must be in [1,32]");
    hipStream_t s = gf_stream(stream);
    const bool ac = align_corners != 0;
    switch (D) {
        case 2: return dispatch_backward_c<2>(C, grad, inputs, offsets, grad_embeddings, B, lv, gridtype, ac, interp, s, level_max);
        case 3: return dispatch_backward_c<3>(C, grad, inputs, offsets, grad_embeddings, B, lv, gridtype, ac, interp, s, level_max);
        default: return gf_set_error(GF_ERR_INVALID, "grid_encode_backward_scaled: D must be 2 or 3");
    }
}

// The same scatter with the BINNING PASS in front (k_grid_bin, round 6): every (point, level) is examined once for the row partitions its
// corners touch, and each of k_grid_backward's workgroups then reads its own partition's list instead of every point.
// workspace: gf_grid_backward_ws_bytes(B, L) bytes of device scratch (32 x 8 counters + L x 8 lists of B indices: sized for 288 GB of HBM, not
// for thrift -- 512 MiB at B = 2^20, L = 16); the table gradient is the one gf_grid_encode_backward_scaled computes up to the rounding of the flush (which points
// share a slice's integer partial changes, and the partials reach the table as float atomics in either form).
GF_EXPORT uint64_t gf_grid_backward_ws_bytes(uint32_t B, uint32_t L) {
    return (uint64_t)gf::kMaxLevels * kGbMaxParts * sizeof(uint32_t) + (uint64_t)L * kGbMaxParts * sizeof(uint32_t) * (uint64_t)B;
}
GF_EXPORT int gf_grid_encode_backward_binned(const float* grad, const float* inputs, const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                             uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                             const uint32_t* level_max, void* workspace, void* stream) {
    if (B == 0) return GF_OK;
    if (!grad || !inputs || !offsets || !grad_embeddings || !level_max || !workspace) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward_binned: null pointer");
    if (gridtype > 1 || interp > 1) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward_binned: gridtype/interp must be 0 or 1");
    gf::GridLevels lv;
    if (gf::fill_grid_levels(lv, L, S, H) != 0) return gf_set_error(GF_ERR_INVALID, "grid_encode_backward_binned: L must be in [1,32]");
    hipStream_t s = gf_stream(stream);
    const bool ac = align_corners != 0;
    uint32_t* counts = static_cast<uint32_t*>(workspace);
    uint32_t* bins = counts + gf::kMaxLevels * kGbMaxParts;
    switch (D) {
        case 2: return dispatch_backward_c<2>(C, grad, inputs, offsets, grad_embeddings, B, lv, gridtype, ac, interp, s, level_max, counts, bins);
        case 3: return dispatch_backward_c<3>(C, grad, inputs, offsets, grad_embeddings, B, lv, gridtype, ac, interp, s, level_max, counts, bins);
        default: return gf_set_error(GF_ERR_INVALID, "grid_encode_backward_binned: D must be 2 or 3");
    }
}

GF_EXPORT int gf_grad_total_variation(const float* inputs, const float* embeddings, float* grad, const int32_t* offsets, float weight, uint32_t B,
                                      uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, void* stream) {
    if (B == 0) return GF_OK;
    if (!inputs || !embeddings || !grad || !offsets) return gf_set_error(GF_ERR_INVALID, "grad_total_variation: null pointer");
    if (gridtype > 1) return gf_set_error(GF_ERR_INVALID, "grad_total_variation: gridtype must be 0 or 1");
    gf::GridLevels lv;
    if (gf::fill_grid_levels(lv, L, S, H) != 0) return gf_set_error(GF_ERR_INVALID, "grad_total_variation: L must be in [1,32]");
    hipStream_t s = gf_stream(stream);
    const bool ac = align_corners != 0;
    switch (D) {
        case 2: return dispatch_tv_c<2>(C, inputs, embeddings, grad, offsets, weight, B, lv, gridtype, ac, s);
        case 3: return dispatch_tv_c<3>(C, inputs, embeddings, grad, offsets, weight, B, lv, gridtype, ac, s);
        case 4: return dispatch_tv_c<4>(C, inputs, embeddings, grad, offsets, weight, B, lv, gridtype, ac, s);
        case 5: return dispatch_tv_c<5>(C, inputs, embeddings, grad, offsets, weight, B, lv, gridtype, ac, s);
        default: return gf_set_error(GF_ERR_INVALID, "GridEncoding: D must be 2, 3, 4, or 5.");
    }
}

GF_EXPORT int gf_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs, void* stream) {
    if (B == 0) return GF_OK;
    if (C != D + 2 * D * deg) return gf_set_error(GF_ERR_INVALID, "freq_encode_forward: C must equal D + 2*D*deg");
    if (!inputs || !outputs) return gf_set_error(GF_ERR_INVALID, "freq_encode_forward: null pointer");
    const uint64_t total = (uint64_t)B * C;
    hipLaunchKernelGGL(k_freq, dim3((uint32_t)gf_div_up<uint64_t>(total, kBlock)), dim3(kBlock), 0, gf_stream(stream), inputs, B, D, C, outputs);
    return gf_check_launch("freq_encode_forward");
}
