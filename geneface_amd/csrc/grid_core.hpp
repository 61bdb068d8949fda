// Multi-resolution tiled / hashed grid lookup shared by the stand-alone encoder op (encoders.hip)
// and the fused field kernels.  Behavioural contract:
//   /root/reference/modules/radnerfs/encoders/gridencoder/src/gridencoder.cu
//     fast_hash :50-63, get_grid_index :66-84, kernel_grid :88-196 (forward), :200-243 (dy_dx)
// Per-level scale / resolution (:138-139) are derived ONCE on the host (glibc exp2f/ceilf, the same
// expressions the CPU oracle evaluates) and handed to the kernels by value, so a device exp2 that is
// an ulp off can never move a level's stride.
#pragma once
#include "common.hpp"
#include <math.h>

namespace gf {

constexpr int kMaxLevels = 32;

struct GridLevels {
    float scale[kMaxLevels];          // exp2f(level*S)*H - 1
    uint32_t resolution[kMaxLevels];  // ceil(scale) + 1
    uint32_t L;
};

inline int fill_grid_levels(GridLevels& g, uint32_t L, float S, uint32_t H) {
    if (L == 0 || L > (uint32_t)kMaxLevels) return -1;
    g.L = L;
    for (uint32_t l = 0; l < L; l++) {
        const float scale = exp2f((float)l * S) * (float)H - 1.0f;
        g.scale[l] = scale;
        g.resolution[l] = (uint32_t)ceilf(scale) + 1;
    }
    return 0;
}

template <uint32_t D>
__device__ __forceinline__ uint32_t fast_hash(const uint32_t (&pos)[D]) {
    constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t r = 0;
#pragma unroll
    for (uint32_t i = 0; i < D; ++i) r ^= pos[i] * primes[i];
    return r;
}

// Row index (in units of C-channel rows) of lattice node `pos` inside one level's table.
// "tiled" stops adding dimensions once the running stride exceeds the table (the reference's
// behaviour: higher dimensions are then ignored); "hash" switches to the xor-prime hash.
template <uint32_t D>
__device__ __forceinline__ uint32_t grid_row(uint32_t gridtype, bool align_corners, uint32_t hashmap_size,
                                             uint32_t resolution, const uint32_t (&pos)[D]) {
    uint32_t stride = 1, index = 0;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= hashmap_size) {
            index += pos[d] * stride;
            stride *= align_corners ? resolution : (resolution + 1);
        }
    }
    if (gridtype == 0 && stride > hashmap_size) index = fast_hash<D>(pos);
    return index % hashmap_size;
}

template <uint32_t C> struct VecOf;
template <> struct VecOf<1> { using type = float; };
template <> struct VecOf<2> { using type = float2; };
template <> struct VecOf<4> { using type = float4; };

template <uint32_t C>
__device__ __forceinline__ void load_row(const float* __restrict__ table, uint32_t row, float (&v)[C]) {
    if constexpr (C == 8) {
        const float4 a = reinterpret_cast<const float4*>(table)[(size_t)row * 2];
        const float4 b = reinterpret_cast<const float4*>(table)[(size_t)row * 2 + 1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if constexpr (C == 4) {
        const float4 a = reinterpret_cast<const float4*>(table)[row];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    } else if constexpr (C == 2) {
        const float2 a = reinterpret_cast<const float2*>(table)[row];
        v[0] = a.x; v[1] = a.y;
    } else {
        v[0] = table[row];
    }
}

// Interpolate one level at one point.  x[] in [0,1] (caller has already ruled out-of-range points out).
// When dy_dx != nullptr it receives D*C derivatives laid out [d][c].
template <uint32_t D, uint32_t C>
__device__ __forceinline__ void grid_level_lookup(const float* __restrict__ table, uint32_t hashmap_size, float scale,
                                                  uint32_t resolution, uint32_t gridtype, bool align_corners,
                                                  uint32_t interp, const float (&x)[D], float (&out)[C], float* dy_dx) {
    float pos[D], pos_deriv[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        pos[d] = __builtin_fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);  // fused on purpose: see oracle/radnerf_kernels.c
        const float fl = floorf(pos[d]);
        pos_grid[d] = (uint32_t)fl;
        pos[d] -= fl;
        if (interp == 1) {
            pos_deriv[d] = 6 * pos[d] * (1.0f - pos[d]);
            pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
        } else {
            pos_deriv[d] = 1.0f;
        }
    }
#pragma unroll
    for (uint32_t c = 0; c < C; c++) out[c] = 0.0f;
#pragma unroll
    for (uint32_t corner = 0; corner < (1u << D); corner++) {
        float w = 1.0f;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t d = 0; d < D; d++) {
            if ((corner & (1u << d)) == 0) { w *= 1 - pos[d]; pl[d] = pos_grid[d]; }
            else { w *= pos[d]; pl[d] = pos_grid[d] + 1; }
        }
        float v[C];
        load_row<C>(table, grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pl), v);
#pragma unroll
        for (uint32_t c = 0; c < C; c++) out[c] += w * v[c];
    }
    if (dy_dx) {
#pragma unroll
        for (uint32_t gd = 0; gd < D; gd++) {
            float rg[C];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) rg[c] = 0.0f;
#pragma unroll
            for (uint32_t corner = 0; corner < (1u << (D - 1)); corner++) {
                float w = scale;
                uint32_t pl[D];
#pragma unroll
                for (uint32_t nd = 0; nd < D - 1; nd++) {
                    const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                    if ((corner & (1u << nd)) == 0) { w *= 1 - pos[d]; pl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pl[d] = pos_grid[d] + 1; }
                }
                float vl[C], vr[C];
                pl[gd] = pos_grid[gd];
                load_row<C>(table, grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pl), vl);
                pl[gd] = pos_grid[gd] + 1;
                load_row<C>(table, grid_row<D>(gridtype, align_corners, hashmap_size, resolution, pl), vr);
#pragma unroll
                for (uint32_t c = 0; c < C; c++) rg[c] += w * (vr[c] - vl[c]) * pos_deriv[gd];
            }
#pragma unroll
            for (uint32_t c = 0; c < C; c++) dy_dx[gd * C + c] = rg[c];
        }
    }
}

// Fused-kernel form: the two 32-lane halves of a wave split a 16-level grid; this lane evaluates levels 8*half .. 8*half+7
// at point x (already mapped to [0,1]) -> f[16] = [level][channel].  meta = LDS table of float4 {scale, resolution, row offset,
// rows} per level (the integer fields bit-cast).
template <uint32_t D>
__device__ __forceinline__ void encode_half(const float* __restrict__ table, const float* __restrict__ meta, int half,
                                            uint32_t gridtype, uint32_t interp, const float (&x)[D], float (&f)[16]) {
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) oob |= (x[d] < 0.0f || x[d] > 1.0f);
#pragma unroll
    for (int l = 0; l < 8; l++) {
        const float4 m = reinterpret_cast<const float4*>(meta)[half * 8 + l];
        float o[2];
        grid_level_lookup<D, 2>(table + (size_t)__float_as_uint(m.z) * 2, __float_as_uint(m.w), m.x, __float_as_uint(m.y),
                                    gridtype, false, interp, x, o, nullptr);
        f[l * 2 + 0] = oob ? 0.0f : o[0];
        f[l * 2 + 1] = oob ? 0.0f : o[1];
    }
}

}  // namespace gf
