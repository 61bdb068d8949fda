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

// Row of a binary16 table (the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF branch, gridencoder.cu:375-398: under autocast its wrapper
// hands the backend half tables, grid.py:41-44), widened to fp32: the arithmetic stays fp32 (the reference's half kernel rounds its running
// sum to half after every corner).
template <uint32_t C>
__device__ __forceinline__ void load_row(const _Float16* __restrict__ table, uint32_t row, float (&v)[C]) {
    static_assert(C == 2 || C == 4 || C == 8, "half tables: C must be even (grid.py:43)");
    typedef _Float16 hvec __attribute__((ext_vector_type(C)));
    const hvec h = reinterpret_cast<const hvec*>(table)[row];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) v[c] = (float)h[c];
}

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
template <uint32_t D, uint32_t C, class T = float>
__device__ __forceinline__ void grid_level_lookup(const T* __restrict__ table, uint32_t hashmap_size, float scale,
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
    for (uint32_t d = 0; d < D; d++) oob |= !(x[d] >= 0.0f && x[d] <= 1.0f);   // NaN: out of range
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

// ---------------------------------------------------------------------------------------------------------------
// Specialised lookup of the fused head kernel.  Per level the generic index rule above (grid_row) reduces to
//   tiled : (x + y*s1 + z*s2) & mask      s_d = stride of dimension d, 0 once the running stride has passed the table size
//   hash  : (x ^ y*P1 ^ z*P2) & mask      on the levels where the full lattice does not fit the table
// with mask = all ones on dense levels (the index cannot reach the table size) and size-1 on wrapped levels, whose size
// GridEncoder caps at 2^log2_hashmap_size (grid.py:118-134).  A wrapped level whose size is not a power of two is
// rejected on the host (gf_grid_levels_fusable); the generic op kernel handles every case.
struct LevelMeta {   // 8 words, kept in LDS
    float scale; uint32_t s1, s2, mask, row_off, use_hash, pad0, pad1;
};

// One lane per level: derive LevelMeta from the table offsets exactly as grid_row walks the strides.
template <uint32_t D>
__device__ __forceinline__ LevelMeta make_level_meta(float scale, uint32_t resolution, const int* __restrict__ offsets, uint32_t l,
                                                     uint32_t gridtype) {
    const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
    uint32_t stride = 1, sd[3] = {0, 0, 0};
#pragma unroll
    for (uint32_t d = 0; d < D; d++) {
        if (stride <= size) { sd[d] = stride; stride *= resolution + 1; }
    }
    LevelMeta m;
    m.scale = scale;
    m.s1 = sd[1]; m.s2 = sd[2];
    m.mask = stride <= size ? 0xFFFFFFFFu : size - 1u;
    m.row_off = (uint32_t)offsets[l];
    m.use_hash = (gridtype == 0 && stride > size) ? 0xFFFFFFFFu : 0u;
    m.pad0 = m.pad1 = 0;
    return m;
}

// A 16-byte read at an 8-byte-aligned address (two x-adjacent 2-channel rows): global memory needs dword alignment only, and the texture
// path charges a wave instruction the same whether its lanes fetch 8 or 16 bytes, at even or odd rows (tools/gather_probe.hip, modes 1 / 4:
// 2x the rows per clock of 8-byte reads at every table size).
struct __attribute__((aligned(8))) RowPair { float x0, y0, x1, y1; };

// encode8 for TILED grids (gridtype 1: no level hashes; the May configuration): the two x-corners of a cell are ADJACENT rows on every
// level -- idx(x + 1, y, z) = (idx(x, y, z) + 1) & mask -- so each (y, z) corner pair travels as ONE 16-byte read: half the load
// instructions and index arithmetic of the per-corner form, the same values consumed in the same (reference) corner order, hence the
// same bits.  The one exception is a wrapped level's last row (idx == mask: its x-neighbour is row 0 of the level): those lanes read
// rows (mask - 1, mask) instead -- never past the level, so never past the table -- and report it (return value): the caller then redoes
// the wave's lookup corner by corner (2^-16 of the pairs on a 2^16-row level).  (A fix-up of just the affected pairs inside this function
// kept the 64 loaded registers live across a branch and pushed all three head kernels into scratch.)
template <uint32_t D>
__device__ __forceinline__ bool encode8_tiled(const float* __restrict__ table, const LevelMeta* __restrict__ meta8, uint32_t interp,
                                              const float (&x)[D], float (&f)[16]) {
    static_assert(D == 2 || D == 3, "head grids are 2-D or 3-D");
    // levels whose reads are in flight together (A/B knobs: smaller batches did not relieve the register pressure that decided this function's
    // shape -- NOTES 8.2 -- so the per-corner form's batch sizes stay)
#ifndef GF_TILED_LB3
#define GF_TILED_LB3 4
#endif
#ifndef GF_TILED_LB2
#define GF_TILED_LB2 8
#endif
    constexpr int LB = D == 3 ? GF_TILED_LB3 : GF_TILED_LB2, NP = 1 << (D - 1), NC = 1 << D;
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) oob |= !(x[d] >= 0.0f && x[d] <= 1.0f);      // (a NaN coordinate is out of range too)
    // An out-of-range point gets zeros (gridencoder.cu:117-131) -- and must not form table addresses from its coordinates: on a dense level
    // nothing wraps the index, and a far-away point (the module API accepts any position) would read far outside the table
    float xs[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) xs[d] = oob ? 0.0f : x[d];
    const float2* __restrict__ rows = reinterpret_cast<const float2*>(table);
    bool wrapped = false;      // some pair of this lane sits on the last row of a wrapped level: the caller redoes the lookup corner by corner
#pragma unroll
    for (int b = 0; b < 8 / LB; b++) {
        RowPair v[LB][NP];
        float pw[LB][D];
#pragma unroll
        for (int k = 0; k < LB; k++) {
            const int l = b * LB + k;
            const uint4 m0 = reinterpret_cast<const uint4*>(meta8)[2 * l];
            const uint32_t row_off = reinterpret_cast<const uint32_t*>(meta8)[8 * l + 4];
            const float scale = __uint_as_float(m0.x);
            const uint32_t s1 = m0.y, s2 = m0.z, mask = m0.w;
            uint32_t g[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                float p = __builtin_fmaf(xs[d], scale, 0.5f);  // fused on purpose: see oracle/radnerf_kernels.c
                const float fl = floorf(p);
                g[d] = (uint32_t)fl;
                p -= fl;
                if (interp == 1) p = p * p * (3.0f - 2.0f * p);
                pw[k][d] = p;
            }
            const uint32_t y0 = g[1] * s1;
            const uint32_t yy[2] = {g[0] + y0, g[0] + y0 + s1};
            uint32_t zz[2] = {0u, 0u};
            if constexpr (D == 3) { zz[0] = g[2] * s2; zz[1] = zz[0] + s2; }
            const uint32_t last = mask - 1u;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                uint32_t i0 = yy[p & 1];
                if constexpr (D == 3) i0 += zz[p >> 1];
                i0 &= mask;
                wrapped |= i0 == mask;
                const uint32_t ild = i0 < last ? i0 : last;      // (dense levels: mask is all ones, nothing changes)
                v[k][p] = *reinterpret_cast<const RowPair*>(rows + row_off + ild);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < LB; k++) {
            const int l = b * LB + k;
            float w1[D], w0[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) { w1[d] = pw[k][d]; w0[d] = 1 - pw[k][d]; }
            float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)NC; c++) {
                const uint32_t bx = c & 1u, by = (c >> 1) & 1u, bz = (c >> 2) & 1u;
                float w = bx ? w1[0] : w0[0];
                w *= by ? w1[1] : w0[1];
                if constexpr (D == 3) w *= bz ? w1[2] : w0[2];
                const RowPair& r = v[k][c >> 1];
                o0 += w * (bx ? r.x1 : r.x0);
                o1 += w * (bx ? r.y1 : r.y0);
            }
            f[l * 2 + 0] = oob ? 0.0f : o0;
            f[l * 2 + 1] = oob ? 0.0f : o1;
        }
    }
    return wrapped;
}

// Eight consecutive levels at one point x (already mapped to [0,1]) -> f[16] = [level][channel].
// Two explicit steps per batch of levels (four 3-D levels = 32 reads, all eight 2-D levels = 32 reads): every table read of the batch is
// ISSUED, then a scheduling barrier, then the interpolation consumes them in the reference's corner order.  Left to itself the scheduler
// either hoists all reads (what it did until the index arithmetic changed) or -- same source shape, other heuristics outcome -- puts a
// full wait behind every single read; the barrier takes the choice away.
template <uint32_t D>
__device__ __forceinline__ void encode8(const float* __restrict__ table, const LevelMeta* __restrict__ meta8, uint32_t gridtype,
                                        uint32_t interp, const float (&x)[D], float (&f)[16]) {
    static_assert(D == 2 || D == 3, "head grids are 2-D or 3-D");
    constexpr uint32_t P1 = 2654435761u, P2 = 805459861u;
    constexpr int LB = D == 3 ? 4 : 8, NC = 1 << D;
#ifndef GF_NO_PAIRED_ROWS
    // wave-uniform (gridtype is a kernel argument): tiled grids take the paired-row form; the per-corner form below serves hashed grids and
    // redoes the rare wave (2-3 % of them at 2^16-row levels) in which some lane's pair wrapped around the end of a level -- same bits for
    // every other lane, the right rows for that one
    if (gridtype == 1u && !__any(encode8_tiled<D>(table, meta8, interp, x, f))) return;
#endif
    bool oob = false;
#pragma unroll
    for (uint32_t d = 0; d < D; d++) oob |= !(x[d] >= 0.0f && x[d] <= 1.0f);      // (a NaN coordinate is out of range too)
    // An out-of-range point gets zeros (gridencoder.cu:117-131) -- and must not form table addresses from its coordinates: on a dense level
    // nothing wraps the index, and a far-away point (the module API accepts any position) would read far outside the table
    float xs[D];
#pragma unroll
    for (uint32_t d = 0; d < D; d++) xs[d] = oob ? 0.0f : x[d];
    const float2* __restrict__ rows = reinterpret_cast<const float2*>(table);
    (void)gridtype;
#pragma unroll
    for (int b = 0; b < 8 / LB; b++) {
        float2 v[LB][NC];
        float pw[LB][D];
#pragma unroll
        for (int k = 0; k < LB; k++) {
            const int l = b * LB + k;
            const uint4 m0 = reinterpret_cast<const uint4*>(meta8)[2 * l];
            const uint2 m1 = reinterpret_cast<const uint2*>(meta8)[4 * l + 2];
            const float scale = __uint_as_float(m0.x);
            const uint32_t s1 = m0.y, s2 = m0.z, mask = m0.w, row_off = m1.x, use_hash = m1.y;
            uint32_t g[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) {
                float p = __builtin_fmaf(xs[d], scale, 0.5f);  // fused on purpose: see oracle/radnerf_kernels.c
                const float fl = floorf(p);
                g[d] = (uint32_t)fl;
                p -= fl;
                if (interp == 1) p = p * p * (3.0f - 2.0f * p);
                pw[k][d] = p;
            }
            // Index of corner (bx, by, bz): hashed levels xor prime multiples, dense levels (and tiled grids) add strided coordinates.  ONE
            // expression serves both, ((x ^ yh) + yt ^ zh) + zt with the unused half of each pair zeroed by the level's mask (use_hash is
            // all ones or zero; it is zero on every level of a tiled grid): two v_xad_u32 per corner instead of both index forms and a
            // select, one quarter-rate integer multiply per dimension instead of two.
            const uint32_t hm = use_hash, dm = ~use_hash;
            const uint32_t my = hm ? P1 : s1;
            const uint32_t y0 = g[1] * my, y1 = y0 + my;
            const uint32_t yh[2] = {y0 & hm, y1 & hm}, yt[2] = {y0 & dm, y1 & dm};
            uint32_t zh[2] = {0u, 0u}, zt[2] = {0u, 0u};
            if constexpr (D == 3) {
                const uint32_t mz = hm ? P2 : s2;
                const uint32_t z0 = g[2] * mz, z1 = z0 + mz;
                zh[0] = z0 & hm; zh[1] = z1 & hm; zt[0] = z0 & dm; zt[1] = z1 & dm;
            }
            const uint32_t ix[2] = {g[0], g[0] + 1u};
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)NC; c++) {
                const uint32_t bx = c & 1u, by = (c >> 1) & 1u, bz = (c >> 2) & 1u;
                uint32_t idx = (ix[bx] ^ yh[by]) + yt[by];
                if constexpr (D == 3) idx = (idx ^ zh[bz]) + zt[bz];
                v[k][c] = rows[row_off + (idx & mask)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < LB; k++) {
            const int l = b * LB + k;
            float w1[D], w0[D];
#pragma unroll
            for (uint32_t d = 0; d < D; d++) { w1[d] = pw[k][d]; w0[d] = 1 - pw[k][d]; }
            float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
            for (uint32_t c = 0; c < (uint32_t)NC; c++) {
                const uint32_t bx = c & 1u, by = (c >> 1) & 1u, bz = (c >> 2) & 1u;
                float w = bx ? w1[0] : w0[0];
                w *= by ? w1[1] : w0[1];
                if constexpr (D == 3) w *= bz ? w1[2] : w0[2];
                o0 += w * v[k][c].x;
                o1 += w * v[k][c].y;
            }
            f[l * 2 + 0] = oob ? 0.0f : o0;
            f[l * 2 + 1] = oob ? 0.0f : o1;
        }
    }
}


// Input gradient of the 2-D lookup for the fused training backward: this lane's eight levels at point x, output gradients gf[16] =
// [level][channel] -> the lane's share of d loss / d x (the caller adds the two lane halves).  Same derivative as kernel_grid's dy_dx
// (gridencoder.cu:163-196) contracted with the output gradient as kernel_input_backward does (:343-368).
__device__ __forceinline__ void encode8_grad2(const float* __restrict__ table, const LevelMeta* __restrict__ meta8, uint32_t gridtype,
                                              uint32_t interp, const float (&x)[2], const float (&gf)[16], float (&dx)[2]) {
    constexpr uint32_t P1 = 2654435761u;
    dx[0] = dx[1] = 0.0f;
    if (!(x[0] >= 0.0f && x[0] <= 1.0f && x[1] >= 0.0f && x[1] <= 1.0f)) return;      // out of range, or NaN
    const float2* __restrict__ rows = reinterpret_cast<const float2*>(table);
#pragma unroll
    for (int l = 0; l < 8; l++) {
        const uint4 m0 = reinterpret_cast<const uint4*>(meta8)[2 * l];
        const uint2 m1 = reinterpret_cast<const uint2*>(meta8)[4 * l + 2];
        const float scale = __uint_as_float(m0.x);
        const uint32_t s1 = m0.y, mask = m0.w, row_off = m1.x, use_hash = m1.y;
        float p[2], der[2];
        uint32_t g[2];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            float q = __builtin_fmaf(x[d], scale, 0.5f);
            const float fl = floorf(q);
            g[d] = (uint32_t)fl;
            q -= fl;
            if (interp == 1) { der[d] = 6 * q * (1.0f - q); q = q * q * (3.0f - 2.0f * q); }
            else der[d] = 1.0f;
            p[d] = q;
        }
        float2 v[4];
#pragma unroll
        for (uint32_t c = 0; c < 4; c++) {
            const uint32_t px = g[0] + (c & 1u), py = g[1] + ((c >> 1) & 1u);
            const uint32_t idx = (gridtype == 0 && use_hash) ? (px ^ (py * P1)) : (px + py * s1);
            v[c] = rows[row_off + (idx & mask)];
        }
        // d/dx0: corners differ in bit 0; d/dx1: in bit 1
        float w = scale * (1 - p[1]);
        float d00 = w * (v[1].x - v[0].x) * der[0], d01 = w * (v[1].y - v[0].y) * der[0];
        w = scale * p[1];
        d00 += w * (v[3].x - v[2].x) * der[0]; d01 += w * (v[3].y - v[2].y) * der[0];
        w = scale * (1 - p[0]);
        float d10 = w * (v[2].x - v[0].x) * der[1], d11 = w * (v[2].y - v[0].y) * der[1];
        w = scale * p[0];
        d10 += w * (v[3].x - v[1].x) * der[1]; d11 += w * (v[3].y - v[1].y) * der[1];
        dx[0] += gf[2 * l] * d00 + gf[2 * l + 1] * d01;
        dx[1] += gf[2 * l] * d10 + gf[2 * l + 1] * d11;
    }
}

// HOST: can the fused kernels' specialised lookup reproduce get_grid_index for these tables?  offsets is a HOST array [L+1].
inline bool grid_levels_fusable(const int* offsets, uint32_t L, uint32_t D, float S, uint32_t H) {
    GridLevels lv;
    if (fill_grid_levels(lv, L, S, H)) return false;
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
        if (size == 0) return false;
        uint32_t stride = 1;
        for (uint32_t d = 0; d < D; d++)
            if (stride <= size) stride *= lv.resolution[l] + 1;
        if (stride > size && (size & (size - 1u)) != 0) return false;  // wrapped level with a non power-of-two table
    }
    return true;
}

}  // namespace gf
