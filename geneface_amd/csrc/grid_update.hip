// Device-side occupancy-grid maintenance for gfx950: the tail of NeRFRenderer.update_extra_state and mark_untrained_grid
// (/root/reference/modules/radnerfs/renderer.py:129-196, :247-260) without the reference's Python loops over cell blocks and
// cascades and without a host round trip between the steps:
//
//   k_grid_density (frame_head.hip)   cell -> jittered centre -> density head on the matrix pipe -> tmp_grid[Morton]
//   k_grid_ema                        Morton-space dilation of tmp_grid (raymarching.cu:300-341) fused with the EMA-max into
//                                     density_grid (renderer.py:249-250) and with per-workgroup partial sums of clamp(grid, 0)
//   k_grid_pack                       mean_density from the partial sums (fixed order: run-to-run identical), threshold
//                                     min(mean, density_thresh), 8 cells -> 1 byte (raymarching.cu:268-289)
//
// i.e. three launches per update.  k_mark_untrained is the camera-frustum test of mark_untrained_grid, one lane per cell.
#include "common.hpp"

namespace {

constexpr int kB = 256;

// One lane per (cascade, Morton index).  partial[block] = sum over the block's cells of max(grid, 0) after the update, in double.
__global__ void __launch_bounds__(kB) k_grid_ema(const float* __restrict__ tmp, float* __restrict__ grid, uint32_t C, uint32_t H, float decay,
                                                 double* __restrict__ partial) {
    __shared__ double red[kB / 64];
    const uint32_t H3 = H * H * H;
    const uint32_t n = blockIdx.x * kB + threadIdx.x;
    float g = 0.0f;
    if (n < C * H3) {
        const uint32_t c = n / H3, ind = n - c * H3;
        const uint32_t x = gf::morton3d_invert(ind), y = gf::morton3d_invert(ind >> 1), z = gf::morton3d_invert(ind >> 2);
        const float* t = tmp + (size_t)c * H3;
        float dil = t[ind];
        if (x + 1 < H) dil = fmaxf(dil, t[gf::morton3d(x + 1, y, z)]);
        if (x > 0) dil = fmaxf(dil, t[gf::morton3d(x - 1, y, z)]);
        if (y + 1 < H) dil = fmaxf(dil, t[gf::morton3d(x, y + 1, z)]);
        if (y > 0) dil = fmaxf(dil, t[gf::morton3d(x, y - 1, z)]);
        if (z + 1 < H) dil = fmaxf(dil, t[gf::morton3d(x, y, z + 1)]);
        if (z > 0) dil = fmaxf(dil, t[gf::morton3d(x, y, z - 1)]);
        g = grid[n];
        if (g >= 0.0f && dil >= 0.0f) {   // cells marked untrained (-1) keep their mark (renderer.py:249)
            g = fmaxf(g * decay, dil);
            grid[n] = g;
        }
    }
    double v = (double)fmaxf(g, 0.0f);
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Every workgroup re-derives the mean from the partial sums in the same fixed order, then packs its cells.
__global__ void __launch_bounds__(kB) k_grid_pack(const float* __restrict__ grid, uint32_t n_bytes, const double* __restrict__ partial,
                                                  uint32_t n_partial, double inv_cells, float density_thresh, uint8_t* __restrict__ bitfield,
                                                  float* __restrict__ stats) {
    __shared__ double red[kB];
    double v = 0.0;
    const uint32_t per = (n_partial + kB - 1) / kB;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t j = threadIdx.x * per + k;
        if (j < n_partial) v += partial[j];
    }
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = kB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float mean = (float)(red[0] * inv_cells);
    const float thresh = fminf(mean, density_thresh);
    if (blockIdx.x == 0 && threadIdx.x == 0) { stats[0] = mean; stats[1] = thresh; }
    const uint32_t n = blockIdx.x * kB + threadIdx.x;
    if (n >= n_bytes) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4*>(grid)[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;   bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;   bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;  bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;  bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}

// mark_untrained_grid (renderer.py:129-196): a cell no camera sees gets density -1.  poses [B][4][4] row-major c2w.
__global__ void __launch_bounds__(kB) k_mark_untrained(const float* __restrict__ poses, uint32_t B, float kx /* cx / fx */, float ky /* cy / fy */,
                                                       uint32_t C, uint32_t H, float bound, float* __restrict__ grid) {
#pragma clang fp contract(off)
    const uint32_t H3 = H * H * H;
    const uint32_t n = blockIdx.x * kB + threadIdx.x;
    if (n >= C * H3) return;
    const uint32_t c = n / H3, ind = n - c * H3;
    const uint32_t x = gf::morton3d_invert(ind), y = gf::morton3d_invert(ind >> 1), z = gf::morton3d_invert(ind >> 2);
    const float bound_c = fminf(scalbnf(1.0f, (int)c), bound);
    const float hgs = (float)((double)bound_c / (double)H);
    const float span = (float)((double)bound_c - (double)bound_c / (double)H);
    const float gm1 = (float)(H - 1);
    const float wx = (2.0f * (float)x / gm1 - 1.0f) * span, wy = (2.0f * (float)y / gm1 - 1.0f) * span, wz = (2.0f * (float)z / gm1 - 1.0f) * span;
    const float margin = hgs * 2.0f;
    bool seen = false;
    for (uint32_t b = 0; b < B && !seen; b++) {
        const float* P = poses + (size_t)b * 16;
        const float dx = wx - P[3], dy = wy - P[7], dz = wz - P[11];
        // cam = d @ R  (columns of R are the camera axes)
        const float camx = dx * P[0] + dy * P[4] + dz * P[8];
        const float camy = dx * P[1] + dy * P[5] + dz * P[9];
        const float camz = dx * P[2] + dy * P[6] + dz * P[10];
        seen = camz > 0.0f && fabsf(camx) < kx * camz + margin && fabsf(camy) < ky * camz + margin;
    }
    if (!seen) grid[n] = -1.0f;
}

}  // namespace

// Dilation + EMA-max + mean + bit packing of NeRFRenderer.update_extra_state (renderer.py:247-256), two launches.
// density_grid [C][H^3] (Morton order) is updated in place from tmp_grid (same layout); bitfield gets C*H^3/8 bytes;
// partial_ws = device scratch of gf_grid_update_ws_bytes(C, H) bytes; stats_dev[0] = new mean_density, [1] = the threshold used.
GF_EXPORT uint64_t gf_grid_update_ws_bytes(uint32_t C, uint32_t H) { return (uint64_t)gf_div_up(C * H * H * H, (uint32_t)kB) * sizeof(double); }

GF_EXPORT int gf_grid_update(float* density_grid, const float* tmp_grid, uint32_t C, uint32_t H, float decay, float density_thresh,
                             uint8_t* bitfield, void* partial_ws, float* stats_dev, void* stream) {
    if (!density_grid || !tmp_grid || !bitfield || !partial_ws || !stats_dev) return gf_set_error(GF_ERR_INVALID, "grid_update: null pointer");
    const uint64_t cells = (uint64_t)C * H * H * H;
    if (C == 0 || H == 0 || H > 1024 || cells >= (1ull << 32) || cells % 8) return gf_set_error(GF_ERR_INVALID, "grid_update: bad grid shape");
    const uint32_t nb = gf_div_up((uint32_t)cells, (uint32_t)kB);
    hipLaunchKernelGGL(k_grid_ema, dim3(nb), dim3(kB), 0, gf_stream(stream), tmp_grid, density_grid, C, H, decay, reinterpret_cast<double*>(partial_ws));
    const uint32_t n_bytes = (uint32_t)(cells / 8);
    hipLaunchKernelGGL(k_grid_pack, dim3(gf_div_up(n_bytes, (uint32_t)kB)), dim3(kB), 0, gf_stream(stream), density_grid, n_bytes,
                       reinterpret_cast<const double*>(partial_ws), nb, 1.0 / (double)cells, density_thresh, bitfield, stats_dev);
    return gf_check_launch("grid_update");
}

// NeRFRenderer.mark_untrained_grid (renderer.py:129-196), one launch: poses [B,4,4] c2w (ngp axes) on the device.
GF_EXPORT int gf_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t C, uint32_t H, float bound,
                                     float* density_grid, void* stream) {
    if (!poses || !density_grid) return gf_set_error(GF_ERR_INVALID, "mark_untrained_grid: null pointer");
    const uint64_t cells = (uint64_t)C * H * H * H;
    if (C == 0 || H == 0 || H > 1024 || cells >= (1ull << 32)) return gf_set_error(GF_ERR_INVALID, "mark_untrained_grid: bad grid shape");
    hipLaunchKernelGGL(k_mark_untrained, dim3(gf_div_up((uint32_t)cells, (uint32_t)kB)), dim3(kB), 0, gf_stream(stream), poses, B,
                       (float)((double)cx / (double)fx), (float)((double)cy / (double)fy), C, H, bound, density_grid);
    return gf_check_launch("mark_untrained_grid");
}
