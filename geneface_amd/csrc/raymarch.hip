// Stand-alone ray-marching operators behind the `_raymarching_face` seam (C-ABI: include/geneface_hip.h).
// Each replaces one launcher of /root/reference/modules/radnerfs/raymarching/src/raymarching.cu; the
// per-ray arithmetic lives in march_core.hpp.  One lane <-> one ray: HBM-streaming kernels, no reuse.
#include "common.hpp"
#include "march_core.hpp"

namespace {

constexpr int kBlock = 256;

__global__ void __launch_bounds__(kBlock) k_near_far(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ aabb, uint32_t N, float min_near,
                                                     float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float* o = rays_o + (size_t)n * 3;
    const float* d = rays_d + (size_t)n * 3;
    float near, far;
    gf::near_far_from_aabb_1(o[0], o[1], o[2], d[0], d[1], d[2], aabb, min_near, near, far);
    nears[n] = near;
    fars[n] = far;
}

__global__ void __launch_bounds__(kBlock) k_march_rays(gf::MarchParams p, uint32_t n_alive, uint32_t n_step,
                                                       const int* __restrict__ rays_alive, const float* __restrict__ rays_t,
                                                       const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ fars, float* __restrict__ xyzs,
                                                       float* __restrict__ dirs, float* __restrict__ deltas,
                                                       const float* __restrict__ noises) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const float* o = rays_o + (size_t)index * 3;
    const float* d = rays_d + (size_t)index * 3;
    const float dx = d[0], dy = d[1], dz = d[2];
    float t = rays_t[index];
    float* xo = xyzs + (size_t)n * n_step * 3;
    float* dout = dirs + (size_t)n * n_step * 3;
    float* de = deltas + (size_t)n * n_step * 2;
    gf::march_ray(p, o[0], o[1], o[2], dx, dy, dz, fars[index], noises[n], n_step, t,
                  [&](uint32_t s, float x, float y, float z, float dt, float t_after, float) {
                      xo[s * 3 + 0] = x; xo[s * 3 + 1] = y; xo[s * 3 + 2] = z;
                      dout[s * 3 + 0] = dx; dout[s * 3 + 1] = dy; dout[s * 3 + 2] = dz;
                      de[s * 2 + 0] = dt; de[s * 2 + 1] = t_after;
                  });
}

__global__ void __launch_bounds__(kBlock) k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                                                           int* __restrict__ rays_alive, float* __restrict__ rays_t,
                                                           const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                           const float* __restrict__ deltas, float* __restrict__ weights_sum,
                                                           float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const float* sg = sigmas + (size_t)n * n_step;
    const float* cl = rgbs + (size_t)n * n_step * 3;
    const float* de = deltas + (size_t)n * n_step * 2;
    gf::RayAcc a;
    a.t = rays_t[index];
    a.weight_sum = weights_sum[index];
    a.depth = depth[index];
    a.r = image[(size_t)index * 3 + 0];
    a.g = image[(size_t)index * 3 + 1];
    a.b = image[(size_t)index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
        const float dt = de[step * 2];
        if (dt == 0) break;  // the marcher ran out of samples for this ray
        if (!gf::composite_sample(a, sg[step], cl[step * 3], cl[step * 3 + 1], cl[step * 3 + 2], dt, de[step * 2 + 1], T_thresh)) break;
        step++;
    }
    if (step < n_step) rays_alive[n] = -1;
    else rays_t[index] = a.t;
    weights_sum[index] = a.weight_sum;
    depth[index] = a.depth;
    image[(size_t)index * 3 + 0] = a.r;
    image[(size_t)index * 3 + 1] = a.g;
    image[(size_t)index * 3 + 2] = a.b;
}

// ---- occupancy-grid maintenance (raymarching.cu:214-341) ----
__global__ void __launch_bounds__(kBlock) k_morton3d(const int* __restrict__ coords, uint32_t N, int* __restrict__ indices) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int)gf::morton3d((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
__global__ void __launch_bounds__(kBlock) k_morton3d_invert(const int* __restrict__ indices, uint32_t N, int* __restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const int ind = indices[n];
    coords[n * 3 + 0] = (int)gf::morton3d_invert((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int)gf::morton3d_invert((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int)gf::morton3d_invert((uint32_t)(ind >> 2));
}
// one lane packs 8 consecutive cells: two 16-byte loads, one byte store
__global__ void __launch_bounds__(kBlock) k_packbits(const float* __restrict__ grid, uint32_t N, float thresh, uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float4 a = reinterpret_cast<const float4*>(grid)[(size_t)n * 2];
    const float4 b = reinterpret_cast<const float4*>(grid)[(size_t)n * 2 + 1];
    uint32_t bits = 0;
    bits |= (a.x > thresh) ? 1u : 0u;   bits |= (a.y > thresh) ? 2u : 0u;
    bits |= (a.z > thresh) ? 4u : 0u;   bits |= (a.w > thresh) ? 8u : 0u;
    bits |= (b.x > thresh) ? 16u : 0u;  bits |= (b.y > thresh) ? 32u : 0u;
    bits |= (b.z > thresh) ? 64u : 0u;  bits |= (b.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
}
__global__ void __launch_bounds__(kBlock) k_morton3d_dilation(const float* __restrict__ grid, uint32_t C, uint32_t H, float* __restrict__ out) {
    const uint32_t H3 = H * H * H;
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= C * H3) return;
    const uint32_t c = n / H3, ind = n - c * H3;
    const uint32_t x = gf::morton3d_invert(ind), y = gf::morton3d_invert(ind >> 1), z = gf::morton3d_invert(ind >> 2);
    const float* g = grid + (size_t)c * H3;
    float res = grid[n];
    if (x + 1 < H) res = fmaxf(res, g[gf::morton3d(x + 1, y, z)]);
    if (x > 0) res = fmaxf(res, g[gf::morton3d(x - 1, y, z)]);
    if (y + 1 < H) res = fmaxf(res, g[gf::morton3d(x, y + 1, z)]);
    if (y > 0) res = fmaxf(res, g[gf::morton3d(x, y - 1, z)]);
    if (z + 1 < H) res = fmaxf(res, g[gf::morton3d(x, y, z + 1)]);
    if (z > 0) res = fmaxf(res, g[gf::morton3d(x, y, z - 1)]);
    out[n] = res;
}

}  // namespace

GF_EXPORT int gf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                                    float* nears, float* fars, void* stream) {
    if (N == 0) return GF_OK;
    if (!rays_o || !rays_d || !aabb || !nears || !fars) return gf_set_error(GF_ERR_INVALID, "near_far_from_aabb: null pointer");
    hipLaunchKernelGGL(k_near_far, dim3(gf_div_up(N, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), rays_o, rays_d, aabb, N, min_near, nears, fars);
    return gf_check_launch("near_far_from_aabb");
}

GF_EXPORT int gf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                            const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                            uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                            float* dirs, float* deltas, const float* noises, void* stream) {
    (void)nears;
    if (n_alive == 0) return GF_OK;
    if (C < 1 || H < 1 || H > 1024 || max_steps < 1) return gf_set_error(GF_ERR_INVALID, "march_rays: bad C/H/max_steps");
    if (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !fars || !xyzs || !dirs || !deltas || !noises)
        return gf_set_error(GF_ERR_INVALID, "march_rays: null pointer");
    gf::MarchParams p;
    gf::fill_march_params(p, grid, bound, dt_gamma, max_steps, C, H);
    hipLaunchKernelGGL(k_march_rays, dim3(gf_div_up(n_alive, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), p, n_alive, n_step,
                       rays_alive, rays_t, rays_o, rays_d, fars, xyzs, dirs, deltas, noises);
    return gf_check_launch("march_rays");
}

GF_EXPORT int gf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                                const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                                float* image, void* stream) {
    if (n_alive == 0) return GF_OK;
    if (!rays_alive || !rays_t || !sigmas || !rgbs || !deltas || !weights_sum || !depth || !image)
        return gf_set_error(GF_ERR_INVALID, "composite_rays: null pointer");
    hipLaunchKernelGGL(k_composite_rays, dim3(gf_div_up(n_alive, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), n_alive, n_step, T_thresh,
                       rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image);
    return gf_check_launch("composite_rays");
}

GF_EXPORT int gf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream) {
    if (N == 0) return GF_OK;
    hipLaunchKernelGGL(k_morton3d, dim3(gf_div_up(N, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), coords, N, indices);
    return gf_check_launch("morton3D");
}
GF_EXPORT int gf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream) {
    if (N == 0) return GF_OK;
    hipLaunchKernelGGL(k_morton3d_invert, dim3(gf_div_up(N, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), indices, N, coords);
    return gf_check_launch("morton3D_invert");
}
GF_EXPORT int gf_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream) {
    if (N == 0) return GF_OK;
    if (((uintptr_t)grid & 15u) != 0) return gf_set_error(GF_ERR_INVALID, "packbits: grid must be 16-byte aligned");
    hipLaunchKernelGGL(k_packbits, dim3(gf_div_up(N, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), grid, N, density_thresh, bitfield);
    return gf_check_launch("packbits");
}
GF_EXPORT int gf_morton3D_dilation(const float* grid, uint32_t C, uint32_t H, float* grid_dilation, void* stream) {
    const uint32_t total = C * H * H * H;
    if (total == 0) return GF_OK;
    hipLaunchKernelGGL(k_morton3d_dilation, dim3(gf_div_up(total, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), grid, C, H, grid_dilation);
    return gf_check_launch("morton3D_dilation");
}
