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

// raymarching.cu:161-198: where the ray leaves the sphere |x| = radius, as (polar, azimuth) angles scaled to [-1, 1] (y up)
__global__ void __launch_bounds__(kBlock) k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float radius, uint32_t N,
                                                         float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[(size_t)n * 3], oy = rays_o[(size_t)n * 3 + 1], oz = rays_o[(size_t)n * 3 + 2];
    const float dx = rays_d[(size_t)n * 3], dy = rays_d[(size_t)n * 3 + 1], dz = rays_d[(size_t)n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;   // half of the linear coefficient
    const float C = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * C)) / A;   // the far root
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float rpi = 0.3183098861837907f;
    coords[(size_t)n * 2] = 2 * atan2f(sqrtf(x * x + z * z), y) * rpi - 1;
    coords[(size_t)n * 2 + 1] = atan2f(z, x) * rpi;
}

GF_EXPORT int gf_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream) {
    if (N == 0) return GF_OK;
    if (!rays_o || !rays_d || !coords) return gf_set_error(GF_ERR_INVALID, "sph_from_ray: null pointer");
    hipLaunchKernelGGL(k_sph_from_ray, dim3(gf_div_up(N, (uint32_t)kBlock)), dim3(kBlock), 0, gf_stream(stream), rays_o, rays_d, radius, N, coords);
    return gf_check_launch("sph_from_ray");
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

// =====================================================================================================================
// Training tier: march_rays_train (+ backward), composite_rays_train forward / backward.
//   /root/reference/modules/radnerfs/raymarching/src/raymarching.cu :353-518, :536-583, :604-687, :712-809
// The CUDA kernel hands out point offsets and ray slots with two atomicAdds per ray, so its sample order differs from run to run
// ("known benign race").  Here a ray's offset is the exclusive prefix sum of the per-ray sample counts, i.e. ray
// order -- one of the orders the reference can produce, deterministic, and no atomics:
//   pass 1  k_train_count : one lane per ray counts its samples (first pass of the reference kernel), a workgroup scan leaves the
//                           in-block exclusive prefix per ray and the block total
//   pass 2  k_train_scan  : one workgroup scans the block totals (N / 256 <= 4096 blocks)
//   pass 3  k_train_write : second pass of the reference kernel at offset = counter[0] + block prefix + in-block prefix
// =====================================================================================================================
namespace {

constexpr int kTB = 256;

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* lds /*[8]*/, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += lds[w];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + incl - v;
}

__global__ void __launch_bounds__(kTB) k_train_count(gf::MarchParams mp, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ nears, const float* __restrict__ fars,
                                                     const float* __restrict__ noises, uint32_t N, uint32_t max_steps,
                                                     uint32_t* __restrict__ counts, uint32_t* __restrict__ prefix, uint32_t* __restrict__ block_tot) {
    __shared__ uint32_t lds[8];
    const uint32_t n = blockIdx.x * kTB + threadIdx.x;
    uint32_t cnt = 0;
    if (n < N) {
        float t = nears[n];
        cnt = gf::march_ray(mp, rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2], rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2], fars[n],
                            noises[n], max_steps, t, [](uint32_t, float, float, float, float, float, float) {});
        counts[n] = cnt;
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan_256(cnt, lds, total);
    if (n < N) prefix[n] = ex;
    if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}

// Point offsets are an exclusive prefix sum of the per-ray counts in ray order, started at block `rot` and wrapping around: when the
// caller's buffer (mean_count of the last steps) is too small, the rays that lose their samples are the tail of THAT order.  rot comes from
// the call's own jitter (noises[0], uniform in [0,1) under perturb=True, 0 otherwise), so with perturbation the dropped block moves from call
// to call like the reference's (arbitrary, atomics decide), instead of always being the bottom rows of a raster-ordered batch; without
// perturbation the order is plain ray order and the result deterministic.
__global__ void __launch_bounds__(1024) k_train_scan(uint32_t* __restrict__ block_tot, uint32_t nblocks, int* __restrict__ counter, uint32_t N,
                                                     uint32_t* __restrict__ base_out, const float* __restrict__ noises) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t s_total;
    // each thread owns a contiguous run of blocks
    const uint32_t per = (nblocks + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < nblocks ? lo + per : nblocks;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += block_tot[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; i++) { const uint32_t v = part[i]; part[i] = run; run += v; }
        *base_out = (uint32_t)counter[0];
        counter[0] += (int)run;     // what the reference's atomicAdd(counter, num_steps) accumulates
        counter[1] += (int)N;       // ... and atomicAdd(counter + 1, 1)
        s_total = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = block_tot[i]; block_tot[i] = run; run += v; }
    uint32_t rot = (uint32_t)(noises[0] * (float)nblocks);
    rot = rot < nblocks ? rot : nblocks - 1;
    if (rot == 0) return;           // uniform
    __syncthreads();
    const uint32_t er = block_tot[rot], total = s_total;
    __syncthreads();
    for (uint32_t i = lo; i < hi; i++) { const uint32_t v = block_tot[i]; block_tot[i] = i >= rot ? v - er : v - er + total; }
}

__global__ void __launch_bounds__(kTB) k_train_write(gf::MarchParams mp, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ nears, const float* __restrict__ fars,
                                                     const float* __restrict__ noises, uint32_t N, uint32_t M, const uint32_t* __restrict__ counts,
                                                     const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ block_off,
                                                     const uint32_t* __restrict__ base, int ray_slot0, float* __restrict__ xyzs,
                                                     float* __restrict__ dirs, float* __restrict__ deltas, int* __restrict__ rays) {
    const uint32_t n = blockIdx.x * kTB + threadIdx.x;
    if (n >= N) return;
    const uint32_t num_steps = counts[n];
    const uint32_t point_index = *base + block_off[blockIdx.x] + prefix[n];
    const uint32_t slot = (uint32_t)ray_slot0 + n;
    rays[slot * 3] = (int)n;
    rays[slot * 3 + 1] = (int)point_index;
    rays[slot * 3 + 2] = (int)num_steps;
    if (num_steps == 0 || point_index + num_steps > M) return;
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    float t = nears[n];
    gf::march_ray(mp, rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2], dx, dy, dz, fars[n], noises[n], num_steps, t,
                  [&](uint32_t q, float x, float y, float z, float dt, float t_after, float) {
                      const size_t p = (size_t)point_index + q;
                      xyzs[p * 3] = x; xyzs[p * 3 + 1] = y; xyzs[p * 3 + 2] = z;
                      dirs[p * 3] = dx; dirs[p * 3 + 1] = dy; dirs[p * 3 + 2] = dz;
                      deltas[p * 2] = dt; deltas[p * 2 + 1] = t_after;
                  });
}

__global__ void __launch_bounds__(kTB) k_march_train_backward(const float* __restrict__ grad_xyzs, const float* __restrict__ grad_dirs,
                                                              const int* __restrict__ rays, const float* __restrict__ deltas, uint32_t N, uint32_t M,
                                                              float* __restrict__ grad_rays_o, float* __restrict__ grad_rays_d) {
    const uint32_t n = blockIdx.x * kTB + threadIdx.x;
    if (n >= N) return;
    const uint32_t offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    float go[3] = {grad_rays_o[n * 3], grad_rays_o[n * 3 + 1], grad_rays_o[n * 3 + 2]};
    float gd[3] = {grad_rays_d[n * 3], grad_rays_d[n * 3 + 1], grad_rays_d[n * 3 + 2]};
    for (uint32_t s = 0; s < num_steps; s++) {
#pragma clang fp contract(off)
        const size_t p = (size_t)offset + s;
        const float t = deltas[p * 2 + 1];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float gx = grad_xyzs[p * 3 + c];
            go[c] += gx;
            gd[c] += gx * t + grad_dirs[p * 3 + c];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { grad_rays_o[n * 3 + c] = go[c]; grad_rays_d[n * 3 + c] = gd[c]; }
}

__global__ void __launch_bounds__(kTB) k_composite_train_forward(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                 const float* __restrict__ ambient, const float* __restrict__ deltas,
                                                                 const int* __restrict__ rays, uint32_t M, uint32_t N, float T_thresh,
                                                                 float* __restrict__ weights_sum, float* __restrict__ ambient_sum,
                                                                 float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * kTB + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0, amb = 0;
    if (!(num_steps == 0 || offset + num_steps > M)) {
#pragma clang fp contract(off)
        for (uint32_t s = 0; s < num_steps; s++) {
            const size_t p = (size_t)offset + s;
            const float alpha = 1.0f - __expf(-sigmas[p] * deltas[p * 2]);
            const float weight = alpha * T;
            r += weight * rgbs[p * 3]; g += weight * rgbs[p * 3 + 1]; b += weight * rgbs[p * 3 + 2];
            d += weight * deltas[p * 2 + 1];
            ws += weight;
            amb += ambient[p];
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
        }
    }
    weights_sum[index] = ws; ambient_sum[index] = amb; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

__global__ void __launch_bounds__(kTB) k_composite_train_backward(const float* __restrict__ grad_weights_sum, const float* __restrict__ grad_ambient_sum,
                                                                  const float* __restrict__ grad_image, const float* __restrict__ sigmas,
                                                                  const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                                  const int* __restrict__ rays, const float* __restrict__ weights_sum,
                                                                  const float* __restrict__ image, uint32_t M, uint32_t N, float T_thresh,
                                                                  float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs,
                                                                  float* __restrict__ grad_ambient) {
    const uint32_t n = blockIdx.x * kTB + threadIdx.x;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) return;
    const float gws = grad_weights_sum[index], gas = grad_ambient_sum[index];
    const float gi0 = grad_image[index * 3], gi1 = grad_image[index * 3 + 1], gi2 = grad_image[index * 3 + 2];
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2], ws_final = weights_sum[index];
    float T = 1.0f, r = 0, g = 0, b = 0;
    for (uint32_t s = 0; s < num_steps; s++) {
#pragma clang fp contract(off)
        const size_t p = (size_t)offset + s;
        const float dt = deltas[p * 2];
        const float c0 = rgbs[p * 3], c1 = rgbs[p * 3 + 1], c2 = rgbs[p * 3 + 2];
        const float alpha = 1.0f - __expf(-sigmas[p] * dt);
        const float weight = alpha * T;
        r += weight * c0; g += weight * c1; b += weight * c2;
        T *= 1.0f - alpha;
        grad_rgbs[p * 3] = gi0 * weight; grad_rgbs[p * 3 + 1] = gi1 * weight; grad_rgbs[p * 3 + 2] = gi2 * weight;
        grad_ambient[p] = gas;
        grad_sigmas[p] = dt * (gi0 * (T * c0 - (r_final - r)) + gi1 * (T * c1 - (g_final - g)) + gi2 * (T * c2 - (b_final - b)) + gws * (1 - ws_final));
        if (T < T_thresh) break;
    }
}

}  // namespace

GF_EXPORT uint64_t gf_march_rays_train_workspace_bytes(uint32_t N) { return ((uint64_t)2 * N + gf_div_up(N, (uint32_t)kTB) + 64) * 4; }

// march_rays_train (raymarching.h:13).  xyzs/dirs [M,3] and deltas [M,2] ZERO-FILLED by the caller; rays int32 [N,3] receives
// (ray, offset, count) in RAY ORDER; counter int32 [2] accumulates (points, rays) like the reference's atomics; `workspace` =
// gf_march_rays_train_workspace_bytes(N) device bytes of scratch.
GF_EXPORT int gf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                                  uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                                  float* deltas, int32_t* rays, int32_t* counter, const float* noises, void* workspace, void* stream) {
    if (N == 0) return GF_OK;
    if (!rays_o || !rays_d || !grid || !nears || !fars || !xyzs || !dirs || !deltas || !rays || !counter || !noises || !workspace)
        return gf_set_error(GF_ERR_INVALID, "march_rays_train: null pointer");
    if (max_steps == 0 || C == 0 || H == 0 || H > 1024) return gf_set_error(GF_ERR_INVALID, "march_rays_train: bad configuration");
    const uint32_t nblocks = gf_div_up(N, (uint32_t)kTB);
    if (nblocks > 1024u * 64u) return gf_set_error(GF_ERR_UNSUPPORTED, "march_rays_train: too many rays for one call");
    gf::MarchParams mp;
    gf::fill_march_params(mp, grid, bound, dt_gamma, max_steps, C, H);
    uint32_t* counts = reinterpret_cast<uint32_t*>(workspace);
    uint32_t* prefix = counts + N;
    uint32_t* block_tot = prefix + N;
    uint32_t* base = block_tot + nblocks;
    hipStream_t s = gf_stream(stream);
    hipLaunchKernelGGL(k_train_count, dim3(nblocks), dim3(kTB), 0, s, mp, rays_o, rays_d, nears, fars, noises, N, max_steps, counts, prefix, block_tot);
    hipLaunchKernelGGL(k_train_scan, dim3(1), dim3(1024), 0, s, block_tot, nblocks, counter, N, base, noises);
    hipLaunchKernelGGL(k_train_write, dim3(nblocks), dim3(kTB), 0, s, mp, rays_o, rays_d, nears, fars, noises, N, M, counts, prefix, block_tot, base, 0,
                       xyzs, dirs, deltas, rays);
    return gf_check_launch("march_rays_train");
}

// march_rays_train_backward (raymarching.h:14): grad_rays_o / grad_rays_d [N,3] accumulate (the wrapper zero-fills them).
GF_EXPORT int gf_march_rays_train_backward(const float* grad_xyzs, const float* grad_dirs, const int32_t* rays, const float* deltas, uint32_t N,
                                           uint32_t M, float* grad_rays_o, float* grad_rays_d, void* stream) {
    if (N == 0) return GF_OK;
    if (!grad_xyzs || !grad_dirs || !rays || !deltas || !grad_rays_o || !grad_rays_d) return gf_set_error(GF_ERR_INVALID, "march_rays_train_backward: null pointer");
    hipLaunchKernelGGL(k_march_train_backward, dim3(gf_div_up(N, (uint32_t)kTB)), dim3(kTB), 0, gf_stream(stream), grad_xyzs, grad_dirs, rays, deltas, N, M,
                       grad_rays_o, grad_rays_d);
    return gf_check_launch("march_rays_train_backward");
}

// composite_rays_train_forward (raymarching.h:15)
GF_EXPORT int gf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ambient, const float* deltas, const int32_t* rays,
                                              uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* ambient_sum, float* depth, float* image,
                                              void* stream) {
    if (N == 0) return GF_OK;
    if (!sigmas || !rgbs || !ambient || !deltas || !rays || !weights_sum || !ambient_sum || !depth || !image)
        return gf_set_error(GF_ERR_INVALID, "composite_rays_train_forward: null pointer");
    hipLaunchKernelGGL(k_composite_train_forward, dim3(gf_div_up(N, (uint32_t)kTB)), dim3(kTB), 0, gf_stream(stream), sigmas, rgbs, ambient, deltas, rays, M,
                       N, T_thresh, weights_sum, ambient_sum, depth, image);
    return gf_check_launch("composite_rays_train_forward");
}

// composite_rays_train_backward (raymarching.h:16): grad_sigmas [M], grad_rgbs [M,3], grad_ambient [M] ZERO-FILLED by the caller.
GF_EXPORT int gf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_ambient_sum, const float* grad_image, const float* sigmas,
                                               const float* rgbs, const float* ambient, const float* deltas, const int32_t* rays, const float* weights_sum,
                                               const float* ambient_sum, const float* image, uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas,
                                               float* grad_rgbs, float* grad_ambient, void* stream) {
    (void)ambient; (void)ambient_sum;
    if (N == 0) return GF_OK;
    if (!grad_weights_sum || !grad_ambient_sum || !grad_image || !sigmas || !rgbs || !deltas || !rays || !weights_sum || !image || !grad_sigmas || !grad_rgbs ||
        !grad_ambient)
        return gf_set_error(GF_ERR_INVALID, "composite_rays_train_backward: null pointer");
    hipLaunchKernelGGL(k_composite_train_backward, dim3(gf_div_up(N, (uint32_t)kTB)), dim3(kTB), 0, gf_stream(stream), grad_weights_sum, grad_ambient_sum,
                       grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient);
    return gf_check_launch("composite_rays_train_backward");
}
