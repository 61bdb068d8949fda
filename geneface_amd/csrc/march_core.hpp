// Per-ray occupancy-grid marching and front-to-back compositing, shared by the stand-alone
// op kernels (raymarch.hip) and the fused frame kernels (frame_head.hip) so both take
// bit-identical sample positions.
//
// Behavioural contract (what the numbers must equal), reference paths relative to
// /root/reference/modules/radnerfs/raymarching/src/raymarching.cu:
//   slab test            : kernel_near_far_from_aabb   :92-145
//   marching + DDA skip  : kernel_march_rays           :828-929
//   compositing          : kernel_composite_rays       :943-1029
// Discrete decisions (voxel index, sample count) have to agree exactly with the CPU oracle, so
// contraction is disabled in this header and the single a*b+c the reference's compiler fuses on
// the decision path (the sample position o + t*d) is an explicit fma.
#pragma once
#include "common.hpp"
#include <float.h>

namespace gf {

struct MarchParams {
    const uint8_t* grid;  // [C*H^3/8] Morton-ordered occupancy bits, LSB first
    float bound;
    float dt_gamma;
    float dt_min, dt_max;
    float rH;    // 1/H
    float H3f;   // (float)(H*H*H)
    float Hf;    // (float)H
    float Hm1f;  // (float)(H-1)
    float Cf;    // (float)C  (number of cascades)
    uint32_t H;
};

// host side: fill the derived fields exactly as the reference derives them per thread (:861-866)
inline void fill_march_params(MarchParams& p, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                              uint32_t C, uint32_t H) {
    const float sqrt3 = 1.7320508075688772f;
    p.grid = grid;
    p.bound = bound;
    p.dt_gamma = dt_gamma;
    p.dt_max = 2 * sqrt3 * (float)(1 << (C - 1)) / (float)H;
    const float alt = 2 * sqrt3 / (float)max_steps;
    p.dt_min = p.dt_max < alt ? p.dt_max : alt;
    p.rH = 1 / (float)H;
    p.H3f = (float)(H * H * H);
    p.Hf = (float)H;
    p.Hm1f = (float)(H - 1);
    p.Cf = (float)C;
    p.H = H;
}

#pragma clang fp contract(off)

__device__ __forceinline__ void near_far_from_aabb_1(float ox, float oy, float oz, float dx, float dy, float dz,
                                                     const float* __restrict__ aabb, float min_near, float& near_out,
                                                     float& far_out) {
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
    if (near > far) { const float c = near; near = far; far = c; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { const float c = near_y; near_y = far_y; far_y = c; }
    if (near > far_y || near_y > far) { near_out = far_out = FLT_MAX; return; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { const float c = near_z; near_z = far_z; far_z = c; }
    if (near > far_z || near_z > far) { near_out = far_out = FLT_MAX; return; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    near_out = near;
    far_out = far;
}

__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float Hf, float max_cascade) {
    const float mx = (float)((double)(dt * Hf) * 0.5);
    int e;
    frexpf(mx, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}

// March one ray from t (updated in place) for at most n_step occupied samples.
// emit(step, x, y, z, dt, t_after, t_at) is called once per sample, in order (t_at = the ray parameter the sample was
// taken at: restarting the march from t_at with noise 0 reproduces this sample and everything after it bit for bit).
// Returns the sample count.  `max_iters` bounds the loop trips (samples + empty-space skips) of this call: a caller that
// keeps t may stop early and resume later (the loop carries no state but t), which lets a workgroup of rays advance in
// bounded, similar-length slices instead of waiting for its longest empty-space traversal.
template <typename Emit>
__device__ __forceinline__ uint32_t march_ray(const MarchParams& p, float ox, float oy, float oz, float dx, float dy,
                                              float dz, float far, float noise, uint32_t n_step, float& t, Emit&& emit,
                                              uint32_t max_iters = 0xFFFFFFFFu) {
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    uint32_t step = 0, iters = 0;
    t += clampf(t * p.dt_gamma, p.dt_min, p.dt_max) * noise;

    while (t < far && step < n_step && iters < max_iters) {
        iters++;
        const float x = clampf(__builtin_fmaf(t, dx, ox), -p.bound, p.bound);
        const float y = clampf(__builtin_fmaf(t, dy, oy), -p.bound, p.bound);
        const float z = clampf(__builtin_fmaf(t, dz, oz), -p.bound, p.bound);
        const float dt = clampf(t * p.dt_gamma, p.dt_min, p.dt_max);

        int level = 0;
        if (p.Cf > 1.0f) {  // a single cascade (bound <= 1) always resolves to level 0
            const int lp = mip_from_pos(x, y, z, p.Cf), ld = mip_from_dt(dt, p.Hf, p.Cf);
            level = lp > ld ? lp : ld;
        }
        const float mip_bound = fminf(scalbnf(1.0f, level), p.bound);
        const float mip_rbound = 1 / mip_bound;

        const int nx = (int)clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)p.Hf), 0.0f, p.Hm1f);
        const int ny = (int)clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)p.Hf), 0.0f, p.Hm1f);
        const int nz = (int)clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)p.Hf), 0.0f, p.Hm1f);

        const uint32_t idx = (uint32_t)((float)level * p.H3f + (float)morton3d((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
        const bool occ = p.grid[idx >> 3] & (1u << (idx & 7u));

        if (occ) {
            const float t_at = t;
            t += dt;
            emit(step, x, y, z, dt, t, t_at);
            step++;
        } else {
            const float tx = ((((float)nx + 0.5f + 0.5f * copysignf(1.0f, dx)) * p.rH * 2 - 1) * mip_bound - x) * rdx;
            const float ty = ((((float)ny + 0.5f + 0.5f * copysignf(1.0f, dy)) * p.rH * 2 - 1) * mip_bound - y) * rdy;
            const float tz = ((((float)nz + 0.5f + 0.5f * copysignf(1.0f, dz)) * p.rH * 2 - 1) * mip_bound - z) * rdz;
            const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
            do {
                t += clampf(t * p.dt_gamma, p.dt_min, p.dt_max);
            } while (t < tt);
        }
    }
    return step;
}

// Accumulators of one ray while it is being composited.
struct RayAcc {
    float weight_sum, depth, r, g, b, t;
};

// Composite one sample; returns false when the ray terminates *after* this sample (T < T_thresh).
__device__ __forceinline__ bool composite_sample(RayAcc& a, float sigma, float cr, float cg, float cb, float dt,
                                                 float t_after, float T_thresh) {
    const float alpha = 1.0f - __expf(-sigma * dt);
    const float T = 1 - a.weight_sum;
    const float w = alpha * T;
    a.weight_sum += w;
    a.t = t_after;
    a.depth += w * t_after;
    a.r += w * cr;
    a.g += w * cg;
    a.b += w * cb;
    return !(T < T_thresh);
}

#pragma clang fp contract(fast)

}  // namespace gf
