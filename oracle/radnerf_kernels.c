/*
 * oracle/radnerf_kernels.c -- TEST INFRASTRUCTURE ONLY (the parity oracle).
 *
 * A plain-C, CPU restatement of the forward arithmetic of the four CUDA extensions
 * that sit under GeneFace's RAD-NeRF renderer.  Nothing under geneface_amd/ may
 * import, link or execute this file: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker.
 *
 * PARITY STATUS: pinned.  The reference ships no tests, golden vectors or checkpoints, but
 * its four CUDA translation units compile unchanged for gfx950 (oracle/refbuild/build_ref.py
 * -> oracle/_ref/), so every kernel below is compared on an MI355X with the running kernel
 * it restates (tests/test_gpu_vs_ref_kernels.py: integer decisions identical, floats within
 * a few ulp; profiles/round1/r1z_ref_kernels_report.json).  In this GPU-less container it is
 * additionally held by (1) analytic known-answer tests (tests/test_oracle_kat.py), (2) the
 * reference's own, unmodified Python layers executing on top of this file (oracle/refshim.py
 * + tests/golden/make_golden.py) and (3) cross-checks against the torch restatement in
 * oracle/radnerf_ref.py.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/modules/radnerfs/).  One thread of the CUDA grid == one
 * iteration of the outermost loop here.
 *
 * Floating point: compiled with -ffp-contract=off.  nvcc's default --fmad=true
 * fuses a*b+c; the only place where that can change a *discrete* decision is the
 * marcher's position  o + t*d  (it selects the occupancy voxel); the only place where
 * it is visible above rounding noise is the grid encoder's  x*scale + 0.5  (a large
 * intermediate).  Those two expressions are written as explicit fmaf() here and in the
 * HIP kernels.  All other expressions are evaluated unfused, in the reference's
 * association order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_EXPORT __attribute__((visibility("default")))

/* ---- raymarching/src/raymarching.cu:19-81 : constants and small helpers ---- */
static const float ORC_SQRT3 = 1.7320508075688772f;

static inline float orc_signf(float x) { return copysignf(1.0f, x); }
static inline float orc_clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

/* raymarching.cu:42-47 */
static inline int orc_mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:49-54 : dt*H in float, *0.5 in double, narrowed to float */
static inline int orc_mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:56-63 */
static inline uint32_t orc_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* raymarching.cu:65-71 */
static inline uint32_t orc_morton3D_1(uint32_t x, uint32_t y, uint32_t z) {
    return orc_expand_bits(x) | (orc_expand_bits(y) << 1) | (orc_expand_bits(z) << 2);
}
/* raymarching.cu:73-81 */
static inline uint32_t orc_morton3D_invert_1(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* ---- raymarching.cu:92-145 kernel_near_far_from_aabb ---- */
ORC_EXPORT void orc_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                                       uint32_t N, float min_near, float* nears, float* fars) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float* o = rays_o + n * 3;
        const float* d = rays_d + n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];

        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx;
        if (near > far) { float c = near; near = far; far = c; }
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }
        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }
        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* ---- raymarching.cu:161-198 kernel_sph_from_ray: far intersection with the sphere |x| = radius -> (theta, phi) in [-1, 1] ---- */
ORC_EXPORT void orc_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
    const float RPI = 0.3183098861837907f;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float A = dx * dx + dy * dy + dz * dz;
        const float B = ox * dx + oy * dy + oz * dz;
        const float C = ox * ox + oy * oy + oz * oz - radius * radius;
        const float t = (-B + sqrtf(B * B - A * C)) / A;
        const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
        const float theta = atan2f(sqrtf(x * x + z * z), y);
        const float phi = atan2f(z, x);
        coords[n * 2] = 2 * theta * RPI - 1;
        coords[n * 2 + 1] = phi * RPI;
    }
}

/* ---- raymarching.cu:214-226 kernel_morton3D ---- */
ORC_EXPORT void orc_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)orc_morton3D_1((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1], (uint32_t)coords[n * 3 + 2]);
}
/* ---- raymarching.cu:237-254 kernel_morton3D_invert ---- */
ORC_EXPORT void orc_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ind = indices[n];
        coords[n * 3 + 0] = (int32_t)orc_morton3D_invert_1((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)orc_morton3D_invert_1((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)orc_morton3D_invert_1((uint32_t)(ind >> 2));
    }
}
/* ---- raymarching.cu:268-289 kernel_packbits : N = number of output bytes ---- */
ORC_EXPORT void orc_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > density_thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}
/* ---- raymarching.cu:304-335 kernel_morton3D_dilation ---- */
ORC_EXPORT void orc_morton3D_dilation(const float* grid, uint32_t C, uint32_t H, float* out) {
    const uint32_t H3 = H * H * H;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)C * H3; n++) {
        const uint32_t c = (uint32_t)(n / H3), ind = (uint32_t)(n - (int64_t)c * H3);
        const uint32_t x = orc_morton3D_invert_1(ind >> 0), y = orc_morton3D_invert_1(ind >> 1), z = orc_morton3D_invert_1(ind >> 2);
        const float* g = grid + (size_t)c * H3;
        float res = grid[n];
        if (x + 1 < H) res = fmaxf(res, g[orc_morton3D_1(x + 1, y, z)]);
        if (x > 0) res = fmaxf(res, g[orc_morton3D_1(x - 1, y, z)]);
        if (y + 1 < H) res = fmaxf(res, g[orc_morton3D_1(x, y + 1, z)]);
        if (y > 0) res = fmaxf(res, g[orc_morton3D_1(x, y - 1, z)]);
        if (z + 1 < H) res = fmaxf(res, g[orc_morton3D_1(x, y, z + 1)]);
        if (z > 0) res = fmaxf(res, g[orc_morton3D_1(x, y, z - 1)]);
        out[n] = res;
    }
}

/* ---- raymarching.cu:828-929 kernel_march_rays ----
 * xyzs/dirs [n_alive*n_step(+pad),3], deltas [...,2] must be zero-filled by the caller
 * (raymarching/raymarching.py:379-386). */
ORC_EXPORT void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                               const float* rays_o, const float* rays_d, float bound, float dt_gamma,
                               uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid,
                               const float* nears, const float* fars, float* xyzs_, float* dirs_,
                               float* deltas_, const float* noises) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int index = rays_alive[n];
        const float noise = noises[n];
        const float* o = rays_o + (size_t)index * 3;
        const float* d = rays_d + (size_t)index * 3;
        float* xyzs = xyzs_ + (size_t)n * n_step * 3;
        float* dirs = dirs_ + (size_t)n * n_step * 3;
        float* deltas = deltas_ + (size_t)n * n_step * 2;

        const float ox = o[0], oy = o[1], oz = o[2];
        const float dx = d[0], dy = d[1], dz = d[2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        const float rH = 1 / (float)H;
        const float H3 = (float)(H * H * H);

        float t = rays_t[index];
        const float far = fars[index];
        (void)nears;

        const float dt_max = 2 * ORC_SQRT3 * (float)(1 << (C - 1)) / (float)H;
        const float dt_min = fminf(dt_max, 2 * ORC_SQRT3 / (float)max_steps);

        uint32_t step = 0;
        t += orc_clampf(t * dt_gamma, dt_min, dt_max) * noise;

        while (t < far && step < n_step) {
            /* nvcc --fmad=true fuses o + t*d : see file header */
            const float x = orc_clampf(fmaf(t, dx, ox), -bound, bound);
            const float y = orc_clampf(fmaf(t, dy, oy), -bound, bound);
            const float z = orc_clampf(fmaf(t, dz, oz), -bound, bound);

            const float dt = orc_clampf(t * dt_gamma, dt_min, dt_max);

            const int lp = orc_mip_from_pos(x, y, z, (float)C), ld = orc_mip_from_dt(dt, (float)H, (float)C);
            const int level = lp > ld ? lp : ld; /* in [0, C-1] */

            const float mip_bound = fminf(scalbnf(1, level), bound);
            const float mip_rbound = 1 / mip_bound;

            /* literal 0.5 is a double: (x*rb+1) float -> double product -> narrowed to float by clamp() */
            const int nx = (int)orc_clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
            const int ny = (int)orc_clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
            const int nz = (int)orc_clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));

            /* level * H3 is float arithmetic; the sum converts back to uint32 */
            const uint32_t idx = (uint32_t)((float)level * H3 + (float)orc_morton3D_1((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
            const int occ = grid[idx / 8] & (1 << (idx % 8));

            if (occ) {
                xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
                t += dt;
                deltas[0] = dt;
                deltas[1] = t;
                xyzs += 3; dirs += 3; deltas += 2;
                step++;
            } else {
                const float tx = ((((float)nx + 0.5f + 0.5f * orc_signf(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
                const float ty = ((((float)ny + 0.5f + 0.5f * orc_signf(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
                const float tz = ((((float)nz + 0.5f + 0.5f * orc_signf(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
                const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
                do {
                    t += orc_clampf(t * dt_gamma, dt_min, dt_max);
                } while (t < tt);
            }
        }
    }
}

/* ---- raymarching.cu:943-1029 kernel_composite_rays ----
 * __expf there is the fast-math exponential; expf here (tolerance-level difference). */
ORC_EXPORT void orc_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive,
                                   float* rays_t, const float* sigmas_, const float* rgbs_, const float* deltas_,
                                   float* weights_sum, float* depth, float* image) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < (int64_t)n_alive; n++) {
        const int index = rays_alive[n];
        const float* sigmas = sigmas_ + (size_t)n * n_step;
        const float* rgbs = rgbs_ + (size_t)n * n_step * 3;
        const float* deltas = deltas_ + (size_t)n * n_step * 2;

        float t = rays_t[index];
        float weight_sum = weights_sum[index];
        float d = depth[index];
        float r = image[(size_t)index * 3], g = image[(size_t)index * 3 + 1], b = image[(size_t)index * 3 + 2];

        uint32_t step = 0;
        while (step < n_step) {
            if (deltas[0] == 0) break;
            const float alpha = 1.0f - expf(-sigmas[0] * deltas[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t = deltas[1];
            d += weight * t;
            r += weight * rgbs[0];
            g += weight * rgbs[1];
            b += weight * rgbs[2];
            if (T < T_thresh) break;
            sigmas++; rgbs += 3; deltas += 2; step++;
        }
        if (step < n_step) rays_alive[n] = -1;
        else rays_t[index] = t;
        weights_sum[index] = weight_sum;
        depth[index] = d;
        image[(size_t)index * 3] = r; image[(size_t)index * 3 + 1] = g; image[(size_t)index * 3 + 2] = b;
    }
}

/* ---- encoders/gridencoder/src/gridencoder.cu ---- */
#define ORC_MAX_D 5
#define ORC_MAX_C 8

/* gridencoder.cu:50-63 fast_hash */
static inline uint32_t orc_fast_hash(const uint32_t* pos_grid, uint32_t D) {
    static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
    uint32_t result = 0;
    for (uint32_t i = 0; i < D; ++i) result ^= pos_grid[i] * primes[i];
    return result;
}

/* gridencoder.cu:66-84 get_grid_index : note the loop stops accumulating dimensions
 * once stride > hashmap_size ("tiled" then silently drops the higher dimensions). */
static inline uint32_t orc_grid_index(uint32_t gridtype, int align_corners, uint32_t D, uint32_t C, uint32_t ch,
                                      uint32_t hashmap_size, uint32_t resolution, const uint32_t* pos_grid) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) index = orc_fast_hash(pos_grid, D);
    return (index % hashmap_size) * C + ch;
}

/* gridencoder.cu:505-596 kernel_grad_tv: normalised total-variation gradient of the table around the node each input falls on,
 * accumulated into grad (the caller's gradient buffer).  One (point, level) pair per iteration, sequential: the reference's
 * atomics make the order arbitrary. */
ORC_EXPORT int orc_grad_total_variation(const float* inputs_, const float* embeddings, float* grad_, const int32_t* offsets, float weight,
                                        uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                        int align_corners) {
    if (D < 2 || D > ORC_MAX_D) return -1;
    if (!(C == 1 || C == 2 || C == 4 || C == 8)) return -2;
    for (uint32_t level = 0; level < L; level++) {
        const float* grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
        float* grad = grad_ + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = exp2f((float)level * S) * (float)H - 1.0f;
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        for (uint32_t b = 0; b < B; b++) {
            const float* inputs = inputs_ + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) oob |= (inputs[d] < 0 || inputs[d] > 1);
            if (oob) continue;
            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) pos_grid[d] = (uint32_t)floorf(fmaf(inputs[d], scale, align_corners ? 0.0f : 0.5f));   /* fused, like the forward */
            float results[ORC_MAX_C] = {0}, idelta[ORC_MAX_C] = {0};
            const uint32_t index = orc_grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pos_grid);
            const float w = weight / (float)(2 * D);
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t cur = pos_grid[d];
                if (cur < resolution) {
                    pos_grid[d] = cur + 1;
                    const uint32_t ir = orc_grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) { const float g = grid[index + ch] - grid[ir + ch]; results[ch] += g; idelta[ch] += g * g; }
                }
                if (cur > 0) {
                    pos_grid[d] = cur - 1;
                    const uint32_t il = orc_grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pos_grid);
                    for (uint32_t ch = 0; ch < C; ch++) { const float g = grid[index + ch] - grid[il + ch]; results[ch] += g; idelta[ch] += g * g; }
                }
                pos_grid[d] = cur;
            }
            for (uint32_t ch = 0; ch < C; ch++) grad[index + ch] += w * results[ch] * (1.0f / sqrtf(idelta[ch] + 1e-9f));
        }
    }
    return 0;
}

/* gridencoder.cu:88-244 kernel_grid (forward, optional dy_dx).
 * outputs is [L,B,C]; dy_dx (may be NULL) is [B, L*D*C]. */
ORC_EXPORT int orc_grid_encode_forward(const float* inputs_, const float* embeddings, const int32_t* offsets,
                                       float* outputs_, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                       uint32_t H, float* dy_dx_, uint32_t gridtype, int align_corners, uint32_t interp) {
    if (D < 2 || D > ORC_MAX_D) return -1;
    if (!(C == 1 || C == 2 || C == 4 || C == 8)) return -2;
    for (uint32_t level = 0; level < L; level++) {
        const float* grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = exp2f((float)level * S) * (float)H - 1.0f;
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float* inputs = inputs_ + (size_t)b * D;
            float* outputs = outputs_ + (size_t)level * B * C + (size_t)b * C;
            float* dy_dx = dy_dx_ ? dy_dx_ + (size_t)b * D * L * C + (size_t)level * D * C : NULL;

            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (inputs[d] < 0 || inputs[d] > 1) oob = 1;
            if (oob) {
                for (uint32_t ch = 0; ch < C; ch++) outputs[ch] = 0;
                if (dy_dx) for (uint32_t i = 0; i < D * C; i++) dy_dx[i] = 0;
                continue;
            }

            float pos[ORC_MAX_D], pos_deriv[ORC_MAX_D];
            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                /* nvcc --fmad=true fuses x*scale+0.5; pos reaches ~2^11 on the finest level, so fused and unfused
                 * differ by one ulp(pos) ~ 1e-4 of a cell -- visible in the output.  Restated as the fused form. */
                pos[d] = fmaf(inputs[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) {
                    pos_deriv[d] = 6 * pos[d] * (1.0f - pos[d]);
                    pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
                } else {
                    pos_deriv[d] = 1.0f;
                }
            }

            float results[ORC_MAX_C] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = orc_grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) results[ch] += w * grid[index + ch];
            }
            for (uint32_t ch = 0; ch < C; ch++) outputs[ch] = results[ch];

            if (dy_dx) {
                for (uint32_t gd = 0; gd < D; gd++) {
                    float rg[ORC_MAX_C] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale;
                        uint32_t pgl[ORC_MAX_D];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                        }
                        pgl[gd] = pos_grid[gd];
                        const uint32_t il = orc_grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pgl);
                        pgl[gd] = pos_grid[gd] + 1;
                        const uint32_t ir = orc_grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pgl);
                        for (uint32_t ch = 0; ch < C; ch++) rg[ch] += w * (grid[ir + ch] - grid[il + ch]) * pos_deriv[gd];
                    }
                    for (uint32_t ch = 0; ch < C; ch++) dy_dx[gd * C + ch] = rg[ch];
                }
            }
        }
    }
    return 0;
}

/* per-level resolution as the kernel derives it (gridencoder.cu:138-139); exposed so the
 * tests can assert it equals grid.py:122's  ceil(base * per_level_scale**l). */
ORC_EXPORT void orc_grid_level_meta(uint32_t L, float S, uint32_t H, float* scale_out, uint32_t* resolution_out) {
    for (uint32_t level = 0; level < L; level++) {
        const float scale = exp2f((float)level * S) * (float)H - 1.0f;
        scale_out[level] = scale;
        resolution_out[level] = (uint32_t)ceilf(scale) + 1;
    }
}

/* ---- encoders/shencoder/src/shencoder.cu:28-121 kernel_sh (forward), degree C in [1,8] ----
 * Bands 0..3 (C <= 4, the only degree GeneFace instantiates: radnerf.py:58 via encoding.py:8,22) restate shencoder.cu:50-68 line by line.
 * Bands 4..7 (shencoder.cu:69-121, derivatives :150-356) are NOT transcribed: the 48 polynomials are derived from the definition of the
 * basis in exact rational arithmetic (tools/gen_sh_tables.py, which also checks orthonormality and the addition theorem) and evaluated here
 * monomial by monomial in DOUBLE -- the value of the polynomial the reference's fp32 expression approximates, and a different algorithm
 * from the product's factorised Horner form.  Pinned by tests/golden/sh_deg8.npz (the reference's own source expressions evaluated by
 * tests/golden/make_golden.py in this container) and by the reference's running kernel on the MI355X (tests/test_gpu_vs_ref_kernels.py). */
#include "sh_high_monomials.inc"

static double orc_ipow(double v, unsigned e) { double r = 1.0; while (e--) r *= v; return r; }

/* basis function k (16 <= k < 64) at (x, y, z): value and, when g != NULL, its three partial derivatives */
static double orc_sh_high_eval(uint32_t k, double x, double y, double z, double* g) {
    double v = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
    for (int t = orc_sh_high_start[k - 16]; t < orc_sh_high_start[k - 15]; t++) {
        const orc_sh_mono_t m = orc_sh_high_mono[t];
        v += m.c * orc_ipow(x, m.ex) * orc_ipow(y, m.ey) * orc_ipow(z, m.ez);
        if (g) {
            if (m.ex) gx += m.c * m.ex * orc_ipow(x, m.ex - 1) * orc_ipow(y, m.ey) * orc_ipow(z, m.ez);
            if (m.ey) gy += m.c * m.ey * orc_ipow(x, m.ex) * orc_ipow(y, m.ey - 1) * orc_ipow(z, m.ez);
            if (m.ez) gz += m.c * m.ez * orc_ipow(x, m.ex) * orc_ipow(y, m.ey) * orc_ipow(z, m.ez - 1);
        }
    }
    if (g) { g[0] = gx; g[1] = gy; g[2] = gz; }
    return v;
}

ORC_EXPORT int orc_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C) {
    if (D != 3) return -1;
    if (C < 1 || C > 8) return -2;
    const uint32_t C2 = C * C;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        const float x = inputs[b * 3], y = inputs[b * 3 + 1], z = inputs[b * 3 + 2];
        float* o = outputs + (size_t)b * C2;
        for (uint32_t k = 16; k < C2; k++) o[k] = (float)orc_sh_high_eval(k, x, y, z, NULL);
        const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        o[0] = 0.28209479177387814f;
        if (C <= 1) continue;
        o[1] = -0.48860251190291987f * y;
        o[2] = 0.48860251190291987f * z;
        o[3] = -0.48860251190291987f * x;
        if (C <= 2) continue;
        o[4] = 1.0925484305920792f * xy;
        o[5] = -1.0925484305920792f * yz;
        o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        o[7] = -1.0925484305920792f * xz;
        o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        if (C <= 3) continue;
        o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        o[10] = 2.8906114426405538f * xy * z;
        o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        o[14] = 1.4453057213202769f * z * (x2 - y2);
        o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);   /* C >= 5: bands 4..7 were written above */
    }
    return 0;
}

/* ---- encoders/freqencoder/src/freqencoder.cu:30-58 kernel_freq ----
 * outputs [B, C], C = D + 2*D*deg; cos is sin(x + pi/2) with pi/2 rounded to float;
 * __sinf there is the fast-math sine; sinf here (tolerance-level difference). */
ORC_EXPORT void orc_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs) {
    const float PI = 3.141592653589793f;
    (void)deg;
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * C; t++) {
        const uint32_t b = (uint32_t)(t / C), c = (uint32_t)(t - (int64_t)b * C);
        const float* in = inputs + (size_t)b * D;
        if (c < D) {
            outputs[t] = in[c];
        } else {
            const uint32_t col = c / D - 1, d = c % D, freq = col / 2;
            const float phase_shift = (float)(col % 2) * (PI / 2);
            outputs[t] = sinf(scalbnf(in[d], (int)freq) + phase_shift);
        }
    }
}


/* =====================================================================================================
 * Training tier: raymarching/src/raymarching.cu:353-518 (march_rays_train),
 * :536-583 (march_rays_train_backward), :604-687 (composite_rays_train_forward), :712-809 (backward).
 * The CUDA kernel assigns point offsets / ray slots with atomicAdd (nondeterministic order);
 * this restatement -- like the HIP kernels -- assigns them in ray order, which is one of the orders the
 * reference can produce: rays[n] = (n, offset_n, count_n), offset_n = sum of the counts before n.
 * ===================================================================================================== */

/* one marcher trip: returns 1 and fills the sample when the cell at t is occupied, else skips to the next cell */
static inline int orc_march_trip(float ox, float oy, float oz, float dx, float dy, float dz, float rdx, float rdy, float rdz,
                                 float bound, float dt_gamma, float dt_min, float dt_max, uint32_t C, uint32_t H,
                                 const uint8_t* grid, float* t_io, float* x_, float* y_, float* z_, float* dt_) {
    const float rH = 1 / (float)H;
    const float H3 = (float)(H * H * H);
    float t = *t_io;
    const float x = orc_clampf(fmaf(t, dx, ox), -bound, bound);
    const float y = orc_clampf(fmaf(t, dy, oy), -bound, bound);
    const float z = orc_clampf(fmaf(t, dz, oz), -bound, bound);
    const float dt = orc_clampf(t * dt_gamma, dt_min, dt_max);
    const int lp = orc_mip_from_pos(x, y, z, (float)C), ld = orc_mip_from_dt(dt, (float)H, (float)C);
    const int level = lp > ld ? lp : ld;
    const float mip_bound = fminf(scalbnf(1, level), bound);
    const float mip_rbound = 1 / mip_bound;
    const int nx = (int)orc_clampf((float)(0.5 * (double)(x * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
    const int ny = (int)orc_clampf((float)(0.5 * (double)(y * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
    const int nz = (int)orc_clampf((float)(0.5 * (double)(z * mip_rbound + 1) * (double)H), 0.0f, (float)(H - 1));
    const uint32_t idx = (uint32_t)((float)level * H3 + (float)orc_morton3D_1((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    if (grid[idx / 8] & (1 << (idx % 8))) {
        *x_ = x; *y_ = y; *z_ = z; *dt_ = dt;
        *t_io = t + dt;
        return 1;
    }
    const float tx = ((((float)nx + 0.5f + 0.5f * orc_signf(dx)) * rH * 2 - 1) * mip_bound - x) * rdx;
    const float ty = ((((float)ny + 0.5f + 0.5f * orc_signf(dy)) * rH * 2 - 1) * mip_bound - y) * rdy;
    const float tz = ((((float)nz + 0.5f + 0.5f * orc_signf(dz)) * rH * 2 - 1) * mip_bound - z) * rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        t += orc_clampf(t * dt_gamma, dt_min, dt_max);
    } while (t < tt);
    *t_io = t;
    return 0;
}

/* raymarching.cu:353-518.  xyzs/dirs [M,3], deltas [M,2] pre-zeroed by the caller; rays int32 [N,3]; counter int32 [2]
 * (accumulated, like the atomicAdd: counter[0] += points, counter[1] += N). */
ORC_EXPORT void orc_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                     uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                     const float* fars, float* xyzs_, float* dirs_, float* deltas_, int32_t* rays, int32_t* counter,
                                     const float* noises) {
    const float dt_max = 2 * ORC_SQRT3 * (float)(1 << (C - 1)) / (float)H;
    const float dt_min = fminf(dt_max, 2 * ORC_SQRT3 / (float)max_steps);
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        const float far = fars[n];
        float t0 = nears[n];
        t0 += orc_clampf(t0 * dt_gamma, dt_min, dt_max) * noises[n];
        /* first pass: count */
        float t = t0, x, y, z, dt;
        uint32_t num_steps = 0;
        while (t < far && num_steps < max_steps)
            num_steps += (uint32_t)orc_march_trip(ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, bound, dt_gamma, dt_min, dt_max, C, H, grid, &t, &x, &y, &z, &dt);
        const uint32_t point_index = (uint32_t)counter[0];
        const uint32_t ray_index = (uint32_t)counter[1];
        counter[0] += (int32_t)num_steps;
        counter[1] += 1;
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps > M) continue;
        /* second pass: write */
        float* xyzs = xyzs_ + (size_t)point_index * 3;
        float* dirs = dirs_ + (size_t)point_index * 3;
        float* deltas = deltas_ + (size_t)point_index * 2;
        t = t0;
        uint32_t step = 0;
        while (t < far && step < num_steps) {
            if (orc_march_trip(ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, bound, dt_gamma, dt_min, dt_max, C, H, grid, &t, &x, &y, &z, &dt)) {
                xyzs[0] = x; xyzs[1] = y; xyzs[2] = z;
                dirs[0] = dx; dirs[1] = dy; dirs[2] = dz;
                deltas[0] = dt; deltas[1] = t;
                xyzs += 3; dirs += 3; deltas += 2;
                step++;
            }
        }
    }
}

/* raymarching.cu:536-583.  grad_rays_o/d [N,3] accumulate (the caller zero-fills). */
ORC_EXPORT void orc_march_rays_train_backward(const float* grad_xyzs_, const float* grad_dirs_, const int32_t* rays, const float* deltas_,
                                              uint32_t N, uint32_t M, float* grad_rays_o_, float* grad_rays_d_) {
    for (uint32_t n = 0; n < N; n++) {
        float* go = grad_rays_o_ + (size_t)n * 3;   /* indexed by the thread id n, as the reference does */
        float* gd = grad_rays_d_ + (size_t)n * 3;
        const uint32_t offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float* gx = grad_xyzs_ + (size_t)offset * 3;
        const float* gdd = grad_dirs_ + (size_t)offset * 3;
        const float* deltas = deltas_ + (size_t)offset * 2;
        for (uint32_t step = 0; step < num_steps; step++) {
            go[0] += gx[0]; go[1] += gx[1]; go[2] += gx[2];
            gd[0] += gx[0] * deltas[1] + gdd[0];
            gd[1] += gx[1] * deltas[1] + gdd[1];
            gd[2] += gx[2] * deltas[1] + gdd[2];
            gx += 3; gdd += 3; deltas += 2;
        }
    }
}

/* raymarching.cu:604-687 */
ORC_EXPORT void orc_composite_rays_train_forward(const float* sigmas_, const float* rgbs_, const float* ambient_, const float* deltas_,
                                                 const int32_t* rays, uint32_t M, uint32_t N, float T_thresh, float* weights_sum,
                                                 float* ambient_sum, float* depth, float* image) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) {
            weights_sum[index] = 0; ambient_sum[index] = 0; depth[index] = 0;
            image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0;
            continue;
        }
        const float* sigmas = sigmas_ + offset;
        const float* rgbs = rgbs_ + (size_t)offset * 3;
        const float* ambient = ambient_ + offset;
        const float* deltas = deltas_ + (size_t)offset * 2;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, d = 0, amb = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-sigmas[0] * deltas[0]);
            const float weight = alpha * T;
            r += weight * rgbs[0]; g += weight * rgbs[1]; b += weight * rgbs[2];
            d += weight * deltas[1];
            ws += weight;
            amb += ambient[0];
            T *= 1.0f - alpha;
            if (T < T_thresh) break;
            sigmas++; rgbs += 3; ambient++; deltas += 2;
        }
        weights_sum[index] = ws; ambient_sum[index] = amb; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* raymarching.cu:712-809 (grad_sigmas / grad_rgbs / grad_ambient pre-zeroed by the caller) */
ORC_EXPORT void orc_composite_rays_train_backward(const float* grad_weights_sum_, const float* grad_ambient_sum_, const float* grad_image_,
                                                  const float* sigmas_, const float* rgbs_, const float* ambient_, const float* deltas_,
                                                  const int32_t* rays, const float* weights_sum_, const float* ambient_sum_,
                                                  const float* image_, uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas_,
                                                  float* grad_rgbs_, float* grad_ambient_) {
    (void)ambient_; (void)ambient_sum_;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < (int64_t)N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps > M) continue;
        const float gws = grad_weights_sum_[index], gas = grad_ambient_sum_[index];
        const float* gi = grad_image_ + (size_t)index * 3;
        const float r_final = image_[index * 3], g_final = image_[index * 3 + 1], b_final = image_[index * 3 + 2], ws_final = weights_sum_[index];
        const float* sigmas = sigmas_ + offset;
        const float* rgbs = rgbs_ + (size_t)offset * 3;
        const float* deltas = deltas_ + (size_t)offset * 2;
        float* grad_sigmas = grad_sigmas_ + offset;
        float* grad_rgbs = grad_rgbs_ + (size_t)offset * 3;
        float* grad_ambient = grad_ambient_ + offset;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-sigmas[0] * deltas[0]);
            const float weight = alpha * T;
            r += weight * rgbs[0]; g += weight * rgbs[1]; b += weight * rgbs[2];
            ws += weight;
            T *= 1.0f - alpha;
            grad_rgbs[0] = gi[0] * weight; grad_rgbs[1] = gi[1] * weight; grad_rgbs[2] = gi[2] * weight;
            grad_ambient[0] = gas;
            grad_sigmas[0] = deltas[0] * (gi[0] * (T * rgbs[0] - (r_final - r)) + gi[1] * (T * rgbs[1] - (g_final - g)) +
                                          gi[2] * (T * rgbs[2] - (b_final - b)) + gws * (1 - ws_final));
            if (T < T_thresh) break;
            sigmas++; rgbs += 3; deltas += 2; grad_sigmas++; grad_rgbs += 3; grad_ambient++;
        }
    }
}


/* =====================================================================================================
 * Training tier, encoders: gridencoder.cu:248-368 (grid backward: scatter-add into the table + input gradient
 * through dy_dx), shencoder.cu:122-356 (dy_dx of the degree <= 4 basis, derived from the polynomials of
 * orc_sh_encode_forward) and :359-383 (backward), freqencoder.cu:63-94 (backward).
 * ===================================================================================================== */

/* grad [L,B,C]; grad_embeddings [sO,C] accumulated (caller zero-fills); dy_dx [B, L*D*C] and grad_inputs [B,D] may both be NULL.
 * The CUDA kernel scatters with atomicAdd (order nondeterministic); here contributions are added in (level, b, corner) order. */
ORC_EXPORT int orc_grid_encode_backward(const float* grad_, const float* inputs_, const float* embeddings, const int32_t* offsets,
                                        float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                        const float* dy_dx_, float* grad_inputs, uint32_t gridtype, int align_corners, uint32_t interp) {
    (void)embeddings;
    if (D < 2 || D > ORC_MAX_D) return -1;
    if (!(C == 1 || C == 2 || C == 4 || C == 8)) return -2;
    for (uint32_t level = 0; level < L; level++) {
        float* grad_grid = grad_embeddings + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const float scale = exp2f((float)level * S) * (float)H - 1.0f;
        const uint32_t resolution = (uint32_t)ceilf(scale) + 1;
        for (uint32_t b = 0; b < B; b++) {
            const float* inputs = inputs_ + (size_t)b * D;
            const float* grad = grad_ + (size_t)level * B * C + (size_t)b * C;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++) if (inputs[d] < 0 || inputs[d] > 1) oob = 1;
            if (oob) continue;
            float pos[ORC_MAX_D];
            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(inputs[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
                if (interp == 1) pos[d] = pos[d] * pos[d] * (3.0f - 2.0f * pos[d]);
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = orc_grid_index(gridtype, align_corners, D, C, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) grad_grid[index + ch] += w * grad[ch];
            }
        }
    }
    if (dy_dx_ && grad_inputs) {   /* kernel_input_backward :343-368 */
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < (int64_t)B * D; t++) {
            const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t - (int64_t)b * D);
            const float* dy_dx = dy_dx_ + (size_t)b * L * D * C;
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++) result += grad_[(size_t)l * B * C + (size_t)b * C + ch] * dy_dx[l * D * C + d * C + ch];
            grad_inputs[t] = result;
        }
    }
    return 0;
}

/* d/dx, d/dy, d/dz of the real SH polynomials of orc_sh_encode_forward: dy_dx [B, 3, degree^2] (bands 4..7 from the monomial lists). */
ORC_EXPORT int orc_sh_encode_dy_dx(const float* inputs, float* dy_dx, uint32_t B, uint32_t degree) {
    if (degree < 1 || degree > 8) return -1;
    const uint32_t C2 = degree * degree, C4 = C2 < 16u ? C2 : 16u;
    const float k1 = 0.48860251190291987f, k2 = 1.0925484305920792f, k3a = 0.59004358992664352f, k3b = 0.45704579946446572f;
    const float a6 = 0.94617469575755997f, c8 = 0.54627421529603959f, c10 = 2.8906114426405538f, c12 = 0.3731763325901154f,
                c14 = 1.4453057213202769f;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++) {
        const float x = inputs[b * 3], y = inputs[b * 3 + 1], z = inputs[b * 3 + 2];
        const float x2 = x * x, y2 = y * y, z2 = z * z;
        float g[3][16];
        memset(g, 0, sizeof(g));
        g[1][1] = -k1; g[2][2] = k1; g[0][3] = -k1;
        g[0][4] = k2 * y; g[1][4] = k2 * x;
        g[1][5] = -k2 * z; g[2][5] = -k2 * y;
        g[2][6] = 2 * a6 * z;
        g[0][7] = -k2 * z; g[2][7] = -k2 * x;
        g[0][8] = 2 * c8 * x; g[1][8] = -2 * c8 * y;
        g[0][9] = -6 * k3a * x * y; g[1][9] = 3 * k3a * (y2 - x2);
        g[0][10] = c10 * y * z; g[1][10] = c10 * x * z; g[2][10] = c10 * x * y;
        g[1][11] = k3b * (1 - 5 * z2); g[2][11] = -10 * k3b * y * z;
        g[2][12] = c12 * (15 * z2 - 3);
        g[0][13] = k3b * (1 - 5 * z2); g[2][13] = -10 * k3b * x * z;
        g[0][14] = 2 * c14 * x * z; g[1][14] = -2 * c14 * y * z; g[2][14] = c14 * (x2 - y2);
        g[0][15] = 3 * k3a * (y2 - x2); g[1][15] = 6 * k3a * x * y;
        float* o = dy_dx + (size_t)b * 3 * C2;
        for (uint32_t d = 0; d < 3; d++)
            for (uint32_t k = 0; k < C4; k++) o[d * C2 + k] = g[d][k];
        for (uint32_t k = 16; k < C2; k++) {
            double gh[3];
            orc_sh_high_eval(k, x, y, z, gh);
            for (uint32_t d = 0; d < 3; d++) o[d * C2 + k] = (float)gh[d];
        }
    }
    return 0;
}

/* shencoder.cu:359-383: grad_inputs[b][d] += sum_k grad[b][k] * dy_dx[b][d][k]   (the caller zero-fills grad_inputs) */
ORC_EXPORT void orc_sh_encode_backward(const float* grad, const float* dy_dx, uint32_t B, uint32_t degree, float* grad_inputs) {
    const uint32_t C2 = degree * degree;
    for (uint32_t b = 0; b < B; b++)
        for (uint32_t d = 0; d < 3; d++) {
            float acc = grad_inputs[b * 3 + d];
            for (uint32_t k = 0; k < C2; k++) acc += grad[(size_t)b * C2 + k] * dy_dx[(size_t)b * 3 * C2 + d * C2 + k];
            grad_inputs[b * 3 + d] = acc;
        }
}

/* freqencoder.cu:63-94: d/dx of [x, sin(2^f x), cos(2^f x), ...] read off the forward OUTPUTS (cos = outputs[D + d] of the pair) */
ORC_EXPORT void orc_freq_encode_backward(const float* grad_, const float* outputs_, uint32_t B, uint32_t D, uint32_t deg, uint32_t C,
                                         float* grad_inputs) {
    for (uint32_t b = 0; b < B; b++)
        for (uint32_t d = 0; d < D; d++) {
            const float* grad = grad_ + (size_t)b * C;
            const float* outputs = outputs_ + (size_t)b * C;
            float result = grad[d];
            grad += D; outputs += D;
            for (uint32_t f = 0; f < deg; f++) {
                result += scalbnf(1.0f, (int)f) * (grad[d] * outputs[D + d] - grad[D + d] * outputs[d]);
                grad += 2 * D; outputs += 2 * D;
            }
            grad_inputs[(size_t)b * D + d] = result;
        }
}
