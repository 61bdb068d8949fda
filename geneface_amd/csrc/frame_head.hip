// Fused head pass of one RAD-NeRF frame for gfx950: ray setup, then one kernel per march iteration that
// marches, evaluates the whole field (3-D grid -> ambient MLP -> tanh -> 2-D grid -> density MLP -> exp; SH + geometry
// feature -> colour MLP -> sigmoid) on f32 MFMA and composites -- with the alive list, n_alive and n_step kept on the
// device, so a frame is enqueued without a single host synchronisation.
//
// What it replaces, per iteration, in the reference (paths relative to /root/reference/modules/radnerfs):
//   renderer.py:316-351 loop body = raymarching.cu:828-929 (march) + radnerf.py:73-105 (~40 launches: 2 grid encodes,
//   8 GEMMs, SH, cats, activations) + raymarching.cu:943-1029 (composite) + `rays_alive[rays_alive >= 0]` (host sync).
//
// Work decomposition: a 256-thread workgroup (4 waves) takes `kPass` = 128 sample slots per pass = 128/n_step rays.
//   A. march   : one lane per ray writes its samples to LDS; a wave-scan packs the valid ones densely
//   B. field   : each wave owns one 32-sample tile.  Activations live in registers in the MFMA accumulator layout,
//                which is exactly the B-operand layout of the next layer's v_mfma_f32_32x32x2_f32 (feature pairs
//                (row, row+4) split over the two 32-lane halves), so layers chain with no data movement; weights
//                stream L2 -> LDS one layer at a time (64 KB buffer, 2 workgroups per CU so one's loads overlap
//                the other's MFMAs); the three skinny output layers (->2, ->1, ->3) run on the VALU
//   C. composite: one lane per ray consumes its samples in order, updates the ray, survivors are appended to the
//                next alive list with one atomic per wave.
// Exactness: the n_step schedule is a function of global alive counts (renderer.py:338); it is reproduced exactly
// because every iteration is its own launch and reads the count its predecessor accumulated.
#include "common.hpp"
#include "frame.hpp"
#include "march_core.hpp"
#include "grid_core.hpp"
#include "sh_core.hpp"
#include "mfma_mlp.hpp"

namespace {

using gf::floatx16;

constexpr int kThreads = 256;
constexpr int kPass = 128;       // sample slots per workgroup pass (4 waves x 32-column MFMA tiles)
constexpr int kWFloats = 16384;  // 64 KB weight buffer
constexpr int kPFloats = gf::HS_TOTAL + 128 /*amb bias*/ + 128 /*level meta: 2 grids x 16 x {scale,res,off,size}*/;

struct HeadArgs {
    gf::MarchParams mp;
    gf::GridLevels lv3, lv2;
    const float* pos_table; const int* pos_offsets;
    const float* amb_table; const int* amb_offsets;
    const float* head_pack; const float* amb_bias;
    const float* rays_o; const float* rays_d; const float* fars;
    float* rays_t; float* weights_sum; float* depth; float* image;
    const int* alive_in; int* alive_out; uint32_t* ctrl;
    uint32_t N, iter, max_steps, gridtype, interp;
    float T_thresh, bound;
};

// ---------------------------------------------------------------------------------------------------- LDS carve
struct Smem {
    float* W;      // [kWFloats] current layer's A-operand stream
    float* P;      // [kPFloats] VALU-layer rows, colour bias, ambient bias, per-level grid meta
    float *sx, *sy, *sz, *sdt, *st;  // [kPass] raw sample slots (slot = ray_local * n_step + s)
    float *osig, *orr, *og, *ob;     // [kPass] field outputs per raw slot
    float *rdx, *rdy, *rdz;          // [kPass] per-ray direction
    uint32_t *d2r, *rcnt, *rbase;    // [kPass] dense->raw map, per-ray sample count / dense base
    uint32_t* misc;                  // [8]
};
constexpr int kSmemFloats = kWFloats + kPFloats + kPass * (5 + 4 + 3 + 3) + 8;
__device__ __forceinline__ Smem carve(char* base) {
    Smem s;
    float* f = reinterpret_cast<float*>(base);
    s.W = f; f += kWFloats;
    s.P = f; f += kPFloats;
    s.sx = f; f += kPass; s.sy = f; f += kPass; s.sz = f; f += kPass; s.sdt = f; f += kPass; s.st = f; f += kPass;
    s.osig = f; f += kPass; s.orr = f; f += kPass; s.og = f; f += kPass; s.ob = f; f += kPass;
    s.rdx = f; f += kPass; s.rdy = f; f += kPass; s.rdz = f; f += kPass;
    s.d2r = reinterpret_cast<uint32_t*>(f); f += kPass;
    s.rcnt = reinterpret_cast<uint32_t*>(f); f += kPass;
    s.rbase = reinterpret_cast<uint32_t*>(f); f += kPass;
    s.misc = reinterpret_cast<uint32_t*>(f);
    return s;
}
// P layout
constexpr int P_SMALL = 0, P_AMBBIAS = gf::HS_TOTAL, P_META = gf::HS_TOTAL + 128;
// meta: grid g (0 = 3-D, 1 = 2-D), level l: P[P_META + g*64 + l*4 + {0 scale, 1 resolution, 2 row offset, 3 rows}]

__device__ __forceinline__ void stage_weights(float* __restrict__ W, const float* __restrict__ src, int nfloats) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(W);
    for (int i = threadIdx.x; i < nfloats / 4; i += kThreads) d4[i] = s4[i];
}

__global__ void __launch_bounds__(kThreads, 2) k_head_iter(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;

    const uint32_t n_alive = a.ctrl[gf::kCtrlAlive + a.iter];
    const uint32_t step0 = a.ctrl[gf::kCtrlStep + a.iter];
    if (n_alive == 0 || step0 >= a.max_steps) return;
    uint32_t n_step = a.N / n_alive;
    n_step = n_step > 8u ? 8u : (n_step < 1u ? 1u : n_step);  // renderer.py:338
    if (blockIdx.x == 0 && tid == 0) a.ctrl[gf::kCtrlStep + a.iter + 1] = step0 + n_step;
    const uint32_t R = kPass / n_step;  // rays per pass
    const uint32_t n_groups = (n_alive + R - 1) / R;
    if (blockIdx.x >= n_groups) return;

    // persistent small data: VALU rows + colour bias, this frame's ambient bias, per-level grid meta
    for (int i = tid; i < (int)gf::HS_TOTAL; i += kThreads) s.P[P_SMALL + i] = a.head_pack[gf::HP_SMALL + i];
    if (tid < 128) s.P[P_AMBBIAS + tid] = a.amb_bias[tid];
    if (tid < 32) {
        const int g = tid >> 4, l = tid & 15;
        const int* off = g ? a.amb_offsets : a.pos_offsets;
        const gf::GridLevels& lv = g ? a.lv2 : a.lv3;
        float* m = s.P + P_META + g * 64 + l * 4;
        m[0] = lv.scale[l];
        m[1] = __uint_as_float(lv.resolution[l]);
        m[2] = __uint_as_float((uint32_t)off[l]);
        m[3] = __uint_as_float((uint32_t)(off[l + 1] - off[l]));
    }
    const float* pack = a.head_pack;

    for (uint32_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
        __syncthreads();  // previous pass fully consumed (staging + weight buffer)
        // ------------------------------------------------------------------ A. march
        const uint32_t slot = group * R + tid;
        const bool has_ray = (uint32_t)tid < R && slot < n_alive;
        int ray = -1;
        float t_ray = 0.0f;
        uint32_t cnt = 0;
        if (has_ray) {
            ray = a.alive_in[slot];
            const float* o = a.rays_o + (size_t)ray * 3;
            const float* d = a.rays_d + (size_t)ray * 3;
            const float dx = d[0], dy = d[1], dz = d[2];
            t_ray = a.rays_t[ray];
            const uint32_t base = tid * n_step;
            cnt = gf::march_ray(a.mp, o[0], o[1], o[2], dx, dy, dz, a.fars[ray], 0.0f, n_step, t_ray,
                                [&](uint32_t st, float x, float y, float z, float dt, float t_after) {
                                    s.sx[base + st] = x; s.sy[base + st] = y; s.sz[base + st] = z;
                                    s.sdt[base + st] = dt; s.st[base + st] = t_after;
                                });
            s.rdx[tid] = dx; s.rdy[tid] = dy; s.rdz[tid] = dz;
        }
        if (tid < kPass) s.rcnt[tid] = cnt;
        stage_weights(s.W, pack + gf::HP_AMB1, 4 * 16 * 64);  // first layer's weights ride the same barrier
        __syncthreads();
        if (wave == 0) {  // exclusive scan of 128 counts, two per lane
            const uint32_t c0 = s.rcnt[2 * lane], c1 = s.rcnt[2 * lane + 1];
            uint32_t incl = c0 + c1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            const uint32_t excl = incl - (c0 + c1);
            s.rbase[2 * lane] = excl;
            s.rbase[2 * lane + 1] = excl + c0;
            if (lane == 63) s.misc[0] = incl;
        }
        __syncthreads();
        const uint32_t Mv = s.misc[0];
        if (tid < kPass) {
            const uint32_t c = s.rcnt[tid], b = s.rbase[tid];
            for (uint32_t q = 0; q < c; q++) s.d2r[b + q] = tid * n_step + q;
        }
        if (tid == 0 && Mv) atomicAdd(&a.ctrl[gf::kCtrlValid + a.iter], Mv);
        __syncthreads();

        // ------------------------------------------------------------------ B. field on this wave's tile
        if (Mv > 0) {  // uniform over the workgroup
            const uint32_t j = wave * 32 + (lane & 31);  // dense sample index
            const bool active = (uint32_t)(wave * 32) < Mv;  // wave-uniform
            const bool valid = j < Mv;
            const uint32_t raw = valid ? s.d2r[j] : 0u;
            const uint32_t rloc = raw / n_step;

            float pf[16], af[16], act[64];
            floatx16 h[4];
            float ambient[2] = {0.0f, 0.0f};

            if (active) {
                const float b2 = 2 * a.bound;
                const float x3[3] = {(s.sx[raw] + a.bound) / b2, (s.sy[raw] + a.bound) / b2, (s.sz[raw] + a.bound) / b2};
                gf::encode_half<3>(a.pos_table, s.P + P_META, half, a.gridtype, a.interp, x3, pf);
                // ambient L1 (3-D grid features; the cond_feat columns are folded into the bias)
                gf::mfma_layer<4, 16, true, false>(s.W, lane, pf, s.P + P_AMBBIAS, h);
                gf::unpack<4>(h, act);
            }
            __syncthreads();
            stage_weights(s.W, pack + gf::HP_AMB2, 4 * 64 * 64);
            __syncthreads();
            if (active) {
                gf::mfma_layer<4, 64, true, false>(s.W, lane, act, nullptr, h);
                gf::unpack<4>(h, act);
                gf::valu_rows<2, 4>(s.P + P_SMALL + gf::HS_AMB3, half, act, ambient);
                ambient[0] = tanhf(ambient[0]);
                ambient[1] = tanhf(ambient[1]);
                const float x2[2] = {(ambient[0] + 1.0f) / 2.0f, (ambient[1] + 1.0f) / 2.0f};
                gf::encode_half<2>(a.amb_table, s.P + P_META + 64, half, a.gridtype, a.interp, x2, af);
            }
            __syncthreads();
            stage_weights(s.W, pack + gf::HP_SIG1, 4 * 32 * 64);
            __syncthreads();
            if (active) {
                float in[32];
#pragma unroll
                for (int t = 0; t < 16; t++) { in[t] = pf[t]; in[16 + t] = af[t]; }
                gf::mfma_layer<4, 32, true, false>(s.W, lane, in, nullptr, h);
                gf::unpack<4>(h, act);
            }
            __syncthreads();
            stage_weights(s.W, pack + gf::HP_SIG2, 4 * 64 * 64);
            __syncthreads();
            float sigma = 0.0f;
            if (active) {
                gf::mfma_layer<4, 64, true, false>(s.W, lane, act, nullptr, h);
                gf::unpack<4>(h, act);
                float h0[1];
                gf::valu_rows<1, 4>(s.P + P_SMALL + gf::HS_SIGROW, half, act, h0);
                sigma = expf(h0[0]);  // trunc_exp forward: plain exp, no clamp (utils.py:41)
            }
            __syncthreads();
            stage_weights(s.W, pack + gf::HP_SIG3, 4 * 64 * 64);
            __syncthreads();
            if (active) {
                gf::mfma_layer<4, 64, false, false>(s.W, lane, act, nullptr, h);  // geometry feature: no activation
                gf::unpack<4>(h, act);
            }
            __syncthreads();
            stage_weights(s.W, pack + gf::HP_COL1S, 4 * 8 * 64);
            __syncthreads();
            if (active) {
                float sh[16], shh[8];
                gf::sh4(s.rdx[rloc], s.rdy[rloc], s.rdz[rloc], sh);
#pragma unroll
                for (int t = 0; t < 8; t++) shh[t] = half ? sh[8 + t] : sh[t];
                gf::mfma_layer<4, 8, false, false>(s.W, lane, shh, s.P + P_SMALL + gf::HS_COLBIAS, h);  // bias = identity-code columns
            }
            __syncthreads();
            stage_weights(s.W, pack + gf::HP_COL1G, 4 * 64 * 64);
            __syncthreads();
            if (active) {
                gf::mfma_layer<4, 64, true, true>(s.W, lane, act, nullptr, h);
                gf::unpack<4>(h, act);
                float c[3];
                gf::valu_rows<3, 4>(s.P + P_SMALL + gf::HS_COL2, half, act, c);
                if (valid && half == 0) {
                    s.osig[raw] = sigma;
                    s.orr[raw] = 1.0f / (1.0f + __expf(-c[0]));
                    s.og[raw] = 1.0f / (1.0f + __expf(-c[1]));
                    s.ob[raw] = 1.0f / (1.0f + __expf(-c[2]));
                }
            }
            __syncthreads();
        }

        // ------------------------------------------------------------------ C. composite + survivor compaction
        bool survive = false;
        if (has_ray) {
            gf::RayAcc acc;
            acc.t = t_ray;
            acc.weight_sum = a.weights_sum[ray];
            acc.depth = a.depth[ray];
            acc.r = a.image[(size_t)ray * 3 + 0];
            acc.g = a.image[(size_t)ray * 3 + 1];
            acc.b = a.image[(size_t)ray * 3 + 2];
            const uint32_t base = tid * n_step;
            uint32_t st = 0;
            while (st < n_step) {
                if (st >= cnt) break;  // the marcher produced no further sample (delta == 0 in the reference)
                if (!gf::composite_sample(acc, s.osig[base + st], s.orr[base + st], s.og[base + st], s.ob[base + st], s.sdt[base + st],
                                          s.st[base + st], a.T_thresh))
                    break;
                st++;
            }
            survive = (st == n_step);
            if (survive) a.rays_t[ray] = acc.t;
            a.weights_sum[ray] = acc.weight_sum;
            a.depth[ray] = acc.depth;
            a.image[(size_t)ray * 3 + 0] = acc.r;
            a.image[(size_t)ray * 3 + 1] = acc.g;
            a.image[(size_t)ray * 3 + 2] = acc.b;
        }
        if (wave < 2) {  // rays only live in the first 128 threads
            const unsigned long long mask = __ballot(survive);
            const uint32_t n = (uint32_t)__popcll(mask);
            uint32_t base = 0;
            if (lane == 0 && n) base = atomicAdd(&a.ctrl[gf::kCtrlAlive + a.iter + 1], n);
            base = __shfl(base, 0);
            if (survive) a.alive_out[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = ray;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- frame setup
struct InitArgs {
    const float* rays_o_in; const float* rays_d_in;  // explicit rays, or NULL
    float pose[12]; float fx, fy, cx, cy; uint32_t img_w;
    const float* aabb; float min_near;
    float *rays_o, *rays_d, *nears, *fars, *rays_t, *weights_sum, *depth, *image;
    int* alive; uint32_t* ctrl; uint32_t N;
};

__global__ void __launch_bounds__(256) k_frame_init(const InitArgs a) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < gf::kCtrlWords; i += 256) a.ctrl[i] = (i == gf::kCtrlAlive) ? a.N : 0u;
    }
    if (n >= a.N) return;
    float ox, oy, oz, dx, dy, dz;
    if (a.rays_o_in) {
        ox = a.rays_o_in[(size_t)n * 3]; oy = a.rays_o_in[(size_t)n * 3 + 1]; oz = a.rays_o_in[(size_t)n * 3 + 2];
        dx = a.rays_d_in[(size_t)n * 3]; dy = a.rays_d_in[(size_t)n * 3 + 1]; dz = a.rays_d_in[(size_t)n * 3 + 2];
    } else {
        // pinhole rays, pixel centres at +0.5, row-major pixels (utils.py:296-363): same operation order as the torch code
#pragma clang fp contract(off)
        const uint32_t row = n / a.img_w, col = n - row * a.img_w;
        const float xs = ((float)col + 0.5f - a.cx) / a.fx, ys = ((float)row + 0.5f - a.cy) / a.fy, zs = 1.0f;
        const float nrm = sqrtf(xs * xs + ys * ys + zs * zs);
        const float ux = xs / nrm, uy = ys / nrm, uz = zs / nrm;
        dx = ux * a.pose[0] + uy * a.pose[1] + uz * a.pose[2];
        dy = ux * a.pose[4] + uy * a.pose[5] + uz * a.pose[6];
        dz = ux * a.pose[8] + uy * a.pose[9] + uz * a.pose[10];
        ox = a.pose[3]; oy = a.pose[7]; oz = a.pose[11];
    }
    a.rays_o[(size_t)n * 3] = ox; a.rays_o[(size_t)n * 3 + 1] = oy; a.rays_o[(size_t)n * 3 + 2] = oz;
    a.rays_d[(size_t)n * 3] = dx; a.rays_d[(size_t)n * 3 + 1] = dy; a.rays_d[(size_t)n * 3 + 2] = dz;
    float near, far;
    gf::near_far_from_aabb_1(ox, oy, oz, dx, dy, dz, a.aabb, a.min_near, near, far);
    a.nears[n] = near;
    a.fars[n] = far;
    a.rays_t[n] = near;
    a.weights_sum[n] = 0.0f;
    a.depth[n] = 0.0f;
    a.image[(size_t)n * 3] = 0.0f; a.image[(size_t)n * 3 + 1] = 0.0f; a.image[(size_t)n * 3 + 2] = 0.0f;
    a.alive[n] = (int)n;
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
    if (f->gridtype > 1 || f->interp > 1) return gf_set_error(GF_ERR_INVALID, "frame: gridtype/interp must be 0 or 1");
    return GF_OK;
}

}  // namespace

GF_EXPORT uint64_t gf_frame_workspace_bytes(uint32_t n_rays) { return gf::carve_workspace(reinterpret_cast<void*>(uintptr_t(1) << 20), n_rays).bytes; }

GF_EXPORT uint32_t gf_frame_ctrl_words(void) { return gf::kCtrlWords; }

// Byte offset of the control block inside the workspace (n_alive / cumulative step / valid samples per iteration).
GF_EXPORT uint64_t gf_frame_ctrl_offset(uint32_t n_rays) {
    char* const base = reinterpret_cast<char*>(uintptr_t(1) << 20);
    const gf::FrameWs w = gf::carve_workspace(base, n_rays);
    return (uint64_t)((char*)w.ctrl - base);
}

GF_EXPORT uint64_t gf_frame_sizeof(void) { return sizeof(gf_frame_t); }

namespace {

int launch_head(const gf_frame_t* f, hipStream_t s, hipEvent_t* ev /* nullable: 2*(iters)+... */, uint32_t* n_iters_out) {
    const gf::FrameWs w = gf::carve_workspace(f->workspace, f->n_rays);
    const uint32_t N = f->n_rays;
    InitArgs ia;
    ia.rays_o_in = f->rays_o; ia.rays_d_in = f->rays_d;
    for (int i = 0; i < 12; i++) ia.pose[i] = f->pose[i];
    ia.fx = f->intrinsics[0]; ia.fy = f->intrinsics[1]; ia.cx = f->intrinsics[2]; ia.cy = f->intrinsics[3];
    ia.img_w = f->img_w ? f->img_w : 1;
    ia.aabb = f->aabb; ia.min_near = f->min_near;
    ia.rays_o = w.rays_o; ia.rays_d = w.rays_d; ia.nears = w.nears; ia.fars = w.fars; ia.rays_t = w.rays_t;
    ia.weights_sum = w.weights_sum; ia.depth = w.depth; ia.image = w.image; ia.alive = w.alive_a; ia.ctrl = w.ctrl; ia.N = N;
    hipLaunchKernelGGL(k_frame_init, dim3(gf_div_up(N, 256u)), dim3(256), 0, s, ia);

    HeadArgs ha;
    gf::fill_march_params(ha.mp, f->bitfield, f->bound, f->dt_gamma, f->max_steps, f->cascade, f->grid_size);
    if (gf::fill_grid_levels(ha.lv3, 16, f->pos_S, f->base_res) || gf::fill_grid_levels(ha.lv2, 16, f->amb_S, f->base_res))
        return gf_set_error(GF_ERR_INVALID, "frame: bad grid levels");
    ha.pos_table = f->pos_table; ha.pos_offsets = f->pos_offsets; ha.amb_table = f->amb_table; ha.amb_offsets = f->amb_offsets;
    ha.head_pack = f->head_pack; ha.amb_bias = f->amb_bias;
    ha.rays_o = w.rays_o; ha.rays_d = w.rays_d; ha.fars = w.fars;
    ha.rays_t = w.rays_t; ha.weights_sum = w.weights_sum; ha.depth = w.depth; ha.image = w.image;
    ha.ctrl = w.ctrl; ha.N = N; ha.max_steps = f->max_steps; ha.gridtype = f->gridtype; ha.interp = f->interp;
    ha.T_thresh = f->T_thresh; ha.bound = f->bound;

    // the n_step schedule needs at most max_steps iterations (n_step >= 1); every launch past the last live one exits at once
    const uint32_t iters = f->max_steps < gf::kMaxIters ? f->max_steps : gf::kMaxIters;
    const size_t smem = (size_t)kSmemFloats * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_iter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return gf_set_error(GF_ERR_HIP, "frame: cannot raise the dynamic LDS limit to %zu bytes", smem);
        attr_set = true;
    }
    // persistent grid: 2 workgroups per CU x 256 CUs, capped by the number of ray groups in the busiest iteration
    const uint32_t max_groups = gf_div_up(N, (uint32_t)(kPass / 8)) ;  // upper bound over every n_step (n_alive*n_step <= N)
    const uint32_t grid = max_groups < 512u ? max_groups : 512u;
    for (uint32_t it = 0; it < iters; it++) {
        ha.iter = it;
        ha.alive_in = (it & 1) ? w.alive_b : w.alive_a;
        ha.alive_out = (it & 1) ? w.alive_a : w.alive_b;
        if (ev) (void)hipEventRecord(ev[2 * it], s);
        hipLaunchKernelGGL(k_head_iter, dim3(grid), dim3(kThreads), smem, s, ha);
        if (ev) (void)hipEventRecord(ev[2 * it + 1], s);
    }
    if (n_iters_out) *n_iters_out = iters;
    return gf_check_launch("render_head");
}

}  // namespace

// Head pass (NeRFRenderer.render inference branch up to the background blend): fills the workspace accumulators.
// When f->torso_pack == NULL the head-only tail (bg blend, clamp, depth) is also enqueued and the outputs are final.
GF_EXPORT int gf_render_head(const gf_frame_t* f, void* stream) {
    int rc = check_frame(f);
    if (rc) return rc;
    if (f->max_steps > gf::kMaxIters) return gf_set_error(GF_ERR_UNSUPPORTED, "frame: max_steps > %u needs the op-by-op path", gf::kMaxIters);
    hipStream_t s = gf_stream(stream);
    rc = launch_head(f, s, nullptr, nullptr);
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

// Same work as gf_render_head's march iterations, bracketed by HIP events on `stream`; synchronises, then reports each
// iteration kernel's duration (ms).  iter_ms_host must hold kMaxIters floats.  For measurement only (bench.py roofline).
GF_EXPORT int gf_render_head_timed(const gf_frame_t* f, void* stream, float* iter_ms_host, uint32_t* n_iters_host) {
    int rc = check_frame(f);
    if (rc) return rc;
    if (f->max_steps > gf::kMaxIters) return gf_set_error(GF_ERR_UNSUPPORTED, "frame: max_steps > %u", gf::kMaxIters);
    static hipEvent_t ev[2 * gf::kMaxIters];
    static bool made = false;
    if (!made) {
        for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return gf_set_error(GF_ERR_HIP, "hipEventCreate failed");
        made = true;
    }
    hipStream_t s = gf_stream(stream);
    uint32_t iters = 0;
    rc = launch_head(f, s, ev, &iters);
    if (rc) return rc;
    if (hipStreamSynchronize(s) != hipSuccess) return gf_set_error(GF_ERR_HIP, "hipStreamSynchronize failed");
    for (uint32_t i = 0; i < iters; i++) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        iter_ms_host[i] = ms;
    }
    *n_iters_host = iters;
    return GF_OK;
}
