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
// WORK DECOMPOSITION.  k_frame_init: one lane per ray -- ray generation, slab test, and the march through empty space up to
// the first occupied sample (a latency-bound, high-occupancy kernel; rays that never hit anything are finished here).
// k_head_phase: persistent 256-thread workgroups (2 per CU) each keep a pool of up to 128 live rays in LDS, refilled from
// a global queue, and loop over rounds of <= 128 samples:
//   A. march    : one lane per pooled ray, samples to LDS, wave-scan packs the valid ones densely
//   B. field    : each of the 4 waves owns a 32-sample MFMA tile.  Activations stay in registers in the accumulator
//                 layout of v_mfma_f32_32x32x2_f32, which is also a legal B-operand layout for the next layer, so layers
//                 chain with no data movement; weights stream L2 -> LDS by asynchronous LDS-DMA in half-layer chunks
//                 (two 32 KB buffers: chunk k+1 lands while chunk k feeds the MFMAs); the three skinny output layers
//                 (->2, ->1, ->3) run on the VALU; grid lookups are split across the two 32-lane halves of the wave
//   C. composite: the owning lane consumes its samples in order; finished rays write their accumulators once.
#include "common.hpp"
#include "frame.hpp"
#include "march_core.hpp"
#include "grid_core.hpp"
#include "sh_core.hpp"
#include "mfma_mlp.hpp"

namespace {

using gf::floatx16;

constexpr int kThreads = 256;
constexpr int kPass = 128;            // sample slots per round (4 waves x 32-column MFMA tiles)
constexpr int kPool = 128;            // live rays per workgroup
constexpr int kBufFloats = 8192;      // one 32 KB weight buffer
constexpr int kPFloats = gf::HS_TOTAL + 128 /*amb bias*/ + 128 /*level meta: 2 grids x 16 x {scale,res,off,rows}*/;
constexpr int kHistBins = gf::kMaxSteps + 2;

// ---- optional per-round timeline (built only with -DGF_TRACE into libgeneface_hip_trace.so; see tools/trace_head.py) ----
#ifdef GF_TRACE
constexpr int kTraceSlots = 40, kTraceRounds = 48, kTraceWGs = 16;
static uint32_t* g_trace_buf = nullptr;
#define GF_STAMP(i)                                                              \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        if (tid == 0) s.tr[(i)] = (uint32_t)__builtin_amdgcn_s_memtime();        \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
#else
#define GF_STAMP(i) do { } while (0)
#endif

struct HeadArgs {
    gf::MarchParams mp;
    gf::GridLevels lv3, lv2;
    const float* pos_table; const int* pos_offsets;
    const float* amb_table; const int* amb_offsets;
    const float* head_pack; const float* amb_bias;
    const float* rays_o; const float* rays_d; const float* fars;
    float* rays_t; float* weights_sum; float* depth; float* image;
    const int* queue;   // phase 0: hit list, phase 1: survivor list
    int* survivors;     // phase 0 output
    uint32_t* ctrl;
    uint32_t N, phase, max_steps, gridtype, interp;
    float T_thresh, bound;
#ifdef GF_TRACE
    uint32_t* trace;
#endif
};

// ---------------------------------------------------------------------------------------------------- LDS carve
struct Smem {
    float* buf[2];   // weight chunk double buffer
    float* P;        // VALU-layer rows, colour bias, ambient bias, per-level grid meta
    // ray pool (slot = owner thread)
    int* p_ray; float *p_ox, *p_oy, *p_oz, *p_dx, *p_dy, *p_dz, *p_t, *p_far, *p_ws, *p_dep, *p_r, *p_g, *p_b; uint32_t* p_done;
    // per-round sample staging (raw slot = rank * n + s); outputs alias the positions
    float *sx, *sy, *sz, *sdt, *st, *ob;
    uint8_t *d2r, *rcnt, *rbase, *rrank;
    uint32_t* hist;  // [kHistBins]
    uint32_t* misc;  // [16]
#ifdef GF_TRACE
    uint32_t* tr;    // [kTraceSlots]
#endif
};
#ifdef GF_TRACE
constexpr int kSmemBytes = (2 * kBufFloats + kPFloats + 15 * kPool + 6 * kPass + kHistBins + 16) * 4 + 4 * kPass + 4 * kTraceSlots;
#else
constexpr int kSmemBytes = (2 * kBufFloats + kPFloats + 15 * kPool + 6 * kPass + kHistBins + 16) * 4 + 4 * kPass;
#endif
static_assert(2 * kSmemBytes <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ Smem carve(char* base) {
    Smem s;
    float* f = reinterpret_cast<float*>(base);
    s.buf[0] = f; f += kBufFloats;
    s.buf[1] = f; f += kBufFloats;
    s.P = f; f += kPFloats;
    s.p_ray = reinterpret_cast<int*>(f); f += kPool;
    s.p_ox = f; f += kPool; s.p_oy = f; f += kPool; s.p_oz = f; f += kPool;
    s.p_dx = f; f += kPool; s.p_dy = f; f += kPool; s.p_dz = f; f += kPool;
    s.p_t = f; f += kPool; s.p_far = f; f += kPool;
    s.p_ws = f; f += kPool; s.p_dep = f; f += kPool; s.p_r = f; f += kPool; s.p_g = f; f += kPool; s.p_b = f; f += kPool;
    s.p_done = reinterpret_cast<uint32_t*>(f); f += kPool;
    s.sx = f; f += kPass; s.sy = f; f += kPass; s.sz = f; f += kPass; s.sdt = f; f += kPass; s.st = f; f += kPass; s.ob = f; f += kPass;
    s.hist = reinterpret_cast<uint32_t*>(f); f += kHistBins;
    s.misc = reinterpret_cast<uint32_t*>(f); f += 16;
    uint8_t* b = reinterpret_cast<uint8_t*>(f);
    s.d2r = b; s.rcnt = b + kPass; s.rbase = b + 2 * kPass; s.rrank = b + 3 * kPass;
#ifdef GF_TRACE
    s.tr = reinterpret_cast<uint32_t*>(b + 4 * kPass);
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

__global__ void __launch_bounds__(kThreads, 2) k_head_phase(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const Smem s = carve(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const bool owner = tid < kPool;

    // ---- phase set-up (uniform) ----
    uint32_t budget, limit;
    const uint32_t qhead = a.phase ? gf::kCtrlQHead1 : gf::kCtrlQHead0;
    if (a.phase == 0) {
        budget = a.max_steps;
        limit = a.ctrl[gf::kCtrlNHit];
    } else {
        if (tid == 0) {
            const uint32_t B = replay_budget(a.ctrl + gf::kCtrlHist, a.N, a.max_steps);
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
    uint32_t pool_cap = (limit + gridDim.x - 1) / gridDim.x;
    pool_cap = pool_cap < 16u ? 16u : (pool_cap > (uint32_t)kPool ? (uint32_t)kPool : pool_cap);
    if ((uint32_t)blockIdx.x * pool_cap >= limit) return;  // not even one refill's worth of work for this workgroup

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
    if (tid < kHistBins) s.hist[tid] = 0;
    if (owner) s.p_ray[tid] = -1;
    const float* pack = a.head_pack;
    uint32_t par = 0;               // which weight buffer the next chunk goes to
    uint32_t st_samples = 0, st_rounds = 0, st_tiles = 0;  // statistics (thread 0)
    bool queue_open = true;         // uniform

#ifdef GF_TRACE
    uint32_t tr_round = 0;
#endif
    for (;;) {
        __syncthreads();  // previous round fully retired (pool, staging)
        GF_STAMP(0);
        // ------------------------------------------------------------------ refill empty pool slots from the queue
        int ray = owner ? s.p_ray[tid] : -1;
        if (queue_open && wave < 2) {  // wave-uniform branch
            const bool want = ray < 0 && (uint32_t)tid < pool_cap;
            const unsigned long long m = __ballot(want);
            const uint32_t nw = (uint32_t)__popcll(m);
            uint32_t base = 0;
            if (lane == 0 && nw) base = atomicAdd(&a.ctrl[qhead], nw);
            base = __shfl(base, 0);
            if (want) {
                const uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (idx < limit) {
                    ray = a.queue[idx];
                    const float* o = a.rays_o + (size_t)ray * 3;
                    const float* d = a.rays_d + (size_t)ray * 3;
                    s.p_ox[tid] = o[0]; s.p_oy[tid] = o[1]; s.p_oz[tid] = o[2];
                    s.p_dx[tid] = d[0]; s.p_dy[tid] = d[1]; s.p_dz[tid] = d[2];
                    s.p_t[tid] = a.rays_t[ray];
                    s.p_far[tid] = a.fars[ray];
                    if (a.phase == 0) {
                        s.p_ws[tid] = 0.0f; s.p_dep[tid] = 0.0f; s.p_r[tid] = 0.0f; s.p_g[tid] = 0.0f; s.p_b[tid] = 0.0f;
                    } else {
                        s.p_ws[tid] = a.weights_sum[ray]; s.p_dep[tid] = a.depth[ray];
                        s.p_r[tid] = a.image[(size_t)ray * 3]; s.p_g[tid] = a.image[(size_t)ray * 3 + 1]; s.p_b[tid] = a.image[(size_t)ray * 3 + 2];
                    }
                    s.p_done[tid] = 0;
                    s.p_ray[tid] = ray;
                }
            }
            if (lane == 0) s.misc[4 + wave] = (nw && base + nw >= limit) ? 1u : 0u;  // this wave saw the end of the queue
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
        uint32_t n = kPass / n_pool;
        n = n > 8u ? 8u : n;          // >= 1 since n_pool <= 128
        // ------------------------------------------------------------------ A. march
        uint32_t cnt = 0, req = 0, rank = 0;
        float t_ray = 0.0f;
        if (alive) {
            rank = (wave ? s.misc[0] : 0u) + (uint32_t)__popcll(amask & ((1ull << lane) - 1ull));
            const uint32_t left = budget - s.p_done[tid];
            req = n < left ? n : left;
            t_ray = s.p_t[tid];
            const uint32_t base = rank * n;
            cnt = gf::march_ray(a.mp, s.p_ox[tid], s.p_oy[tid], s.p_oz[tid], s.p_dx[tid], s.p_dy[tid], s.p_dz[tid], s.p_far[tid], 0.0f, req, t_ray,
                                [&](uint32_t q, float x, float y, float z, float dt, float t_after, float) {
                                    s.sx[base + q] = x; s.sy[base + q] = y; s.sz[base + q] = z;
                                    s.sdt[base + q] = dt; s.st[base + q] = t_after;
                                });
        }
        GF_STAMP(3);
        if (owner) s.rcnt[tid] = (uint8_t)cnt;
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
            for (uint32_t q = 0; q < cnt; q++) { s.d2r[b + q] = (uint8_t)(rank * n + q); s.rrank[b + q] = (uint8_t)tid; }
        }
        if (tid == 0) { st_samples += Mv; st_rounds++; st_tiles += (Mv + 31) / 32; }
        GF_STAMP(5);

        // ------------------------------------------------------------------ B. field on this wave's tile
        if (Mv > 0) {  // uniform over the workgroup
            // chunk 0 (ambient L1) starts streaming now; the d2r table is published by the first barrier below
            gf::dma_to_lds(s.buf[par], pack + gf::HP_AMB1, 4 * 16 * 64, wave, lane);
            __syncthreads();
            GF_STAMP(6);
            const uint32_t j = wave * 32 + (lane & 31);      // dense sample index
            const bool active = (uint32_t)(wave * 32) < Mv;  // wave-uniform
            const bool valid = j < Mv;
            const uint32_t jj = valid ? j : (active ? (uint32_t)(wave * 32) : 0u);
            const uint32_t raw = s.d2r[jj];
            const uint32_t slot = s.rrank[jj];

            float pf[16], af[16], act[64];
            floatx16 h[4];
            float sigma = 0.0f;
            const float* cur;
            [[maybe_unused]] int tr_i = 7;
#define GF_NEXT_CHUNK(SRC, NFLOATS)                                   \
            GF_STAMP(tr_i); tr_i++;                                    \
            __syncthreads(); /* chunk landed; previous chunk's readers are done */ \
            GF_STAMP(tr_i); tr_i++;                                    \
            cur = s.buf[par]; par ^= 1u;                               \
            gf::dma_to_lds(s.buf[par], pack + (SRC), (NFLOATS), wave, lane);
#define GF_LAST_CHUNK()                                               \
            GF_STAMP(tr_i); tr_i++;                                    \
            __syncthreads();                                          \
            GF_STAMP(tr_i); tr_i++;                                    \
            cur = s.buf[par]; par ^= 1u;

            if (active) {
                const float b2 = 2 * a.bound;
                const float x3[3] = {(s.sx[raw] + a.bound) / b2, (s.sy[raw] + a.bound) / b2, (s.sz[raw] + a.bound) / b2};
                gf::encode_half<3>(a.pos_table, s.P + P_META, half, a.gridtype, a.interp, x3, pf);
            }
            GF_NEXT_CHUNK(gf::HP_AMB2, 2 * 64 * 64)                       // -> ambient L2, out-blocks 0-1
            if (active) {
                gf::mfma_layer<4, 16, true, false>(cur, lane, pf, s.P + P_AMBBIAS, h);   // ambient L1 (cond_feat folded into the bias)
                gf::unpack<4>(h, act);
            }
            GF_NEXT_CHUNK(gf::HP_AMB2 + 2 * 64 * 64, 2 * 64 * 64)         // -> ambient L2, out-blocks 2-3
            if (active) gf::mfma_part<4, 0, 2, 64, true, false>(cur, lane, act, nullptr, h);
            GF_NEXT_CHUNK(gf::HP_SIG1, 4 * 32 * 64)                       // -> density L1
            if (active) {
                gf::mfma_part<4, 2, 2, 64, true, false>(cur, lane, act, nullptr, h);
                gf::unpack<4>(h, act);
                float ambient[2];
                gf::valu_rows<2, 4>(s.P + P_SMALL + gf::HS_AMB3, half, act, ambient);
                const float x2[2] = {(tanhf(ambient[0]) + 1.0f) / 2.0f, (tanhf(ambient[1]) + 1.0f) / 2.0f};
                gf::encode_half<2>(a.amb_table, s.P + P_META + 64, half, a.gridtype, a.interp, x2, af);
            }
            GF_NEXT_CHUNK(gf::HP_SIG2, 2 * 64 * 64)                       // -> density L2, 0-1
            if (active) {
                float in[32];
#pragma unroll
                for (int t = 0; t < 16; t++) { in[t] = pf[t]; in[16 + t] = af[t]; }
                gf::mfma_layer<4, 32, true, false>(cur, lane, in, nullptr, h);
                gf::unpack<4>(h, act);
            }
            GF_NEXT_CHUNK(gf::HP_SIG2 + 2 * 64 * 64, 2 * 64 * 64)         // -> density L2, 2-3
            if (active) gf::mfma_part<4, 0, 2, 64, true, false>(cur, lane, act, nullptr, h);
            GF_NEXT_CHUNK(gf::HP_SIG3, 2 * 64 * 64)                       // -> density L3 (geo), 0-1
            if (active) {
                gf::mfma_part<4, 2, 2, 64, true, false>(cur, lane, act, nullptr, h);
                gf::unpack<4>(h, act);
                float h0[1];
                gf::valu_rows<1, 4>(s.P + P_SMALL + gf::HS_SIGROW, half, act, h0);
                sigma = expf(h0[0]);  // trunc_exp forward: plain exp, no clamp (utils.py:41)
            }
            GF_NEXT_CHUNK(gf::HP_SIG3 + 2 * 64 * 64, 2 * 64 * 64)         // -> density L3 (geo), 2-3
            if (active) gf::mfma_part<4, 0, 2, 64, false, false>(cur, lane, act, nullptr, h);  // geometry feature: no activation
            GF_NEXT_CHUNK(gf::HP_COL1S, 4 * 8 * 64)                       // -> colour L1, SH columns
            if (active) {
                gf::mfma_part<4, 2, 2, 64, false, false>(cur, lane, act, nullptr, h);
                gf::unpack<4>(h, act);
            }
            GF_NEXT_CHUNK(gf::HP_COL1G, 2 * 64 * 64)                      // -> colour L1, geo columns, 0-1
            if (active) {
                float sh[16], shh[8];
                gf::sh4(s.p_dx[slot], s.p_dy[slot], s.p_dz[slot], sh);
#pragma unroll
                for (int t = 0; t < 8; t++) shh[t] = half ? sh[8 + t] : sh[t];
                gf::mfma_layer<4, 8, false, false>(cur, lane, shh, s.P + P_SMALL + gf::HS_COLBIAS, h);  // bias = identity-code columns
            }
            GF_NEXT_CHUNK(gf::HP_COL1G + 2 * 64 * 64, 2 * 64 * 64)        // -> colour L1, geo columns, 2-3
            if (active) gf::mfma_part<4, 0, 2, 64, true, true>(cur, lane, act, nullptr, h);
            GF_LAST_CHUNK()
            if (active) {
                gf::mfma_part<4, 2, 2, 64, true, true>(cur, lane, act, nullptr, h);
                gf::unpack<4>(h, act);
                float c[3];
                gf::valu_rows<3, 4>(s.P + P_SMALL + gf::HS_COL2, half, act, c);
                if (valid && half == 0) {  // outputs reuse the position slots (every wave read its positions 11 barriers ago)
                    s.sx[raw] = sigma;
                    s.sy[raw] = 1.0f / (1.0f + __expf(-c[0]));
                    s.sz[raw] = 1.0f / (1.0f + __expf(-c[1]));
                    s.ob[raw] = 1.0f / (1.0f + __expf(-c[2]));
                }
            }
#undef GF_NEXT_CHUNK
#undef GF_LAST_CHUNK
            GF_STAMP(31);
            __syncthreads();
        }
        GF_STAMP(32);

        // ------------------------------------------------------------------ C. composite, retire
        bool survivor = false;
        if (alive) {
            gf::RayAcc acc;
            acc.t = t_ray;
            acc.weight_sum = s.p_ws[tid]; acc.depth = s.p_dep[tid];
            acc.r = s.p_r[tid]; acc.g = s.p_g[tid]; acc.b = s.p_b[tid];
            uint32_t done = s.p_done[tid];
            const uint32_t base = rank * n;
            bool died = false;
            uint32_t d = 0;
            for (uint32_t q = 0; q < cnt; q++) {
                done++;
                if (!gf::composite_sample(acc, s.sx[base + q], s.sy[base + q], s.sz[base + q], s.ob[base + q], s.sdt[base + q], s.st[base + q], a.T_thresh)) {
                    died = true;  // T < T_thresh: terminates at this sample (raymarching.cu:1004)
                    d = done;
                    break;
                }
            }
            if (!died && cnt < req) {  // the marcher ran out: the next request finds nothing (raymarching.cu:977)
                died = true;
                d = done + 1;
            }
            const bool finished = !died && done == budget;
            if (died || finished) {
                a.weights_sum[ray] = acc.weight_sum;
                a.depth[ray] = acc.depth;
                a.image[(size_t)ray * 3 + 0] = acc.r; a.image[(size_t)ray * 3 + 1] = acc.g; a.image[(size_t)ray * 3 + 2] = acc.b;
                if (finished) a.rays_t[ray] = t_ray;
                if (died && a.phase == 0) atomicAdd(&s.hist[d], 1u);
                survivor = finished && a.phase == 0;
                s.p_ray[tid] = -1;
            } else {
                s.p_ws[tid] = acc.weight_sum; s.p_dep[tid] = acc.depth; s.p_r[tid] = acc.r; s.p_g[tid] = acc.g; s.p_b[tid] = acc.b;
                s.p_t[tid] = t_ray;
                s.p_done[tid] = done;
            }
        }
        if (a.phase == 0 && wave < 2) {
            const unsigned long long m = __ballot(survivor);
            const uint32_t ns = (uint32_t)__popcll(m);
            uint32_t base = 0;
            if (lane == 0 && ns) base = atomicAdd(&a.ctrl[gf::kCtrlNSurv], ns);
            base = __shfl(base, 0);
            if (survivor) a.survivors[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ray;
        }
        GF_STAMP(33);
#ifdef GF_TRACE
        if (tid == 0) { s.tr[34] = Mv; s.tr[35] = n_pool; s.tr[36] = n; s.tr[37] = __builtin_amdgcn_s_getreg(63492 /* HW_REG_HW_ID, 32 bits */); }
        __syncthreads();
        if (a.trace && blockIdx.x < kTraceWGs && tr_round < kTraceRounds && tid < kTraceSlots)
            a.trace[((a.phase * kTraceWGs + blockIdx.x) * kTraceRounds + tr_round) * kTraceSlots + tid] = s.tr[tid];
        tr_round++;
#endif
    }

    __syncthreads();
    if (a.phase == 0 && tid >= 1 && tid <= (int)a.max_steps) {
        const uint32_t v = s.hist[tid];
        if (v) atomicAdd(&a.ctrl[gf::kCtrlHist + tid], v);
    }
    if (tid == 0) {
        atomicAdd(&a.ctrl[gf::kCtrlSamples + a.phase], st_samples);
        atomicAdd(&a.ctrl[gf::kCtrlRounds + a.phase], st_rounds);
        atomicAdd(&a.ctrl[gf::kCtrlTiles + a.phase], st_tiles);
    }
}

// ---------------------------------------------------------------------------------------------------- frame setup
struct InitArgs {
    gf::MarchParams mp;
    const float* rays_o_in; const float* rays_d_in;  // explicit rays, or NULL
    float pose[12]; float fx, fy, cx, cy; uint32_t img_w;
    const float* aabb; float min_near;
    float *rays_o, *rays_d, *nears, *fars, *rays_t, *weights_sum, *depth, *image;
    int* hit_list; uint32_t* ctrl; uint32_t N;
};

__global__ void __launch_bounds__(256) k_frame_init(const InitArgs a) {
    const uint32_t n = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool hit = false, miss = false;
    if (n < a.N) {
        float ox, oy, oz, dx, dy, dz;
        if (a.rays_o_in) {
            ox = a.rays_o_in[(size_t)n * 3]; oy = a.rays_o_in[(size_t)n * 3 + 1]; oz = a.rays_o_in[(size_t)n * 3 + 2];
            dx = a.rays_d_in[(size_t)n * 3]; dy = a.rays_d_in[(size_t)n * 3 + 1]; dz = a.rays_d_in[(size_t)n * 3 + 2];
        } else {
            // pinhole rays, pixel centres at +0.5, row-major pixels (utils.py:296-363), same operation order as the torch code
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
        a.weights_sum[n] = 0.0f;
        a.depth[n] = 0.0f;
        a.image[(size_t)n * 3] = 0.0f; a.image[(size_t)n * 3 + 1] = 0.0f; a.image[(size_t)n * 3 + 2] = 0.0f;
        // march through empty space to the first occupied sample; the field kernel restarts the marcher exactly there
        float t = near, t_first = near;
        const uint32_t got = gf::march_ray(a.mp, ox, oy, oz, dx, dy, dz, far, 0.0f, 1u, t,
                                           [&](uint32_t, float, float, float, float, float, float t_at) { t_first = t_at; });
        a.rays_t[n] = t_first;
        hit = got > 0;
        miss = !hit;
    }
    // rays with a sample -> hit list (order is irrelevant: rays are independent); rays without one terminate at index 1
    const unsigned long long hm = __ballot(hit), mm = __ballot(miss);
    const uint32_t nh = (uint32_t)__popcll(hm), nm = (uint32_t)__popcll(mm);
    uint32_t base = 0;
    if (lane == 0) {
        if (nh) base = atomicAdd(&a.ctrl[gf::kCtrlNHit], nh);
        if (nm) atomicAdd(&a.ctrl[gf::kCtrlHist + 1], nm);
    }
    base = __shfl(base, 0);
    if (hit) a.hit_list[base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = (int)n;
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
    if (f->max_steps > gf::kMaxSteps) return gf_set_error(GF_ERR_UNSUPPORTED, "frame: max_steps > %u needs the op-by-op path", gf::kMaxSteps);
    if (f->gridtype > 1 || f->interp > 1) return gf_set_error(GF_ERR_INVALID, "frame: gridtype/interp must be 0 or 1");
    return GF_OK;
}

int launch_head(const gf_frame_t* f, hipStream_t s, hipEvent_t* ev /* nullable: 4 events around the two phase kernels */) {
    const gf::FrameWs w = gf::carve_workspace(f->workspace, f->n_rays);
    const uint32_t N = f->n_rays;
    if (hipMemsetAsync(w.ctrl, 0, gf::kCtrlWords * sizeof(uint32_t), s) != hipSuccess) return gf_set_error(GF_ERR_HIP, "frame: hipMemsetAsync failed");

    InitArgs ia;
    gf::fill_march_params(ia.mp, f->bitfield, f->bound, f->dt_gamma, f->max_steps, f->cascade, f->grid_size);
    ia.rays_o_in = f->rays_o; ia.rays_d_in = f->rays_d;
    for (int i = 0; i < 12; i++) ia.pose[i] = f->pose[i];
    ia.fx = f->intrinsics[0]; ia.fy = f->intrinsics[1]; ia.cx = f->intrinsics[2]; ia.cy = f->intrinsics[3];
    ia.img_w = f->img_w ? f->img_w : 1;
    ia.aabb = f->aabb; ia.min_near = f->min_near;
    ia.rays_o = w.rays_o; ia.rays_d = w.rays_d; ia.nears = w.nears; ia.fars = w.fars; ia.rays_t = w.rays_t;
    ia.weights_sum = w.weights_sum; ia.depth = w.depth; ia.image = w.image; ia.hit_list = w.alive_b; ia.ctrl = w.ctrl; ia.N = N;
    hipLaunchKernelGGL(k_frame_init, dim3(gf_div_up(N, 256u)), dim3(256), 0, s, ia);

    HeadArgs ha;
    ha.mp = ia.mp;
    if (gf::fill_grid_levels(ha.lv3, 16, f->pos_S, f->base_res) || gf::fill_grid_levels(ha.lv2, 16, f->amb_S, f->base_res))
        return gf_set_error(GF_ERR_INVALID, "frame: bad grid levels");
    ha.pos_table = f->pos_table; ha.pos_offsets = f->pos_offsets; ha.amb_table = f->amb_table; ha.amb_offsets = f->amb_offsets;
    ha.head_pack = f->head_pack; ha.amb_bias = f->amb_bias;
    ha.rays_o = w.rays_o; ha.rays_d = w.rays_d; ha.fars = w.fars;
    ha.rays_t = w.rays_t; ha.weights_sum = w.weights_sum; ha.depth = w.depth; ha.image = w.image;
    ha.survivors = w.alive_a;
    ha.ctrl = w.ctrl; ha.N = N; ha.max_steps = f->max_steps; ha.gridtype = f->gridtype; ha.interp = f->interp;
    ha.T_thresh = f->T_thresh; ha.bound = f->bound;
#ifdef GF_TRACE
    ha.trace = g_trace_buf;
#endif

    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_phase), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes) != hipSuccess)
            return gf_set_error(GF_ERR_HIP, "frame: cannot raise the dynamic LDS limit to %d bytes", kSmemBytes);
        attr_set = true;
    }
    // persistent grid: 2 workgroups per CU x 256 CUs, never more workgroups than pools' worth of rays
    const uint32_t pools = gf_div_up(N, (uint32_t)kPool);
    const uint32_t grid = pools < 512u ? pools : 512u;
    for (uint32_t phase = 0; phase < 2; phase++) {
        ha.phase = phase;
        ha.queue = phase ? w.alive_a : w.alive_b;
        if (ev) (void)hipEventRecord(ev[2 * phase], s);
        hipLaunchKernelGGL(k_head_phase, dim3(grid), dim3(kThreads), kSmemBytes, s, ha);
        if (ev) (void)hipEventRecord(ev[2 * phase + 1], s);
    }
    return gf_check_launch("render_head");
}

}  // namespace

#ifdef GF_TRACE
// trace build only: device buffer of 2 * kTraceWGs * kTraceRounds * kTraceSlots uint32 (phase, workgroup, round, slot)
GF_EXPORT void gf_trace_set(void* dev_buf) { g_trace_buf = reinterpret_cast<uint32_t*>(dev_buf); }
GF_EXPORT uint32_t gf_trace_dims(uint32_t which) { return which == 0 ? kTraceWGs : which == 1 ? kTraceRounds : kTraceSlots; }
#endif

GF_EXPORT uint64_t gf_frame_workspace_bytes(uint32_t n_rays) { return gf::carve_workspace(reinterpret_cast<void*>(uintptr_t(1) << 20), n_rays).bytes; }

GF_EXPORT uint32_t gf_frame_ctrl_words(void) { return gf::kCtrlWords; }

// Byte offset of the control block inside the workspace (see frame.hpp for the word layout).
GF_EXPORT uint64_t gf_frame_ctrl_offset(uint32_t n_rays) {
    char* const base = reinterpret_cast<char*>(uintptr_t(1) << 20);
    const gf::FrameWs w = gf::carve_workspace(base, n_rays);
    return (uint64_t)((char*)w.ctrl - base);
}

GF_EXPORT uint64_t gf_frame_sizeof(void) { return sizeof(gf_frame_t); }

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
// their durations (ms): phase_ms_host[0] = first max_steps samples, [1] = the remaining B - max_steps.  Measurement only.
GF_EXPORT int gf_render_head_timed(const gf_frame_t* f, void* stream, float* phase_ms_host, uint32_t* n_phases_host) {
    int rc = check_frame(f);
    if (rc) return rc;
    static hipEvent_t ev[4];
    static bool made = false;
    if (!made) {
        for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return gf_set_error(GF_ERR_HIP, "hipEventCreate failed");
        made = true;
    }
    hipStream_t s = gf_stream(stream);
    rc = launch_head(f, s, ev);
    if (rc) return rc;
    if (hipStreamSynchronize(s) != hipSuccess) return gf_set_error(GF_ERR_HIP, "hipStreamSynchronize failed");
    for (uint32_t i = 0; i < 2; i++) {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        phase_ms_host[i] = ms;
    }
    *n_phases_host = 2;
    return GF_OK;
}
