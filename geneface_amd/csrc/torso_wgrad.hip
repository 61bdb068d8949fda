// Weight gradients of the torso field (gfx950): the seven tall products dW = dZ^T X of RADNeRFTorso.forward_torso's backward
// (/root/reference/modules/radnerfs/radnerf_torso.py:51-84 under autograd: torso_deform_net and torso_canonicial_net, three bias-free Linear
// layers each), the two column sums behind the per-frame-constant columns (pose encoding | identity code) and d code, in three launches.
//
// What it replaces: geneface_amd/train_torso.py's seven batched library products + sums, two column sums, two outer products, two GEMVs and
// four concatenations -- ~25 launches of a step that runs at the host's launch rate (NOTES 10.3, 10.11).  22 KFLOP per pixel, 65 536 pixels:
// nothing for the matrix pipe to do.  A workgroup takes rows in stages of 32, all thirteen matrices of a stage in LDS (57 KB), every lane owns up
// to three 4 x 4 blocks of the 668 the products have (register blocking: eight LDS reads per 16 FMAs); the column sums ride along as a ones
// column of the frequency encoding.  Partial sums per workgroup, added in workgroup order by the second launch, which writes every gradient
// tensor whole (outer-product columns and d v included): no atomics, the same bits every run.
#include "common.hpp"
#include "geneface_hip.h"

namespace {

constexpr int kThreads = 256, kR = 32, kWGs = 256;
// LDS layout of a stage (floats per row, offset of the matrix): X side enc 48 (cols 0..41 real, col 42 := 1), h_d1 64, h_d2 64, g 32, h_c1 32,
// h_c2 32; G side dz_d1 64, dz_d2 64, dz_d3 4 (2 real), dz_c1 32, dz_c2 32, dz_c3 4
enum { X_ENC, X_HD1, X_HD2, X_G, X_HC1, X_HC2, G_D1, G_D2, G_D3, G_C1, G_C2, G_C3, N_MAT };
constexpr int kW[N_MAT] = {48, 64, 64, 32, 32, 32, 64, 64, 4, 32, 32, 4};        // LDS row width
constexpr int kSrcW[N_MAT] = {48, 64, 64, 32, 32, 32, 64, 64, 2, 32, 32, 4};     // width in global memory
constexpr int mat_off(int m) { int o = 0; for (int i = 0; i < m; i++) o += kR * kW[i]; return o; }
constexpr int kLdsFloats = mat_off(N_MAT);
static_assert(kLdsFloats * 4 + 256 <= 64 * 1024, "one stage in static LDS");

// the products: gradient matrix (G side), O rows; activation matrix (X side), I columns (the +1 of the two enc products = the ones column)
struct Prod { int g, O, x, I; };
constexpr Prod kProd[7] = {{G_D1, 64, X_ENC, 43}, {G_D2, 64, X_HD1, 64}, {G_D3, 2, X_HD2, 64}, {G_C1, 32, X_G, 32},
                           {G_C1, 32, X_ENC, 43}, {G_C2, 32, X_HC1, 32}, {G_C3, 4, X_HC2, 32}};
constexpr int nbo(int p) { return (kProd[p].O + 3) / 4; }
constexpr int nbi(int p) { return (kProd[p].I + 3) / 4; }
constexpr int blk_base(int p) { int b = 0; for (int i = 0; i < p; i++) b += nbo(i) * nbi(i); return b; }
constexpr int kBlocks = blk_base(7);
static_assert(kBlocks <= 3 * kThreads, "three blocks per lane");

struct Tables { int g_off[7], g_w[7], x_off[7], x_w[7], nbi[7], base[8]; };
__host__ __device__ inline Tables tables() {
    Tables t = {};
    const int w[N_MAT] = {48, 64, 64, 32, 32, 32, 64, 64, 4, 32, 32, 4};
    int off[N_MAT], o = 0;
    for (int m = 0; m < N_MAT; m++) { off[m] = o; o += kR * w[m]; }
    const int pg[7] = {G_D1, G_D2, G_D3, G_C1, G_C1, G_C2, G_C3}, px[7] = {X_ENC, X_HD1, X_HD2, X_G, X_ENC, X_HC1, X_HC2};
    const int pO[7] = {64, 64, 2, 32, 32, 32, 4}, pI[7] = {43, 64, 64, 32, 43, 32, 32};
    int b = 0;
    for (int p = 0; p < 7; p++) {
        t.g_off[p] = off[pg[p]]; t.g_w[p] = w[pg[p]]; t.x_off[p] = off[px[p]]; t.x_w[p] = w[px[p]];
        t.nbi[p] = (pI[p] + 3) / 4; t.base[p] = b; b += ((pO[p] + 3) / 4) * t.nbi[p];
    }
    t.base[7] = b;
    return t;
}

struct WArgs {
    const float* src[N_MAT];
    const float *v, *w_d1, *w_c1;
    float *g_wd1, *g_wd2, *g_wd3, *g_wc1, *g_wc2, *g_wc3, *g_v;
    float* ws;                 // [kWGs][kBlocks][16]
    uint32_t M, rows_per_wg;   // rows_per_wg: a multiple of kR
};

__global__ void __launch_bounds__(kThreads) k_torso_wgrad(const WArgs a) {
    __shared__ __attribute__((aligned(16))) float L[kLdsFloats];
    __shared__ Tables T;                                               // indexed by a run-time product number: LDS, not private memory
    const int tid = threadIdx.x;
    if (tid == 0) T = tables();
    __syncthreads();
    // this lane's (up to) three blocks: LDS pointers of its four gradient columns and its four activation columns
    int goff[3], gw[3], xoff[3], xw[3];
    bool on[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int b = tid + k * kThreads;
        on[k] = b < kBlocks;
        int p = 0;
#pragma unroll
        for (int q = 1; q < 7; q++) p += (on[k] && b >= T.base[q]) ? 1 : 0;
        const int lb = on[k] ? b - T.base[p] : 0, bo = lb / T.nbi[p], bi = lb % T.nbi[p];
        goff[k] = T.g_off[p] + 4 * bo; gw[k] = T.g_w[p];
        xoff[k] = T.x_off[p] + 4 * bi; xw[k] = T.x_w[p];
    }
    float acc[3][16];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[k][e] = 0.0f;
    const uint32_t row_begin = blockIdx.x * a.rows_per_wg;
    const uint32_t row_end = row_begin + a.rows_per_wg < a.M ? row_begin + a.rows_per_wg : a.M;
    for (uint32_t row0 = row_begin; row0 < row_end; row0 += kR) {
        __syncthreads();
        int off = 0;
#pragma unroll
        for (int m = 0; m < N_MAT; m++) {                              // stage: rows [row0, row0 + kR) of every matrix, zeros beyond the end
            const int w = kW[m], sw = kSrcW[m];
            const float* __restrict__ src = a.src[m] + (size_t)row0 * sw;
            for (int i = tid; i < kR * w; i += kThreads) {
                const int r = i / w, c = i - r * w;
                float val = 0.0f;
                if (row0 + (uint32_t)r < row_end && c < sw) val = src[r * sw + c];
                if (m == X_ENC && c == 42) val = row0 + (uint32_t)r < row_end ? 1.0f : 0.0f;      // the ones column: column sums of dz_d1 / dz_c1
                L[off + i] = val;
            }
            off += kR * w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (!on[k]) continue;
            const float* gp = L + goff[k];
            const float* xp = L + xoff[k];
#pragma unroll 4
            for (int r = 0; r < kR; r++) {
                const float4 g4 = *reinterpret_cast<const float4*>(gp + r * gw[k]);
                const float4 x4 = *reinterpret_cast<const float4*>(xp + r * xw[k]);
                const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int o = 0; o < 4; o++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[k][4 * o + i] = __builtin_fmaf(gv[o], xv[i], acc[k][4 * o + i]);
            }
        }
    }
    float* __restrict__ out = a.ws + (size_t)blockIdx.x * kBlocks * 16;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (on[k]) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4*>(out + (size_t)(tid + k * kThreads) * 16 + 4 * q) = float4{acc[k][4 * q], acc[k][4 * q + 1], acc[k][4 * q + 2], acc[k][4 * q + 3]};
        }
}

// element (o, i) of product p summed over the workgroups in workgroup order
__device__ __forceinline__ float reduced(const float* __restrict__ ws, const Tables& T, int n_wg, int p, int o, int i) {
    const size_t e = (size_t)(T.base[p] + (o >> 2) * T.nbi[p] + (i >> 2)) * 16 + (o & 3) * 4 + (i & 3);
    float s0 = 0.0f, s1 = 0.0f;
    int k = 0;
    for (; k + 2 <= n_wg; k += 2) { s0 += ws[(size_t)k * kBlocks * 16 + e]; s1 += ws[(size_t)(k + 1) * kBlocks * 16 + e]; }
    if (k < n_wg) s0 += ws[(size_t)k * kBlocks * 16 + e];
    return s0 + s1;
}

// the 96 column sums (dz_d1: 64, dz_c1: 32) first, four lanes each -- the reduction's outer-product columns and d v read them from the tail of the
// workspace (the first version let the 62 lanes of d v add 96 x n_wg partials each, one after the other: 1.5 ms)
__global__ void __launch_bounds__(kThreads) k_torso_wgrad_sums(const WArgs a, int n_wg) {
    const Tables T = tables();
    const int t = (int)(blockIdx.x * kThreads + threadIdx.x), c = t >> 2, part = t & 3;
    if (c >= 96) return;
    const int p = c < 64 ? 0 : 4, o = c < 64 ? c : c - 64;
    const size_t e = (size_t)(T.base[p] + (o >> 2) * T.nbi[p] + (42 >> 2)) * 16 + (o & 3) * 4 + (42 & 3);
    float sum = 0.0f;
    for (int k = part; k < n_wg; k += 4) sum += a.ws[(size_t)k * kBlocks * 16 + e];
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (part == 0) a.ws[(size_t)kWGs * kBlocks * 16 + c] = sum;
}

// one lane per element of the seven gradient tensors and d v
__global__ void __launch_bounds__(kThreads) k_torso_wgrad_reduce(const WArgs a, int n_wg) {
    const Tables T = tables();
    const float* __restrict__ colsum = a.ws + (size_t)kWGs * kBlocks * 16;      // [0:64] sum of dz_d1 over the pixels, [64:96] of dz_c1
    int e = (int)(blockIdx.x * kThreads + threadIdx.x);
    if (e < 64 * 104) {                                                // dW_deform0 [64,104]: frequency encoding 42 | per-frame constants 62
        const int o = e / 104, c = e % 104;
        a.g_wd1[e] = c < 42 ? reduced(a.ws, T, n_wg, 0, o, c) : colsum[o] * a.v[c - 42];
        return;
    }
    e -= 64 * 104;
    if (e < 64 * 64) { a.g_wd2[e] = reduced(a.ws, T, n_wg, 1, e / 64, e % 64); return; }
    e -= 64 * 64;
    if (e < 2 * 64) { a.g_wd3[e] = reduced(a.ws, T, n_wg, 2, e / 64, e % 64); return; }
    e -= 2 * 64;
    if (e < 32 * 136) {                                                // dW_canon0 [32,136]: grid features 32 | frequency encoding 42 | constants 62
        const int o = e / 136, c = e % 136;
        a.g_wc1[e] = c < 32 ? reduced(a.ws, T, n_wg, 3, o, c) : c < 74 ? reduced(a.ws, T, n_wg, 4, o, c - 32) : colsum[64 + o] * a.v[c - 74];
        return;
    }
    e -= 32 * 136;
    if (e < 32 * 32) { a.g_wc2[e] = reduced(a.ws, T, n_wg, 5, e / 32, e % 32); return; }
    e -= 32 * 32;
    if (e < 4 * 32) { a.g_wc3[e] = reduced(a.ws, T, n_wg, 6, e / 32, e % 32); return; }
    e -= 4 * 32;
    if (e < 62) {                                                      // d v = W_deform0[:, 42:]^T s_d1 + W_canon0[:, 74:]^T s_c1
        float sum = 0.0f;
        for (int o = 0; o < 64; o++) sum = __builtin_fmaf(a.w_d1[o * 104 + 42 + e], colsum[o], sum);
        float sum2 = 0.0f;
        for (int o = 0; o < 32; o++) sum2 = __builtin_fmaf(a.w_c1[o * 136 + 74 + e], colsum[64 + o], sum2);
        a.g_v[e] = sum + sum2;
    }
}

constexpr int kReduceElems = 64 * 104 + 64 * 64 + 2 * 64 + 32 * 136 + 32 * 32 + 4 * 32 + 62;

}  // namespace

GF_EXPORT uint64_t gf_torso_wgrad_ws_bytes(void) { return ((uint64_t)kWGs * kBlocks * 16 + 96) * sizeof(float); }

GF_EXPORT int gf_torso_wgrad(const gf_torso_wgrad_t* w, void* stream) {
    if (!w) return gf_set_error(GF_ERR_INVALID, "torso_wgrad: null descriptor");
    const void* out[] = {w->g_wd1, w->g_wd2, w->g_wd3, w->g_wc1, w->g_wc2, w->g_wc3, w->g_v, w->workspace, w->v, w->w_d1, w->w_c1};
    for (const void* p : out) if (!p) return gf_set_error(GF_ERR_INVALID, "torso_wgrad: null buffer");
    const void* in[] = {w->enc, w->h_d1, w->h_d2, w->g, w->h_c1, w->h_c2, w->dz_d1, w->dz_d2, w->dz_d3, w->dz_c1, w->dz_c2, w->dz_c3};
    if (w->M) for (const void* p : in) if (!p) return gf_set_error(GF_ERR_INVALID, "torso_wgrad: null input matrix");
    WArgs a = {};
    const float* src[N_MAT] = {w->enc, w->h_d1, w->h_d2, w->g, w->h_c1, w->h_c2, w->dz_d1, w->dz_d2, w->dz_d3, w->dz_c1, w->dz_c2, w->dz_c3};
    for (int m = 0; m < N_MAT; m++) a.src[m] = src[m];
    a.v = w->v; a.w_d1 = w->w_d1; a.w_c1 = w->w_c1;
    a.g_wd1 = w->g_wd1; a.g_wd2 = w->g_wd2; a.g_wd3 = w->g_wd3; a.g_wc1 = w->g_wc1; a.g_wc2 = w->g_wc2; a.g_wc3 = w->g_wc3; a.g_v = w->g_v;
    a.ws = w->workspace; a.M = w->M;
    const uint32_t stages = gf_div_up(w->M ? w->M : 1u, (uint32_t)kR);
    const uint32_t n_wg = stages < (uint32_t)kWGs ? stages : (uint32_t)kWGs;
    a.rows_per_wg = gf_div_up(stages, n_wg) * (uint32_t)kR;
    if (w->M) {
        hipLaunchKernelGGL(k_torso_wgrad, dim3(n_wg), dim3(kThreads), 0, gf_stream(stream), a);
        if (const int e = gf_check_launch("torso_wgrad")) return e;
    }
    hipLaunchKernelGGL(k_torso_wgrad_sums, dim3(gf_div_up(96 * 4, kThreads)), dim3(kThreads), 0, gf_stream(stream), a, w->M ? (int)n_wg : 0);
    hipLaunchKernelGGL(k_torso_wgrad_reduce, dim3(gf_div_up(kReduceElems, kThreads)), dim3(kThreads), 0, gf_stream(stream), a, w->M ? (int)n_wg : 0);
    return gf_check_launch("torso_wgrad (reduce)");
}
