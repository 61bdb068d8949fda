// Weight gradients of the RAD-NeRF head's field on the AMP tier (gfx950): every tall product  dW = G^T X  of the training backward
// (G = a layer's pre-activation gradient [M,128], X = the activations it multiplied, M ~ 10^6 points) in ONE launch + one reduction.
//
// What it replaces: geneface_amd/train_field.py::_tall_tn -- ten hipBLASLt batched products over 4096-row slabs, each followed by a cast
// and a sum over the slabs (~45 launches, 1.25 ms of a 7.2 ms AMP step, profiles/round6/r6f_*), i.e. the weight gradients autograd
// derives for the Linear layers of /root/reference/modules/radnerfs/radnerf.py:73-105 (cond_encoder.py:106-111 MLP) under
// utils/commons/trainer.py:307-382's autocast: half operands, fp32 accumulation.
//
// The products are HBM-bound: 3.26 KB of binary16 rows per point against 179 KFLOP of f16 MFMA work (0.09 of the matrix pipe at the
// HBM rate).  So the kernel is organised around reading every row ONCE, fully coalesced:
//   * five GROUPS of products that share operands (g_hc1 x [sh | geo] and g_zc x hc1; g_geo x hs2 and g_h0 x hs2; g_hs2 x hs1;
//     g_hs1 x [f3 | f2] and g_ha1 x f3; g_ha2 x ha1 and g_za x ha2).  Each workgroup belongs to one group and owns a contiguous range
//     of rows; workgroups are dealt to the groups in proportion to the bytes a row of the group costs, so all finish together;
//   * a stage = 32 rows of every operand of the group, global -> registers (prefetched one stage ahead, 16 B per lane, whole 256 B rows)
//     -> LDS row-major (rows padded to 320 B) -> MFMA operands read TRANSPOSED from LDS (the reduction index of these products is the
//     row, the slow index of both operands): two ds_read_b64_tr_b16 per operand, gfx950's transposing LDS read;
//   * wave w owns gradient features [32 w, 32 w + 32) (the M index of v_mfma_f32_32x32x16_f16) against all activation tiles, and the
//     n-tile w of the group's skinny product (fp32 gradients g_zc / g_h0 / g_za rounded to binary16 on the way into LDS, as the op graph's
//     `.half()` did); accumulators fp32, in registers for the workgroup's whole row range;
//   * partial sums go to a workspace [workgroup][wave][tile][16][64]; k_wgrad_reduce adds them in a FIXED order and writes the eight
//     gradient tensors in place (sub-blocks of W_color0 / W_ambient0 included).  Same inputs, same bits, run to run.
#include "common.hpp"
#include "geneface_hip.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
constexpr int kR = 32;             // rows per stage
constexpr int kGroups = 5;
// LDS row strides in elements.  binary16: 160 halves = 80 dwords for a 128-wide operand, so the four rows x 64 B a transposing read's half-wave
// touches fall on 4 x 16 distinct banks; 32 halves for a narrow one (<= 32 columns: 16 dwords, the same property without padding).
// fp32 (ds_read_b32, 32 lanes = one row's 32 consecutive floats per pass): no padding needed.
template <class E> constexpr int kWS = sizeof(E) == 2 ? 160 : 128;
template <class E> constexpr int kNS = 32;
template <class E> constexpr int kEPC = 16 / (int)sizeof(E);     // elements per 16-byte chunk
constexpr int kMaxTiles = 6;       // accumulator tiles per wave, the largest group

// sources, in the order of WgArgs::src
enum { S_F3, S_HA1, S_HA2, S_F2, S_HS1, S_HS2, S_GEO, S_HC1, S_SH, S_G_HC1, S_G_GEO, S_G_HS2, S_G_HS1, S_G_HA2, S_G_HA1, S_COUNT };
enum { K_G_ZC, K_G_H0, K_G_ZA, K_COUNT };

struct WgArgs {
    const void* src[S_COUNT];         // binary16 (k_field_wgrad<_Float16>) or fp32 (k_field_wgrad<float>) row-major matrices
    const float* skinny[K_COUNT];
    float* ws;
    uint32_t M;
    uint32_t wg_first[kGroups + 1];   // workgroup ranges of the groups
    uint32_t rows_per_wg[kGroups];    // multiples of kR
    uint32_t ws_base[kGroups];        // floats
};

// A group: NS operands staged per stage (widths 128 / 32 / 16 halves), NP products A[pa] (m-tile = wave) x B[pb] (pn n-tiles), an optional
// skinny product (skc fp32 columns of skinny[sks], rounded to binary16) x n-tile `wave` of operand skb.
template <int G> struct Desc;
template <> struct Desc<0> {   // colour net: g_hc1 x [sh | geo] -> dW_color0[:, 0:144];  g_zc x hc1 -> dW_color1
    static constexpr int NS = 4, NP = 2, SKC = 3, SKS = K_G_ZC, SKB = 2;
    static constexpr int S[4] = {S_G_HC1, S_GEO, S_HC1, S_SH}, W[4] = {128, 128, 128, 16};
    static constexpr int PA[2] = {0, 0}, PB[2] = {3, 1}, PN[2] = {1, 4};
};
template <> struct Desc<1> {   // sigma net, last layer: g_geo x hs2 -> dW_sigma2[1:129];  g_h0 x hs2 -> dW_sigma2[0]
    static constexpr int NS = 2, NP = 1, SKC = 1, SKS = K_G_H0, SKB = 1;
    static constexpr int S[2] = {S_G_GEO, S_HS2}, W[2] = {128, 128};
    static constexpr int PA[1] = {0}, PB[1] = {1}, PN[1] = {4};
};
template <> struct Desc<2> {   // sigma net, middle layer: g_hs2 x hs1 -> dW_sigma1
    static constexpr int NS = 2, NP = 1, SKC = 0;
    static constexpr int S[2] = {S_G_HS2, S_HS1}, W[2] = {128, 128};
    static constexpr int PA[1] = {0}, PB[1] = {1}, PN[1] = {4};
};
template <> struct Desc<3> {   // the two grid-fed layers: g_hs1 x [f3 | f2] -> dW_sigma0;  g_ha1 x f3 -> dW_ambient0[:, 0:32]
    static constexpr int NS = 4, NP = 3, SKC = 0;
    static constexpr int S[4] = {S_G_HS1, S_G_HA1, S_F3, S_F2}, W[4] = {128, 128, 32, 32};
    static constexpr int PA[3] = {0, 0, 1}, PB[3] = {2, 3, 2}, PN[3] = {1, 1, 1};
};
template <> struct Desc<4> {   // ambient net: g_ha2 x ha1 -> dW_ambient1;  g_za x ha2 -> dW_ambient2
    static constexpr int NS = 3, NP = 1, SKC = 2, SKS = K_G_ZA, SKB = 2;
    static constexpr int S[3] = {S_G_HA2, S_HA1, S_HA2}, W[3] = {128, 128, 128};
    static constexpr int PA[1] = {0}, PB[1] = {1}, PN[1] = {4};
};

template <class D, class E> constexpr int lds_stride(int s) { return D::W[s] == 128 ? kWS<E> : kNS<E>; }
template <class D, class E> constexpr int lds_offset(int s) {            // elements
    int o = 0;
    for (int i = 0; i < s; i++) o += kR * lds_stride<D, E>(i);
    return o;
}
template <class D, class E> constexpr int lds_elems() { return lds_offset<D, E>(D::NS) + (D::SKC ? kR * kNS<E> : 0); }
template <class D> constexpr int n_tiles() {
    int n = D::SKC ? 1 : 0;
    for (int p = 0; p < D::NP; p++) n += D::PN[p];
    return n;
}
template <class D, class E> constexpr int chunks_per_thread(int s) { return (kR * D::W[s] / kEPC<E> + kThreads - 1) / kThreads; }
template <class D, class E> constexpr int stage_regs() {
    int n = 0;
    for (int s = 0; s < D::NS; s++) n += chunks_per_thread<D, E>(s);
    return n;
}
template <class E> constexpr int lds_elems_max() {
    int m = lds_elems<Desc<0>, E>();
    if (lds_elems<Desc<1>, E>() > m) m = lds_elems<Desc<1>, E>();
    if (lds_elems<Desc<2>, E>() > m) m = lds_elems<Desc<2>, E>();
    if (lds_elems<Desc<3>, E>() > m) m = lds_elems<Desc<3>, E>();
    if (lds_elems<Desc<4>, E>() > m) m = lds_elems<Desc<4>, E>();
    return m;
}
static_assert(lds_elems_max<float>() * 4 <= 65536, "static LDS of the fp32 kernel");
static_assert(n_tiles<Desc<0>>() <= kMaxTiles && n_tiles<Desc<1>>() <= kMaxTiles && n_tiles<Desc<3>>() <= kMaxTiles, "tiles per wave");

// rows [row, row + 8) of column col0 + (lane & 31) of an LDS tile: the k-slots 8 (lane >> 5) .. + 8 of an MFMA operand whose k index is the
// row.  ds_read_b64_tr_b16 (gfx950): each 16-lane group reads a [4 rows][16 columns] block -- lane i supplies the address of the four
// consecutive halves at row i / 4, columns 4 (i % 4) .. + 4 -- and lane i receives column i of the block, rows 0..3.  Two reads per operand.
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#define GF_LDS __attribute__((address_space(3)))
__device__ __forceinline__ half8 read_rows(const _Float16* T, int stride, int row, int col0, int lane) {
    const int i = lane & 15;
    const _Float16* p = T + (row + (i >> 2)) * stride + col0 + (lane & 16) + 4 * (i & 3);
    const half4 lo = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((GF_LDS fp16x4*)(p)));
    const half4 hi = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((GF_LDS fp16x4*)(p + 4 * stride)));
    return half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <class D, class E>
struct Stage {
    float4 q[stage_regs<D, E>()];
    float sk;
};

// global -> registers: rows [row0, row0 + kR) of every operand of the group; rows at or beyond `row_end` read as zeros
template <class D, class E>
__device__ __forceinline__ void stage_load(Stage<D, E>& st, const WgArgs& a, uint32_t row0, uint32_t row_end, int tid) {
    int k = 0;
#pragma unroll
    for (int s = 0; s < D::NS; s++) {
        const int cpr = D::W[s] / kEPC<E>;                    // 16-byte chunks per row
        const E* __restrict__ p = static_cast<const E*>(a.src[D::S[s]]);
#pragma unroll
        for (int i = 0; i < chunks_per_thread<D, E>(s); i++, k++) {
            const int c = tid + kThreads * i;
            const uint32_t row = row0 + (uint32_t)(c / cpr);
            float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (c < kR * cpr && row < row_end) v = *reinterpret_cast<const float4*>(p + (size_t)row * D::W[s] + (c % cpr) * kEPC<E>);
            st.q[k] = v;
        }
    }
    if constexpr (D::SKC > 0) {
        const uint32_t row = row0 + (uint32_t)(tid / D::SKC);
        st.sk = (tid < kR * D::SKC && row < row_end) ? a.skinny[D::SKS][(size_t)row * D::SKC + tid % D::SKC] : 0.0f;
    }
}

// registers -> LDS (row-major tiles)
template <class D, class E>
__device__ __forceinline__ void stage_store(const Stage<D, E>& st, E* L, int tid) {
    int k = 0;
#pragma unroll
    for (int s = 0; s < D::NS; s++) {
        const int cpr = D::W[s] / kEPC<E>;
#pragma unroll
        for (int i = 0; i < chunks_per_thread<D, E>(s); i++, k++) {
            const int c = tid + kThreads * i;
            if (c < kR * cpr) *reinterpret_cast<float4*>(L + lds_offset<D, E>(s) + (c / cpr) * lds_stride<D, E>(s) + (c % cpr) * kEPC<E>) = st.q[k];
        }
    }
    if constexpr (D::SKC > 0) {
        if (tid < kR * D::SKC) L[lds_offset<D, E>(D::NS) + (tid / D::SKC) * kNS<E> + tid % D::SKC] = (E)st.sk;
    }
}

// The products of one stage.  binary16: one v_mfma_f32_32x32x16_f16 per tile and 16 rows, operands through the transposing read.
// fp32: one v_mfma_f32_32x32x2_f32 per tile and 2 rows -- lane (j, h) supplies row 2 k + h of column j, a plain ds_read_b32 (the 32 lanes
// of a half-wave read 32 consecutive floats); nothing but LDS reads and MFMAs in the loop, because the f32 MFMA hides no VALU work (NOTES 10.1).
template <class D, class E, int NT>
__device__ __forceinline__ void stage_products(const E* L, floatx16 (&acc)[NT], int wave, int lane) {
    const int half = lane >> 5, j = lane & 31;
    if constexpr (sizeof(E) == 2) {
#pragma unroll
        for (int ks = 0; ks < kR / 16; ks++) {
            const int row = 16 * ks + 8 * half;
            int t = 0;
            half8 A = {};
#pragma unroll
            for (int p = 0; p < D::NP; p++) {
                if (p == 0 || D::PA[p] != D::PA[p > 0 ? p - 1 : 0])
                    A = read_rows(L + lds_offset<D, E>(D::PA[p]), lds_stride<D, E>(D::PA[p]), row, 32 * wave, lane);
#pragma unroll
                for (int n = 0; n < D::PN[p]; n++, t++) {
                    const half8 B = read_rows(L + lds_offset<D, E>(D::PB[p]), lds_stride<D, E>(D::PB[p]), row, 32 * n, lane);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[t], 0, 0, 0);
                }
            }
            if constexpr (D::SKC > 0) {
                const half8 As = read_rows(L + lds_offset<D, E>(D::NS), kNS<E>, row, 0, lane);
                const half8 B = read_rows(L + lds_offset<D, E>(D::SKB), lds_stride<D, E>(D::SKB), row, 32 * wave, lane);
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(As, B, acc[NT - 1], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int k2 = 0; k2 < kR / 2; k2++) {
            const int row = 2 * k2 + half;
            int t = 0;
            float A = 0.0f;
#pragma unroll
            for (int p = 0; p < D::NP; p++) {
                if (p == 0 || D::PA[p] != D::PA[p > 0 ? p - 1 : 0])
                    A = L[lds_offset<D, E>(D::PA[p]) + row * lds_stride<D, E>(D::PA[p]) + 32 * wave + j];
#pragma unroll
                for (int n = 0; n < D::PN[p]; n++, t++) {
                    const float B = L[lds_offset<D, E>(D::PB[p]) + row * lds_stride<D, E>(D::PB[p]) + 32 * n + j];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, acc[t], 0, 0, 0);
                }
            }
            if constexpr (D::SKC > 0) {
                const float As = L[lds_offset<D, E>(D::NS) + row * kNS<E> + j];
                const float B = L[lds_offset<D, E>(D::SKB) + row * lds_stride<D, E>(D::SKB) + 32 * wave + j];
                acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(As, B, acc[NT - 1], 0, 0, 0);
            }
        }
    }
}

template <int G, class E>
__device__ __forceinline__ void run_group(const WgArgs& a, E* L, uint32_t wg_local) {
    using D = Desc<G>;
    constexpr int NT = n_tiles<D>();
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t rpw = a.rows_per_wg[G];
    const uint32_t row_begin = wg_local * rpw;
    const uint32_t row_end = row_begin + rpw < a.M ? row_begin + rpw : a.M;

    // narrow tiles are read 32 columns wide: the columns nobody stages stay zero
    for (int i = tid; i < lds_elems_max<E>() * (int)sizeof(E) / 4; i += kThreads) reinterpret_cast<uint32_t*>(L)[i] = 0u;

    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;

    Stage<D, E> st;
    if (row_begin < row_end) stage_load<D, E>(st, a, row_begin, row_end, tid);
    __syncthreads();
    for (uint32_t row0 = row_begin; row0 < row_end; row0 += kR) {
        stage_store<D, E>(st, L, tid);
        __syncthreads();
        if (row0 + kR < row_end) stage_load<D, E>(st, a, row0 + kR, row_end, tid);
        stage_products<D, E, NT>(L, acc, wave, lane);
        __syncthreads();
    }
    float* __restrict__ out = a.ws + a.ws_base[G] + ((size_t)(wg_local * 4 + (uint32_t)wave) * NT) * 1024;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) out[(t * 16 + r) * 64 + lane] = acc[t][r];
}

template <class E>
__device__ __forceinline__ void wgrad_body(const WgArgs& a, E* L) {
    const uint32_t b = blockIdx.x;
    if (b < a.wg_first[1]) run_group<0, E>(a, L, b - a.wg_first[0]);
    else if (b < a.wg_first[2]) run_group<1, E>(a, L, b - a.wg_first[1]);
    else if (b < a.wg_first[3]) run_group<2, E>(a, L, b - a.wg_first[2]);
    else if (b < a.wg_first[4]) run_group<3, E>(a, L, b - a.wg_first[3]);
    else run_group<4, E>(a, L, b - a.wg_first[4]);
}

__global__ void __launch_bounds__(kThreads, 3) k_field_wgrad16(const WgArgs a) {
    __shared__ __attribute__((aligned(16))) _Float16 L[lds_elems_max<_Float16>()];
    wgrad_body<_Float16>(a, L);
}

// The exact tier's products: fp32 rows (gf_field_forward_train's saves, gf_field_backward's fp32 outputs), v_mfma_f32_32x32x2_f32.
__global__ void __launch_bounds__(kThreads, 2) k_field_wgrad32(const WgArgs a) {
    __shared__ __attribute__((aligned(16))) float L[lds_elems_max<float>()];
    wgrad_body<float>(a, L);
}

// One accumulator tile of the result: where its partial sums are and where its 32 x 32 values go.
struct TileOut {
    float* dst;                  // element (0, 0) of the tile in the gradient tensor
    uint32_t ld;                 // row stride of the tensor (floats)
    uint32_t ws_off, ws_stride;  // floats: the first workgroup's partial tile, and the distance to the next workgroup's
    uint32_t n_wg;
    uint16_t rows, cols;         // the real extent of the tile (gradient features x activation features)
    uint32_t _pad;
};
constexpr int kTilesTotal = 4 * (n_tiles<Desc<0>>() + n_tiles<Desc<1>>() + n_tiles<Desc<2>>() + n_tiles<Desc<3>>() + n_tiles<Desc<4>>());
struct ReduceArgs {
    const float* ws;
    TileOut t[kTilesTotal];
};
static_assert(sizeof(ReduceArgs) <= 4096, "kernel arguments");

// block = a quarter of one tile; a lane adds its element over the group's workgroups in workgroup order (a fixed order: reproducible bits)
__global__ void __launch_bounds__(256) k_wgrad_reduce(const ReduceArgs a) {
    const TileOut& t = a.t[blockIdx.x >> 2];
    const uint32_t e = (blockIdx.x & 3u) * 256u + threadIdx.x;
    const float* __restrict__ p = a.ws + t.ws_off + e;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    uint32_t k = 0;
    for (; k + 4 <= t.n_wg; k += 4) {
        s0 += p[(size_t)(k + 0) * t.ws_stride];
        s1 += p[(size_t)(k + 1) * t.ws_stride];
        s2 += p[(size_t)(k + 2) * t.ws_stride];
        s3 += p[(size_t)(k + 3) * t.ws_stride];
    }
    for (; k < t.n_wg; k++) s0 += p[(size_t)k * t.ws_stride];
    const float s = (s0 + s1) + (s2 + s3);
    const uint32_t r = e >> 6, lane = e & 63u;
    const uint32_t m = 8u * (r >> 2) + 4u * (lane >> 5) + (r & 3u), n = lane & 31u;
    if (m < t.rows && n < t.cols) t.dst[(size_t)m * t.ld + n] = s;
}

constexpr uint32_t kBytesPerRow[kGroups] = {3 * 256 + 32 + 12, 2 * 256 + 4, 2 * 256, 2 * 256 + 2 * 64, 3 * 256 + 8};
constexpr uint32_t kTilesPerWave[kGroups] = {(uint32_t)n_tiles<Desc<0>>(), (uint32_t)n_tiles<Desc<1>>(), (uint32_t)n_tiles<Desc<2>>(),
                                             (uint32_t)n_tiles<Desc<3>>(), (uint32_t)n_tiles<Desc<4>>()};
constexpr uint32_t kWgTotal16 = 768;   // three per CU (166 VGPRs, 34 KB of LDS): one resident generation of workgroups, no tail
constexpr uint32_t kWgTotal32 = 512;   // fp32: two per CU (56 KB of LDS)

// the deal of workgroups to groups for M rows: proportional to what a row of the group costs its workgroup -- bytes for the binary16 kernel
// (HBM-bound), accumulator tiles = MFMAs for the fp32 one (matrix-pipe bound) --, at least one, never more than there are stages
void plan(uint32_t M, bool f32, uint32_t (&n_wg)[kGroups], uint32_t (&rows_per_wg)[kGroups]) {
    const uint32_t wg_total = f32 ? kWgTotal32 : kWgTotal16;
    const uint32_t* cost = f32 ? kTilesPerWave : kBytesPerRow;
    uint32_t total = 0;
    for (int g = 0; g < kGroups; g++) total += cost[g];
    const uint32_t stages = gf_div_up(M, (uint32_t)kR);
    uint32_t dealt = 0;
    for (int g = 0; g < kGroups; g++) {
        uint32_t n = (uint32_t)(((uint64_t)wg_total * cost[g] + total / 2) / total);
        if (g == kGroups - 1 && dealt < wg_total) n = wg_total - dealt;      // the rounding remainder: exactly one resident generation
        dealt += n;
        if (n < 1) n = 1;
        if (n > stages) n = stages;
        n_wg[g] = n;
        rows_per_wg[g] = gf_div_up(stages, n) * (uint32_t)kR;
    }
}

}  // namespace

namespace {
uint64_t ws_bytes(bool f32) {
    uint32_t n_wg[kGroups], rpw[kGroups];
    plan(~0u - 64u, f32, n_wg, rpw);                 // the largest deal
    uint64_t floats = 0;
    for (int g = 0; g < kGroups; g++) floats += (uint64_t)n_wg[g] * 4u * kTilesPerWave[g] * 1024u;
    return floats * sizeof(float);
}

int wgrad_impl(bool f32, uint32_t M, const gf_field_wgrad_t* w, void* stream) {
    if (!w) return gf_set_error(GF_ERR_INVALID, "field_wgrad: null pointer");
    const void* in[] = {w->f3, w->ha1, w->ha2, w->f2, w->hs1, w->hs2, w->geo, w->hc1, w->sh, w->g_hc1, w->g_geo, w->g_hs2, w->g_hs1, w->g_ha2, w->g_ha1,
                        w->g_zc, w->g_h0, w->g_za};
    const void* out[] = {w->gw_color1, w->gw_color0, w->gw_sigma2, w->gw_sigma1, w->gw_sigma0, w->gw_ambient2, w->gw_ambient1, w->gw_ambient0, w->workspace};
    for (const void* p : out) if (!p) return gf_set_error(GF_ERR_INVALID, "field_wgrad: null output buffer");
    if (w->ld_color0 < 144 || w->ld_ambient0 < 32) return gf_set_error(GF_ERR_INVALID, "field_wgrad: row strides of W_color0 / W_ambient0 gradients too small");
    if (M > 0) for (const void* p : in) if (!p) return gf_set_error(GF_ERR_INVALID, "field_wgrad: null input buffer");

    WgArgs a = {};
    const void* src[S_COUNT] = {w->f3, w->ha1, w->ha2, w->f2, w->hs1, w->hs2, w->geo, w->hc1, w->sh, w->g_hc1, w->g_geo, w->g_hs2, w->g_hs1, w->g_ha2, w->g_ha1};
    for (int s = 0; s < S_COUNT; s++) a.src[s] = src[s];
    a.skinny[K_G_ZC] = w->g_zc; a.skinny[K_G_H0] = w->g_h0; a.skinny[K_G_ZA] = w->g_za;
    a.ws = w->workspace; a.M = M;
    uint32_t n_wg[kGroups];
    plan(M ? M : 1u, f32, n_wg, a.rows_per_wg);
    uint32_t first = 0, base = 0;
    for (int g = 0; g < kGroups; g++) {
        a.wg_first[g] = first; first += n_wg[g];
        a.ws_base[g] = base; base += n_wg[g] * 4u * kTilesPerWave[g] * 1024u;
    }
    a.wg_first[kGroups] = first;
    if (M > 0) {
        if (f32) hipLaunchKernelGGL(k_field_wgrad32, dim3(first), dim3(kThreads), 0, gf_stream(stream), a);
        else hipLaunchKernelGGL(k_field_wgrad16, dim3(first), dim3(kThreads), 0, gf_stream(stream), a);
        if (const int e = gf_check_launch("field_wgrad16")) return e;
    }

    ReduceArgs r = {};
    r.ws = w->workspace;
    int k = 0;
    auto tile = [&](int g, int wave, int slot, float* dst, uint32_t ld, uint32_t rows, uint32_t cols) {
        TileOut& t = r.t[k++];
        t.dst = dst; t.ld = ld; t.rows = (uint16_t)rows; t.cols = (uint16_t)cols; t.n_wg = n_wg[g];
        t.ws_stride = 4u * kTilesPerWave[g] * 1024u;
        t.ws_off = a.ws_base[g] + ((uint32_t)wave * kTilesPerWave[g] + (uint32_t)slot) * 1024u;
    };
    for (int wv = 0; wv < 4; wv++) {
        // group 0: g_hc1 x sh | geo -> dW_color0[32 wv.., 0:16 | 16:144];  g_zc x hc1 -> dW_color1[0:3, 32 wv..]
        tile(0, wv, 0, w->gw_color0 + (size_t)32 * wv * w->ld_color0, w->ld_color0, 32, 16);
        for (int n = 0; n < 4; n++) tile(0, wv, 1 + n, w->gw_color0 + (size_t)32 * wv * w->ld_color0 + 16 + 32 * n, w->ld_color0, 32, 32);
        tile(0, wv, 5, w->gw_color1 + 32 * wv, 128, 3, 32);
        // group 1: g_geo x hs2 -> dW_sigma2[1 + 32 wv.., :];  g_h0 x hs2 -> dW_sigma2[0, 32 wv..]
        for (int n = 0; n < 4; n++) tile(1, wv, n, w->gw_sigma2 + (size_t)(1 + 32 * wv) * 128 + 32 * n, 128, 32, 32);
        tile(1, wv, 4, w->gw_sigma2 + 32 * wv, 128, 1, 32);
        // group 2: g_hs2 x hs1 -> dW_sigma1
        for (int n = 0; n < 4; n++) tile(2, wv, n, w->gw_sigma1 + (size_t)32 * wv * 128 + 32 * n, 128, 32, 32);
        // group 3: g_hs1 x f3 | f2 -> dW_sigma0[32 wv.., 0:32 | 32:64];  g_ha1 x f3 -> dW_ambient0[32 wv.., 0:32]
        tile(3, wv, 0, w->gw_sigma0 + (size_t)32 * wv * 64, 64, 32, 32);
        tile(3, wv, 1, w->gw_sigma0 + (size_t)32 * wv * 64 + 32, 64, 32, 32);
        tile(3, wv, 2, w->gw_ambient0 + (size_t)32 * wv * w->ld_ambient0, w->ld_ambient0, 32, 32);
        // group 4: g_ha2 x ha1 -> dW_ambient1;  g_za x ha2 -> dW_ambient2[0:2, 32 wv..]
        for (int n = 0; n < 4; n++) tile(4, wv, n, w->gw_ambient1 + (size_t)32 * wv * 128 + 32 * n, 128, 32, 32);
        tile(4, wv, 4, w->gw_ambient2 + 32 * wv, 128, 2, 32);
    }
    if (k != kTilesTotal) return gf_set_error(GF_ERR_INVALID, "field_wgrad: internal tile table");
    if (M == 0) for (int i = 0; i < kTilesTotal; i++) r.t[i].n_wg = 0;     // no rows: the gradients are zeros
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(4 * kTilesTotal), dim3(256), 0, gf_stream(stream), r);
    return gf_check_launch("field_wgrad16 (reduce)");
}
}  // namespace

GF_EXPORT uint64_t gf_field_wgrad16_ws_bytes(void) { return ws_bytes(false); }
GF_EXPORT uint64_t gf_field_wgrad32_ws_bytes(void) { return ws_bytes(true); }
GF_EXPORT int gf_field_wgrad16(uint32_t M, const gf_field_wgrad_t* w, void* stream) { return wgrad_impl(false, M, w, stream); }
GF_EXPORT int gf_field_wgrad32(uint32_t M, const gf_field_wgrad_t* w, void* stream) { return wgrad_impl(true, M, w, stream); }
