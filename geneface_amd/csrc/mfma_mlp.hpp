// f32 MFMA building blocks for the small fused MLPs (head: 128-wide, torso: 64/32-wide) on gfx950.
//
// A wave owns a tile of 32 samples (the 32 columns of v_mfma_f32_32x32x2_f32).  A layer with NOB*32 outputs is NOB
// accumulator blocks of 32x32; weights are the A operand, streamed from LDS as [out_block][step/4][lane][step%4]
// (one ds_read_b128 feeds four MFMAs), the previous layer's activations are the B operand.  The accumulator layout
// (lane half h, register r  <->  feature row (r&3) + 8*(r>>2) + 4*h, column lane&31) IS a valid B-operand layout
// for the next layer if step (block, r) is defined to consume the feature pair (row(r,0), row(r,1)), so activations
// never leave registers between layers.  f32 MFMA is an exact k-ordered fmaf chain (no reduced-precision path).
#pragma once
#include "common.hpp"

namespace gf {

using floatx16 = __attribute__((ext_vector_type(16))) float;

// Out-blocks [OB0, OB0+NOBP) of a layer whose accumulator array has NOB blocks.  Wl points at the LDS stream of exactly these
// NOBP blocks ([block][step/4][lane][step%4]).  bin[t] = this lane's B operand for step t.  bias = LDS vector of the WHOLE layer
// in accumulator-layout order [ob][half][16], or nullptr.  ACCUM continues accumulating into out[] (used when a layer's K
// range is split over two weight streams).
template <int NOB, int OB0, int NOBP, int NSTEPS, bool RELU, bool ACCUM>
__device__ __forceinline__ void mfma_part(const float* __restrict__ Wl, int lane, const float (&bin)[NSTEPS],
                                          const float* __restrict__ bias, floatx16 (&out)[NOB]) {
    static_assert(NSTEPS % 4 == 0, "steps come in groups of four (one 16-byte LDS read)");
    static_assert(OB0 + NOBP <= NOB, "block range");
    const float4* W4 = reinterpret_cast<const float4*>(Wl);
    const int half = lane >> 5;
    floatx16 acc[NOBP];
#pragma unroll
    for (int o = 0; o < NOBP; o++) {
        const int ob = OB0 + o;
        if (ACCUM) {
            acc[o] = out[ob];
        } else if (bias) {
            const float4* b4 = reinterpret_cast<const float4*>(bias + ob * 32 + half * 16);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 b = b4[q];
                acc[o][q * 4 + 0] = b.x; acc[o][q * 4 + 1] = b.y; acc[o][q * 4 + 2] = b.z; acc[o][q * 4 + 3] = b.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) acc[o][r] = 0.0f;
        }
    }
    // The out-blocks advance TOGETHER, one group of four steps each in turn (round 5; they used to run one after the other): consecutive MFMAs
    // then belong to different accumulators wherever a layer has two blocks, so none waits for the result of the one issued just before it
    // (tools/trace_torso.py: a 64-MFMA layer took 88 cycles per MFMA as one dependent chain per block).  Each accumulator still sees its own
    // steps in the same order: same bits.
#pragma unroll
    for (int t4 = 0; t4 < NSTEPS / 4; t4++) {
#pragma unroll
        for (int o = 0; o < NOBP; o++) {
            const float4 w = W4[(o * (NSTEPS / 4) + t4) * 64 + lane];
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, bin[t4 * 4 + 0], acc[o], 0, 0, 0);
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, bin[t4 * 4 + 1], acc[o], 0, 0, 0);
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, bin[t4 * 4 + 2], acc[o], 0, 0, 0);
            acc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, bin[t4 * 4 + 3], acc[o], 0, 0, 0);
        }
    }
#pragma unroll
    for (int o = 0; o < NOBP; o++) {
        if (RELU) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc[o][r] = fmaxf(acc[o][r], 0.0f);
        }
        out[OB0 + o] = acc[o];
    }
}

// whole layer from one stream
template <int NOB, int NSTEPS, bool RELU, bool ACCUM>
__device__ __forceinline__ void mfma_layer(const float* __restrict__ Wl, int lane, const float (&bin)[NSTEPS],
                                           const float* __restrict__ bias, floatx16 (&out)[NOB]) {
    mfma_part<NOB, 0, NOB, NSTEPS, RELU, ACCUM>(Wl, lane, bin, bias, out);
}

// Asynchronous L2/HBM -> LDS copy of `nfloats` (multiple of 256) contiguous floats by the whole 256-thread workgroup:
// each wave-instruction moves 1 KiB (64 lanes x 16 B) without touching VGPRs; completion = this wave's vmcnt reaching 0
// (a following __syncthreads() waits for it).  Destination is lane-linear, which is exactly the stream layout above.
__device__ __forceinline__ void dma_to_lds(float* lds_dst, const float* __restrict__ gsrc, int nfloats, int wave, int lane) {
    using gptr_t = __attribute__((address_space(1))) void*;
    using lptr_t = __attribute__((address_space(3))) void*;
    const int n_inst = nfloats >> 8;
    for (int k = wave; k < n_inst; k += 4) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + (size_t)k * 256 + lane * 4), (lptr_t)(lds_dst + k * 256), 16, 0, 0);
    }
}

template <int NOB>
__device__ __forceinline__ void unpack(const floatx16 (&v)[NOB], float (&a)[NOB * 16]) {
#pragma unroll
    for (int ob = 0; ob < NOB; ob++)
#pragma unroll
        for (int r = 0; r < 16; r++) a[ob * 16 + r] = v[ob][r];
}

// NOUT skinny outputs on the VALU over NOB*32 input features: rows are stored in accumulator-layout order so a lane reads
// its weights contiguously; the two lane halves hold complementary features and are summed with one cross-half exchange.
template <int NOUT, int NOB>
__device__ __forceinline__ void valu_rows(const float* __restrict__ rows, int half, const float (&act)[NOB * 16], float (&res)[NOUT]) {
#pragma unroll
    for (int c = 0; c < NOUT; c++) {
        float sum = 0.0f;
#pragma unroll
        for (int ob = 0; ob < NOB; ob++) {
            const float4* w4 = reinterpret_cast<const float4*>(rows + c * (NOB * 32) + ob * 32 + half * 16);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 w = w4[q];
                sum = __builtin_fmaf(w.x, act[ob * 16 + q * 4 + 0], sum);
                sum = __builtin_fmaf(w.y, act[ob * 16 + q * 4 + 1], sum);
                sum = __builtin_fmaf(w.z, act[ob * 16 + q * 4 + 2], sum);
                sum = __builtin_fmaf(w.w, act[ob * 16 + q * 4 + 3], sum);
            }
        }
        res[c] = sum + __shfl_xor(sum, 32);
    }
}

}  // namespace gf
