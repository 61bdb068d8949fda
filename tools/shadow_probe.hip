// Probe for the 1-wave-per-SIMD design point of k_head_phase: how much single-issue VALU work does ONE wave hide in the
// shadow of its own MFMA stream, against what two co-resident waves achieve that each alternate an MFMA segment with a VALU segment (the
// product's structure: two workgroups per CU, barrier-delimited segments).
//
//   arm I  (interleaved, 1 wave/SIMD):   per MFMA, K independent v_fma_f32 placed behind it with sched_group_barrier
//   arm S1 (segments,    1 wave/SIMD):   16 MFMAs, then 16 K v_fma_f32, no overlap possible inside the wave
//   arm S2 (segments,    2 waves/SIMD):  the same stream on two free-running waves per SIMD (they overlap by drifting apart)
//   arm I2 (interleaved, 2 waves/SIMD):  arm I twice per SIMD
// Reported: shader cycles per MFMA per SIMD (s_memtime over the whole loop, median over waves), i.e. the matrix pipe's period; 32 (f16
// 32x32x16) / 64 (f32 32x32x2) is the pipe-bound floor.  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/shadow_probe.hip -o tools/shadow_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int F16>
__device__ __forceinline__ floatx16 mfma(const half8& ah, const half8& bh, float af, float bf, const floatx16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c, 0, 0, 0);
}

// K fillers per MFMA, 16 MFMAs per trip on 8 accumulators (an accumulator is reused after 8 MFMAs: no dependent-issue stall)
template <int F16, int K, int INTERLEAVE, int LDSR>
__global__ void __launch_bounds__(512) probe(float* out, int trips, unsigned long long* ticks, const float* seed) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    floatx16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = seed[(i + r) & 15];
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = seed[i] + (float)lane;
    half8 ah, bh;
#pragma unroll
    for (int i = 0; i < 8; i++) { ah[i] = (_Float16)seed[i]; bh[i] = (_Float16)seed[8 + i]; }
    float af = seed[3], bf = seed[5];
    const float c1 = seed[1], c2 = seed[2];
    float4 ld = {0, 0, 0, 0};
    const float4* lp = reinterpret_cast<const float4*>(lds) + lane;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; t++) {
        if constexpr (INTERLEAVE) {
#pragma unroll
            for (int m = 0; m < 16; m++) {
                acc[m & 7] = mfma<F16>(ah, bh, af, bf, acc[m & 7]);
#pragma unroll
                for (int k = 0; k < K; k++) v[(m * K + k) & 15] = __builtin_fmaf(v[(m * K + k) & 15], c1, c2);
                if constexpr (LDSR) { if ((m & 1) == 0) { const float4 x = lp[64 * (m >> 1)]; ld.x += x.x; ld.y += x.y; ld.z += x.z; ld.w += x.w; } }
            }
#pragma unroll
            for (int m = 0; m < 16; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (K > 0) __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) acc[m & 7] = mfma<F16>(ah, bh, af, bf, acc[m & 7]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 16; m++) {
#pragma unroll
                for (int k = 0; k < K; k++) v[(m * K + k) & 15] = __builtin_fmaf(v[(m * K + k) & 15], c1, c2);
                if constexpr (LDSR) { if ((m & 1) == 0) { const float4 x = lp[64 * (m >> 1)]; ld.x += x.x; ld.y += x.y; ld.z += x.z; ld.w += x.w; } }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = ld.x + ld.y + ld.z + ld.w;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][lane & 15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int F16, int K, int INTERLEAVE, int LDSR>
double run(int waves_per_simd, float* out, unsigned long long* ticks, const float* seed) {
    const int threads = 256 * waves_per_simd, blocks = 256, trips = 2000;
    const size_t shmem = 100 * 1024;   // one workgroup per CU
    auto k = probe<F16, K, INTERLEAVE, LDSR>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), shmem, 0, out, 50, ticks, seed);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), shmem, 0, out, trips, ticks, seed);
    hipDeviceSynchronize();
    const int nw = blocks * threads / 64;
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), ticks, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // pipe period per SIMD: a wave's loop time / its MFMAs, divided by the waves that share the SIMD's pipe
    return (double)h[nw / 2] / (trips * 16.0) / waves_per_simd;
}

template <int F16, int LDSR>
void sweep(float* out, unsigned long long* ticks, const float* seed) {
    printf("%s MFMA%s: cycles of matrix pipe per MFMA and SIMD (floor %d)\n", F16 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_32x32x2_f32",
           LDSR ? " + one ds_read_b128 per 2 MFMAs" : "", F16 ? 32 : 64);
    printf("  K fillers/MFMA |  I: 1 wave interleaved |  S1: 1 wave segments |  S2: 2 waves segments |  I2: 2 waves interleaved\n");
#define ROW(K)                                                                                                                       \
    printf("  %14d | %22.1f | %20.1f | %21.1f | %24.1f\n", K, run<F16, K, 1, LDSR>(1, out, ticks, seed), run<F16, K, 0, LDSR>(1, out, ticks, seed), \
           run<F16, K, 0, LDSR>(2, out, ticks, seed), run<F16, K, 1, LDSR>(2, out, ticks, seed));
    ROW(0) ROW(2) ROW(4) ROW(5) ROW(6) ROW(8) ROW(10) ROW(12) ROW(16)
#undef ROW
}

int main() {
    float *out, *seed;
    unsigned long long* ticks;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&ticks, 256 * 8 * 8);
    hipMalloc(&seed, 64);
    float hs[16];
    for (int i = 0; i < 16; i++) hs[i] = 0.25f + 0.03125f * i;
    hs[1] = 0.999f; hs[2] = 0.001f;
    hipMemcpy(seed, hs, 64, hipMemcpyHostToDevice);
    sweep<1, 0>(out, ticks, seed);
    sweep<1, 1>(out, ticks, seed);
    sweep<0, 0>(out, ticks, seed);
    sweep<0, 1>(out, ticks, seed);
    return 0;
}
