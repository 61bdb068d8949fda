// Stand-alone probe (no torch): what f32-MFMA rate does gfx950 sustain for the head kernel's instruction mix?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o gpurun_out/mfma_probe && gpurun_out/mfma_probe
// Variants (all: 512 workgroups x 256 threads, 2 per CU, 4 accumulators of v_mfma_f32_32x32x2_f32 per wave):
//   0 registers only            1 + one ds_read_b128 per tile per 4 MFMAs (B operand from LDS)
//   2 + one 1-KiB global_load_dwordx4 per 16 MFMAs (A operand from L2)        3 = 2 with one workgroup per CU
//   4 = 2 on zero-filled data (data-dependent power)                          5 = 0 with one workgroup per CU
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using floatx16 = __attribute__((ext_vector_type(16))) float;
constexpr int kHS = 132;

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_probe(const float* __restrict__ W, float* __restrict__ out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* H = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 128 * kHS; i += 256) H[i] = seed * (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    floatx16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    const float* Hb = H + (lane & 31) * kHS + 4 * (lane >> 5);
    const char* Ws = reinterpret_cast<const char*>(W) + (size_t)wave * 78 * 1024;
    uint32_t lane16 = lane * 16u;
    asm volatile("" : "+v"(lane16));
    float4 a = {seed, seed * 0.5f, seed * 0.25f, seed * 0.125f};
    float4 b[4];
    for (int t = 0; t < 4; t++) b[t] = float4{seed, -seed, seed * 0.3f, seed * 0.7f};
    float4 q[3] = {a, a, a};
    float4 dummy = {0, 0, 0, 0};
    if (MODE == 6) for (int g = 0; g < 3; g++) q[g] = *reinterpret_cast<const float4*>(Ws + (size_t)g * 1024 + (size_t)lane16);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            if (MODE >= 2 && MODE <= 5) a = *reinterpret_cast<const float4*>(Ws + (size_t)((it * 16 + g) % 78) * 1024 + (size_t)lane16);
            if (MODE == 6) {   // FIFO three groups ahead, pinned like the head kernel
                a = q[(it * 16 + g) % 3];
            }
            if (MODE == 7) {   // the load is issued but nothing waits for it until the end
                const float4 v = *reinterpret_cast<const float4*>(Ws + (size_t)((it * 16 + g) % 78) * 1024 + (size_t)lane16);
                dummy.x += v.x;
            }
            if (MODE >= 1) {
#pragma unroll
                for (int t = 0; t < 4; t++) b[t] = *reinterpret_cast<const float4*>(Hb + t * 32 * kHS + 8 * (g & 15));
            }
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
            if (MODE == 6) {
                q[(it * 16 + g) % 3] = *reinterpret_cast<const float4*>(Ws + (size_t)((it * 16 + g + 3) % 78) * 1024 + (size_t)lane16);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    acc[0][0] += dummy.x;
    float s = 0.0f;
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
double run(int grid, const float* W, float* out, int iters, float seed, int reps) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * kHS * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 128 * kHS * 4, 0, W, out, iters, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 128 * kHS * 4, 0, W, out, iters, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 /*waves*/ * iters * 16 * 16 /*mfma per group*/ * 4096.0 * reps;
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    float *W, *out;
    const size_t wbytes = 4 * 78 * 1024;
    hipMalloc(&W, wbytes); hipMalloc(&out, 512 * 256 * 4);
    std::vector<float> h(wbytes / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2246822519u) >> 12) * 1e-6f - 0.5f;
    hipMemcpy(W, h.data(), wbytes, hipMemcpyHostToDevice);
    const int iters = 400, reps = 20;   // ~1.3 ms per launch at peak, like one head phase
    printf("mode0 regs only            2wg/CU: %7.1f TFLOP/s\n", run<0>(512, W, out, iters, 1.0f, reps));
    printf("mode1 +LDS B operand       2wg/CU: %7.1f TFLOP/s\n", run<1>(512, W, out, iters, 1.0f, reps));
    printf("mode2 +L2 A operand        2wg/CU: %7.1f TFLOP/s\n", run<2>(512, W, out, iters, 1.0f, reps));
    printf("mode2 +L2 A operand        1wg/CU: %7.1f TFLOP/s\n", run<2>(256, W, out, iters, 1.0f, reps));
    hipMemset(W, 0, wbytes);
    printf("mode2 zero data            2wg/CU: %7.1f TFLOP/s\n", run<2>(512, W, out, iters, 0.0f, reps));
    printf("mode0 regs only            1wg/CU: %7.1f TFLOP/s\n", run<0>(256, W, out, iters, 1.0f, reps));
    hipMemcpy(W, h.data(), wbytes, hipMemcpyHostToDevice);
    printf("mode6 L2 A operand, FIFO 3 ahead 2wg/CU: %7.1f TFLOP/s\n", run<6>(512, W, out, iters, 1.0f, reps));
    printf("mode6 L2 A operand, FIFO 3 ahead 1wg/CU: %7.1f TFLOP/s\n", run<6>(256, W, out, iters, 1.0f, reps));
    printf("mode7 loads issued, never waited 2wg/CU: %7.1f TFLOP/s\n", run<7>(512, W, out, iters, 1.0f, reps));
    printf("mode1 +LDS B again               2wg/CU: %7.1f TFLOP/s\n", run<1>(512, W, out, iters, 1.0f, reps));
    return 0;
}
