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

// Faithful main loop of the head kernel's MFMA segments: per group 16 MFMAs, 4 ds_read_b128 issued one group ahead, one
// global_load_dwordx4 three groups ahead in a compile-time register FIFO, sched_barrier pinned.  LDSB / L2A switch the operand
// sources off individually.
template <bool LDSB, bool L2A, int g>
__device__ __forceinline__ void probe_step(const char* Ws, uint32_t lane16, const float* Hb, floatx16 (&acc)[4], float4 (&q)[3], const float4 (&b)[4]) {
    if constexpr (g < 48) {
        float4 bn[4];
#pragma unroll
        for (int t = 0; t < 4; t++) bn[t] = LDSB ? *reinterpret_cast<const float4*>(Hb + t * 32 * kHS + 8 * ((g + 1) & 15)) : b[t];
        __builtin_amdgcn_sched_barrier(0);
        const float4 a = q[g % 3];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
        if (L2A) q[g % 3] = *reinterpret_cast<const float4*>(Ws + (size_t)((g + 3) % 48) * 1024 + (size_t)lane16);
        __builtin_amdgcn_sched_barrier(0);
        probe_step<LDSB, L2A, g + 1>(Ws, lane16, Hb, acc, q, bn);
    }
}

template <bool LDSB, bool L2A>
__global__ void __launch_bounds__(256, 2) k_faithful(const float* __restrict__ W, float* __restrict__ out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* H = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 128 * kHS; i += 256) H[i] = seed * (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    floatx16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    const char* Ws = reinterpret_cast<const char*>(W) + (size_t)wave * 78 * 1024;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const unsigned long long r_start = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: ticks / realtime = the true shader clock
    float4 q[3], b[4];
    for (int g = 0; g < 3; g++) q[g] = float4{seed, seed * 0.5f, seed * 0.25f, seed * 0.125f};
    for (int t = 0; t < 4; t++) b[t] = float4{seed, -seed, seed * 0.3f, seed * 0.7f};
    for (int it = 0; it < iters; it++) {
        uint32_t lane16 = lane * 16u;
        uint32_t hoff = (uint32_t)((lane & 31) * kHS + 4 * (lane >> 5));
        asm volatile("" : "+v"(lane16), "+v"(hoff));   // per iteration: nothing is loop invariant
        probe_step<LDSB, L2A, 0>(Ws, lane16, H + hoff, acc, q, b);
    }
    float s = 0.0f;
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) {   // shader ticks this workgroup lived: ticks / wall time = what s_memtime counts
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        reinterpret_cast<unsigned long long*>(out + 512 * 256)[0] = t_end - t_start;
        reinterpret_cast<unsigned long long*>(out + 512 * 256)[2] = __builtin_amdgcn_s_memrealtime() - r_start;
    }
}

template <bool LDSB, bool L2A>
double run_faithful(int grid, const float* W, float* out, int iters, float seed, int reps) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_faithful<LDSB, L2A>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * kHS * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_faithful<LDSB, L2A>), dim3(grid), dim3(256), 128 * kHS * 4, 0, W, out, iters, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_faithful<LDSB, L2A>), dim3(grid), dim3(256), 128 * kHS * 4, 0, W, out, iters, seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 48 * 16 * 4096.0 * reps;
    unsigned long long ticks = 0, real = 0;
    hipMemcpy(&ticks, out + 512 * 256, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&real, reinterpret_cast<char*>(out + 512 * 256) + 16, 8, hipMemcpyDeviceToHost);
    printf("   [s_memtime: %llu ticks per workgroup lifetime, %.1f us per launch -> %.3f GHz; %.2f ticks per MFMA; ticks / s_memrealtime -> %.3f GHz]\n", ticks,
           ms * 1e3 / reps, (double)ticks / (ms * 1e-3 / reps) / 1e9, (double)ticks / ((double)iters * 48 * 16 * (grid > 256 ? 2 : 1)),
           real ? (double)ticks / (double)real * 0.1 : 0.0);
    return flop / (ms * 1e-3) / 1e12;
}

// Co-residency probe: even workgroups run the faithful MFMA loop, odd workgroups (same CUs, 2 per CU) run a partner loop:
//   PARTNER 0 idle (exit), 1 dense VALU fma chains, 2 random 8-byte gathers from an 8 MB table, 3 LDS read/write traffic,
//   4 VALU + gathers.  Reports the MFMA workgroups' rate and the effective shader clock.
template <int PARTNER>
__global__ void __launch_bounds__(256, 2) k_mixed(const float* __restrict__ W, const float2* __restrict__ table, float* __restrict__ out,
                                                  int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* H = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 128 * kHS; i += 256) H[i] = seed * (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    if (blockIdx.x < 256) {   // blocks b and b + 256 share an XCD (b % 8) and, with round-robin CU fill, a CU
        floatx16 acc[4];
        for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
        const char* Ws = reinterpret_cast<const char*>(W) + (size_t)wave * 78 * 1024;
        float4 q[3], b[4];
        for (int g = 0; g < 3; g++) q[g] = float4{seed, seed * 0.5f, seed * 0.25f, seed * 0.125f};
        for (int t = 0; t < 4; t++) b[t] = float4{seed, -seed, seed * 0.3f, seed * 0.7f};
        for (int it = 0; it < iters; it++) {
            uint32_t lane16 = lane * 16u;
            uint32_t hoff = (uint32_t)((lane & 31) * kHS + 4 * (lane >> 5));
            asm volatile("" : "+v"(lane16), "+v"(hoff));
            probe_step<true, true, 0>(Ws, lane16, H + hoff, acc, q, b);
        }
        float s = 0.0f;
        for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
        out[blockIdx.x * 256 + tid] = s;
        if (blockIdx.x == 0 && tid == 0) reinterpret_cast<unsigned long long*>(out + 512 * 256)[0] = __builtin_amdgcn_s_memtime() - t_start;
    } else {
        if (PARTNER == 0) return;
        float x0 = seed + tid, x1 = seed * 0.5f, x2 = 0.3f, x3 = 0.1f, acc = 0.0f;
        uint32_t idx = tid * 2654435761u + blockIdx.x;
        const int n = PARTNER == 1 ? iters * 48 * 6 : (PARTNER == 3 ? iters * 48 * 3 : iters * 16);   // tuned to outlive the MFMA workgroups slightly
        for (int i = 0; i < n; i++) {
            if (PARTNER == 1 || PARTNER == 4) {
#pragma unroll
                for (int k = 0; k < 16; k++) { x0 = __builtin_fmaf(x0, 1.0001f, x1); x1 = __builtin_fmaf(x1, 0.9999f, x2); x2 = __builtin_fmaf(x2, 1.0002f, x3); x3 = __builtin_fmaf(x3, 0.9998f, x0); }
            }
            if (PARTNER == 2 || PARTNER == 4) {
#pragma unroll
                for (int k = 0; k < 4; k++) { idx = idx * 1664525u + 1013904223u; const float2 v = table[idx & ((1u << 20) - 1u)]; acc += v.x + v.y; }
            }
            if (PARTNER == 3) {
#pragma unroll
                for (int k = 0; k < 4; k++) { const float4 v = *reinterpret_cast<const float4*>(H + ((tid * 4 + k * 1024 + i) & 8191)); acc += v.x; }
                H[(tid + i) & 8191] = acc;
            }
        }
        out[blockIdx.x * 256 + tid] = x0 + x1 + x2 + x3 + acc;
        if (blockIdx.x == 256 && tid == 0) reinterpret_cast<unsigned long long*>(out + 512 * 256)[1] = __builtin_amdgcn_s_memtime() - t_start;
    }
}

template <int PARTNER>
void run_mixed(const float* W, const float2* table, float* out, int iters, int reps) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_mixed<PARTNER>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * kHS * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mixed<PARTNER>, dim3(512), dim3(256), 128 * kHS * 4, 0, W, table, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_mixed<PARTNER>, dim3(512), dim3(256), 128 * kHS * 4, 0, W, table, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ticks[2] = {0, 0};
    hipMemcpy(ticks, out + 512 * 256, 16, hipMemcpyDeviceToHost);
    const double mfma = (double)iters * 48 * 16;
    const unsigned long long span = ticks[0] > ticks[1] ? ticks[0] : ticks[1];
    printf("partner %d: launch %.1f us; MFMA workgroup %llu ticks = %.2f ticks/MFMA, partner workgroup %llu ticks; clock >= %.3f GHz\n", PARTNER,
           ms * 1e3 / reps, ticks[0], (double)ticks[0] / mfma, ticks[1], (double)span / (ms * 1e-3 / reps) / 1e9);
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
    hipMalloc(&W, wbytes); hipMalloc(&out, 512 * 256 * 4 + 64); hipMemset(out, 0, 512 * 256 * 4 + 64);
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
    const int fit = 133;   // 133 * 48 groups ~ 400 * 16
    printf("faithful regs only               2wg/CU: %7.1f TFLOP/s\n", (run_faithful<false, false>(512, W, out, fit, 1.0f, reps)));
    printf("faithful LDS B                   2wg/CU: %7.1f TFLOP/s\n", (run_faithful<true, false>(512, W, out, fit, 1.0f, reps)));
    printf("faithful L2 A                    2wg/CU: %7.1f TFLOP/s\n", (run_faithful<false, true>(512, W, out, fit, 1.0f, reps)));
    printf("faithful LDS B + L2 A            2wg/CU: %7.1f TFLOP/s\n", (run_faithful<true, true>(512, W, out, fit, 1.0f, reps)));
    printf("faithful LDS B + L2 A            1wg/CU: %7.1f TFLOP/s\n", (run_faithful<true, true>(256, W, out, fit, 1.0f, reps)));
    float2* table;
    hipMalloc(&table, (size_t)8 << 20);
    hipMemset(table, 0, (size_t)8 << 20);
    run_mixed<0>(W, table, out, fit, 10);
    run_mixed<1>(W, table, out, fit, 10);
    run_mixed<2>(W, table, out, fit, 10);
    run_mixed<3>(W, table, out, fit, 10);
    run_mixed<4>(W, table, out, fit, 10);
    return 0;
}
