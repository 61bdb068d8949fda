// Does the SHAPE of the load -- matrix-pipe bursts between barriers, with gather / VALU / LDS pieces in between -- hold the shader clock
// below what a steady MFMA stream gets?  (NOTES.md 4.2: k_head_phase runs at 2.13 GHz, tools/mfma_probe.hip at 2.40 GHz.)
// Two workgroups per CU (68 KB of LDS each), free running.  Per iteration: an MFMA burst of `groups` x 16 v_mfma_f32_32x32x2_f32, a barrier,
// a filler piece of `fill` steps, a barrier.  Clock = s_memtime ticks / s_memrealtime (100 MHz) over the workgroup's life.
//   filler 0 none   1 VALU transcendentals   2 LDS read/write   3 random 8-byte gathers (8 MB table)   4 s_sleep   5 gathers + transcendentals
//   hipcc -O3 --offload-arch=gfx950 tools/burst_probe.hip -o tools/burst_probe.bin && tools/burst_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int FILL>
__global__ void __launch_bounds__(256) k_burst(float* out, const float2* table, int iters, int groups, int fill, float seed) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 17000; i += 256) lds[i] = seed * (float)i;
    __syncthreads();
    floatx16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    float a = seed + (float)lane, b = seed * 0.5f - (float)lane, v = seed;
    uint32_t h = (uint32_t)tid * 2654435761u + blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        for (int g = 0; g < groups; g++) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            asm volatile("" : "+v"(a), "+v"(b));
        }
        __syncthreads();
        if (FILL == 1 || FILL == 5) {
            for (int f = 0; f < fill; f++) { v = __expf(v * 0.999f) * 0.5f + __sinf(v); v = v * 0.25f + 0.1f; }
        }
        if (FILL == 2) {
            for (int f = 0; f < fill; f++) { const int i = (tid * 4 + f * 1024) % 16384; float4 x = *reinterpret_cast<float4*>(lds + i); x.x += v; *reinterpret_cast<float4*>(lds + i) = x; v += x.y * 1e-9f; }
        }
        if (FILL == 3 || FILL == 5) {
            for (int f = 0; f < fill; f += 8) {
                float2 s[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { h = h * 1664525u + 1013904223u; s[j] = table[h >> 12]; }
#pragma unroll
                for (int j = 0; j < 8; j++) v += s[j].x * 1e-9f + s[j].y * 1e-9f;
            }
        }
        if (FILL == 4) { for (int f = 0; f < fill; f++) __builtin_amdgcn_s_sleep(8); }
        __syncthreads();
    }
    float s = v;
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(out + 512 * 256) + (blockIdx.x & 1) * 2;
        if (blockIdx.x < 2) { o[0] = __builtin_amdgcn_s_memtime() - t0; o[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    }
}

template <int FILL>
void run(const char* name, float* out, const float2* table, int iters, int groups, int fill) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_burst<FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_burst<FILL>, dim3(512), dim3(256), 68 * 1024, 0, out, table, iters, groups, fill, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 4;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_burst<FILL>, dim3(512), dim3(256), 68 * 1024, 0, out, table, iters, groups, fill, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long o[4];
    (void)hipMemcpy(o, out + 512 * 256, 32, hipMemcpyDeviceToHost);
    const double flop = 512.0 * 4 * iters * groups * 16 * 4096.0 * reps;
    const double mfma_ticks = (double)iters * groups * 16 * 64 * 2;   // two workgroups share each SIMD's pipe
    printf("%-44s groups %3d fill %5d: %7.1f TFLOP/s  clock %.3f GHz  (ticks per iteration %.0f, matrix pipe busy %.0f %%)\n", name, groups, fill,
           flop / (ms * 1e-3) / 1e12, (double)o[0] / (double)o[1] * 0.1, (double)o[0] / iters, 100.0 * mfma_ticks / (double)o[0]);
}

int main() {
    float* out; float2* table;
    (void)hipMalloc(&out, 512 * 256 * 4 + 64); (void)hipMalloc(&table, (size_t)8 << 20);
    (void)hipMemset(table, 0, (size_t)8 << 20);
    const int it = 400;
    run<0>("steady MFMA (barriers only)", out, table, it, 16, 0);
    run<4>("MFMA bursts + sleep", out, table, it, 16, 40);
    run<1>("MFMA bursts + VALU transcendentals", out, table, it, 16, 600);
    run<2>("MFMA bursts + LDS traffic", out, table, it, 16, 600);
    run<3>("MFMA bursts + gathers", out, table, it, 16, 256);
    run<5>("MFMA bursts + gathers + transcendentals", out, table, it, 16, 256);
    run<5>("short bursts + gathers + transcendentals", out, table, it * 4, 4, 64);
    run<5>("long bursts + gathers + transcendentals", out, table, it / 2, 32, 512);
    run<0>("steady MFMA again", out, table, it, 16, 0);
    return 0;
}
