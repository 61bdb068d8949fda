// LDS atomic-add throughput on gfx950 by operand type: what bounds the table-partitioned grid backward (encoders.hip, k_grid_backward:
// 268 M ds_add_f32 per 1 M-point batch in 1.3 ms = 0.4 adds per clock and CU).  1024-lane workgroups, 128 KiB of LDS, pseudo-random
// addresses (like hashed / tiled fine levels), 1 workgroup per CU.
//     hipcc --offload-arch=gfx950 -O2 tools/lds_atomic_probe.hip -o tools/lds_atomic_probe.bin && tools/lds_atomic_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int kT = 1024, kWords = 32768;

template <int MODE>   // 0 f32 add, 1 u32 add, 2 u64 add (half as many slots), 3 plain read-modify-write (racy, bandwidth reference)
__global__ void __launch_bounds__(kT) k(uint32_t iters, float* out) {
    __shared__ uint32_t tab[kWords];
    for (int i = threadIdx.x; i < kWords; i += kT) tab[i] = 0;
    __syncthreads();
    uint32_t x = (blockIdx.x * kT + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t a = (x >> 9) & (kWords - 1);
            if (MODE == 0) atomicAdd(reinterpret_cast<float*>(&tab[a]), 1.0f);
            else if (MODE == 1) atomicAdd(&tab[a], 3u);
            else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(&tab[a & ~1u]), 3ull);
            else tab[a] += 3u;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)tab[7];
}

template <int MODE>
static void run(const char* name, float* d_out) {
    const uint32_t iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(kT), 0, 0, 10u, d_out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(kT), 0, 0, iters, d_out);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double adds = 256.0 * kT * iters * 16;
    printf("%-28s %8.3f ms  %7.1f G adds/s  %5.2f adds per clock and CU (2.1 GHz)\n", name, ms, adds / ms / 1e6, adds / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
    float* d_out;
    (void)hipMalloc(&d_out, 256 * 4);
    run<0>("ds_add_f32", d_out);
    run<1>("ds_add_u32", d_out);
    run<2>("ds_add_u64", d_out);
    run<3>("read + add + write (racy)", d_out);
    return 0;
}
