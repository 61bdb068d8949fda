// Probe: float atomic-add throughput into a small table, (a) one table shared by all XCDs, (b) one private copy per XCD.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_probe.hip -o tools/atomic_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void __launch_bounds__(256) k(float* table, uint32_t rows, uint32_t per_thread, int per_xcd, uint32_t* xcc_hist) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xF;
    if (threadIdx.x == 0) atomicAdd(&xcc_hist[xcc], 1u);
    float* t = table + (per_xcd ? (size_t)xcc * rows : 0);
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t i = 0; i < per_thread; i++) {
        h = h * 1664525u + 1013904223u;
        // neighbouring lanes hit neighbouring rows most of the time, like samples along a ray
        const uint32_t row = ((h >> 8) + threadIdx.x) % rows;
        unsafeAtomicAdd(t + row, 1.0f);
    }
}

int main() {
    const uint32_t per_thread = 256, blocks = 4096;
    uint32_t* hist;
    hipMalloc(&hist, 64);
    for (uint32_t rows : {4913u * 2, 65536u * 2, 1u << 20}) {
        float* table;
        hipMalloc(&table, (size_t)rows * 8 * sizeof(float));
        for (int per_xcd = 0; per_xcd < 2; per_xcd++) {
            hipMemset(table, 0, (size_t)rows * 8 * sizeof(float));
            hipMemset(hist, 0, 64);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, rows, 8u, per_xcd, hist);   // warm-up
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, rows, per_thread, per_xcd, hist);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<float> h((size_t)rows * 8);
            hipMemcpy(h.data(), table, h.size() * 4, hipMemcpyDeviceToHost);
            double sum = 0;
            for (float v : h) sum += v;
            uint32_t hh[16];
            hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost);
            const double n = (double)blocks * 256 * (per_thread + 8);
            printf("rows %8u per_xcd %d: %8.3f ms  %7.2f G atomics/s  sum %.0f (expect %.0f)  xcc hist %u %u %u %u %u %u %u %u\n", rows, per_xcd, ms,
                   (double)blocks * 256 * per_thread / ms * 1e-6, sum, n, hh[0], hh[1], hh[2], hh[3], hh[4], hh[5], hh[6], hh[7]);
        }
        hipFree(table);
    }
    return 0;
}
