// Probe: what bounds the hash-grid gathers of the head kernels -- the per-CU rate of DIVERGENT 8-byte loads (one table row of two floats per
// lane and instruction, 64 different cache lines per wave instruction) against the same rows fetched as 16-byte pairs, and whether a load
// instruction with half of its lanes masked off costs half.
// hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe.bin
//   mode 0: 8 x  8-byte loads per lane and iteration, all lanes            (today's lookup: 8 corners of one 3-D level)
//   mode 1: 4 x 16-byte loads per lane and iteration, all lanes            (x-corner pairs in one load; same bytes as mode 0)
//   mode 2: 4 x 16-byte loads (all lanes) + 4 x 8-byte loads (every other lane, by a hashed predicate)   (pairs where the hash allows)
//   mode 3: 8 x  8-byte loads, half of the lanes active                    (does cost follow the active-lane count?)
//   mode 4: 4 x 16-byte loads at 8-byte (not 16-byte) aligned addresses    (dense levels: the pair starts at any row)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const float2* __restrict__ table, uint32_t mask, uint32_t iters, float* out) {
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t idx[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            h = h * 1664525u + 1013904223u;
            idx[c] = (h >> 7) & mask;
        }
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 8; c++) { const float2 v = table[idx[c]]; acc += v.x + v.y; }
        } else if (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t i = MODE == 1 ? (idx[c] & ~1u) : (idx[c] | 1u) & (mask - 1u);
                const float4 v = *reinterpret_cast<const float4*>(table + i);
                acc += v.x + v.y + v.z + v.w;
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float4 v = *reinterpret_cast<const float4*>(table + (idx[c] & ~1u));
                acc += v.x + v.y + v.z + v.w;
            }
#pragma unroll
            for (int c = 4; c < 8; c++)
                if (idx[c - 4] & 4u) { const float2 v = table[idx[c]]; acc += v.x + v.y; }
        } else if (MODE == 3) {
#pragma unroll
            for (int c = 0; c < 8; c++)
                if (idx[c & 3] & 4u) { const float2 v = table[idx[c]]; acc += v.x + v.y; }
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int MODE>
static void run(const float2* table, uint32_t rows, float* out, const char* what) {
    const uint32_t blocks = 512, iters = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, table, rows - 1, 16u, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, table, rows - 1, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // rows fetched per lane and iteration: 8, but 4 on average in mode 3 and 8 + 2 in mode 2
    const double rows_fetched = (double)blocks * 256 * iters * (MODE == 3 ? 4 : MODE == 2 ? 10 : 8);
    printf("  mode %d (%-46s): %7.3f ms  %6.2f G rows/s  %5.2f rows per clock and CU at 2.1 GHz\n", MODE, what, ms, rows_fetched / ms * 1e-6,
           rows_fetched / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 256 * 4);
    for (uint32_t rows : {1u << 12, 1u << 16, 1u << 20, 1u << 24}) {   // 32 KB (L1-sized), 512 KB (one hashed level), 8 MB (the table), 128 MB
        float2* table;
        hipMalloc(&table, (size_t)rows * 8 + 64);
        hipMemset(table, 0, (size_t)rows * 8 + 64);
        printf("table %u rows (%.1f KB)\n", rows, rows * 8 / 1024.0);
        run<0>(table, rows, out, "8 x 8 B, all lanes");
        run<1>(table, rows, out, "4 x 16 B aligned, all lanes");
        run<4>(table, rows, out, "4 x 16 B at odd rows, all lanes");
        run<2>(table, rows, out, "4 x 16 B all lanes + 4 x 8 B half the lanes");
        run<3>(table, rows, out, "8 x 8 B, half the lanes");
        hipFree(table);
    }
    return 0;
}
