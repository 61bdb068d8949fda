// Probe: operand / accumulator layout of v_mfma_f32_32x32x16_f16 on gfx950, checked against a host matmul.
// Assumed (and used by frame_head.hip's fast path): A lane l = row l%32, k = 8*(l/32)+0..7; B lane l = col l%32, same k;
// C register r of lane l = row (r%4) + 8*(r/4) + 4*(l/32), col l%32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16* A, const _Float16* B, float* C) {   // A [32][16], B [16][32], C [32][32]
    const int l = threadIdx.x, r0 = l & 31, h = l >> 5;
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = A[r0 * 16 + 8 * h + i]; b[i] = B[(8 * h + i) * 32 + r0]; }
    floatx16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r0] = c[r];
}

int main() {
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    for (int i = 0; i < 512; i++) { A[i] = (_Float16)((float)((i * 37) % 17 - 8) / 8.0f); B[i] = (_Float16)((float)((i * 53) % 13 - 6) / 4.0f); }
    _Float16 *dA, *dB; float* dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double err = 0;
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) {
        double s = 0;
        for (int kk = 0; kk < 16; kk++) s += (double)(float)A[m * 16 + kk] * (double)(float)B[kk * 32 + n];
        err = fmax(err, fabs(s - C[m * 32 + n]));
    }
    printf("mfma_f32_32x32x16_f16 layout probe: max |err| = %g (%s)\n", err, err < 1e-4 ? "layout as assumed" : "LAYOUT MISMATCH");
    return err < 1e-4 ? 0 : 1;
}
