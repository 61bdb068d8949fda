// Probe: is v_mfma_f32_32x32x16_f16 position-invariant?  Identical B columns (and identical accumulator inputs) must give bitwise
// identical C columns; chained accumulation over several K groups included.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16* A, const _Float16* Bcol, float* C, int groups) {   // A [groups][32][16], Bcol [groups][16]: one column, replicated
    const int l = threadIdx.x, r0 = l & 31, h = l >> 5;
    floatx16 c;
    for (int r = 0; r < 16; r++) c[r] = 0.125f * (float)((r & 3) + 8 * (r >> 2) + 4 * h);      // depends on the row only
    for (int g = 0; g < groups; g++) {
        half8 a, b;
        for (int i = 0; i < 8; i++) { a[i] = A[(g * 32 + r0) * 16 + 8 * h + i]; b[i] = Bcol[g * 16 + 8 * h + i]; }
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r0] = c[r];
}

int main() {
    const int groups = 8;
    std::vector<_Float16> A(groups * 32 * 16), B(groups * 16);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 9) % 2001 - 1000) / 997.0f; };
    for (auto& v : A) v = (_Float16)(rnd() * 0.37f);
    for (auto& v : B) v = (_Float16)(rnd() * 1.91f);
    _Float16 *dA, *dB; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, groups);
    std::vector<float> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; m++) for (int n = 1; n < 32; n++) if (memcmp(&C[m * 32 + n], &C[m * 32], 4) != 0) bad++;
    printf("mfma_f32_32x32x16_f16 column invariance: %d of %d entries differ from column 0 (%s)\n", bad, 32 * 31, bad ? "POSITION DEPENDENT" : "invariant");
    return bad != 0;
}
