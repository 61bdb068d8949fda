// Does v_mfma_f32_32x32x16_f16 honour f16 DENORMAL inputs on gfx950, and does v_cvt_f16_f32 produce them?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_denorm_probe.hip -o tools/mfma_denorm_probe.bin && tools/mfma_denorm_probe.bin
// A = 2^-20 (an f16 denormal) everywhere, B = 1: every output element is 16 * 2^-20 = 2^-16 if denormals are honoured, 0 if they are flushed.
// Decides nothing in the product (field_round_split keeps denormals out of its hi terms either way); it documents the hardware.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void probe(float a_val, float b_val, float* out, uint16_t* bits) {
    half8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    floatx16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        _Float16 h = (_Float16)a_val;
        bits[0] = __builtin_bit_cast(uint16_t, h);
        out[1] = (float)h;
    }
}

int main() {
    float* out; uint16_t* bits;
    hipMalloc(&out, 8); hipMalloc(&bits, 2);
    const float vals[3] = {9.5367431640625e-07f /* 2^-20 */, 3.0517578125e-05f /* 2^-15 */, 6.103515625e-05f /* 2^-14, smallest normal */};
    for (float v : vals) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, v, 1.0f, out, bits);
        float h[2]; uint16_t b;
        hipMemcpy(h, out, 8, hipMemcpyDeviceToHost); hipMemcpy(&b, bits, 2, hipMemcpyDeviceToHost);
        printf("a = %.6e  cvt_f16 bits 0x%04x back %.6e | mfma(a x16, 1) = %.6e  expected %.6e  -> %s\n", v, b, h[1], h[0], 16.0f * v,
               h[0] == 16.0f * v ? "denormals honoured" : (h[0] == 0.0f ? "FLUSHED" : "other"));
    }
    return 0;
}
