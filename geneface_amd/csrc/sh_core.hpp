// Direction / frequency encodings shared by the op kernels and the fused field kernels.
//   real spherical harmonics, bands 0..3 (16 values): contract = kernel_sh,
//     /root/reference/modules/radnerfs/encoders/shencoder/src/shencoder.cu:50-68
//   bands 4..7 (degrees 5..8 of the `_shencoder` seam; shencoder.cu:69-121 values, :150-356 derivatives): sh_high, table driven
//   NeRF frequency encoding: contract = kernel_freq, .../freqencoder/src/freqencoder.cu:30-58
#pragma once
#include "common.hpp"

namespace gf {

// sh[0..15] for a (unit) direction.  Band ordering and signs follow the reference's basis.
__device__ __forceinline__ void sh4(float x, float y, float z, float (&sh)[16]) {
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    constexpr float k1 = 0.48860251190291987f;   // sqrt(3/(4pi))
    constexpr float k2 = 1.0925484305920792f;    // sqrt(15/(4pi))
    constexpr float k3a = 0.59004358992664352f;  // sqrt(35/(32pi))
    constexpr float k3b = 0.45704579946446572f;  // sqrt(21/(32pi))
    sh[0] = 0.28209479177387814f;
    sh[1] = -k1 * y;
    sh[2] = k1 * z;
    sh[3] = -k1 * x;
    sh[4] = k2 * xy;
    sh[5] = -k2 * yz;
    sh[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    sh[7] = -k2 * xz;
    sh[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    sh[9] = k3a * y * (-3.0f * x2 + y2);
    sh[10] = 2.8906114426405538f * xy * z;
    const float m5z = 1.0f - 5.0f * z2;
    sh[11] = k3b * y * m5z;
    sh[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    sh[13] = k3b * x * m5z;
    sh[14] = 1.4453057213202769f * z * (x2 - y2);
    sh[15] = k3a * x * (-x2 + 3.0f * y2);
}

// d sh[k] / d(x, y, z) of the basis above (what the reference's kernel_sh writes to dy_dx, shencoder.cu:122-356, for degree <= 4),
// derived from the polynomials of sh4.
__device__ __forceinline__ void sh4_grad(float x, float y, float z, float (&gx)[16], float (&gy)[16], float (&gz)[16]) {
    constexpr float k1 = 0.48860251190291987f, k2 = 1.0925484305920792f, k3a = 0.59004358992664352f, k3b = 0.45704579946446572f;
    constexpr float a6 = 0.94617469575755997f, c8 = 0.54627421529603959f, c10 = 2.8906114426405538f, c12 = 0.3731763325901154f,
                    c14 = 1.4453057213202769f;
    const float x2 = x * x, y2 = y * y, z2 = z * z;
#pragma unroll
    for (int k = 0; k < 16; k++) { gx[k] = 0.0f; gy[k] = 0.0f; gz[k] = 0.0f; }
    gy[1] = -k1; gz[2] = k1; gx[3] = -k1;
    gx[4] = k2 * y; gy[4] = k2 * x;
    gy[5] = -k2 * z; gz[5] = -k2 * y;
    gz[6] = 2 * a6 * z;
    gx[7] = -k2 * z; gz[7] = -k2 * x;
    gx[8] = 2 * c8 * x; gy[8] = -2 * c8 * y;
    gx[9] = -6 * k3a * x * y; gy[9] = 3 * k3a * (y2 - x2);
    gx[10] = c10 * y * z; gy[10] = c10 * x * z; gz[10] = c10 * x * y;
    gy[11] = k3b * (1 - 5 * z2); gz[11] = -10 * k3b * y * z;
    gz[12] = c12 * (15 * z2 - 3);
    gx[13] = k3b * (1 - 5 * z2); gz[13] = -10 * k3b * x * z;
    gx[14] = 2 * c14 * x * z; gy[14] = -2 * c14 * y * z; gz[14] = c14 * (x2 - y2);
    gx[15] = 3 * k3a * (y2 - x2); gy[15] = 6 * k3a * x * y;
}

// ---- bands 4..7 (basis functions 16..63): the op seam serves degree <= 8 like the reference's extension; GeneFace itself stops at 4.
// The reference lists 48 polynomials and 144 partial derivatives term by term.  Here the basis is evaluated from its structure,
//     Y_l^m = N_l^m Q_l^|m|(z) * { A_m (m > 0) | 1 | B_|m| (m < 0) },   A_m + i B_m = (x + i y)^m,   Q_l^m = d^m P_l / dz^m,
// and because dQ_l^m/dz = Q_l^(m+1), dA_m/dx = m A_(m-1), dA_m/dy = -m B_(m-1), dB_m/dx = m B_(m-1), dB_m/dy = m A_(m-1), values
// and all three partials come from one family of Horner polynomials in z and one (A, B) recurrence: ~26 Horner chains + 14 products
// instead of 192 expressions.  Same polynomials (on R^3, x^2 + y^2 eliminated in favour of z: the representative the reference
// differentiates), other association order: agrees with the reference's kernel to a few ulp of the output scale
// (tests/test_gpu_vs_ref_kernels.py).  Tables: tools/gen_sh_tables.py (derived from the definition in exact rational arithmetic).
#include "sh_high_tables.inc"

__device__ __forceinline__ float sh_horner8(const float (&c)[8], float z) {
    float acc = c[7];
#pragma unroll
    for (int k = 6; k >= 0; k--) acc = fmaf(acc, z, c[k]);
    return acc;
}

// Writes out[16 .. degree^2) and, when gx != nullptr, the three derivative rows over the same index range.  degree in 5..8.
__device__ __forceinline__ void sh_high(float x, float y, float z, uint32_t degree, float* __restrict__ out, float* __restrict__ gx,
                                        float* __restrict__ gy, float* __restrict__ gz) {
    float A[8], B[8];
    A[0] = 1.0f; B[0] = 0.0f;
#pragma unroll
    for (int m = 1; m < 8; m++) {
        A[m] = x * A[m - 1] - y * B[m - 1];
        B[m] = x * B[m - 1] + y * A[m - 1];
    }
#pragma unroll
    for (int l = 4; l < 8; l++) {
        if ((uint32_t)l >= degree) break;
        const int c0 = l * l + l;
#pragma unroll
        for (int m = 0; m <= l; m++) {
            const float q = sh_horner8(kShHighQ[kShHighRow[l - 4] + m], z);
            if (m == 0) {
                out[c0] = q;
            } else {
                out[c0 + m] = q * A[m];
                out[c0 - m] = q * B[m];
            }
            if (gx) {
                const float dq = sh_horner8(kShHighDQ[kShHighRow[l - 4] + m], z);
                if (m == 0) {
                    gx[c0] = 0.0f; gy[c0] = 0.0f; gz[c0] = dq;
                } else {
                    const float qm = q * (float)m;
                    gx[c0 + m] = qm * A[m - 1];  gy[c0 + m] = -qm * B[m - 1];  gz[c0 + m] = dq * A[m];
                    gx[c0 - m] = qm * B[m - 1];  gy[c0 - m] = qm * A[m - 1];   gz[c0 - m] = dq * B[m];
                }
            }
        }
    }
}

// sin(x) for the frequency encodings (round 5).  The reference calls the hardware's fast sine (__sinf, freqencoder.cu:52: absolute error that
// grows with |x|); oracle and product evaluate the function it approximates.  libm's sinf carries a Payne-Hanek reduction for arguments up to
// 3e38 that the compiler must inline in full -- ~100 VALU operations per call, 2 400 of the 4 000 VALU instructions of a torso tile (24
// encodings per lane), the largest single item of k_torso_field.  The encodings' arguments are 2^f x + {0, pi/2} with f <= 9 and |x| of order 1:
// a three-term Cody-Waite reduction with fused multiply-adds (pi/2 = C1 + C2 + C3 to 2^-75) and the two degree-9 / degree-8 minimax
// polynomials on [-pi/4, pi/4] do it in ~24 operations, absolute error <= 9.3e-8 against the exact sine over |x| <= 512 (measured on 12 M
// arguments in fp32 emulation; glibc's sinf: 7.0e-8), <= 1.6 ulp wherever |sin| > 0.01.  Beyond |x| = 8192 (never reached by an encoding of a
// bounded coordinate; the op seam accepts any float) and for NaN the call falls back to sinf.
// sin_reduced: the branch-free core, valid for |x| <= 8192 (anything else gives a finite, meaningless value).  Callers whose arguments are
// bounded by construction use it directly: 24 of them in a row then interleave freely (the torso tile: with the fallback's branch between
// them each was a serial chain, 480 cycles apiece where 24 operations need ~100; tools/trace_torso.py).
__device__ __forceinline__ float sin_reduced(float x) {
    const float n = rintf(x * 0.63661977236758134f);
    float r = __builtin_fmaf(-n, 1.5707963705062866f, x);
    r = __builtin_fmaf(-n, -4.371138828673793e-08f, r);
    r = __builtin_fmaf(-n, -1.7763568394002505e-15f, r);
    const float r2 = r * r;
    float ps = __builtin_fmaf(r2, 2.7557314297e-06f, -1.9841270114e-04f);
    ps = __builtin_fmaf(r2, ps, 8.3333337680e-03f);
    ps = __builtin_fmaf(r2, ps, -1.6666667163e-01f);
    const float s = __builtin_fmaf(r * r2, ps, r);
    float pc = __builtin_fmaf(r2, -2.7557314297e-07f, 2.4801587642e-05f);
    pc = __builtin_fmaf(r2, pc, -1.3888889225e-03f);
    pc = __builtin_fmaf(r2, pc, 4.1666667908e-02f);
    const float c = __builtin_fmaf(r2 * r2, pc, __builtin_fmaf(r2, -0.5f, 1.0f));
    const int q = (int)n;
    const float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
__device__ __forceinline__ float sin_bounded(float x) {
    if (!(fabsf(x) <= 8192.0f)) return sinf(x);
    return sin_reduced(x);
}

// element c of the [D + 2*D*deg] frequency encoding of in[0..D):
//   [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...], cos evaluated as sin(. + pi/2) with pi/2 rounded to fp32
__device__ __forceinline__ float freq_element(const float* __restrict__ in, uint32_t D, uint32_t c) {
    if (c < D) return in[c];
    const uint32_t col = c / D - 1, d = c - (col + 1) * D, freq = col >> 1;
    const float phase = (col & 1u) ? (3.141592653589793f / 2) : 0.0f;
    // the reference's __sinf is a hardware-specific fast sine (error grows with |x|, and the torso's
    // 2^9 x reaches ~400 rad); the accurate sine is the centroid every fast sine approximates
    return sin_bounded(scalbnf(in[d], (int)freq) + phase);
}

}  // namespace gf
