// Stand-alone probe for the instruction pair behind the fast tier's run-to-run deviations (NOTES.md 4.7):
//
//     v_pk_mul_f32 W, P, Q op_sel:[0,1] op_sel_hi:[0,1]     ; W.lo = W.hi = P.lo * Q.hi   (the compiler's "broadcast" form)
//     [s_waitcnt vmcnt(k)]
//     v_pk_fma_f32 A, W, V, A                                ; A.lo += W.lo * V.lo,  A.hi += W.hi * V.hi
//
// In k_head_phase<true> the low half of exactly this v_pk_fma_f32 lost its product (A.lo came back unchanged) in lanes 32..63, only
// when two waves shared a SIMD.  The probe runs the pair in a loop next to a scalar reference (v_mul_f32 + 2 x v_fma_f32) with many waves
// per SIMD and counts mismatching halves per lane; variant 1 feeds V from a global load that the wait in between retires, variant 2
// puts an s_nop 0 between the two instructions.
//
//     hipcc --offload-arch=gfx950 -O2 tools/pk_fwd_probe.hip -o tools/pk_fwd_probe.bin && tools/pk_fwd_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

typedef float f16x __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// BG: what the odd-numbered workgroups (the likely co-residents on a SIMD) run instead of the pair: 0 = the pair as well,
// 1 = v_mfma_f32_32x32x16_f16 back to back, 2 = v_mfma_f32_32x32x2_f32 back to back
template <int VARIANT, int BG = 0>
__global__ void __launch_bounds__(256, 2) k_probe(const float* __restrict__ src, unsigned* __restrict__ bad_lo, unsigned* __restrict__ bad_hi,
                                                  int iters, int n_src) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (BG != 0 && (blockIdx.x & 1)) {
        f16x acc = {0};
        if (BG == 1) {
            h8 a, b;
            for (int i = 0; i < 8; i++) { a[i] = (_Float16)src[tid % 1024 + i]; b[i] = (_Float16)src[tid % 512 + 8 + i]; }
            for (int i = 0; i < iters * 2; i++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        } else {
            const float a = src[tid % 1024], b = src[tid % 512 + 8];
            for (int i = 0; i < iters * 2; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (acc[0] == 12345.678f) bad_lo[0] = 0xFFFFFFFFu;   // keep the loop alive
        return;
    }
    unsigned nlo = 0, nhi = 0;
    unsigned idx = (unsigned)tid * 2654435761u;
    for (int i = 0; i < iters; i++) {
        idx = idx * 1664525u + 1013904223u;
        const unsigned j = (idx >> 8) % (unsigned)(n_src - 8);
        f2 P = {src[j], src[j + 1]}, Q = {src[j + 2], src[j + 3]};
        f2 A = {src[j + 4], src[j + 5]};
        const f2 A0 = A;
        f2 W = {0.0f, 0.0f}, V;
        if (VARIANT == 1) {
            const f2* vp = reinterpret_cast<const f2*>(src + ((j * 7u) % (unsigned)(n_src - 8) & ~1u));
            asm volatile(
                "global_load_dwordx2 %2, %5, off\n"
                "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1] op_sel_hi:[0,1]\n"
                "s_waitcnt vmcnt(0)\n"
                "v_pk_fma_f32 %1, %0, %2, %1\n"
                : "+v"(W), "+v"(A), "=&v"(V)
                : "v"(P), "v"(Q), "v"(vp)
                : "memory");
        } else {
            V = f2{src[j + 6], src[j + 7]};
            if (VARIANT == 0)
                asm volatile(
                    "v_pk_mul_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[0,1]\n"
                    "v_pk_fma_f32 %1, %0, %4, %1\n"
                    : "+v"(W), "+v"(A)
                    : "v"(P), "v"(Q), "v"(V));
            else
                asm volatile(
                    "v_pk_mul_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[0,1]\n"
                    "s_nop 0\n"
                    "v_pk_fma_f32 %1, %0, %4, %1\n"
                    : "+v"(W), "+v"(A)
                    : "v"(P), "v"(Q), "v"(V));
        }
        float w, rlo, rhi;
        asm volatile(
            "v_mul_f32 %0, %3, %4\n"
            "s_nop 1\n"
            "v_fma_f32 %1, %0, %5, %7\n"
            "v_fma_f32 %2, %0, %6, %8\n"
            : "=&v"(w), "=&v"(rlo), "=&v"(rhi)
            : "v"(P.x), "v"(Q.y), "v"(V.x), "v"(V.y), "v"(A0.x), "v"(A0.y));
        nlo += __float_as_uint(rlo) != __float_as_uint(A.x);
        nhi += __float_as_uint(rhi) != __float_as_uint(A.y);
    }
    if (nlo) atomicAdd(&bad_lo[lane], nlo);
    if (nhi) atomicAdd(&bad_hi[lane], nhi);
}

// The whole last level of the 2-D lookup exactly as the compiler emitted it in k_head_phase<true> (four dwordx2 gathers retired by partial
// vmcnt waits, the broadcast v_pk_mul_f32 feeding the third v_pk_fma_f32, the scalar v_mul_f32 that overwrites the low weight right after).
__global__ void __launch_bounds__(256, 2) k_level(const float* __restrict__ src, unsigned* __restrict__ bad_lo, unsigned* __restrict__ bad_hi,
                                                  int iters, int n_src, int bg) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (bg && (blockIdx.x & 1)) {
        f16x acc = {0};
        h8 a, b;
        for (int i = 0; i < 8; i++) { a[i] = (_Float16)src[tid % 1024 + i]; b[i] = (_Float16)src[tid % 512 + 8 + i]; }
        for (int i = 0; i < iters * 4; i++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (acc[0] == 12345.678f) bad_lo[0] = 0xFFFFFFFFu;
        return;
    }
    unsigned nlo = 0, nhi = 0;
    unsigned idx = (unsigned)tid * 2654435761u;
    const unsigned rows = (unsigned)n_src / 2 - 4096;
    for (int i = 0; i < iters; i++) {
        idx = idx * 1664525u + 1013904223u;
        const float px = (float)((idx >> 9) & 0x7FFF) / 32768.0f, py = (float)((idx >> 3) & 0x7FFF) / 32768.0f;
        const unsigned r0 = (idx >> 7) % rows;
        const f2* a0 = reinterpret_cast<const f2*>(src) + r0;
        const f2* a1 = a0 + 1;
        const f2* a2 = a0 + 2049;
        const f2* a3 = a0 + 2050;
        float olo, ohi;
        asm volatile(
            "v_mov_b32 v100, %[px]\n v_mov_b32 v101, %[py]\n"
            "v_pk_add_f32 v[102:103], v[100:101], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
            "global_load_dwordx2 v[110:111], %[a0], off\n"
            "global_load_dwordx2 v[112:113], %[a1], off\n"
            "global_load_dwordx2 v[114:115], %[a2], off\n"
            "global_load_dwordx2 v[116:117], %[a3], off\n"
            "v_pk_mul_f32 v[104:105], v[102:103], v[102:103] op_sel:[0,1] op_sel_hi:[0,1]\n"
            "v_mul_f32 v106, v101, v102\n"
            "s_waitcnt vmcnt(3)\n"
            "v_pk_fma_f32 v[104:105], v[104:105], v[110:111], 0 op_sel_hi:[1,1,0]\n"
            "s_waitcnt vmcnt(2)\n"
            "v_pk_fma_f32 v[104:105], v[106:107], v[112:113], v[104:105] op_sel_hi:[0,1,1]\n"
            "v_pk_mul_f32 v[106:107], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[0,1]\n"
            "s_waitcnt vmcnt(1)\n"
            "v_pk_fma_f32 v[104:105], v[106:107], v[114:115], v[104:105]\n"
            "v_mul_f32 v106, v100, v101\n"
            "s_waitcnt vmcnt(0)\n"
            "v_pk_fma_f32 v[104:105], v[106:107], v[116:117], v[104:105] op_sel_hi:[0,1,1]\n"
            "s_nop 4\n"
            "v_mov_b32 %[olo], v104\n v_mov_b32 %[ohi], v105\n"
            : [olo] "=&v"(olo), [ohi] "=&v"(ohi)
            : [px] "v"(px), [py] "v"(py), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3)
            : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117");
        const float qx = 1.0f - px, qy = 1.0f - py;
        const f2 L0 = *a0, L1 = *a1, L2 = *a2, L3 = *a3;
        const float w0 = qx * qy, w1 = py * qx, w2 = px * qy, w3 = px * py;
        const float rlo = __builtin_fmaf(w3, L3.x, __builtin_fmaf(w2, L2.x, __builtin_fmaf(w1, L1.x, __builtin_fmaf(w0, L0.x, 0.0f))));
        const float rhi = __builtin_fmaf(w3, L3.y, __builtin_fmaf(w2, L2.y, __builtin_fmaf(w1, L1.y, __builtin_fmaf(w0, L0.y, 0.0f))));
        nlo += __float_as_uint(rlo) != __float_as_uint(olo);
        nhi += __float_as_uint(rhi) != __float_as_uint(ohi);
    }
    if (nlo) atomicAdd(&bad_lo[lane], nlo);
    if (nhi) atomicAdd(&bad_hi[lane], nhi);
}

static void run_level(const char* name, const float* d_src, int n_src, unsigned* d_lo, unsigned* d_hi, int blocks, int iters, int bg) {
    (void)hipMemset(d_lo, 0, 64 * 4);
    (void)hipMemset(d_hi, 0, 64 * 4);
    hipLaunchKernelGGL(k_level, dim3(blocks), dim3(256), 0, 0, d_src, d_lo, d_hi, iters, n_src, bg);
    (void)hipDeviceSynchronize();
    unsigned lo[64], hi[64];
    (void)hipMemcpy(lo, d_lo, sizeof lo, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hi, d_hi, sizeof hi, hipMemcpyDeviceToHost);
    unsigned long long tlo = 0, thi = 0, lo_upper = 0;
    for (int l = 0; l < 64; l++) { tlo += lo[l]; thi += hi[l]; if (l >= 32) lo_upper += lo[l]; }
    printf("%-34s blocks %5d iters %6d : low-half mismatches %llu (lanes 32..63: %llu), high-half mismatches %llu of %.3g lookups\n", name, blocks, iters,
           tlo, lo_upper, thi, (double)blocks * 256.0 * iters);
}

template <int VARIANT, int BG = 0>
static void run(const char* name, const float* d_src, int n_src, unsigned* d_lo, unsigned* d_hi, int blocks, int iters) {
    (void)hipMemset(d_lo, 0, 64 * 4);
    (void)hipMemset(d_hi, 0, 64 * 4);
    hipLaunchKernelGGL((k_probe<VARIANT, BG>), dim3(blocks), dim3(256), 0, 0, d_src, d_lo, d_hi, iters, n_src);
    (void)hipDeviceSynchronize();
    unsigned lo[64], hi[64];
    (void)hipMemcpy(lo, d_lo, sizeof lo, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hi, d_hi, sizeof hi, hipMemcpyDeviceToHost);
    unsigned long long tlo = 0, thi = 0, lo_upper = 0;
    for (int l = 0; l < 64; l++) { tlo += lo[l]; thi += hi[l]; if (l >= 32) lo_upper += lo[l]; }
    printf("%-34s blocks %5d iters %6d : low-half mismatches %llu (lanes 32..63: %llu), high-half mismatches %llu of %.3g pairs\n", name, blocks, iters,
           tlo, lo_upper, thi, (double)blocks * 256.0 * iters);
}

int main(int argc, char** argv) {
    const int n_src = 1 << 22;
    std::vector<float> h(n_src);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    float* d_src; unsigned *d_lo, *d_hi;
    (void)hipMalloc(&d_src, n_src * 4); (void)hipMalloc(&d_lo, 256); (void)hipMalloc(&d_hi, 256);
    (void)hipMemcpy(d_src, h.data(), n_src * 4, hipMemcpyHostToDevice);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int blocks : {256, 512, 2048}) {   // 1, 2 and 8 workgroups' worth per CU
        run<0>("pair back to back", d_src, n_src, d_lo, d_hi, blocks, iters);
        run<1>("pair across s_waitcnt vmcnt(0)", d_src, n_src, d_lo, d_hi, blocks, iters);
        run<2>("pair with s_nop 0 in between", d_src, n_src, d_lo, d_hi, blocks, iters);
        run<0, 1>("back to back, beside f16 MFMA", d_src, n_src, d_lo, d_hi, blocks, iters);
        run<1, 1>("across waitcnt, beside f16 MFMA", d_src, n_src, d_lo, d_hi, blocks, iters);
        run<0, 2>("back to back, beside f32 MFMA", d_src, n_src, d_lo, d_hi, blocks, iters);
        run<1, 2>("across waitcnt, beside f32 MFMA", d_src, n_src, d_lo, d_hi, blocks, iters);
        run_level("whole level", d_src, n_src, d_lo, d_hi, blocks, iters, 0);
        run_level("whole level, beside f16 MFMA", d_src, n_src, d_lo, d_hi, blocks, iters, 1);
    }
    return 0;
}
