// Shared helpers for the gfx950 kernels of the RAD-NeRF render path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GF_EXPORT extern "C" __attribute__((visibility("default")))

// error codes returned by every C-ABI entry point (0 == success); gf_last_error() has the text
enum {
    GF_OK = 0,
    GF_ERR_INVALID = 1,   // bad argument (the reference throws std::runtime_error / TORCH_CHECK here)
    GF_ERR_HIP = 2,       // a HIP runtime call or launch failed
    GF_ERR_UNSUPPORTED = 3,
};

int gf_set_error(int code, const char* fmt, ...);
int gf_check_launch(const char* what);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to ONE device: remembered per call site and per device, so a process that
// drives several GPUs raises the limit on each of them (a bit per device ordinal).
struct GfLdsAttr { unsigned long long done = 0; };
int gf_raise_lds_limit(GfLdsAttr& st, const void* fn, int bytes, const char* what);

static inline hipStream_t gf_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
__host__ __device__ inline T gf_div_up(T a, T b) { return (a + b - 1) / b; }

namespace gf {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// 10-bit -> 30-bit spread used by the Morton-ordered occupancy grid
__host__ __device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__host__ __device__ __forceinline__ uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__host__ __device__ __forceinline__ uint32_t morton3d_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

}  // namespace gf
