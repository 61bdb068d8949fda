// Error plumbing + library identity for libgeneface_hip.so (declared in include/geneface_hip.h).
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_last_error[512] = "";

int gf_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

int gf_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gf_set_error(GF_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return GF_OK;
}

int gf_raise_lds_limit(GfLdsAttr& st, const void* fn, int bytes, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return gf_set_error(GF_ERR_HIP, "%s: hipGetDevice failed", what);
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&st.done, __ATOMIC_ACQUIRE) & bit) return GF_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return gf_set_error(GF_ERR_HIP, "%s: cannot raise the dynamic LDS limit to %d bytes on device %d", what, bytes, dev);
    __atomic_fetch_or(&st.done, bit, __ATOMIC_RELEASE);
    return GF_OK;
}

GF_EXPORT const char* gf_last_error(void) { return g_last_error; }

GF_EXPORT const char* gf_version(void) { return "geneface_hip 0.1 (gfx950)"; }

// number of HIP devices visible to this process; <0 on runtime error (lets the host side fail loudly)
GF_EXPORT int gf_device_count(void) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { gf_set_error(GF_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); return -1; }
    return n;
}
