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

GF_EXPORT const char* gf_last_error(void) { return g_last_error; }

GF_EXPORT const char* gf_version(void) { return "geneface_hip 0.1 (gfx950)"; }

// number of HIP devices visible to this process; <0 on runtime error (lets the host side fail loudly)
GF_EXPORT int gf_device_count(void) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { gf_set_error(GF_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); return -1; }
    return n;
}
