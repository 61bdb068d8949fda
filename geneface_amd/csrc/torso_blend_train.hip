// The tail of RADNeRFTorso.render's TRAINING branch (gfx950; /root/reference/modules/radnerfs/radnerf_torso.py:181-192 under autograd): mask the
// torso field's outputs, blend torso over background, head over that, clamp -- one launch forward, one launch backward.
//
//   alpha = a * m,  colour = c * m                                     (m: the torso mask as 0 / 1; the field ran on every sampled pixel)
//   torso_rgb = colour * alpha + bg * (1 - alpha)                      (radnerf_torso.py:186)
//   rgb = clamp(image + (1 - weights_sum) * torso_rgb, 0, 1)           (:190-191)
// Through torch these are ~10 elementwise launches forward and ~15 backward in a step that runs at the host's launch rate (NOTES 10.3, 10.11).
// Every product, sum and difference is rounded on its own, in the order of the torch expressions (no contraction): the same bits.
// Gradients go to a and c only: the head is frozen in the torso task, image / weights_sum / bg are data.
#include "common.hpp"
#include "geneface_hip.h"

namespace {

__global__ void __launch_bounds__(256) k_torso_blend_train_fwd(const gf_torso_blend_t t) {
#pragma clang fp contract(off)
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= t.N) return;
    const float m = t.mask[n];
    const float alpha = t.a[n] * m;
    const float one_minus = 1.0f - alpha;
    const float w = 1.0f - t.weights_sum[n];
    t.torso_alpha[n] = alpha;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float colour = t.c[n * 3 + ch] * m;
        const float bg = t.bg[(size_t)n * t.bg_stride + ch];
        const float tr = colour * alpha + bg * one_minus;
        t.torso_rgb[n * 3 + ch] = tr;
        const float pre = t.image[n * 3 + ch] + w * tr;
        t.rgb[n * 3 + ch] = fminf(fmaxf(pre, 0.0f), 1.0f);
    }
}

// g_a = m * (g_alpha + sum_ch g_tr[ch] * (colour[ch] - bg[ch])),  g_c[ch] = m * alpha * g_tr[ch],
// g_tr[ch] = g_torso_rgb[ch] + g_rgb[ch] * (1 - weights_sum) * [0 <= pre <= 1]      (torch's clamp passes the gradient on the closed interval)
__global__ void __launch_bounds__(256) k_torso_blend_train_bwd(const gf_torso_blend_t t) {
#pragma clang fp contract(off)
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= t.N) return;
    const float m = t.mask[n];
    const float alpha = t.a[n] * m;
    const float w = 1.0f - t.weights_sum[n];
    float ga = t.g_alpha ? t.g_alpha[n] : 0.0f;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float colour = t.c[n * 3 + ch] * m;
        const float bg = t.bg[(size_t)n * t.bg_stride + ch];
        const float tr = colour * alpha + bg * (1.0f - alpha);
        const float pre = t.image[n * 3 + ch] + w * tr;
        float g_tr = t.g_torso_rgb ? t.g_torso_rgb[n * 3 + ch] : 0.0f;
        if (t.g_rgb && pre >= 0.0f && pre <= 1.0f) g_tr += t.g_rgb[n * 3 + ch] * w;
        ga += g_tr * (colour - bg);
        t.g_c[n * 3 + ch] = g_tr * alpha * m;
    }
    t.g_a[n] = ga * m;
}

int check_common(const gf_torso_blend_t* t, const char* what) {
    if (!t) return gf_set_error(GF_ERR_INVALID, "%s: null descriptor", what);
    if (t->N == 0) return GF_OK;
    if (!t->a || !t->c || !t->mask || !t->bg || !t->image || !t->weights_sum) return gf_set_error(GF_ERR_INVALID, "%s: null input", what);
    if (t->bg_stride != 0 && t->bg_stride != 3) return gf_set_error(GF_ERR_INVALID, "%s: bg_stride must be 0 (one colour) or 3 (per pixel)", what);
    return GF_OK;
}

}  // namespace

GF_EXPORT int gf_torso_blend_train_forward(const gf_torso_blend_t* t, void* stream) {
    if (const int rc = check_common(t, "torso_blend_train_forward")) return rc;
    if (t->N == 0) return GF_OK;
    if (!t->torso_alpha || !t->torso_rgb || !t->rgb) return gf_set_error(GF_ERR_INVALID, "torso_blend_train_forward: null output");
    hipLaunchKernelGGL(k_torso_blend_train_fwd, dim3(gf_div_up(t->N, 256u)), dim3(256), 0, gf_stream(stream), *t);
    return gf_check_launch("torso_blend_train_forward");
}

GF_EXPORT int gf_torso_blend_train_backward(const gf_torso_blend_t* t, void* stream) {
    if (const int rc = check_common(t, "torso_blend_train_backward")) return rc;
    if (t->N == 0) return GF_OK;
    if (!t->g_a || !t->g_c) return gf_set_error(GF_ERR_INVALID, "torso_blend_train_backward: null output");
    hipLaunchKernelGGL(k_torso_blend_train_bwd, dim3(gf_div_up(t->N, 256u)), dim3(256), 0, gf_stream(stream), *t);
    return gf_check_launch("torso_blend_train_backward");
}
