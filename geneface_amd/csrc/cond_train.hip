// The condition encoder of the RAD-NeRF head under TRAINING (gfx950): RADNeRF.cal_cond_feat (/root/reference/modules/radnerfs/radnerf.py:61-71)
//   = AudioNet.forward (cond_encoder.py:44-52: four k=3 Conv1d + LeakyReLU(0.02), FC 64->64 LeakyReLU, FC 64->dim_aud)
//   + AudioAttNet.forward (cond_encoder.py:79-89: five k=3 Conv1d over the window axis, Linear(S,S), softmax, weighted sum)
// as ONE forward launch that keeps every layer's activations and TWO backward launches (the activation-gradient chain on one workgroup, then
// every element of the 24 parameter gradients as an independent sum on a grid).  Under torch these are ~50 + ~50 launches per training step (MIOpen convolutions of a [5, 204, 1] window -- half of them its
// "naive" fallback kernels --, GEMVs, activations, softmax, and under autocast a cast of every weight): 0.4 ms of kernels and as many
// microseconds of host time in a 5.5 ms step (profiles/round6/r6u_train_step_kernel_stats_amp_final.csv; NOTES 10.9: HIP graphs around the
// torch ops made it slower).  53 K parameters, ~1 MFLOP: one workgroup, latency bound; activations and their gradients live in a small
// global scratch (L2), layers are separated by workgroup barriers.  fp32 throughout -- under autocast the reference runs these layers in
// half; the master weights and the gradients it hands the optimizer are fp32 either way.
// The inference-side twin is cond_encode.hip::k_cond_encode (same arithmetic, same operation order per output: bit-identical cond_feat).
#include "common.hpp"
#include "geneface_hip.h"

namespace {

constexpr int kThreads = 256;
// attentionConvNet channels: dim_aud -> 16 -> 8 -> 4 -> 2 -> 1 (index 0 = the input, given by the caller)
__host__ __device__ inline int att_ch(int l) { return l == 1 ? 16 : l == 2 ? 8 : l == 3 ? 4 : l == 4 ? 2 : 1; }

__device__ __forceinline__ float leaky(float x) { return x > 0.0f ? x : 0.02f * x; }
__device__ __forceinline__ float dleaky(float y) { return y > 0.0f ? 1.0f : 0.02f; }      // from the OUTPUT: leaky keeps the sign

struct Dims { int S, T, C, A; int ch[5], st[4], len[5]; };      // len[l]: window length entering conv l (len[4] = 1)

// offsets (floats) of every activation in the scratch: the permuted input, the four conv outputs, fc1's output, the features, the
// attention net's inputs / outputs, the softmax
struct Layout { int act[5], h1, feat, y[6], p, total; };
__host__ __device__ inline Layout layout(const Dims& d) {
    Layout L;
    int o = 0;
    for (int l = 0; l < 5; l++) { L.act[l] = o; o += d.S * d.ch[l] * d.len[l]; }
    L.h1 = o; o += d.S * 64;
    L.feat = o; o += d.S * d.A;
    for (int l = 0; l < 6; l++) { L.y[l] = o; o += (l == 0 ? d.A : att_ch(l)) * d.S; }
    L.p = o; o += d.S;
    L.total = o;
    return L;
}

constexpr int kMaxC = 224;
constexpr int kWFloats = kMaxC * 32 * 3;          // the largest layer (first conv, C x 32 x 3) staged in LDS
constexpr int kSmemBytes = kWFloats * 4;

// a layer's weights into LDS with coalesced 16-byte loads, eight per lane in flight (the copy is pure latency); cond_encode.hip::stage
__device__ __forceinline__ void stage(float* __restrict__ dst, const float* __restrict__ src, int n) {
    const int tid = threadIdx.x;
    if (((uintptr_t)src & 15u) == 0) {
        const int n4 = n >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        int i = tid;
        for (; i + 7 * kThreads < n4; i += 8 * kThreads) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = s4[i + k * kThreads];
#pragma unroll
            for (int k = 0; k < 8; k++) d4[i + k * kThreads] = v[k];
        }
        for (; i < n4; i += kThreads) d4[i] = s4[i];
        for (int j = (n4 << 2) + tid; j < n; j += kThreads) dst[j] = src[j];
    } else {
        for (int i = tid; i < n; i += kThreads) dst[i] = src[i];
    }
}

// out[n][co][p] = act(b[co] + sum_ci sum_k w[co][ci][k] in[n][ci][p stride + k - 1]), zero padding 1 (torch Conv1d k = 3); the same loop
// order over (ci, k) as cond_encode.hip::conv1d_k3 (four lanes share an output and split ci; partial sums added by two butterflies).  w in LDS.
__device__ void conv_fwd(const float* __restrict__ w, const float* __restrict__ b, const float* in, float* out, int N, int cin, int cout, int lin,
                         int lout, int stride) {
    const int total = N * cout * lout, part = threadIdx.x & 3;
    for (int base = 0; base < total; base += kThreads / 4) {
        const int idx = base + (threadIdx.x >> 2);
        float sum = 0.0f;
        if (idx < total) {
            const int p = idx % lout, co = (idx / lout) % cout, n = idx / (lout * cout);
            const float* wr = w + (size_t)co * cin * 3;
            const float* xr = in + (size_t)n * cin * lin;
            const int pos0 = p * stride - 1;
            const int k_lo = pos0 < 0 ? -pos0 : 0, k_hi = pos0 + 2 >= lin ? lin - 1 - pos0 : 2;
            for (int ci = part; ci < cin; ci += 4)
                for (int k = k_lo; k <= k_hi; k++) sum = __builtin_fmaf(wr[ci * 3 + k], xr[ci * lin + pos0 + k], sum);
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        if (idx < total && part == 0) out[idx] = leaky(sum + b[(idx / lout) % cout]);
    }
}

// rows: out[s][o] = act(b[o] + sum_i w[o][i] in[s][i]); w in LDS
template <bool LEAKY>
__device__ void fc_fwd(const float* __restrict__ w, const float* __restrict__ b, const float* in, float* out, int S, int I, int O) {
    for (int idx = threadIdx.x; idx < S * O; idx += kThreads) {
        const int o = idx % O, s = idx / O;
        float sum = b[o];
        for (int c = 0; c < I; c++) sum = __builtin_fmaf(w[o * I + c], in[s * I + c], sum);
        out[idx] = LEAKY ? leaky(sum) : sum;
    }
}

// ---- backward, chain part (one workgroup): g_out of a layer -> g_pre = g_out * leaky'(out) IN PLACE -> gradient of the layer's input.
__device__ void pre_inplace(float* g, const float* out, int n) {
    for (int i = threadIdx.x; i < n; i += kThreads) g[i] *= dleaky(out[i]);
}
//   g_in[n][ci][q] = sum_co sum_{k, p: p stride + k - 1 = q} w[co][ci][k] g_pre[n][co][p];  w in LDS
__device__ void conv_bwd_input(const float* __restrict__ w, const float* g_pre, float* g_in, int N, int cin, int cout, int lin, int lout, int stride) {
    for (int e = threadIdx.x; e < N * cin * lin; e += kThreads) {
        const int q = e % lin, ci = (e / lin) % cin, n = e / (lin * cin);
        float sum = 0.0f;
        for (int k = 0; k < 3; k++) {
            const int t = q + 1 - k;
            if (t < 0 || t % stride != 0) continue;
            const int p = t / stride;
            if (p >= lout) continue;
            for (int co = 0; co < cout; co++) sum = __builtin_fmaf(w[(co * cin + ci) * 3 + k], g_pre[(n * cout + co) * lout + p], sum);
        }
        g_in[e] = sum;
    }
}
__device__ void fc_bwd_input(const float* __restrict__ w, const float* g_pre, float* g_in, int S, int I, int O) {
    for (int e = threadIdx.x; e < S * I; e += kThreads) {
        const int i = e % I, s = e / I;
        float sum = 0.0f;
        for (int o = 0; o < O; o++) sum = __builtin_fmaf(w[o * I + i], g_pre[s * O + o], sum);
        g_in[e] = sum;
    }
}

// ---- backward, weight part (many workgroups; lane `gt` of `gn`): every element of every parameter gradient is one independent sum
//   g_w[co][ci][k] = sum_n sum_p g_pre[n][co][p] in[n][ci][p stride + k - 1],  g_b[co] = sum_n sum_p g_pre[n][co][p]
__device__ void conv_bwd_weight(const float* in, const float* g_pre, float* __restrict__ g_w, float* __restrict__ g_b, int N, int cin, int cout, int lin,
                                int lout, int stride, int gt, int gn) {
    for (int e = gt; e < cout * cin * 3; e += gn) {
        const int k = e % 3, ci = (e / 3) % cin, co = e / (3 * cin);
        float sum = 0.0f;
        for (int n = 0; n < N; n++)
            for (int p = 0; p < lout; p++) {
                const int q = p * stride + k - 1;
                if (q >= 0 && q < lin) sum = __builtin_fmaf(g_pre[(n * cout + co) * lout + p], in[(n * cin + ci) * lin + q], sum);
            }
        g_w[e] = sum;
    }
    for (int co = gt; co < cout; co += gn) {
        float sum = 0.0f;
        for (int n = 0; n < N; n++)
            for (int p = 0; p < lout; p++) sum += g_pre[(n * cout + co) * lout + p];
        g_b[co] = sum;
    }
}
__device__ void fc_bwd_weight(const float* in, const float* g_pre, float* __restrict__ g_w, float* __restrict__ g_b, int S, int I, int O, int gt, int gn) {
    for (int e = gt; e < O * I; e += gn) {
        const int i = e % I, o = e / I;
        float sum = 0.0f;
        for (int s = 0; s < S; s++) sum = __builtin_fmaf(g_pre[s * O + o], in[s * I + i], sum);
        g_w[e] = sum;
    }
    for (int o = gt; o < O; o += gn) {
        float sum = 0.0f;
        for (int s = 0; s < S; s++) sum += g_pre[s * O + o];
        g_b[o] = sum;
    }
}

struct TrainArgs {
    gf_cond_t enc;
    Dims d;
    float* acts;          // Layout::total floats: the forward's activations
    float* grads;         // Layout::total floats: the backward's activation gradients (scratch)
    const float* g_feat;  // [A]
    float* g_conv_w[4]; float* g_conv_b[4];
    float *g_fc1_w, *g_fc1_b, *g_fc2_w, *g_fc2_b;
    float* g_att_w[5]; float* g_att_b[5];
    float *g_att_lin_w, *g_att_lin_b;
};

__global__ void __launch_bounds__(kThreads) k_cond_train_fwd(const TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wbuf = reinterpret_cast<float*>(smem_raw);
    const Dims& d = a.d;
    const Layout L = layout(d);
    const gf_cond_t& e = a.enc;
    const int tid = threadIdx.x, S = d.S, T = d.T, C = d.C, A = d.A;
    float* X = a.acts;
    for (int i = tid; i < S * T * C; i += kThreads) {                    // [S, T, C] -> permute(0, 2, 1) -> [S, C, T]
        const int c = i % C, t = (i / C) % T, s = i / (C * T);
        X[L.act[0] + (s * C + c) * T + t] = e.cond[i];
    }
    for (int l = 0; l < 4; l++) {
        __syncthreads();
        stage(wbuf, e.conv_w[l], d.ch[l + 1] * d.ch[l] * 3);
        __syncthreads();
        conv_fwd(wbuf, e.conv_b[l], X + L.act[l], X + L.act[l + 1], S, d.ch[l], d.ch[l + 1], d.len[l], d.len[l + 1], d.st[l]);
    }
    __syncthreads();
    stage(wbuf, e.fc1_w, 64 * 64);
    __syncthreads();
    fc_fwd<true>(wbuf, e.fc1_b, X + L.act[4], X + L.h1, S, 64, 64);
    __syncthreads();
    stage(wbuf, e.fc2_w, A * 64);
    __syncthreads();
    fc_fwd<false>(wbuf, e.fc2_b, X + L.h1, X + L.feat, S, 64, A);
    __syncthreads();
    if (!e.att_lin_w) {                                                   // with_att = false: cond_feat = the (single) window's features
        for (int c = tid; c < A; c += kThreads) e.cond_feat[c] = X[L.feat + c];
        return;
    }
    for (int i = tid; i < A * S; i += kThreads) X[L.y[0] + i] = X[L.feat + (i % S) * A + i / S];      // x[:, :A].permute(1, 0): [A][S]
    for (int l = 0; l < 5; l++) {
        const int cin = l == 0 ? A : att_ch(l);
        __syncthreads();
        stage(wbuf, e.att_w[l], att_ch(l + 1) * cin * 3);
        __syncthreads();
        conv_fwd(wbuf, e.att_b[l], X + L.y[l], X + L.y[l + 1], 1, cin, att_ch(l + 1), S, S, 1);
    }
    __syncthreads();
    __shared__ float vec[64];
    if (tid < S) {
        float sum = e.att_lin_b[tid];
        for (int j = 0; j < S; j++) sum = __builtin_fmaf(e.att_lin_w[tid * S + j], X[L.y[5] + j], sum);
        vec[tid] = sum;
    }
    __syncthreads();
    if (tid == 0) {
        float m = vec[0];
        for (int j = 1; j < S; j++) m = fmaxf(m, vec[j]);
        float den = 0.0f;
        for (int j = 0; j < S; j++) { const float ex = expf(vec[j] - m); vec[16 + j] = ex; den += ex; }
        for (int j = 0; j < S; j++) X[L.p + j] = vec[32 + j] = vec[16 + j] / den;
    }
    __syncthreads();
    for (int c = tid; c < A; c += kThreads) {
        float sum = 0.0f;
        for (int s = 0; s < S; s++) sum += vec[32 + s] * X[L.feat + s * A + c];
        e.cond_feat[c] = sum;
    }
}

// Backward, part 1 (one workgroup): the activation gradients of every layer, each left in `grads` as the layer's PRE-activation gradient.
__global__ void __launch_bounds__(kThreads) k_cond_train_bwd_chain(const TrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wbuf = reinterpret_cast<float*>(smem_raw);
    const Dims& d = a.d;
    const Layout L = layout(d);
    const gf_cond_t& e = a.enc;
    const int tid = threadIdx.x, S = d.S, A = d.A;
    const float* X = a.acts;
    float* G = a.grads;
    if (!e.att_lin_w) {
        for (int c = tid; c < A; c += kThreads) G[L.feat + c] = a.g_feat[c];
    } else {
        __shared__ float gp[16], gl[16];
        if (tid < S) {                                                    // out[c] = sum_s p[s] feat[s][c]
            float sum = 0.0f;
            for (int c = 0; c < A; c++) sum = __builtin_fmaf(a.g_feat[c], X[L.feat + tid * A + c], sum);
            gp[tid] = sum;
        }
        for (int i = tid; i < S * A; i += kThreads) G[L.feat + i] = X[L.p + i / A] * a.g_feat[i % A];
        __syncthreads();
        if (tid < S) {                                                    // softmax: g_logit[s] = p[s] (g_p[s] - sum_t p[t] g_p[t])
            float dot = 0.0f;
            for (int t = 0; t < S; t++) dot = __builtin_fmaf(X[L.p + t], gp[t], dot);
            gl[tid] = X[L.p + tid] * (gp[tid] - dot);
        }
        __syncthreads();
        if (tid < S) {                                                    // logits = W y5 + b: the gradient of the logits is kept in p's slot
            G[L.p + tid] = gl[tid];
            float sum = 0.0f;
            for (int i = 0; i < S; i++) sum = __builtin_fmaf(e.att_lin_w[i * S + tid], gl[i], sum);
            G[L.y[5] + tid] = sum;
        }
        for (int l = 4; l >= 0; l--) {
            const int cin = l == 0 ? A : att_ch(l), cout = att_ch(l + 1);
            __syncthreads();
            stage(wbuf, e.att_w[l], cout * cin * 3);
            pre_inplace(G + L.y[l + 1], X + L.y[l + 1], cout * S);
            __syncthreads();
            conv_bwd_input(wbuf, G + L.y[l + 1], G + L.y[l], 1, cin, cout, S, S, 1);
        }
        __syncthreads();
        for (int i = tid; i < S * A; i += kThreads) G[L.feat + i] += G[L.y[0] + (i % A) * S + i / A];      // the attention net read feat^T
    }
    __syncthreads();
    stage(wbuf, e.fc2_w, A * 64);                                         // fc2 has no activation: g_pre = g_out
    __syncthreads();
    fc_bwd_input(wbuf, G + L.feat, G + L.h1, S, 64, A);
    __syncthreads();
    stage(wbuf, e.fc1_w, 64 * 64);
    pre_inplace(G + L.h1, X + L.h1, S * 64);
    __syncthreads();
    fc_bwd_input(wbuf, G + L.h1, G + L.act[4], S, 64, 64);
    for (int l = 3; l >= 0; l--) {
        __syncthreads();
        if (l) stage(wbuf, e.conv_w[l], d.ch[l + 1] * d.ch[l] * 3);
        pre_inplace(G + L.act[l + 1], X + L.act[l + 1], S * d.ch[l + 1] * d.len[l + 1]);
        __syncthreads();
        if (l) conv_bwd_input(wbuf, G + L.act[l + 1], G + L.act[l], S, d.ch[l], d.ch[l + 1], d.len[l], d.len[l + 1], d.st[l]);
    }
}

// Backward, part 2 (a grid of workgroups): all 24 parameter gradients from the activations and the pre-activation gradients.
__global__ void __launch_bounds__(kThreads) k_cond_train_bwd_weights(const TrainArgs a) {
    const Dims& d = a.d;
    const Layout L = layout(d);
    const gf_cond_t& e = a.enc;
    const int S = d.S, A = d.A;
    const int gt = (int)(blockIdx.x * kThreads + threadIdx.x), gn = (int)(gridDim.x * kThreads);
    const float* X = a.acts;
    const float* G = a.grads;
    for (int l = 0; l < 4; l++)
        conv_bwd_weight(X + L.act[l], G + L.act[l + 1], a.g_conv_w[l], a.g_conv_b[l], S, d.ch[l], d.ch[l + 1], d.len[l], d.len[l + 1], d.st[l], gt, gn);
    fc_bwd_weight(X + L.act[4], G + L.h1, a.g_fc1_w, a.g_fc1_b, S, 64, 64, gt, gn);
    fc_bwd_weight(X + L.h1, G + L.feat, a.g_fc2_w, a.g_fc2_b, S, 64, A, gt, gn);
    if (e.att_lin_w) {
        for (int l = 0; l < 5; l++)
            conv_bwd_weight(X + L.y[l], G + L.y[l + 1], a.g_att_w[l], a.g_att_b[l], 1, l == 0 ? A : att_ch(l), att_ch(l + 1), S, S, 1, gt, gn);
        for (int i = gt; i < S * S; i += gn) a.g_att_lin_w[i] = G[L.p + i / S] * X[L.y[5] + i % S];
        for (int i = gt; i < S; i += gn) a.g_att_lin_b[i] = G[L.p + i];
    }
}

int fill(const gf_cond_train_t* t, TrainArgs& a, bool backward) {
    if (!t || !t->enc) return gf_set_error(GF_ERR_INVALID, "cond_train: null descriptor");
    const gf_cond_t& c = *t->enc;
    gf_cond_t probe = c;
    const bool with_att = c.att_lin_w != nullptr;
    if (!with_att) {      // gf_cond_check wants the attention weights: validate the rest, then require a single window
        static const float dummy = 0.0f;
        probe.att_lin_w = probe.att_lin_b = &dummy;
        for (int l = 0; l < 5; l++) probe.att_w[l] = probe.att_b[l] = &dummy;
    }
    if (const int rc = gf_cond_check(&probe)) return rc;
    if (!with_att && c.S != 1) return gf_set_error(GF_ERR_UNSUPPORTED, "cond_train: without the attention net the window must be a single frame");
    if (c.S > 16 || c.dim_aud > 64) return gf_set_error(GF_ERR_UNSUPPORTED, "cond_train: S <= 16, dim_aud <= 64");
    if (!c.cond || !c.cond_feat || !t->acts || !t->grads) return gf_set_error(GF_ERR_INVALID, "cond_train: null buffer");
    a.enc = c;
    a.d.S = (int)c.S; a.d.T = (int)c.T; a.d.C = (int)c.C; a.d.A = (int)c.dim_aud;
    int len = (int)c.T;
    for (int l = 0; l < 5; l++) {
        a.d.ch[l] = (int)c.conv_ch[l];
        a.d.len[l] = len;
        if (l < 4) { a.d.st[l] = (int)c.conv_stride[l]; len = (len + 2 - 3) / (int)c.conv_stride[l] + 1; }
    }
    a.acts = t->acts; a.grads = t->grads; a.g_feat = t->g_feat;
    if (backward) {
        const void* need[] = {t->g_feat, t->g_conv_w[0], t->g_conv_w[1], t->g_conv_w[2], t->g_conv_w[3], t->g_conv_b[0], t->g_conv_b[1], t->g_conv_b[2],
                              t->g_conv_b[3], t->g_fc1_w, t->g_fc1_b, t->g_fc2_w, t->g_fc2_b};
        for (const void* p : need) if (!p) return gf_set_error(GF_ERR_INVALID, "cond_train_backward: null gradient buffer");
        if (with_att) {
            for (int l = 0; l < 5; l++) if (!t->g_att_w[l] || !t->g_att_b[l]) return gf_set_error(GF_ERR_INVALID, "cond_train_backward: null gradient buffer");
            if (!t->g_att_lin_w || !t->g_att_lin_b) return gf_set_error(GF_ERR_INVALID, "cond_train_backward: null gradient buffer");
        }
        for (int l = 0; l < 4; l++) { a.g_conv_w[l] = t->g_conv_w[l]; a.g_conv_b[l] = t->g_conv_b[l]; }
        for (int l = 0; l < 5; l++) { a.g_att_w[l] = t->g_att_w[l]; a.g_att_b[l] = t->g_att_b[l]; }
        a.g_fc1_w = t->g_fc1_w; a.g_fc1_b = t->g_fc1_b; a.g_fc2_w = t->g_fc2_w; a.g_fc2_b = t->g_fc2_b;
        a.g_att_lin_w = t->g_att_lin_w; a.g_att_lin_b = t->g_att_lin_b;
    }
    return GF_OK;
}

}  // namespace

// floats of each of the two scratch buffers (acts, grads) for a window [S, T, C] (strides as gf_cond_t::conv_stride) and dim_aud
GF_EXPORT uint32_t gf_cond_train_scratch_floats(const gf_cond_t* c) {
    if (!c) return 0;
    Dims d = {};
    d.S = (int)c->S; d.T = (int)c->T; d.C = (int)c->C; d.A = (int)c->dim_aud;
    int len = (int)c->T;
    for (int l = 0; l < 5; l++) {
        d.ch[l] = (int)c->conv_ch[l];
        d.len[l] = len;
        if (l < 4) { d.st[l] = (int)c->conv_stride[l] ? (int)c->conv_stride[l] : 1; len = (len + 2 - 3) / d.st[l] + 1; }
    }
    return (uint32_t)layout(d).total;
}

GF_EXPORT int gf_cond_train_forward(const gf_cond_train_t* t, void* stream) {
    TrainArgs a = {};
    if (const int rc = fill(t, a, false)) return rc;
    static GfLdsAttr lds;
    if (const int e = gf_raise_lds_limit(lds, reinterpret_cast<const void*>(k_cond_train_fwd), kSmemBytes, "cond_train_forward")) return e;
    hipLaunchKernelGGL(k_cond_train_fwd, dim3(1), dim3(kThreads), kSmemBytes, gf_stream(stream), a);
    return gf_check_launch("cond_train_forward");
}

GF_EXPORT int gf_cond_train_backward(const gf_cond_train_t* t, void* stream) {
    TrainArgs a = {};
    if (const int rc = fill(t, a, true)) return rc;
    static GfLdsAttr lds;
    if (const int e = gf_raise_lds_limit(lds, reinterpret_cast<const void*>(k_cond_train_bwd_chain), kSmemBytes, "cond_train_backward")) return e;
    hipLaunchKernelGGL(k_cond_train_bwd_chain, dim3(1), dim3(kThreads), kSmemBytes, gf_stream(stream), a);
    hipLaunchKernelGGL(k_cond_train_bwd_weights, dim3(64), dim3(kThreads), 0, gf_stream(stream), a);
    return gf_check_launch("cond_train_backward");
}
