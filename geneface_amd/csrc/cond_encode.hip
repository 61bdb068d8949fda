// Per-frame condition encoder of the RAD-NeRF head for gfx950: one single-workgroup launch replaces the ~40 tiny launches
// (MIOpen convolutions, GEMVs, activations, softmax, cats) torch issues per frame for
//   AudioNet.forward      /root/reference/modules/radnerfs/cond_encoder.py:44-52   ([S, T, C] landmark / audio windows ->
//                          four k=3 Conv1d + LeakyReLU(0.02) shrinking T to 1, FC 64->64 LeakyReLU, FC 64->dim_aud)
//   AudioAttNet.forward   cond_encoder.py:79-89  (five k=3 Conv1d over the S window axis, Linear(S,S), softmax, weighted sum)
// and the two per-frame bias folds of the fused field kernels (frame_head.hip / frame_torso.hip):
//   amb_bias   = W_amb0[:, 32:96] @ cond_feat                      (radnerf.py:80-84: cond_feat is concatenated to every sample)
//   torso_bias = [W_deform0[:, 42:] ; W_canon0[:, 74:]] @ [freq(pose) | identity code]   (radnerf_torso.py:57-80)
// 53 K parameters, ~0.3 MFLOP: latency bound; every layer's weights are staged through LDS with coalesced 16-byte reads.
#include "common.hpp"
#include "sh_core.hpp"
#include "geneface_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxS = 16, kMaxT = 16, kMaxC = 224;
constexpr int kWFloats = kMaxC * 32 * 3;          // largest layer: first conv, C x 32 x 3
constexpr int kActFloats = 7168;                  // one activation buffer: S*T*max(C, 64) floats, enforced on the host
static_assert((kWFloats + 2 * kActFloats + 256) * 4 <= 160 * 1024, "single workgroup, all of LDS");
constexpr int kSmemBytes = (kWFloats + 2 * kActFloats + 256) * 4;

__device__ __forceinline__ float leaky(float x) { return x > 0.0f ? x : 0.02f * x; }

// Copy a layer's weights into LDS: eight 16-byte loads per lane in flight before the first store (the copy is pure latency).
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

// out[s][co][p] = act(b[co] + sum_ci sum_k w[co][ci][k] * in[s][ci][p*stride + k - 1]),  zero padding 1 (torch Conv1d k=3)
// Four adjacent lanes share one output and split the input channels (ci = part, part + 4, ...); only the taps that fall inside
// the window are visited.
__device__ void conv1d_k3(const float* __restrict__ w /*LDS*/, const float* __restrict__ b /*global*/, const float* in, float* out,
                          int S, int cin, int cout, int lin, int lout, int stride) {
    const int total = S * cout * lout;
    const int part = threadIdx.x & 3;
    for (int base = 0; base < total; base += kThreads / 4) {
        const int idx = base + (threadIdx.x >> 2);
        float sum = 0.0f;
        if (idx < total) {
            const int p = idx % lout, co = (idx / lout) % cout, s = idx / (lout * cout);
            const float* wr = w + (size_t)co * cin * 3;
            const float* xr = in + (size_t)s * cin * lin;
            const int pos0 = p * stride - 1;
            const int k_lo = pos0 < 0 ? -pos0 : 0, k_hi = pos0 + 2 >= lin ? lin - 1 - pos0 : 2;   // valid taps k_lo..k_hi
            for (int ci = part; ci < cin; ci += 4)
                for (int k = k_lo; k <= k_hi; k++) sum = __builtin_fmaf(wr[ci * 3 + k], xr[ci * lin + pos0 + k], sum);
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        if (idx < total && part == 0) out[idx] = leaky(sum + b[(idx / lout) % cout]);
    }
}

// One workgroup per frame: workgroup k serves frame k of a batch whose windows, poses and outputs are stacked along the leading axis
// (gf_cond_encode_batch; gf_cond_encode is the batch of one).  A frame's arithmetic does not depend on the batch it is part of.
__global__ void __launch_bounds__(kThreads) k_cond_encode(gf_cond_t a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    {
        const size_t k = blockIdx.x;
        a.cond += k * a.S * a.T * a.C;
        a.cond_feat += k * a.dim_aud;
        if (a.amb_bias) a.amb_bias += k * 128;
        if (a.torso_bias) { a.torso_bias += k * 96; a.pose6 += k * 6; }
    }
    float* wbuf = reinterpret_cast<float*>(smem_raw);
    float* act0 = wbuf + kWFloats;
    float* act1 = act0 + kActFloats;
    float* vec = act1 + kActFloats;   // [256] scratch vectors
    const int tid = threadIdx.x;
    const int S = (int)a.S, T = (int)a.T, C = (int)a.C, A = (int)a.dim_aud;

    // ---- AudioNet: [S, T, C] -> permute(0, 2, 1) -> [S, C, T]
    for (int i = tid; i < S * T * C; i += kThreads) {
        const int c = i % C, t = (i / C) % T, s = i / (C * T);
        act0[(s * C + c) * T + t] = a.cond[i];
    }
    float *cur = act0, *nxt = act1;
    int lin = T;
    for (int l = 0; l < 4; l++) {
        const int cin = (int)a.conv_ch[l], cout = (int)a.conv_ch[l + 1], st = (int)a.conv_stride[l];
        const int lout = (lin + 2 - 3) / st + 1;
        __syncthreads();
        stage(wbuf, a.conv_w[l], cout * cin * 3);
        __syncthreads();
        conv1d_k3(wbuf, a.conv_b[l], cur, nxt, S, cin, cout, lin, lout, st);
        float* t2 = cur; cur = nxt; nxt = t2;
        lin = lout;
    }
    // cur = [S][64][1] (host checked that the window shrinks to 1)
    __syncthreads();
    stage(wbuf, a.fc1_w, 64 * 64);
    __syncthreads();
    for (int idx = tid; idx < S * 64; idx += kThreads) {
        const int o = idx % 64, s = idx / 64;
        float sum = a.fc1_b[o];
        for (int c = 0; c < 64; c++) sum = __builtin_fmaf(wbuf[o * 64 + c], cur[s * 64 + c], sum);
        nxt[idx] = leaky(sum);
    }
    __syncthreads();
    stage(wbuf, a.fc2_w, A * 64);
    __syncthreads();
    float* feat = cur;   // [S][A]
    for (int idx = tid; idx < S * A; idx += kThreads) {
        const int o = idx % A, s = idx / A;
        float sum = a.fc2_b[o];
        for (int c = 0; c < 64; c++) sum = __builtin_fmaf(wbuf[o * 64 + c], nxt[s * 64 + c], sum);
        feat[idx] = sum;
    }
    __syncthreads();

    // ---- AudioAttNet: x[:, :A].permute(1, 0) = [1][A channels][S positions]
    float* y0 = nxt;            // [A][S]
    float* y1 = nxt + A * S;    // ping-pong inside act1 (A*S <= 1024 floats each)
    for (int i = tid; i < A * S; i += kThreads) y0[i] = feat[(i % S) * A + i / S];
    {
        float *yc = y0, *yn = y1;
        int cin = A;
        const int couts[5] = {16, 8, 4, 2, 1};
        for (int l = 0; l < 5; l++) {
            __syncthreads();
            stage(wbuf, a.att_w[l], couts[l] * cin * 3);
            __syncthreads();
            conv1d_k3(wbuf, a.att_b[l], yc, yn, 1, cin, couts[l], S, S, 1);
            float* t2 = yc; yc = yn; yn = t2;
            cin = couts[l];
        }
        __syncthreads();
        // Linear(S, S) + softmax over the S outputs (torch: softmax(dim=1) of [1, S])
        if (tid < S) {
            float sum = a.att_lin_b[tid];
            for (int j = 0; j < S; j++) sum = __builtin_fmaf(a.att_lin_w[tid * S + j], yc[j], sum);
            vec[tid] = sum;
        }
        __syncthreads();
        if (tid == 0) {
            float m = vec[0];
            for (int j = 1; j < S; j++) m = fmaxf(m, vec[j]);
            float den = 0.0f;
            for (int j = 0; j < S; j++) { const float e = expf(vec[j] - m); vec[32 + j] = e; den += e; }
            for (int j = 0; j < S; j++) vec[64 + j] = vec[32 + j] / den;
        }
        __syncthreads();
    }
    // cond_feat[c] = sum_s att[s] * feat[s][c]
    float* cf = vec + 96;   // [A] (A <= 128)
    if (tid < A) {
        float sum = 0.0f;
        for (int s = 0; s < S; s++) sum += vec[64 + s] * feat[s * A + tid];
        cf[tid] = sum;
        a.cond_feat[tid] = sum;
    }
    __syncthreads();

    // ---- per-frame bias folds
    if (a.amb_bias) {
        for (int r = tid; r < 128; r += kThreads) {
            float sum = 0.0f;
            const float* w = a.W_cond + (size_t)r * A;
            for (int c = 0; c < A; c++) sum = __builtin_fmaf(w[c], cf[c], sum);
            a.amb_bias[r] = sum;
        }
    }
    if (a.torso_bias) {
        float* v = wbuf;   // [54 + code_dim]
        const int nv = 54 + (int)a.code_dim;
        __syncthreads();
        if (tid < 54) v[tid] = gf::freq_element(a.pose6, 6, (uint32_t)tid);          // FreqEncoder(input_dim=6, degree=4): freq.py:66-76
        else if (tid < nv) v[tid] = a.torso_code[tid - 54];
        __syncthreads();
        for (int r = tid; r < 96; r += kThreads) {
            float sum = 0.0f;
            const float* w = a.W_tconst + (size_t)r * nv;
            for (int c = 0; c < nv; c++) sum = __builtin_fmaf(w[c], v[c], sum);
            a.torso_bias[r] = sum;
        }
    }
}

}  // namespace

// Validation only (HOST): 0 when gf_cond_encode can serve this encoder / window, else an error code + gf_last_error() text.
// Pointers that change per frame (cond, outputs, pose6) are not examined.
GF_EXPORT int gf_cond_check(const gf_cond_t* c) {
    if (!c) return gf_set_error(GF_ERR_INVALID, "cond_encode: null descriptor");
    if (c->S == 0 || c->S > (uint32_t)kMaxS || c->T == 0 || c->T > (uint32_t)kMaxT || c->C == 0 || c->C > (uint32_t)kMaxC)
        return gf_set_error(GF_ERR_UNSUPPORTED, "cond_encode: window [%u, %u, %u] outside the fused encoder's limits", c->S, c->T, c->C);
    if (c->dim_aud == 0 || c->dim_aud > 128) return gf_set_error(GF_ERR_UNSUPPORTED, "cond_encode: dim_aud must be 1..128");
    if (c->conv_ch[0] != c->C || c->conv_ch[4] != 64) return gf_set_error(GF_ERR_INVALID, "cond_encode: conv channels must run C -> ... -> 64");
    uint32_t len = c->T, biggest = c->S * c->C * c->T;
    for (int l = 0; l < 4; l++) {
        if (!c->conv_w[l] || !c->conv_b[l] || c->conv_stride[l] == 0 || c->conv_ch[l + 1] == 0 || c->conv_ch[l + 1] > 64)
            return gf_set_error(GF_ERR_INVALID, "cond_encode: bad conv layer %d", l);
        if (c->conv_ch[l + 1] * c->conv_ch[l] * 3 > (uint32_t)kWFloats) return gf_set_error(GF_ERR_UNSUPPORTED, "cond_encode: conv layer %d does not fit the LDS weight stage", l);
        len = (len + 2 - 3) / c->conv_stride[l] + 1;
        const uint32_t sz = c->S * c->conv_ch[l + 1] * len;
        biggest = sz > biggest ? sz : biggest;
    }
    if (len != 1) return gf_set_error(GF_ERR_INVALID, "cond_encode: the conv stack must shrink the window to length 1 (got %u)", len);
    if (biggest > (uint32_t)kActFloats || 2 * c->S * c->dim_aud > (uint32_t)kActFloats)
        return gf_set_error(GF_ERR_UNSUPPORTED, "cond_encode: activations of window [%u, %u, %u] exceed the LDS buffers", c->S, c->T, c->C);
    if (!c->fc1_w || !c->fc1_b || !c->fc2_w || !c->fc2_b || !c->att_lin_w || !c->att_lin_b) return gf_set_error(GF_ERR_INVALID, "cond_encode: null FC / attention weights");
    for (int l = 0; l < 5; l++) if (!c->att_w[l] || !c->att_b[l]) return gf_set_error(GF_ERR_INVALID, "cond_encode: null attention conv %d", l);
    if (c->code_dim > 64) return gf_set_error(GF_ERR_UNSUPPORTED, "cond_encode: identity code longer than 64");
    return GF_OK;
}

// cond [S, T, C] -> cond_feat [dim_aud] (= RADNeRF.cal_cond_feat with with_att, radnerf.py:61-71) and, when the output pointers
// are given, the folded first-layer biases of the fused field kernels.  Enqueues one launch on `stream`.
GF_EXPORT int gf_cond_encode(const gf_cond_t* c, void* stream) { return gf_cond_encode_batch(c, 1, stream); }

// The same for n_frames frames in ONE launch (one workgroup per frame): cond [n, S, T, C], pose6 [n, 6] -> cond_feat [n, dim_aud],
// amb_bias [n, 128], torso_bias [n, 96].  Row k is bit-identical to what gf_cond_encode writes for frame k alone.  The frame loop calls it
// once per pass over its shard (all landmark windows are resident before the loop), so no frame waits for its own 70 us single-workgroup launch.
GF_EXPORT int gf_cond_encode_batch(const gf_cond_t* c, uint32_t n_frames, void* stream) {
    if (!c || !c->cond || !c->cond_feat) return gf_set_error(GF_ERR_INVALID, "cond_encode: null pointer");
    if (n_frames == 0) return GF_OK;
    const int rc = gf_cond_check(c);
    if (rc) return rc;
    if (c->amb_bias && !c->W_cond) return gf_set_error(GF_ERR_INVALID, "cond_encode: amb_bias needs W_cond");
    if (c->torso_bias && (!c->W_tconst || !c->pose6 || (c->code_dim && !c->torso_code)))
        return gf_set_error(GF_ERR_INVALID, "cond_encode: torso_bias needs W_tconst, pose6 and the identity code");
    static GfLdsAttr lds;
    if (const int e = gf_raise_lds_limit(lds, reinterpret_cast<const void*>(k_cond_encode), kSmemBytes, "cond_encode")) return e;
    hipLaunchKernelGGL(k_cond_encode, dim3(n_frames), dim3(kThreads), kSmemBytes, gf_stream(stream), *c);
    return gf_check_launch("cond_encode");
}

GF_EXPORT uint64_t gf_cond_sizeof(void) { return sizeof(gf_cond_t); }
