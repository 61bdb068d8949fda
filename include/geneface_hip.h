/*
 * geneface_hip.h -- C ABI of libgeneface_hip.so, the MI355X (gfx950) RAD-NeRF render path for GeneFace.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; tensors are contiguous, row-major, fp32
 *     unless stated; outputs are pre-allocated by the caller and written in place (the convention of the
 *     reference's pybind modules, where every at::Tensor output is allocated by the Python wrapper);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work;
 *   - return value: 0 on success, non-zero otherwise (GF_ERR_*); gf_last_error() returns the message the
 *     reference would have thrown (std::runtime_error / TORCH_CHECK text where one exists).
 * Each entry point cites the reference interface it replaces (paths relative to the GeneFace repository).
 */
#ifndef GENEFACE_HIP_H
#define GENEFACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_OK 0
#define GF_ERR_INVALID 1
#define GF_ERR_HIP 2
#define GF_ERR_UNSUPPORTED 3

const char* gf_last_error(void);
const char* gf_version(void);
int gf_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * `_raymarching_face`   modules/radnerfs/raymarching/src/bindings.cpp:5-21, raymarching.h:7-20
 * ---------------------------------------------------------------------------------------------- */

/* near_far_from_aabb (raymarching.h:7, kernel raymarching.cu:92-145).  rays_o/rays_d [N,3], aabb [6], nears/fars [N]. */
int gf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                          float* nears, float* fars, void* stream);

/* sph_from_ray (raymarching.h:8, kernel raymarching.cu:161-198).  coords [N,2] = (polar, azimuth) of the ray's far intersection
 * with the sphere of `radius`, scaled to [-1,1].  No caller in GeneFace; exported for completeness of the seam. */
int gf_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream);

/* march_rays (raymarching.h:19, kernel raymarching.cu:828-929).  rays_alive i32[>=n_alive], rays_t/nears/fars [N],
 * grid u8[C*H^3/8], xyzs/dirs [n_alive*n_step,3] and deltas [n_alive*n_step,2] ZERO-FILLED by the caller, noises [n_alive]. */
int gf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                  const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                  const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                  const float* noises, void* stream);

/* composite_rays (raymarching.h:20, kernel raymarching.cu:943-1029).  In-place on rays_alive (-1 = terminated), rays_t,
 * weights_sum [N], depth [N], image [N,3]. */
int gf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                      const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                      float* image, void* stream);

/* Training tier.  march_rays_train (raymarching.h:13, kernel raymarching.cu:353-518): rays_o/rays_d [N,3], nears/fars/noises [N];
 * xyzs/dirs [M,3], deltas [M,2] ZERO-FILLED by the caller; rays i32 [N,3] = (ray, point offset, point count); counter i32 [2]
 * accumulates (points, rays).  Offsets are handed out in RAY ORDER (exclusive prefix sum of the counts): deterministic, one of the
 * orders the reference's two atomicAdds per ray can produce.  workspace: gf_march_rays_train_workspace_bytes(N) device bytes. */
uint64_t gf_march_rays_train_workspace_bytes(uint32_t N);
int gf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma, uint32_t max_steps,
                        uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                        float* deltas, int32_t* rays, int32_t* counter, const float* noises, void* workspace, void* stream);
/* march_rays_train_backward (raymarching.h:14, .cu:536-583): grad_rays_o/grad_rays_d [N,3] accumulate. */
int gf_march_rays_train_backward(const float* grad_xyzs, const float* grad_dirs, const int32_t* rays, const float* deltas, uint32_t N,
                                 uint32_t M, float* grad_rays_o, float* grad_rays_d, void* stream);
/* composite_rays_train_forward / _backward (raymarching.h:15-16, .cu:604-687, :712-809); the backward's grad outputs are
 * ZERO-FILLED by the caller. */
int gf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* ambient, const float* deltas, const int32_t* rays,
                                    uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* ambient_sum, float* depth, float* image,
                                    void* stream);
int gf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_ambient_sum, const float* grad_image, const float* sigmas,
                                     const float* rgbs, const float* ambient, const float* deltas, const int32_t* rays, const float* weights_sum,
                                     const float* ambient_sum, const float* image, uint32_t M, uint32_t N, float T_thresh, float* grad_sigmas,
                                     float* grad_rgbs, float* grad_ambient, void* stream);

/* occupancy-grid maintenance: morton3D (raymarching.h:9, .cu:214-226), morton3D_invert (:10, .cu:237-254),
 * packbits (:11, .cu:268-289; N = number of output bytes), morton3D_dilation (:12, .cu:304-335). */
int gf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream);
int gf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream);
int gf_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream);
int gf_morton3D_dilation(const float* grid, uint32_t C, uint32_t H, float* grid_dilation, void* stream);

/* ------------------------------------------------------------------------------------------------
 * `_gridencoder`   modules/radnerfs/encoders/gridencoder/src/bindings.cpp:5-9, gridencoder.h:11
 * ---------------------------------------------------------------------------------------------- */

/* grid_encode_forward (kernel gridencoder.cu:88-244).  inputs [B,D] in [0,1]; embeddings [sO,C]; offsets i32[L+1];
 * outputs [L,B,C]; S = log2(per_level_scale); H = base resolution; dy_dx NULL or [B, L*D*C];
 * gridtype 0 hash / 1 tiled; interp 0 linear / 1 smoothstep.  D in {2,3,4,5}, C in {1,2,4,8}. */
int gf_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                           uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype,
                           int align_corners, uint32_t interp, void* stream);
/* grid_encode_backward (gridencoder.h:12, kernels gridencoder.cu:248-368).  grad [L,B,C]; grad_embeddings [sO,C] ZERO-FILLED by the
 * caller, accumulated with f32 atomics; dy_dx [B, L*D*C] (from the forward) and grad_inputs [B,D]: both NULL or both given. */
int gf_grid_encode_backward(const float* grad, const float* inputs, const float* embeddings, const int32_t* offsets, float* grad_embeddings,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                            uint32_t gridtype, int align_corners, uint32_t interp, void* stream);
/* grad_total_variation (gridencoder.h:14, kernel gridencoder.cu:505-596).  grad [sO,C] is ACCUMULATED into (f32 atomics): the
 * normalised total-variation gradient of the table around the lattice node every input [B,D] in [0,1] falls on, times weight/(2D). */
int gf_grad_total_variation(const float* inputs, const float* embeddings, float* grad, const int32_t* offsets, float weight, uint32_t B,
                            uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, void* stream);
/* The same forward for a HALF table (binary16 bit patterns): the branch the reference reaches through AT_DISPATCH_FLOATING_TYPES_AND_HALF
 * (gridencoder.cu:375-398) when autocast is on -- its wrapper casts the table to half on every call (grid.py:41-44).  Rows are widened to fp32
 * on load; inputs, arithmetic, outputs and dy_dx stay fp32 (the caller narrows them if its buffers are half).  C must be 2, 4 or 8. */
int gf_grid_encode_forward_f16(const float* inputs, const uint16_t* embeddings_f16, const int32_t* offsets, float* outputs, uint32_t B,
                               uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners,
                               uint32_t interp, void* stream);
/* same arithmetic as gf_grid_encode_forward, outputs laid out [B, L*C] (what GridEncoder.forward returns after grid.py:57's permute). */
int gf_grid_encode_forward_blc(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                               uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, float* dy_dx, uint32_t gridtype,
                               int align_corners, uint32_t interp, void* stream);
/* The fused head kernel's specialised 16-level C=2 lookup (strides / mask / hash flag per level), stand-alone: inputs [B,D] in [0,1], outputs [B,32].  Same results as gf_grid_encode_forward_blc; exists so
 * tests can compare the two point by point (level sizes must satisfy gf_grid_levels_fusable). */
int gf_grid_encode_fused_lookup(const float* inputs, const float* embeddings, const int32_t* offsets, float* outputs, uint32_t B,
                                uint32_t D, float S, uint32_t H, uint32_t gridtype, uint32_t interp, void* stream);
/* HOST helper: per-level scale / resolution exactly as the kernels use them (gridencoder.cu:138-139). */
int gf_grid_level_meta(uint32_t L, float S, uint32_t H, float* scale_out_host, uint32_t* resolution_out_host);

/* ------------------------------------------------------------------------------------------------
 * `_shencoder`   modules/radnerfs/encoders/shencoder/src/bindings.cpp, shencoder.h:9
 * `_freqencoder` modules/radnerfs/encoders/freqencoder/src/bindings.cpp, freqencoder.h:7
 * ---------------------------------------------------------------------------------------------- */

/* sh_encode_forward (kernel shencoder.cu:28-356).  inputs [B,3], outputs [B,degree^2]; degree 1..8 like the reference (bands 4..7: table driven,
 * csrc/sh_core.hpp::sh_high); dy_dx NULL or [B,3,degree^2]. */
int gf_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t degree, float* dy_dx, void* stream);
/* sh_encode_backward (shencoder.h:10, kernel shencoder.cu:359-383).  grad_inputs [B,3] accumulates. */
int gf_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t degree, const float* dy_dx,
                          float* grad_inputs, void* stream);

/* freq_encode_forward (kernel freqencoder.cu:30-58).  inputs [B,D], outputs [B,C], C = D + 2*D*deg. */
int gf_freq_encode_forward(const float* inputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* outputs, void* stream);
/* freq_encode_backward (freqencoder.h:10, kernel freqencoder.cu:63-94).  grad / outputs [B,C] -> grad_inputs [B,D]. */
int gf_freq_encode_backward(const float* grad, const float* outputs, uint32_t B, uint32_t D, uint32_t deg, uint32_t C, float* grad_inputs,
                            void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused frame path.  One call enqueues a whole head (and torso) pass of
 *   NeRFRenderer.render    modules/radnerfs/renderer.py:263-367 (inference branch :314-367)
 *   RADNeRF.forward        modules/radnerfs/radnerf.py:73-105   (evaluated inside the march iterations)
 *   RADNeRFTorso.render    modules/radnerfs/radnerf_torso.py:86-198, forward_torso :51-84
 * with no host synchronisation: the alive list, n_alive and the n_step schedule (renderer.py:338) stay on the device.
 * Weights are handed over once in the packed layouts produced by gf_head_pack / gf_torso_pack (HOST functions).
 * ---------------------------------------------------------------------------------------------- */
typedef struct gf_frame_t {
    /* rays: explicit tensors, or generated from pose + intrinsics when rays_o == NULL (utils.py:282-363, N = -1 branch) */
    uint32_t n_rays, img_h, img_w, _pad0;
    const float* rays_o;        /* [N,3] or NULL */
    const float* rays_d;        /* [N,3] or NULL */
    float pose[12];             /* row-major 3x4 cam2world, ngp axes */
    float intrinsics[4];        /* fx, fy, cx, cy */
    /* marcher state (renderer.py:65-99) */
    const float* aabb;          /* [6] aabb_infer */
    const uint8_t* bitfield;    /* density_bitfield [cascade*grid^3/8] */
    float min_near, bound, dt_gamma, T_thresh;
    uint32_t max_steps, cascade, grid_size, has_occ_aabb;
    float occ_aabb[6];          /* gf_occupancy_aabb(bitfield): box around the occupied cells; has_occ_aabb = 0 -> march to aabb's far */
    float _pad1[2];
    /* head field (radnerf.py:41-59) */
    const float* pos_table;  const int32_t* pos_offsets;   /* position_embedder.embeddings / offsets (3-D, 16x2) */
    const float* amb_table;  const int32_t* amb_offsets;   /* ambient_embedder.embeddings / offsets  (2-D, 16x2) */
    float pos_S, amb_S;         /* log2(per_level_scale) of each grid */
    uint32_t base_res, gridtype, interp;
    uint32_t precision;         /* 0: fp32 everywhere (strict parity).  1: "fast" -- f16 MFMA operands and activations, fp32 accumulate
                                   (what the reference's autocast / .half() viewer path computes); needs head_pack16.
                                   2: "split" -- fp32 VALUES carried as two-term f16 splits (hi + lo' * 2^-11) on the f16 matrix pipe, three MFMAs
                                   per product term set, fp32 accumulate: fp32-level accuracy (strict tolerance), not fp32 bit patterns; needs
                                   head_pack_split */
    const float* head_pack;     /* device copy of gf_head_pack() output */
    const uint16_t* head_pack16;/* device copy of gf_head_pack16() output, or NULL when precision != 1 */
    const uint16_t* head_pack_split;/* device copy of gf_head_pack_split() output, or NULL when precision != 2 */
    const float* amb_bias;      /* [128] ambient_net.net.0.weight[:, 32:96] @ cond_feat, rows permuted by gf_clayout_perm */
    /* torso field (radnerf_torso.py:20-49); torso_pack == NULL renders the head only */
    const float* torso_pack;    /* device copy of gf_torso_pack() output */
    const float* torso_bias;    /* [96] folded per-frame constants (deform L1 64 | canonical L1 32), permuted rows */
    const float* torso_table; const int32_t* torso_offsets;
    const float* torso_occ;     /* density_grid_torso [grid*grid] */
    const float* bg_coords;     /* [N,2]; any finite value: beyond |bg_coords * torso_shrink| = 15.9 the encodings use the full-range sine, like gf_freq_encode_forward */
    float torso_S, torso_thresh, torso_shrink, _pad3;
    /* compositing */
    const float* bg_color;      /* [N,3] */
    float* out_rgb;             /* [N,3] rgb_map */
    float* out_depth;           /* [N]   depth_map */
    uint8_t* out_rgb8;          /* [N,3] or NULL: (rgb*255) truncated, base_nerf_infer.py:96-97 */
    float* out_torso_alpha;     /* [N]   or NULL: torso_alpha_map */
    float* out_torso_rgb;       /* [N,3] or NULL: torso_rgb_map */
    float* out_deform;          /* [N,2] or NULL: dx of the masked pixels (others untouched) */
    void* workspace;            /* gf_frame_workspace_bytes(n_rays) bytes, 256-byte aligned */
    /* perturb=True at inference (renderer.py:338-342: `perturb if step == 0 else False`): the FIRST march iteration starts every ray at
     * t = near + clamp(near * dt_gamma, dt_min, dt_max) * noise (raymarching.cu:851).  [N] U[0,1) draws in ray order, or NULL = no jitter. */
    const float* perturb_noise;
    /* torso_head_aware (radnerf_torso.py:36-46,68-74,175-179): device copy of gf_torso_pack_ha()'s second output, or NULL.  With
     * torso_ha_branch = 1 the torso field of a masked pixel also sees the head's accumulated colour and opacity at that pixel through
     * head_color_weights_encoder; the other outcome of the reference's per-frame coin (zeros in) is a per-frame constant the host folds
     * into torso_bias, so it needs nothing here. */
    const float* torso_ha_pack;
    float* torso_ha_ws;         /* [N,16] scratch for the encoder's outputs; needed when torso_ha_branch = 1 */
    uint32_t torso_ha_branch, _pad4;
    /* The torso mask as a dense list (round 5): which pixels `grid_sample(density_grid_torso, bg_coords) > torso_thresh` selects
     * (radnerf_torso.py:166-172) is a property of (bg_coords, torso_occ, torso_thresh, n_rays) -- constant over a frame loop.  A caller that
     * renders many frames builds the list once with gf_torso_mask_list() and hands it in here; with torso_mask_list == NULL gf_render_torso
     * builds it for every frame in the workspace.  All three device pointers or none. */
    const uint32_t* torso_mask_list;      /* [count] pixel indices of the masked pixels (any order) */
    const uint32_t* torso_mask_dense_of;  /* [N] position of pixel n in the list, 0xFFFFFFFF when the mask does not select it */
    const uint32_t* torso_mask_count;     /* [1] length of the list */
} gf_frame_t;

/* gf_frame_t.torso_mask_*: the masked pixels of a torso pass as a dense list, built once for many frames.  bg_coords [N,2], torso_occ [G*G],
 * list / dense_of [N] uint32, count [1] uint32 (overwritten).  Same mask arithmetic as gf_render_torso (one device function). */
int gf_torso_mask_list(const float* bg_coords, const float* torso_occ, uint32_t n_rays, uint32_t grid_size, float torso_thresh, uint32_t* list,
                       uint32_t* dense_of, uint32_t* count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-frame condition encoder: RADNeRF.cal_cond_feat (modules/radnerfs/radnerf.py:61-71) =
 *   AudioNet.forward (cond_encoder.py:44-52) + AudioAttNet.forward (:79-89), in one launch, plus the two bias folds the
 *   fused field kernels consume (amb_bias / torso_bias below).  All weights are the nn.Module tensors as they are.
 * ---------------------------------------------------------------------------------------------- */
typedef struct gf_cond_t {
    const float* cond;                  /* [S, T, C] window (smo_win_size, cond_win_size, cond dim), e.g. [5,1,204] */
    uint32_t S, T, C, dim_aud;
    const float* conv_w[4];             /* cond_prenet.encoder_conv.{0,2,4,6}.weight [Cout, Cin, 3] */
    const float* conv_b[4];             /* ....bias */
    uint32_t conv_stride[4];            /* cond_encoder.py:14-41: strides by window size */
    uint32_t conv_ch[5];                /* C, 32, 32, 64, 64 */
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;   /* cond_prenet.encoder_fc1.{0,2} */
    const float* att_w[5];              /* cond_att_net.attentionConvNet.{0,2,4,6,8}.weight: dim_aud->16->8->4->2->1 */
    const float* att_b[5];
    const float *att_lin_w, *att_lin_b; /* cond_att_net.attentionNet.0: Linear(S, S) */
    float* cond_feat;                   /* out [dim_aud] */
    const float* W_cond;                /* [128, dim_aud] = ambient_net.net.0.weight[gf_clayout_perm, 32:]; NULL with amb_bias NULL */
    float* amb_bias;                    /* out [128] or NULL */
    const float* pose6;                 /* [6] euler + translation (utils.py:263-269) */
    const float* torso_code;            /* [code_dim] torso_individual_codes[0] or NULL */
    uint32_t code_dim, _pad;
    const float* W_tconst;              /* [96, 54 + code_dim] per-frame-constant columns of the torso first layers */
    float* torso_bias;                  /* out [96] or NULL */
} gf_cond_t;
uint64_t gf_cond_sizeof(void);
int gf_cond_check(const gf_cond_t* cond);            /* HOST: can gf_cond_encode serve this encoder / window? (no launch) */
int gf_cond_encode(const gf_cond_t* cond, void* stream);
/* n_frames frames in one launch (one workgroup each): cond [n,S,T,C], pose6 [n,6] -> cond_feat [n,dim_aud], amb_bias [n,128], torso_bias [n,96];
 * row k equals gf_cond_encode on frame k alone, bit for bit.  What the frame loop of base_nerf_infer.py:81-106 runs per frame
 * (tasks/radnerfs/radnerf.py:119-128 -> cal_cond_feat) hoisted in front of the loop: every window of the shard is resident by then. */
int gf_cond_encode_batch(const gf_cond_t* cond, uint32_t n_frames, void* stream);

/* The same encoder under TRAINING (round 6): cal_cond_feat as one forward launch that keeps every layer's activations and one backward launch
 * that writes the gradient of all 24 parameter tensors (two launches: the chain, then the weight sums on a grid) (cond_encoder.py:7-89 under autograd: ~100 torch launches per step).  fp32.
 *   enc      the encoder as for gf_cond_encode (cond, S, T, C, dim_aud, weights; cond_feat = the forward's output; the bias folds are not
 *            used).  att_lin_w = NULL: no attention net (with_att = false), S must be 1.  S <= 16, dim_aud <= 64.
 *   acts / grads   gf_cond_train_scratch_floats(enc) floats each: the forward fills acts, the backward reads acts and uses grads.
 *   backward g_feat [dim_aud] in; every g_* out, shaped like its weight, fully written (no accumulation). */
typedef struct gf_cond_train {
    const gf_cond_t* enc;
    float* acts; float* grads;
    const float* g_feat;
    float* g_conv_w[4]; float* g_conv_b[4];
    float* g_fc1_w; float* g_fc1_b; float* g_fc2_w; float* g_fc2_b;
    float* g_att_w[5]; float* g_att_b[5];
    float* g_att_lin_w; float* g_att_lin_b;
} gf_cond_train_t;
uint32_t gf_cond_train_scratch_floats(const gf_cond_t* enc);
int gf_cond_train_forward(const gf_cond_train_t* t, void* stream);
int gf_cond_train_backward(const gf_cond_train_t* t, void* stream);

uint64_t gf_frame_sizeof(void);
uint64_t gf_frame_workspace_bytes(uint32_t n_rays);
uint64_t gf_frame_ctrl_offset(uint32_t n_rays);   /* byte offset of the uint32 control block inside the workspace */
uint32_t gf_frame_ctrl_words(void);               /* word layout: geneface_amd/csrc/frame.hpp (queue heads, counts, terminal-index histogram).
                                                      Size of the block.  A frame DEFINES (clears, then writes) only the words below
                                                      64 + max_steps + 2 -- the histogram is as long as the frame's max_steps; words beyond
                                                      that keep whatever an earlier frame with a larger max_steps left there (a viewer
                                                      moving its step slider from 1024 to 16) and must not be read. */
/* Byte offset of one per-ray array inside the workspace, for tests that inspect what k_frame_init derived from the pose (the
 * reference materialises the same arrays: get_rays utils.py:282-363, near_far_from_aabb raymarching.cu:92-145).
 * field: 0 nears [N], 1 fars [N], 2 rays_t [N], 3 weights_sum [N], 4 depth [N], 5 image [N,3], 6 rays_o [N,3], 7 rays_d [N,3],
 * 8 far_occ [N], 9 hit list int32 [N].  Fields 2, 7, 8 are written only for the rays of the hit list (the first ctrl[1] entries of
 * field 9: rays with at least one sample), field 6 only for those and only when explicit rays were passed in (rays generated from
 * the pose share one origin, which travels by value).  Returns UINT64_MAX for an unknown field. */
uint64_t gf_frame_field_offset(uint32_t n_rays, uint32_t field);
/* get_rays, full-image branch (modules/radnerfs/utils.py:282-363 with N = -1; the reference's dataset calls it on the GPU,
 * tasks/radnerfs/dataset_utils.py:172-178) in one launch: rays_o, rays_d [img_h*img_w, 3], row-major pixels.  pose12_host = the 3x4
 * cam2world (row-major, ngp axes), intrinsics4_host = fx, fy, cx, cy: HOST pointers, read before the call returns.  Bit for bit the rays a
 * pose-mode frame (gf_frame_t.rays_o == NULL) generates for itself: both run the same device function. */
int gf_pinhole_rays(const float* pose12_host, const float* intrinsics4_host, uint32_t img_h, uint32_t img_w, float* rays_o, float* rays_d,
                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * Occupancy-grid maintenance on the device: NeRFRenderer.update_extra_state (renderer.py:199-260) in three launches,
 * mark_untrained_grid (:129-196) in one.  The reference walks cell blocks and cascades in Python, one field query per block.
 * gf_grid_density: sigma * density_scale of every cell of every cascade (RADNeRF.density, radnerf.py:107-126, on the matrix pipe) into
 *   tmp_grid [cascade][grid_size^3], Morton order.  Reads f's tables, level scales, packed head weights, amb_bias, bound, cascade,
 *   grid_size, gridtype, interp.  noise_or_null: [cascade][grid_size^3][3] U[0,1) jitter in meshgrid order (x slowest), NULL = centres.
 * gf_grid_update: Morton dilation of tmp_grid (raymarching.cu:300-341) + EMA-max into density_grid + mean of clamp(grid, 0) (fixed
 *   summation order) + packbits with min(mean, density_thresh) (raymarching.cu:268-289).  partial_ws: gf_grid_update_ws_bytes() bytes of
 *   device scratch; stats_dev[0] = mean_density, [1] = threshold used.
 * gf_mark_untrained_grid: cells outside every camera frustum get density -1.  poses: device [B,4,4] c2w, ngp axes. */
int gf_grid_density(const gf_frame_t* f, const float* noise_or_null, float density_scale, float* tmp_grid, void* stream);
/* RADNeRF.forward (radnerf.py:73-105) for a dense point list in one launch, inference arithmetic (no autograd): what the reference runs
 * as ~40 launches per call for the viewer, for the frozen head of torso training (radnerf_torso.py:97-150, under no_grad) and for any
 * eval-mode model(x, d, cond_feat, code).  xyz, dirs [M,3]; sigma [M], rgb [M,3], ambient_or_null [M,2] (the tanh output).
 * col_bias_or_null: [128] = W_color0[:, 144:148] @ individual_code in accumulator order (gf_clayout_perm); NULL = the code packed into
 * f->head_pack.  Reads f's tables, level scales, head_pack, amb_bias, bound, gridtype, interp. */
int gf_field_forward(const gf_frame_t* f, const float* xyz, const float* dirs, uint32_t M, const float* col_bias_or_null,
                     float* sigma, float* rgb, float* ambient_or_null, void* stream);
/* Training forward of the same field (radnerf.py:73-105 under autograd): the one launch also leaves what the backward pass needs of
 * every layer -- post-ReLU activations and the two grid feature sets -- as row-major [M, width] fp32 matrices. */
typedef struct gf_field_saves {
    float* f3;    /* [M,32]  3-D grid features (position_embedder output) */
    float* ha1;   /* [M,128] ambient_net layer 0 output after ReLU */
    float* ha2;   /* [M,128] ambient_net layer 1 output after ReLU */
    float* f2;    /* [M,32]  2-D grid features (ambient_embedder output) */
    float* hs1;   /* [M,128] sigma_net layer 0 output after ReLU */
    float* hs2;   /* [M,128] sigma_net layer 1 output after ReLU */
    float* geo;   /* [M,128] geometry feature (sigma_net output rows 1..128, no activation) */
    float* hc1;   /* [M,128] color_net layer 0 output after ReLU */
    /* optional (all five or none): ReLU masks of ha1, ha2, hs1, hs2, hc1 for gf_field_backward, 2 bytes per lane in the kernels'
     * accumulator layout: uint16 [ceil(M/128)][4 tiles][4 waves][64 lanes], bit r set when the lane's r-th value of that tile is > 0 */
    uint16_t* m_ha1; uint16_t* m_ha2; uint16_t* m_hs1; uint16_t* m_hs2; uint16_t* m_hc1;
    float* sh;    /* optional: [M,16] SH basis of the directions (the other input of color_net layer 0) */
} gf_field_saves_t;
int gf_field_forward_train(const gf_frame_t* f, const float* xyz, const float* dirs, uint32_t M, const float* col_bias_or_null,
                           float* sigma, float* rgb, float* ambient, const gf_field_saves_t* saves, void* stream);
/* The same launch on the f16 tier (round 6; the arithmetic of the reference's AMP training: /root/reference/egs/egs_bases/radnerf/base.yaml:49
 * amp: true, utils/commons/trainer.py:307-382 autocast + GradScaler, cond_encoder.py:106-111 Linear layers in half): f16 MFMA operands
 * (f->head_pack16 beside f->head_pack), fp32 accumulation, fp32 outputs.  The nine matrices of `saves` are BINARY16 here (the activations
 * the MFMAs consumed), the five masks as above and all required. */
int gf_field_forward_train16(const gf_frame_t* f, const float* xyz, const float* dirs, uint32_t M, const float* col_bias_or_null,
                             float* sigma, float* rgb, float* ambient, const gf_field_saves_t* saves, void* stream);
/* The input-gradient (dX) chain of the same field in one launch: from the gradients of the three outputs back through every layer, the
 * ReLU derivatives taken from the forward's mask bits, the 2-D lookup's input gradient re-gathered.  It writes the pre-activation gradient
 * of every layer (what the weight gradients are tall products of, with the saved activations) and the gradients of both grid feature sets
 * (what gf_grid_encode_backward scatters into the tables).  bwd_stream: gf_field_bwd_stream_floats() floats = the transposed blocks
 * W_color0[:, 16:144]^T, W_sigma2[1:]^T, W_sigma1^T, W_sigma0^T (64 rows, zero-padded to 128), W_ambient1^T, W_ambient0[:, :32]^T (32 rows,
 * padded) as A-operand streams [wave 4][layer 6][group 16][lane 64][4]: element = Wt[32 wave + (lane & 31)][8 group + 4 (lane >> 5) + i]. */
typedef struct gf_field_grads {
    const float* g_sigma; const float* g_rgb; const float* g_amb;   /* in: [M], [M,3], [M,2] */
    const float* sigma; const float* rgb; const float* amb;         /* in: the forward's outputs */
    const uint16_t* m_hc1; const uint16_t* m_hs2; const uint16_t* m_hs1; const uint16_t* m_ha2; const uint16_t* m_ha1;   /* in: forward masks */
    float* g_zc; float* g_h0; float* g_za;                          /* out: [M,3] colour pre-sigmoid, [M] log-density, [M,2] ambient pre-tanh */
    float* g_hc1; float* g_geo; float* g_hs2; float* g_hs1; float* g_ha2; float* g_ha1;   /* out: [M,128] each, pre-activation gradients */
    float* g_f3; float* g_f2;                                       /* out: [16 levels][M][2] each (the [L,B,C] layout gf_grid_encode_backward reads) */
    float* s_hc1; float* s_ha1;                                     /* out, ZEROED by the caller: [128] column sums of g_hc1 / g_ha1 over the points */
    uint32_t* level_max;                                            /* out or NULL, ZEROED by the caller: [2][16] max |g_f3| (first 16) and |g_f2| per level,
                                                                       as bit patterns of non-negative floats: what gf_grid_encode_backward_scaled takes */
    uint32_t out16; uint32_t _pad;                                  /* 1: the six [M,128] outputs are binary16 (AMP tier: half operands for the weight-gradient
                                                                       products); 2: also the chain itself on the f16 matrix pipe -- bwd_stream then points at
                                                                       gf_field_bwd16_stream_halves() binary16 values: the same six transposed blocks as
                                                                       [wave 4][layer 6][group 8][lane 64][8], element = Wt[32 wave + (lane & 31)][16 group + 8 (lane >> 5) + i] */
} gf_field_grads_t;
uint32_t gf_field_bwd_stream_floats(void);
uint32_t gf_field_bwd16_stream_halves(void);
int gf_field_backward(const gf_frame_t* f, const float* bwd_stream, uint32_t M, const gf_field_grads_t* g, void* stream);
/* The weight gradients of the same field on the AMP tier (round 6): the eight tall products dW = G^T X that autograd derives for the Linear
 * layers of radnerf.py:73-105 (cond_encoder.py:106-111) under the trainer's autocast (utils/commons/trainer.py:307-382: half operands, fp32
 * accumulation), in one launch + one fixed-order reduction -- every binary16 row read once.  In: the nine binary16 saves of
 * gf_field_forward_train16, the six binary16 [M,128] pre-activation gradients of gf_field_backward (out16 != 0) and its fp32 g_zc [M,3],
 * g_h0 [M], g_za [M,2] (rounded to binary16 inside, as a half GEMM operand would be).  Out (fp32, fully written for the listed blocks):
 *   gw_color1 [3,128]; gw_color0 [128, ld_color0] columns 0..143 (SH | geometry feature; the identity-code columns are the caller's outer
 *   product); gw_sigma2 [129,128] (row 0 = the density row); gw_sigma1 [128,128]; gw_sigma0 [128,64] (3-D | 2-D grid features);
 *   gw_ambient2 [2,128]; gw_ambient1 [128,128]; gw_ambient0 [128, ld_ambient0] columns 0..31 (the condition columns are the caller's).
 * workspace: gf_field_wgrad16_ws_bytes() bytes of device scratch.  The result does not depend on the run (no atomics). */
typedef struct gf_field_wgrad {
    const void* f3; const void* ha1; const void* ha2; const void* f2; const void* hs1; const void* hs2; const void* geo; const void* hc1; const void* sh;
    const void* g_hc1; const void* g_geo; const void* g_hs2; const void* g_hs1; const void* g_ha2; const void* g_ha1;
    const float* g_zc; const float* g_h0; const float* g_za;
    float* gw_color1; float* gw_color0; float* gw_sigma2; float* gw_sigma1; float* gw_sigma0; float* gw_ambient2; float* gw_ambient1; float* gw_ambient0;
    uint32_t ld_color0; uint32_t ld_ambient0;
    float* workspace;
} gf_field_wgrad_t;
uint64_t gf_field_wgrad16_ws_bytes(void);
int gf_field_wgrad16(uint32_t M, const gf_field_wgrad_t* w, void* stream);
/* The same products for the exact tier: every matrix fp32 (gf_field_forward_train's saves, gf_field_backward's outputs with out16 = 0),
 * v_mfma_f32_32x32x2_f32 -- an exact fmaf chain per accumulator, partial sums per workgroup added in workgroup order. */
uint64_t gf_field_wgrad32_ws_bytes(void);
int gf_field_wgrad32(uint32_t M, const gf_field_wgrad_t* w, void* stream);
/* gf_grid_encode_backward (gridencoder.cu:248-339) for a gradient already in [L, B, C] order whose per-level max |g| is known on the device
 * (level_max[L]; gf_field_backward produces both): the table scatter without its max pass; no input gradient.  D = 2 or 3. */
int gf_grid_encode_backward_scaled(const float* grad, const float* inputs, const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                   uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                   const uint32_t* level_max, void* stream);
/* The same scatter with a BINNING PASS in front (round 6): one lane per point walks the levels once and appends the point to the list of every
 * row partition its corners touch; each scatter workgroup then reads its own partition's list instead of examining every point (the reference
 * has no counterpart: its kernel issues one global atomic per point, level, corner and channel, gridencoder.cu:248-339).  workspace:
 * gf_grid_backward_ws_bytes(B, L) bytes of device scratch.  Same table gradient as gf_grid_encode_backward_scaled. */
uint64_t gf_grid_backward_ws_bytes(uint32_t B, uint32_t L);
int gf_grid_encode_backward_binned(const float* grad, const float* inputs, const int32_t* offsets, float* grad_embeddings, uint32_t B, uint32_t D,
                                   uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, uint32_t interp,
                                   const uint32_t* level_max, void* workspace, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Torso field for TRAINING (round 6): RADNeRFTorso.forward_torso (modules/radnerfs/radnerf_torso.py:51-84) on a list of M pixel coordinates
 * as two launches -- forward with every layer's activations saved, and the input-gradient chain -- replacing the ~60 + ~120 torch launches
 * of the op graph in the torso task's step (tasks/radnerfs/radnerf_torso.py:74-122; the default architecture, torso_head_aware = false).
 * Same building blocks as gf_render_torso's field kernel (f32 MFMA, register-chained layers, the 2-D tiled grid lookup).
 *   forward : x [M,2] = bg_coords of the masked pixels (shrunk inside, :57) -> out [M,4] = sigmoid(alpha, r, g, b), dx [M,2], and the saves:
 *             enc [M,48] frequency encoding of x (42 entries + 6 zeros), h_d1 / h_d2 [M,64] deform net activations after ReLU, x01 [M,2] the
 *             clamped canonical coordinate mapped to [0,1] (what the grid encoder was evaluated at), g [M,32] its grid features,
 *             h_c1 / h_c2 [M,32] canonical net activations after ReLU.  torso_pack / torso_bias as gf_frame_t's.
 *   backward: g_out [M,4], g_dx [M,2] or NULL in; pre-activation gradients out: dz_c3 [M,4], dz_c2 / dz_c1 [M,32], dz_d3 [M,2] (the total
 *             gradient of dx: through the grid lookup's input gradient and the clamp, plus g_dx), dz_d2 / dz_d1 [M,64], the grid feature
 *             gradient g_grid [16 levels][M][2] (the layout gf_grid_encode_backward_scaled reads) and its per-level max |g| (level_max [16],
 *             ZEROED by the caller).  Weight gradients are tall products of these with the saves (host side, geneface_amd/train_torso.py).
 *             bwd_streams: gf_torso_bwd_stream_floats() floats = the transposed blocks W_c2^T | W_c1[:, grid columns]^T (rows in the lane
 *             order of the lookup: gf_torso_bwd_grid_row_perm) | W_d2^T as A-operand streams (gf_mlp_stream_pack).
 * ---------------------------------------------------------------------------------------------- */
typedef struct gf_torso_train {
    uint32_t M; float torso_shrink; float torso_S; uint32_t base_res;
    const float* x;                                  /* [M,2] */
    const float* torso_pack; const float* torso_bias; const float* torso_table; const int32_t* torso_offsets;
    float* out; float* dx;                           /* [M,4], [M,2]: written by the forward, read by the backward */
    float* enc; float* h_d1; float* h_d2; float* x01; float* g; float* h_c1; float* h_c2;   /* saves (forward out, backward in: h_*, x01) */
    /* backward only */
    const float* bwd_streams; const float* g_out; const float* g_dx;
    float* dz_c3; float* dz_c2; float* dz_c1; float* dz_d3; float* dz_d2; float* dz_d1; float* g_grid; uint32_t* level_max;
} gf_torso_train_t;
int gf_torso_train_forward(const gf_torso_train_t* t, void* stream);
int gf_torso_train_backward(const gf_torso_train_t* t, void* stream);
/* The weight gradients of the same field in two launches (round 6): the seven tall products dZ^T X, the column sums behind the 62 per-frame-constant
 * columns of both first layers, and d v.  In: the saves of gf_torso_train_forward and the dz_* of gf_torso_train_backward (row-major, as those
 * calls leave them), v [62] = [frequency encoding of the pose | identity code], the two first-layer weights [64,104] / [32,136].
 * Out, fully written: g_wd1 [64,104], g_wd2 [64,64], g_wd3 [2,64], g_wc1 [32,136], g_wc2 [32,32], g_wc3 [4,32], g_v [62] (d code = g_v[54:]).
 * workspace: gf_torso_wgrad_ws_bytes() bytes.  No atomics: the same bits every run. */
typedef struct gf_torso_wgrad {
    uint32_t M; uint32_t _pad;
    const float* enc; const float* h_d1; const float* h_d2; const float* g; const float* h_c1; const float* h_c2;
    const float* dz_d1; const float* dz_d2; const float* dz_d3; const float* dz_c1; const float* dz_c2; const float* dz_c3;
    const float* v; const float* w_d1; const float* w_c1;
    float* g_wd1; float* g_wd2; float* g_wd3; float* g_wc1; float* g_wc2; float* g_wc3; float* g_v;
    float* workspace;
} gf_torso_wgrad_t;
uint64_t gf_torso_wgrad_ws_bytes(void);
int gf_torso_wgrad(const gf_torso_wgrad_t* w, void* stream);
/* The tail of RADNeRFTorso.render's training branch (radnerf_torso.py:181-192) in one launch each way: alpha = a m, colour = c m (m = the torso
 * mask as 0 / 1), torso_rgb = colour alpha + bg (1 - alpha), rgb = clamp(image + (1 - weights_sum) torso_rgb, 0, 1); every operation rounded on
 * its own in the order of the torch expressions.  a [N], c [N,3], mask [N], bg [N,3] (bg_stride 3) or one colour [3] (bg_stride 0), image [N,3],
 * weights_sum [N] -> torso_alpha [N], torso_rgb [N,3], rgb [N,3].  Backward: g_alpha [N] / g_torso_rgb [N,3] / g_rgb [N,3] (each or NULL) ->
 * g_a [N], g_c [N,3] (the head is frozen in the torso task: image, weights_sum and bg are data). */
typedef struct gf_torso_blend {
    uint32_t N; uint32_t bg_stride;
    const float* a; const float* c; const float* mask; const float* bg; const float* image; const float* weights_sum;
    float* torso_alpha; float* torso_rgb; float* rgb;
    const float* g_alpha; const float* g_torso_rgb; const float* g_rgb;
    float* g_a; float* g_c;
} gf_torso_blend_t;
int gf_torso_blend_train_forward(const gf_torso_blend_t* t, void* stream);
int gf_torso_blend_train_backward(const gf_torso_blend_t* t, void* stream);
uint32_t gf_torso_bwd_stream_floats(void);
/* HOST: one weight matrix W [nob*32 rows][ld] as an MFMA A-operand stream [ob][step/4][lane][step%4] whose step t consumes the hidden feature
 * of accumulator position t (hidden -> hidden layers: the layout of gf_torso_pack's second layers); out [nob * nsteps * 64] floats */
int gf_mlp_stream_pack(const float* W_host, uint32_t ld, uint32_t nob, uint32_t nsteps, float* out_host);
/* HOST: perm[32]: row p of the transposed grid block W_c1[:, 0:32]^T handed to gf_mlp_stream_pack must be grid feature perm[p], so that a
 * lane's accumulator registers hold the gradients of ITS eight levels (16 half + r) */
int gf_torso_bwd_grid_row_perm(uint32_t* perm32_host);

uint64_t gf_grid_update_ws_bytes(uint32_t C, uint32_t H);
int gf_grid_update(float* density_grid, const float* tmp_grid, uint32_t C, uint32_t H, float decay, float density_thresh,
                   uint8_t* bitfield, void* partial_ws, float* stats_dev, void* stream);
int gf_mark_untrained_grid(const float* poses, uint32_t B, float fx, float fy, float cx, float cy, uint32_t C, uint32_t H, float bound,
                           float* density_grid, void* stream);
/* HOST: {xmin,ymin,zmin,xmax,ymax,zmax} of the occupied cells of a density_bitfield (HOST pointer).  No sample of
 * kernel_march_rays (raymarching.cu:828-929) can lie outside it, so the fused marcher stops at a ray's exit from it. */
int gf_occupancy_aabb(const uint8_t* bitfield_host, uint32_t cascade, uint32_t H, float bound, float* out6_host);
/* HOST: can the fused kernels index these grid tables (GridEncoder.offsets as a HOST array [L+1], gridencoder.cu:66-84)? */
int gf_grid_levels_fusable(const int32_t* offsets_host, uint32_t L, uint32_t D, float S, uint32_t H);
uint32_t gf_head_pack_floats(void);
uint32_t gf_torso_pack_floats(void);
uint32_t gf_head_pack_colbias_offset(void);   /* where gf_head_pack puts the 128 folded identity-code biases (float offset) */
int gf_clayout_perm(uint32_t* perm128_host);
/* HOST pointers: nn.Linear weights [out,in] of ambient_net / sigma_net / color_net (cond_encoder.py:92-111), individual code or NULL */
int gf_head_pack(const float* amb0_host, const float* amb1_host, const float* amb2_host, const float* sig0_host,
                 const float* sig1_host, const float* sig2_host, const float* col0_host, const float* col1_host,
                 const float* ind_code_host, float* out_host);
/* fast path (gf_frame_t.precision = 1): the six MFMA layers of the head as f16 A-operand streams (binary16 bit patterns) */
uint32_t gf_head_pack16_halves(void);
int gf_head_pack16(const float* amb0_host, const float* amb1_host, const float* sig0_host, const float* sig1_host,
                   const float* sig2_host, const float* col0_host, uint16_t* out_halves_host);
/* HOST: out_index_host [gf_head_pack16_halves()]: the 1-based flat index into cat(amb0, amb1, sig0, sig1, sig2, col0) every half of that layout
 * is taken from (0: a zero slot) -- for re-gathering the streams on the device after every optimizer step (the AMP training tier) */
int gf_head_pack16_index(uint32_t* out_index_host);
/* split path (gf_frame_t.precision = 2): the same six layers as two-term f16 splits; GF_ERR_UNSUPPORTED when a weight is outside the f16 range */
uint32_t gf_head_pack_split_halves(void);
int gf_head_pack_split(const float* amb0_host, const float* amb1_host, const float* sig0_host, const float* sig1_host,
                       const float* sig2_host, const float* col0_host, uint16_t* out_halves_host);
/* HOST pointers: torso_deform_net / torso_canonicial_net weights */
int gf_torso_pack(const float* d0_host, const float* d1_host, const float* d2_host, const float* c0_host,
                  const float* c1_host, const float* c2_host, float* out_host);
/* torso_head_aware models: d0 [64,120], c0 [32,152] (16 encoder columns appended to both first layers), the encoder's three Linear layers
 * e0 [16,4]+[16], e1 [32,16]+[32], e2 [16,32]+[16].  out_main_host [gf_torso_pack_floats()] as gf_torso_pack; out_ha_host
 * [gf_torso_ha_pack_floats()] = the encoder columns as two extra A-operand streams + the encoder itself. */
uint32_t gf_torso_ha_pack_floats(void);
int gf_torso_pack_ha(const float* d0_host, const float* d1_host, const float* d2_host, const float* c0_host, const float* c1_host,
                     const float* c2_host, const float* e0w_host, const float* e0b_host, const float* e1w_host, const float* e1b_host,
                     const float* e2w_host, const float* e2b_host, float* out_main_host, float* out_ha_host);
/* head pass; with torso_pack == NULL also the head-only tail (renderer.py:354-364) so outputs are final */
int gf_render_head(const gf_frame_t* frame, void* stream);
/* torso pass + final blend (radnerf_torso.py:156-198); gf_render_head must precede it on the same stream */
int gf_render_torso(const gf_frame_t* frame, void* stream);
/* measurement only: gf_render_head's launches bracketed by HIP events on `stream`; synchronises.  phase_ms_host [4]: [0], [1] the two field
 * launches (k_head_phase), [2] k_frame_init (ray generation + the march through empty space); *n_phases_host = 3 */
int gf_render_head_timed(const gf_frame_t* frame, void* stream, float* phase_ms_host, uint32_t* n_phases_host);

/* ------------------------------------------------------------------------------------------------
 * Frame files.  inference/nerfs/base_nerf_infer.py:97-101 writes every frame as <tmp_imgs_dir>/<idx:05d>.png (cv2.imwrite) between two
 * frames; here native worker threads deflate and write while the GPU renders on.  HOST pointers only; 8-bit RGB PNG, filter type 0.
 * level 0..9 and strategy 0..4 are zlib's (Z_DEFAULT_STRATEGY 0, Z_FILTERED 1, Z_HUFFMAN_ONLY 2, Z_RLE 3, Z_FIXED 4).
 * ---------------------------------------------------------------------------------------------- */
void* gf_png_writer_create(const char* dir, uint32_t H, uint32_t W, uint32_t workers, int level, int strategy, uint32_t max_pending);
int gf_png_writer_submit(void* handle, uint32_t idx, const uint8_t* rgb_host);   /* copies the frame; blocks while max_pending frames wait */
int gf_png_writer_close(void* handle, double* stats_out_host);                   /* drains, joins, frees; stats [6] or NULL (see png_writer.cpp) */
int gf_png_encode_rgb8(const uint8_t* rgb_host, uint32_t H, uint32_t W, int level, int strategy, uint8_t* out_host, uint64_t cap_bytes,
                       uint64_t* n_bytes_host);

#ifdef __cplusplus
}
#endif
#endif /* GENEFACE_HIP_H */
