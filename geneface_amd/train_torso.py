"""Training-time torso field: RADNeRFTorso.forward_torso (modules/radnerfs/radnerf_torso.py:51-84) as ONE autograd node (round 6).

The torso task's step (tasks/radnerfs/radnerf_torso.py:74-122) renders the frozen head under no_grad and trains only the 2-D deformable
torso field on the masked pixels.  As a torch op graph that field is ~60 launches forward and ~120 backward (two frequency encodings, a
grid encode, six Linear layers, cats / expands of the per-frame constants, clamps, sigmoids) for < 6 GFLOP: the step is bound by the host's
launch rate (276 launches, 2.1 ms of kernels in a 4.0 ms step: profiles/round6/r6b_torso_train_step_kernel_stats_before_fusing.csv).

Forward: one launch (gf_torso_train_forward, csrc/frame_torso.hip::k_torso_train_fwd -- the renderer's own torso field chain on f32 MFMA)
that also leaves every layer's activations as [M, width] matrices.  Backward: the input-gradient chain in one launch
(gf_torso_train_backward: transposed weight blocks through the same register-chained MFMA layers, ReLU masks from the saved activations,
the 2-D lookup's input gradient re-gathered, the clamp's derivative), the six weight gradients as tall products of its outputs with the
saves (per-frame constants -- pose encoding, identity code -- enter as outer products with column sums, never as [M, 62] copies), and
the table gradient through the library's scatter (gf_grid_encode_backward_scaled).  Gradients flow to: the torso grid table, the six
Linear weights, the torso identity code.  Pixel coordinates and the pose get none (they are data).
Default architecture only (torso_head_aware = false, base.yaml:90); head-aware models train through the op graph as before.
"""
import ctypes as C

import numpy as np
import torch

from .lib import check, current_stream, lib, ptr
from .train_field import _grid_backward, _tall_tn

_vp = C.c_void_p


class GfTorsoTrain(C.Structure):
    """ctypes mirror of gf_torso_train_t (include/geneface_hip.h)."""
    _fields_ = [("M", C.c_uint32), ("torso_shrink", C.c_float), ("torso_S", C.c_float), ("base_res", C.c_uint32)] + \
               [(n, _vp) for n in ("x", "torso_pack", "torso_bias", "torso_table", "torso_offsets", "out", "dx", "enc", "h_d1", "h_d2", "x01", "g",
                                   "h_c1", "h_c2", "bwd_streams", "g_out", "g_dx", "dz_c3", "dz_c2", "dz_c1", "dz_d3", "dz_d2", "dz_d1", "g_grid",
                                   "level_max")]


_BWD_IDX = {}


class GfTorsoWgrad(C.Structure):
    """ctypes mirror of gf_torso_wgrad_t (include/geneface_hip.h)."""
    _fields_ = [("M", C.c_uint32), ("_pad", C.c_uint32)] + [(n, C.c_void_p) for n in (
        "enc", "h_d1", "h_d2", "g", "h_c1", "h_c2", "dz_d1", "dz_d2", "dz_d3", "dz_c1", "dz_c2", "dz_c3", "v", "w_d1", "w_c1",
        "g_wd1", "g_wd2", "g_wd3", "g_wc1", "g_wc2", "g_wc3", "g_v", "workspace")]


def _bwd_stream_index(dev):
    """Index map of the backward's A-operand streams into cat([0], W_c2^T, W_c1[:, :32]^T (rows permuted), W_d2^T) (1-based flat indices):
    the HOST packer run once on index-valued matrices, so a weight update is one device-side gather (fused._pack_index)."""
    key = str(dev)
    if key not in _BWD_IDX:
        from .fused import _hp, _pack_index
        L = lib()

        def pack(arr, out):
            check(L.gf_mlp_stream_pack(_hp(arr[0]), 32, 1, 16, _hp(out[0:1024])))
            check(L.gf_mlp_stream_pack(_hp(arr[1]), 32, 1, 16, _hp(out[1024:2048])))
            check(L.gf_mlp_stream_pack(_hp(arr[2]), 64, 2, 32, _hp(out[2048:6144])))
        assert L.gf_torso_bwd_stream_floats() == 6144
        perm = (C.c_uint32 * 32)()
        check(L.gf_torso_bwd_grid_row_perm(perm))
        _BWD_IDX[key] = (_pack_index(pack, [(32, 32), (32, 32), (64, 64)], 6144).to(dev), torch.tensor(list(perm), dtype=torch.long, device=dev))
    return _BWD_IDX[key]


def supported(model) -> bool:
    """The fused training field covers the default torso architecture on a GPU (what gf_torso_pack packs)."""
    return (not bool(getattr(model, "torso_head_aware", False))) and model.density_grid_torso.is_cuda and model.torso_individual_embedding_dim == 8 \
        and tuple(model.torso_deform_net.net[0].weight.shape) == (64, 104) and tuple(model.torso_canonicial_net.net[0].weight.shape) == (32, 136)


class _TorsoField(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, model, x, poses, code, table, wd1, wd2, wd3, wc1, wc2, wc3):
        from . import fused
        st = fused.get_state(model)          # the packed torso weights follow the current parameter values (one device-side gather per change)
        dev = x.device
        x = x.detach().reshape(-1, 2).float().contiguous()
        M = x.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        te = model.torso_embedder
        with torch.no_grad():
            v = torch.cat([model.torso_pose_embedder(poses.detach().reshape(1, 6).float()).reshape(-1), code.detach().reshape(-1).float()])
            bias = torch.mv(st.W_tconst, v)
        out, dx = torch.empty(M, 4, **f32), torch.empty(M, 2, **f32)
        sv = {n: torch.empty(M, w, **f32) for n, w in (("enc", 48), ("h_d1", 64), ("h_d2", 64), ("x01", 2), ("g", 32), ("h_c1", 32), ("h_c2", 32))}
        t = GfTorsoTrain()
        t.M, t.torso_shrink, t.torso_S, t.base_res = M, float(model.torso_shrink), st.torso_S, int(te.base_resolution)
        t.x, t.torso_pack, t.torso_bias = ptr(x), ptr(st.torso_pack), ptr(bias, torch.float32)
        t.torso_table, t.torso_offsets = ptr(te.embeddings, torch.float32), ptr(te.offsets, torch.int32)
        t.out, t.dx = ptr(out), ptr(dx)
        for n, buf in sv.items():
            setattr(t, n, buf.data_ptr())
        if M > 0:
            check(lib().gf_torso_train_forward(C.byref(t), current_stream(dev)))
        ctx.model, ctx.t_fields = model, (M, float(model.torso_shrink), st.torso_S, int(te.base_resolution))
        ctx.save_for_backward(x, v, out, dx, *sv.values(), table, wd1, wd2, wd3, wc1, wc2, wc3)
        return out[:, :1], out[:, 1:], dx

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_alpha, g_rgb, g_dx):
        from . import fused
        model = ctx.model
        x, v, out, dx, enc, h_d1, h_d2, x01, g, h_c1, h_c2, table, wd1, wd2, wd3, wc1, wc2, wc3 = ctx.saved_tensors
        M, shrink, S, base_res = ctx.t_fields
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        st = fused.get_state(model)
        te = model.torso_embedder
        if M == 0:
            z = lambda w: torch.zeros_like(w)
            return (None, None, None, torch.zeros(8, **f32), z(table), z(wd1), z(wd2), z(wd3), z(wc1), z(wc2), z(wc3))
        g_out = torch.cat([g_alpha.reshape(M, 1) if g_alpha is not None else torch.zeros(M, 1, **f32),
                           g_rgb.reshape(M, 3) if g_rgb is not None else torch.zeros(M, 3, **f32)], dim=1).float().contiguous()
        idx, perm = _bwd_stream_index(dev)
        zero = torch.zeros(1, **f32)
        flat = torch.cat([zero, wc2.detach().t().reshape(-1).float(), wc1.detach()[:, :32].t()[perm].reshape(-1).float(), wd2.detach().t().reshape(-1).float()])
        streams = flat[idx]
        dz = {n: torch.empty(M, w, **f32) for n, w in (("dz_c3", 4), ("dz_c2", 32), ("dz_c1", 32), ("dz_d3", 2), ("dz_d2", 64), ("dz_d1", 64))}
        g_grid = torch.empty(16, M, 2, **f32)
        level_max = torch.zeros(16, dtype=torch.int32, device=dev)
        t = GfTorsoTrain()
        t.M, t.torso_shrink, t.torso_S, t.base_res = M, shrink, S, base_res
        t.x, t.torso_pack, t.torso_bias = ptr(x), ptr(st.torso_pack), None
        t.torso_table, t.torso_offsets = ptr(te.embeddings, torch.float32), ptr(te.offsets, torch.int32)
        t.out, t.dx = ptr(out), ptr(dx)
        t.enc, t.g = ptr(enc), ptr(g)
        t.h_d1, t.h_d2, t.x01, t.h_c1, t.h_c2 = ptr(h_d1), ptr(h_d2), ptr(x01), ptr(h_c1), ptr(h_c2)
        t.bwd_streams, t.g_out = ptr(streams), ptr(g_out)
        g_dx_c = g_dx.reshape(M, 2).float().contiguous() if g_dx is not None else None      # (kept in a local until the launch is enqueued)
        t.g_dx = ptr(g_dx_c) if g_dx_c is not None else None
        for n, buf in dz.items():
            setattr(t, n, buf.data_ptr())
        t.g_grid, t.level_max = ptr(g_grid), ptr(level_max)
        check(lib().gf_torso_train_backward(C.byref(t), current_stream(dev)))
        # weight gradients: tall products of the pre-activation gradients with the saved layer inputs; the 62 per-frame constant columns of
        # both first layers (pose encoding 54 | identity code 8) are outer products of the column sums with that one vector -- all of it in two
        # launches (gf_torso_wgrad, csrc/torso_wgrad.hip; `model.torso_wgrad_impl = "gemm"` keeps the batched library products and their glue)
        if getattr(model, "torso_wgrad_impl", "fused") == "fused" and tuple(wd1.shape) == (64, 104) and tuple(wc1.shape) == (32, 136) \
                and all(w.dtype == torch.float32 for w in (wd1, wd2, wd3, wc1, wc2, wc3)):
            if getattr(st, "_torso_wgrad_ws", None) is None:
                st._torso_wgrad_ws = torch.empty(lib().gf_torso_wgrad_ws_bytes() // 4, **f32)
            g_wd1, g_wd2, g_wd3, g_wc1, g_wc2, g_wc3 = (torch.empty(w.shape, **f32) for w in (wd1, wd2, wd3, wc1, wc2, wc3))
            g_v = torch.empty(62, **f32)
            vv, w1, w2 = v.float().contiguous(), wd1.detach().contiguous(), wc1.detach().contiguous()
            wg = GfTorsoWgrad(M=M, enc=ptr(enc), h_d1=ptr(h_d1), h_d2=ptr(h_d2), g=ptr(g), h_c1=ptr(h_c1), h_c2=ptr(h_c2),
                              dz_d1=ptr(dz["dz_d1"]), dz_d2=ptr(dz["dz_d2"]), dz_d3=ptr(dz["dz_d3"]), dz_c1=ptr(dz["dz_c1"]), dz_c2=ptr(dz["dz_c2"]),
                              dz_c3=ptr(dz["dz_c3"]), v=ptr(vv), w_d1=ptr(w1), w_c1=ptr(w2), g_wd1=ptr(g_wd1), g_wd2=ptr(g_wd2), g_wd3=ptr(g_wd3),
                              g_wc1=ptr(g_wc1), g_wc2=ptr(g_wc2), g_wc3=ptr(g_wc3), g_v=ptr(g_v), workspace=ptr(st._torso_wgrad_ws))
            check(lib().gf_torso_wgrad(C.byref(wg), current_stream(dev)))
        else:
            e42 = enc[:, :42]
            s_d1, s_c1 = dz["dz_d1"].sum(0), dz["dz_c1"].sum(0)
            g_wd1 = torch.cat([_tall_tn(dz["dz_d1"], e42), torch.outer(s_d1, v)], dim=1)
            g_wd2 = _tall_tn(dz["dz_d2"], h_d1)
            g_wd3 = _tall_tn(dz["dz_d3"], h_d2)
            g_wc1 = torch.cat([_tall_tn(dz["dz_c1"], g), _tall_tn(dz["dz_c1"], e42), torch.outer(s_c1, v)], dim=1)
            g_wc2 = _tall_tn(dz["dz_c2"], h_c1)
            g_wc3 = _tall_tn(dz["dz_c3"], h_c2)
            g_v = torch.mv(wd1.detach()[:, 42:].t().float(), s_d1) + torch.mv(wc1.detach()[:, 74:].t().float(), s_c1)
        g_code = g_v[54:]
        g_table, _ = _grid_backward(te, x01, g_grid, want_input_grad=False, level_major=True, level_max=level_max)
        cast = lambda gw, w: gw.to(w.dtype)
        return (None, None, None, g_code, cast(g_table, table), cast(g_wd1, wd1), cast(g_wd2, wd2), cast(g_wd3, wd3), cast(g_wc1, wc1),
                cast(g_wc2, wc2), cast(g_wc3, wc3))


def torso_field_no_grad(model, x, poses, code):
    """The same one-launch forward WITHOUT autograd (update_extra_state's query of the field on the 128 x 128 jittered cell centres,
    radnerf_torso.py:225-232): alpha [M,1], colour [M,3], dx [M,2].  The activations go to scratch buffers nobody reads."""
    with torch.no_grad():
        d, c = model.torso_deform_net.net, model.torso_canonicial_net.net
        return _TorsoField.forward(_NoCtx(), model, x, poses, code, model.torso_embedder.embeddings, d[0].weight, d[1].weight, d[2].weight, c[0].weight,
                                   c[1].weight, c[2].weight)


class _NoCtx:
    """Stand-in for the autograd context when the forward runs outside autograd."""
    def save_for_backward(self, *a):
        pass


def forward_torso_fused(model, x, poses, code):
    """forward_torso (radnerf_torso.py:51-84) through the fused node: x [M,2], poses [1,6], code [8] -> alpha [M,1], colour [M,3], dx [M,2]."""
    d, c = model.torso_deform_net.net, model.torso_canonicial_net.net
    return _TorsoField.apply(model, x, poses, code, model.torso_embedder.embeddings, d[0].weight, d[1].weight, d[2].weight, c[0].weight, c[1].weight,
                             c[2].weight)


class GfTorsoBlend(C.Structure):
    """ctypes mirror of gf_torso_blend_t (include/geneface_hip.h)."""
    _fields_ = [("N", C.c_uint32), ("bg_stride", C.c_uint32)] + [(n, C.c_void_p) for n in (
        "a", "c", "mask", "bg", "image", "weights_sum", "torso_alpha", "torso_rgb", "rgb", "g_alpha", "g_torso_rgb", "g_rgb", "g_a", "g_c")]


class _TorsoBlend(torch.autograd.Function):
    """alpha = a m, colour = c m, torso_rgb = colour alpha + bg (1 - alpha), rgb = clamp(image + (1 - weights_sum) torso_rgb, 0, 1)
    (radnerf_torso.py:181-192): one launch forward, one backward; gradients to a and c (the head is frozen in the torso task)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, a, c, m, bg, image, weights_sum):
        dev = a.device
        N = a.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        a_, c_ = a.detach().reshape(N).float().contiguous(), c.detach().reshape(N, 3).float().contiguous()
        m_, img, ws = m.reshape(N).float().contiguous(), image.detach().reshape(N, 3).float().contiguous(), weights_sum.detach().reshape(N).float().contiguous()
        bg_ = bg.detach().float().contiguous()
        stride = 3 if bg_.numel() == 3 * N and N > 1 else 0
        if stride == 0 and bg_.numel() != 3:
            bg_ = bg_.reshape(-1)[:1].expand(3).contiguous()
        torso_alpha, torso_rgb, rgb = torch.empty(N, 1, **f32), torch.empty(N, 3, **f32), torch.empty(N, 3, **f32)
        t = GfTorsoBlend(N=N, bg_stride=stride, a=ptr(a_), c=ptr(c_), mask=ptr(m_), bg=ptr(bg_), image=ptr(img), weights_sum=ptr(ws),
                         torso_alpha=ptr(torso_alpha), torso_rgb=ptr(torso_rgb), rgb=ptr(rgb))
        check(lib().gf_torso_blend_train_forward(C.byref(t), current_stream(dev)))
        ctx.save_for_backward(a_, c_, m_, bg_, img, ws)
        ctx.stride = stride
        return torso_alpha, torso_rgb, rgb

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_alpha, g_torso_rgb, g_rgb):
        a_, c_, m_, bg_, img, ws = ctx.saved_tensors
        dev, N = a_.device, a_.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        keep = [None if g is None else g.detach().float().contiguous() for g in (g_alpha, g_torso_rgb, g_rgb)]
        g_a, g_c = torch.empty(N, 1, **f32), torch.empty(N, 3, **f32)
        t = GfTorsoBlend(N=N, bg_stride=ctx.stride, a=ptr(a_), c=ptr(c_), mask=ptr(m_), bg=ptr(bg_), image=ptr(img), weights_sum=ptr(ws),
                         g_alpha=None if keep[0] is None else keep[0].data_ptr(), g_torso_rgb=None if keep[1] is None else keep[1].data_ptr(),
                         g_rgb=None if keep[2] is None else keep[2].data_ptr(), g_a=ptr(g_a), g_c=ptr(g_c))
        check(lib().gf_torso_blend_train_backward(C.byref(t), current_stream(dev)))
        return g_a, g_c, None, None, None, None


def torso_blend_train(a, c, m, bg_color, image, weights_sum):
    """a [N,1], c [N,3] (the torso field on every sampled pixel), m [N] the mask as 0 / 1, bg_color a number, [3], or [.., N, 3], the frozen head's
    image [N,3] and weights_sum [N] -> torso_alpha_map [N,1], torso_rgb_map [N,3], rgb_map [N,3]."""
    if not torch.is_tensor(bg_color):
        bg_color = torch.full((3,), float(bg_color), dtype=torch.float32, device=a.device)
    return _TorsoBlend.apply(a, c, m, bg_color, image, weights_sum)
