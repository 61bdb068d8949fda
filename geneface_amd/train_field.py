"""Training-time field of the RAD-NeRF head: RADNeRF.forward (modules/radnerfs/radnerf.py:73-105) as ONE autograd node.

The reference -- and this package's op-by-op route -- builds the graph out of ~40 torch ops per call: two grid encodes, SH, 8 GEMMs,
cats of per-point copies of the condition vector and the identity code, ReLUs, tanh / exp / sigmoid.  On a 1 M-point training batch the
GEMMs are 8 ms and the glue between them 5.6 ms of an 18 ms step (profiles/round2/r2t_train_kernel_stats.csv).

Forward: one launch of the renderer's own field core over the point list (gf_field_forward_train, csrc/frame_head.hip), which also leaves
every layer's activations as [M, width] matrices and the ReLU masks as bits.  Backward:
  * the input-gradient chain through all layers is ONE launch as well (gf_field_backward: transposed weight blocks through the same MFMA
    core, ReLU derivatives from the mask bits, the 2-D lookup's input gradient re-gathered); it writes every layer's pre-activation
    gradient, both grid feature gradients (level-major, as the scatter kernel reads them) and the column sums behind d cond_feat / d code;
  * weight gradients are tall products of those with the saved activations (the split reduction of cond_encoder._linear_tall); per-point
    constants (condition vector, identity code) enter as outer products, never as [M, 64] / [M, 4] copies, and no `cat` is materialised;
  * the two grid tables go through the library's own scatter kernels (gf_grid_encode_backward).
Gradients flow to: both grid tables, the eight MLP weights, cond_feat, individual_code.  Positions and directions get none (the
reference's marcher outputs are not differentiable either: raymarching.py:262-265).
"""
import ctypes as C

import numpy as np
import torch

from .lib import check, current_stream, lib, ptr

_vp = C.c_void_p


class GfFieldSaves(C.Structure):
    """ctypes mirror of gf_field_saves_t (include/geneface_hip.h)."""
    _fields_ = [(n, _vp) for n in ("f3", "ha1", "ha2", "f2", "hs1", "hs2", "geo", "hc1", "m_ha1", "m_ha2", "m_hs1", "m_hs2", "m_hc1", "sh")]


class GfFieldGrads(C.Structure):
    """ctypes mirror of gf_field_grads_t."""
    _fields_ = [(n, _vp) for n in ("g_sigma", "g_rgb", "g_amb", "sigma", "rgb", "amb", "m_hc1", "m_hs2", "m_hs1", "m_ha2", "m_ha1", "g_zc", "g_h0", "g_za",
                                   "g_hc1", "g_geo", "g_hs2", "g_hs1", "g_ha2", "g_ha1", "g_f3", "g_f2", "s_hc1", "s_ha1", "level_max")] + \
               [("out16", C.c_uint32), ("_pad", C.c_uint32)]


class GfFieldWgrad(C.Structure):
    """ctypes mirror of gf_field_wgrad_t."""
    _fields_ = [(n, _vp) for n in ("f3", "ha1", "ha2", "f2", "hs1", "hs2", "geo", "hc1", "sh", "g_hc1", "g_geo", "g_hs2", "g_hs1", "g_ha2", "g_ha1",
                                   "g_zc", "g_h0", "g_za", "gw_color1", "gw_color0", "gw_sigma2", "gw_sigma1", "gw_sigma0", "gw_ambient2", "gw_ambient1",
                                   "gw_ambient0")] + [("ld_color0", C.c_uint32), ("ld_ambient0", C.c_uint32), ("workspace", _vp)]


def _bwd_stream_index(shapes):
    """Index map of gf_field_backward's A-operand stream into cat([0], W_a1, W_a2, W_a3, W_s1, W_s2, W_s3, W_c1, W_c2) (1-based, 0 = zero
    padding): six transposed 128 x 128 blocks, element [wave][layer][group][lane][i] = Wt[32 wave + (lane & 31)][8 group + 4 (lane >> 5) + i]
    with Wt[n][o] = W[o0 + o][n0 + n] for n < the block's real width."""
    names = ["a1", "a2", "a3", "s1", "s2", "s3", "c1", "c2"]
    base, off = {}, 1
    for n, shp in zip(names, shapes):
        base[n] = (off, shp)
        off += shp[0] * shp[1]
    #          weight, first fwd-output row o0, first fwd-input column n0, real number of fwd-input columns
    layers = [("c1", 0, 16, 128), ("s3", 1, 0, 128), ("s2", 0, 0, 128), ("s1", 0, 0, 64), ("a2", 0, 0, 128), ("a1", 0, 0, 32)]
    w = np.arange(4).reshape(4, 1, 1, 1, 1)
    u = np.arange(16).reshape(1, 1, 16, 1, 1)
    l = np.arange(64).reshape(1, 1, 1, 64, 1)
    i = np.arange(4).reshape(1, 1, 1, 1, 4)
    n = 32 * w + (l & 31)                       # bwd output = forward input feature
    o = 8 * u + 4 * (l >> 5) + i                # bwd reduction index = forward output feature
    idx = np.zeros((4, 6, 16, 64, 4), dtype=np.int64)
    for k, (name, o0, n0, width) in enumerate(layers):
        b0, (rows, cols) = base[name]
        src = b0 + (o0 + o) * cols + (n0 + n)
        idx[:, k] = np.where(np.broadcast_to(n < width, src.shape), src, 0)[:, 0]
    return torch.from_numpy(idx.reshape(-1))


def _bwd16_stream_index(shapes):
    """The same six transposed blocks for the f16 chain (k_field_backward16): [wave 4][layer 6][group 8][lane 64][8] halves, element =
    Wt[32 wave + (lane & 31)][16 group + 8 (lane >> 5) + i]; indices into cat([0], weights) as in _bwd_stream_index."""
    names = ["a1", "a2", "a3", "s1", "s2", "s3", "c1", "c2"]
    base, off = {}, 1
    for n, shp in zip(names, shapes):
        base[n] = (off, shp)
        off += shp[0] * shp[1]
    layers = [("c1", 0, 16, 128), ("s3", 1, 0, 128), ("s2", 0, 0, 128), ("s1", 0, 0, 64), ("a2", 0, 0, 128), ("a1", 0, 0, 32)]
    w = np.arange(4).reshape(4, 1, 1, 1, 1)
    u = np.arange(8).reshape(1, 1, 8, 1, 1)
    l = np.arange(64).reshape(1, 1, 1, 64, 1)
    i = np.arange(8).reshape(1, 1, 1, 1, 8)
    n = 32 * w + (l & 31)
    o = 16 * u + 8 * (l >> 5) + i
    idx = np.zeros((4, 6, 8, 64, 8), dtype=np.int64)
    for k, (name, o0, n0, width) in enumerate(layers):
        b0, (rows, cols) = base[name]
        src = b0 + (o0 + o) * cols + (n0 + n)
        idx[:, k] = np.where(np.broadcast_to(n < width, src.shape), src, 0)[:, 0]
    return torch.from_numpy(idx.reshape(-1))


def _tall_tn(g, x, out_dtype=None):
    """g^T @ x for g [M,O], x [M,I] with M ~ 10^6: batched partial products + a sum (see cond_encoder._linear_tall).
    out_dtype: the partial products are added (and returned) in this dtype -- the AMP tier multiplies half operands and adds in fp32."""
    B, O, I = x.shape[0], g.shape[1], x.shape[1]
    S = max(1, B // 4096)
    rows = B // S
    main = S * rows
    part = torch.bmm(g[:main].view(S, rows, O).transpose(1, 2), x[:main].view(S, rows, I))
    gw = (part if out_dtype is None else part.to(out_dtype)).sum(0)
    if main < B:
        tail = g[main:].t() @ x[main:]
        gw = gw + (tail if out_dtype is None else tail.to(out_dtype))
    return gw


def _fused_wgrad(st, half, M, dev, weights, mats):
    """The eight weight gradients (W_color1, W_color0, W_sigma2, W_sigma1, W_sigma0, W_ambient2, W_ambient1, W_ambient0 order) as fresh fp32
    tensors from gf_field_wgrad16 (half=True: binary16 saves and gradient rows) or gf_field_wgrad32 (fp32 ones).  The identity-code columns
    of W_color0 and the condition columns of W_ambient0 are left for the caller."""
    wc2, wc1, ws3, ws2, ws1, wa3, wa2, wa1 = weights
    assert tuple(wc2.shape) == (3, 128) and wc1.shape[0] == 128 and wc1.shape[1] >= 144 and tuple(ws3.shape) == (129, 128) and \
        tuple(ws2.shape) == (128, 128) and tuple(ws1.shape) == (128, 64) and tuple(wa3.shape) == (2, 128) and tuple(wa2.shape) == (128, 128) and \
        wa1.shape[0] == 128 and wa1.shape[1] >= 32
    want = torch.float16 if half else torch.float32
    for n, t in mats.items():
        if t.dtype != (torch.float32 if n in ("g_zc", "g_h0", "g_za") else want) or not t.is_contiguous():
            raise RuntimeError(f"weight-gradient kernel: {n} must be a contiguous {want} matrix")
    key = "_wgrad_ws16" if half else "_wgrad_ws32"
    if getattr(st, key, None) is None:
        nbytes = lib().gf_field_wgrad16_ws_bytes() if half else lib().gf_field_wgrad32_ws_bytes()
        setattr(st, key, torch.empty(nbytes // 4, dtype=torch.float32, device=dev))
    outs = tuple(torch.empty(w.shape, dtype=torch.float32, device=dev) for w in weights)
    names = ("gw_color1", "gw_color0", "gw_sigma2", "gw_sigma1", "gw_sigma0", "gw_ambient2", "gw_ambient1", "gw_ambient0")
    wg = GfFieldWgrad(**{n: t.data_ptr() for n, t in mats.items()}, **{n: t.data_ptr() for n, t in zip(names, outs)},
                      ld_color0=wc1.shape[1], ld_ambient0=wa1.shape[1], workspace=getattr(st, key).data_ptr())
    check((lib().gf_field_wgrad16 if half else lib().gf_field_wgrad32)(M, C.byref(wg), current_stream(dev)))
    return outs


_GRID_BINNING = __import__("os").environ.get("GF_GRID_BINNING", "1") != "0"     # False (GF_GRID_BINNING=0): gf_grid_encode_backward_scaled without the binning pass (A/B; tests compare the two bit for bit)
_GRID_WS = {}            # device -> uint8 workspace of gf_grid_encode_backward_binned


def _grid_backward(enc, x01, grad, want_input_grad, level_major=False, level_max=None):
    """Table gradient (and d/d x01 through a freshly evaluated dy_dx) of GridEncoder `enc` at inputs x01 [M,D] for an output gradient
    `grad` [M, L*C] (or already [L, M, C] with level_major): the library's backward kernels, without re-entering autograd.  level_max: the
    per-level max |grad| (int32 view of float bits, [L]) when the producer of `grad` has it -- spares the scatter its max pass."""
    L_ = lib()
    dev = x01.device
    B, D = x01.shape
    L, Cc = enc.num_levels, enc.level_dim
    S = float(np.log2(enc.per_level_scale))
    if level_max is not None and level_major and not want_input_grad:
        g_tab = torch.zeros_like(enc.embeddings)
        if _GRID_BINNING and B >= 4096:
            # the binning pass in front of the scatter (round 6): one workspace per device, shared by the calls of a stream (they are ordered)
            need = L_.gf_grid_backward_ws_bytes(B, L)
            ws = _GRID_WS.get(dev)
            if ws is None or ws.numel() < need:
                ws = _GRID_WS[dev] = torch.empty(need, dtype=torch.uint8, device=dev)
            check(L_.gf_grid_encode_backward_binned(ptr(grad, torch.float32), ptr(x01, torch.float32), ptr(enc.offsets, torch.int32), ptr(g_tab, torch.float32),
                                                    B, D, Cc, L, S, int(enc.base_resolution), enc.gridtype_id, int(bool(enc.align_corners)), enc.interp_id,
                                                    ptr(level_max, torch.int32), ws.data_ptr(), current_stream(dev)))
            return g_tab, None
        check(L_.gf_grid_encode_backward_scaled(ptr(grad, torch.float32), ptr(x01, torch.float32), ptr(enc.offsets, torch.int32), ptr(g_tab, torch.float32),
                                                B, D, Cc, L, S, int(enc.base_resolution), enc.gridtype_id, int(bool(enc.align_corners)), enc.interp_id,
                                                ptr(level_max, torch.int32), current_stream(dev)))
        return g_tab, None
    dy_dx = gin = None
    if want_input_grad:
        dy_dx = torch.empty(B, L * D * Cc, device=dev, dtype=torch.float32)
        scratch = torch.empty(B, L * Cc, device=dev, dtype=torch.float32)
        check(L_.gf_grid_encode_forward_blc(ptr(x01, torch.float32), ptr(enc.embeddings, torch.float32), ptr(enc.offsets, torch.int32), ptr(scratch),
                                            B, D, Cc, L, S, int(enc.base_resolution), ptr(dy_dx), enc.gridtype_id, int(bool(enc.align_corners)),
                                            enc.interp_id, current_stream(dev)))
        gin = torch.zeros_like(x01)
    g_lbc = grad if level_major else grad.view(B, L, Cc).permute(1, 0, 2).contiguous()
    g_tab = torch.zeros_like(enc.embeddings)
    check(L_.gf_grid_encode_backward(ptr(g_lbc, torch.float32), ptr(x01, torch.float32), ptr(enc.embeddings, torch.float32), ptr(enc.offsets, torch.int32),
                                     ptr(g_tab, torch.float32), B, D, Cc, L, S, int(enc.base_resolution), ptr(dy_dx, torch.float32, allow_none=True),
                                     ptr(gin, torch.float32, allow_none=True), enc.gridtype_id, int(bool(enc.align_corners)), enc.interp_id,
                                     current_stream(dev)))
    return g_tab, gin


class _HeadField(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, model, xyz, dirs, cond_feat, ind_code, pos_tab, amb_tab, wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2):
        from . import fused
        st = fused.get_state(model)
        dev = xyz.device
        x = xyz.detach().reshape(-1, 3).float().contiguous()
        d = dirs.detach().reshape(-1, 3).float().contiguous()
        M = x.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        sigma, rgb, amb = torch.empty(M, **f32), torch.empty(M, 3, **f32), torch.empty(M, 2, **f32)
        sv = {n: torch.empty(M, w, **f32) for n, w in (("f3", 32), ("ha1", 128), ("ha2", 128), ("f2", 32), ("hs1", 128), ("hs2", 128),
                                                      ("geo", 128), ("hc1", 128), ("sh", 16))}
        chunks = (M + 127) // 128
        masks = {n: torch.empty(chunks * 1024, dtype=torch.int16, device=dev) for n in ("m_ha1", "m_ha2", "m_hs1", "m_hs2", "m_hc1")}
        if M > 0:
            amb_bias = torch.mv(st.W_cond, cond_feat.detach().reshape(-1).float())
            col_bias = torch.mv(st.W_ind, ind_code.detach().reshape(-1).float()) if (st.W_ind is not None and ind_code is not None) else None
            if st.W_ind is not None and col_bias is None:
                col_bias = torch.zeros(128, **f32)
            f = fused.GfFrame()
            pe, ae = model.position_embedder, model.ambient_embedder
            f.bound = float(model.bound)
            f.pos_table, f.pos_offsets = ptr(pe.embeddings, torch.float32), ptr(pe.offsets, torch.int32)
            f.amb_table, f.amb_offsets = ptr(ae.embeddings, torch.float32), ptr(ae.offsets, torch.int32)
            f.pos_S, f.amb_S, f.base_res, f.gridtype, f.interp = st.pos_S, st.amb_S, st.base_res, st.gridtype, st.interp
            f.head_pack, f.amb_bias = ptr(st.head_pack), ptr(amb_bias, torch.float32)
            saves = GfFieldSaves(**{n: t.data_ptr() for n, t in {**sv, **masks}.items()})
            check(lib().gf_field_forward_train(C.byref(f), ptr(x, torch.float32), ptr(d, torch.float32), M, ptr(col_bias, torch.float32, allow_none=True),
                                               ptr(sigma), ptr(rgb), ptr(amb), C.byref(saves), current_stream(dev)))
        ctx.model = model
        ctx.has_code = ind_code is not None
        ctx.save_for_backward(x, d, cond_feat, ind_code if ind_code is not None else torch.zeros(0, device=dev), sigma, rgb, amb,
                              wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2, *[sv[n] for n in ("f3", "ha1", "ha2", "f2", "hs1", "hs2", "geo", "hc1", "sh")],
                              *[masks[n] for n in ("m_hc1", "m_hs2", "m_hs1", "m_ha2", "m_ha1")])
        return sigma, rgb, amb

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_sigma, g_rgb, g_amb):
        (x, d, cond_feat, ind_code, sigma, rgb, amb, wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2, f3, ha1, ha2, f2, hs1, hs2, geo, hc1, sh,
         m_hc1, m_hs2, m_hs1, m_ha2, m_ha1) = ctx.saved_tensors
        from . import fused
        model = ctx.model
        M, dev = x.shape[0], x.device
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda *shape: torch.zeros(*shape, **f32)
        g_sigma = g_sigma.float().contiguous() if g_sigma is not None else z(M)
        g_rgb = g_rgb.float().contiguous() if g_rgb is not None else z(M, 3)
        g_amb = g_amb.float().contiguous() if g_amb is not None else z(M, 2)
        cond = cond_feat.reshape(-1).float()
        # ---- the dX chain: one launch (transposed weight blocks re-gathered on the device from the current weights)
        st = fused.get_state(model)
        if getattr(st, "_bwd_idx", None) is None:
            st._bwd_idx = _bwd_stream_index([tuple(w.shape) for w in (wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2)]).to(dev)
        flat = torch.cat([z(1)] + [w.detach().reshape(-1).float() for w in (wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2)])
        stream = flat[st._bwd_idx]
        out = {n: torch.empty(M, w, **f32) for n, w in (("g_zc", 3), ("g_za", 2), ("g_hc1", 128), ("g_geo", 128), ("g_hs2", 128), ("g_hs1", 128),
                                                        ("g_ha2", 128), ("g_ha1", 128), ("g_f3", 32), ("g_f2", 32))}
        out["g_h0"] = torch.empty(M, **f32)
        out["s_hc1"], out["s_ha1"] = z(128), z(128)
        level_max = torch.zeros(32, dtype=torch.int32, device=dev)      # [0:16] max |g_f3| per level, [16:32] max |g_f2| (float bit patterns)
        if M > 0:
            f = fused.GfFrame()
            pe, ae = model.position_embedder, model.ambient_embedder
            f.bound = float(model.bound)
            f.pos_offsets = ptr(pe.offsets, torch.int32)
            f.amb_table, f.amb_offsets = ptr(ae.embeddings, torch.float32), ptr(ae.offsets, torch.int32)
            f.pos_S, f.amb_S, f.base_res, f.gridtype, f.interp = st.pos_S, st.amb_S, st.base_res, st.gridtype, st.interp
            f.head_pack = ptr(st.head_pack)
            g = GfFieldGrads(g_sigma=g_sigma.data_ptr(), g_rgb=g_rgb.data_ptr(), g_amb=g_amb.data_ptr(), sigma=sigma.data_ptr(), rgb=rgb.data_ptr(),
                             amb=amb.data_ptr(), m_hc1=m_hc1.data_ptr(), m_hs2=m_hs2.data_ptr(), m_hs1=m_hs1.data_ptr(), m_ha2=m_ha2.data_ptr(),
                             m_ha1=m_ha1.data_ptr(), level_max=level_max.data_ptr(), **{n: t.data_ptr() for n, t in out.items()})
            check(lib().gf_field_backward(C.byref(f), ptr(stream, torch.float32), M, C.byref(g), current_stream(dev)))
        g_zc, g_h0, g_za = out["g_zc"], out["g_h0"], out["g_za"]
        g_hc1, g_geo, g_hs2, g_hs1, g_ha2, g_ha1 = (out[n] for n in ("g_hc1", "g_geo", "g_hs2", "g_hs1", "g_ha2", "g_ha1"))
        # ---- weight gradients: tall products of the pre-activation gradients with the saved activations -- all eight in one launch + one
        # fixed-order reduction on the f32 matrix pipe (gf_field_wgrad32, csrc/field_wgrad.hip, round 6; `model.wgrad_impl = "gemm"` keeps
        # the batched library products)
        s_hc1, s_ha1 = out["s_hc1"], out["s_ha1"]
        g_code = None
        if getattr(model, "wgrad_impl", "fused") == "fused":
            g_wc2, g_wc1, g_ws3, g_ws2, g_ws1, g_wa3, g_wa2, g_wa1 = _fused_wgrad(
                st, False, M, dev, (wc2, wc1, ws3, ws2, ws1, wa3, wa2, wa1),
                dict(f3=f3, ha1=ha1, ha2=ha2, f2=f2, hs1=hs1, hs2=hs2, geo=geo, hc1=hc1, sh=sh, g_hc1=g_hc1, g_geo=g_geo, g_hs2=g_hs2, g_hs1=g_hs1,
                     g_ha2=g_ha2, g_ha1=g_ha1, g_zc=g_zc, g_h0=g_h0, g_za=g_za))
            if ctx.has_code:
                g_wc1[:, 144:] = torch.outer(s_hc1, ind_code.reshape(-1).float())
            elif wc1.shape[1] > 144:
                g_wc1[:, 144:] = 0
            g_wa1[:, 32:] = torch.outer(s_ha1, cond)
        else:
            g_wc2 = _tall_tn(g_zc, hc1)
            parts = [_tall_tn(g_hc1, sh), _tall_tn(g_hc1, geo)]
            if ctx.has_code:
                parts.append(torch.outer(s_hc1, ind_code.reshape(-1).float()))
            g_wc1 = torch.cat(parts, dim=1)
            g_ws3 = torch.cat([_tall_tn(g_h0.unsqueeze(1), hs2), _tall_tn(g_geo, hs2)], dim=0)
            g_ws2 = _tall_tn(g_hs2, hs1)
            g_ws1 = torch.cat([_tall_tn(g_hs1, f3), _tall_tn(g_hs1, f2)], dim=1)
            g_wa3 = _tall_tn(g_za, ha2)
            g_wa2 = _tall_tn(g_ha2, ha1)
            g_wa1 = torch.cat([_tall_tn(g_ha1, f3), torch.outer(s_ha1, cond)], dim=1)
        if ctx.has_code:
            g_code = (s_hc1 @ wc1[:, 144:]).view_as(ind_code)
        g_cond = (s_ha1 @ wa1[:, 32:]).view_as(cond_feat)
        # ---- grid tables
        lm = level_max if M > 0 else None
        g_amb_tab, _ = _grid_backward(model.ambient_embedder, (amb + 1) / 2, out["g_f2"], False, level_major=True, level_max=None if lm is None else lm[16:])
        g_pos_tab, _ = _grid_backward(model.position_embedder, (x + model.bound) / (2 * model.bound), out["g_f3"], False, level_major=True,
                                      level_max=None if lm is None else lm[:16])
        return (None, None, None, g_cond, g_code, g_pos_tab, g_amb_tab, g_wa1, g_wa2, g_wa3, g_ws1, g_ws2, g_ws3, g_wc1, g_wc2)


def _pack16_device(st, weights6):
    """gf_head_pack16's layout gathered on the device from the current fp32 master weights (amb0, amb1, sig0, sig1, sig2, col0): one cat, one
    cast to half (round to nearest even, like the host packer), one gather.  Returned as int16 bit patterns, as FusedState.pack16 does."""
    if getattr(st, "_pack16_idx", None) is None:
        L = lib()
        idx = np.zeros(L.gf_head_pack16_halves(), dtype=np.uint32)
        check(L.gf_head_pack16_index(idx.ctypes.data))
        st._pack16_idx = torch.from_numpy(idx.astype(np.int64)).to(st.device)
    flat = torch.cat([torch.zeros(1, dtype=torch.float32, device=st.device)] + [w.detach().reshape(-1).float() for w in weights6]).half()
    return flat[st._pack16_idx].view(torch.int16)


class _HeadFieldAMP(torch.autograd.Function):
    """The same node on the f16 tier, taken under torch.autocast(float16) (round 6; VERDICT r5 missing #3).  The reference's AMP step
    (egs/egs_bases/radnerf/base.yaml:49 amp: true; utils/commons/trainer.py:307-382 autocast + GradScaler) runs its Linear layers in half:
    operands f16, accumulation fp32, master weights fp32, loss scaling outside.  Here: forward = gf_field_forward_train16 (f16 MFMA operands
    re-gathered from the fp32 master weights on the device, fp32 accumulators, fp32 outputs, every layer's activations saved as BINARY16 --
    half the save traffic of the fp32 node); backward = the dX chain on the f16 matrix pipe as well (k_field_backward16: transposed f16
    blocks, binary16 gradient rows in LDS, fp32 accumulators / masks / column sums / skinny transposes / lookup gradient / grid-feature
    gradients), writing its six [M,128] pre-activation gradients as binary16 so that the weight-gradient products run on half operands
    like autocast's own; every returned gradient is fp32.  `model.amp_backward = "f32"` keeps the fp32 chain (binary16 outputs only).  A scaled loss whose gradients leave the f16 range gives inf there, which
    is what GradScaler looks for (it skips the step and lowers the scale), exactly as with the reference's half Linear layers."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, model, xyz, dirs, cond_feat, ind_code, pos_tab, amb_tab, wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2):
        from . import fused
        with torch.autocast("cuda", enabled=False):
            st = fused.get_state(model)
            dev = xyz.device
            x = xyz.detach().reshape(-1, 3).float().contiguous()
            d = dirs.detach().reshape(-1, 3).float().contiguous()
            M = x.shape[0]
            f32, f16 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.float16, device=dev)
            sigma, rgb, amb = torch.empty(M, **f32), torch.empty(M, 3, **f32), torch.empty(M, 2, **f32)
            sv = {n: torch.empty(M, w, **f16) for n, w in (("f3", 32), ("ha1", 128), ("ha2", 128), ("f2", 32), ("hs1", 128), ("hs2", 128),
                                                          ("geo", 128), ("hc1", 128), ("sh", 16))}
            chunks = (M + 127) // 128
            masks = {n: torch.empty(chunks * 1024, dtype=torch.int16, device=dev) for n in ("m_ha1", "m_ha2", "m_hs1", "m_hs2", "m_hc1")}
            if M > 0:
                amb_bias = torch.mv(st.W_cond, cond_feat.detach().reshape(-1).float())
                col_bias = torch.mv(st.W_ind, ind_code.detach().reshape(-1).float()) if (st.W_ind is not None and ind_code is not None) else None
                if st.W_ind is not None and col_bias is None:
                    col_bias = torch.zeros(128, **f32)
                pack16 = _pack16_device(st, (wa1, wa2, ws1, ws2, ws3, wc1))
                f = fused.GfFrame()
                pe, ae = model.position_embedder, model.ambient_embedder
                f.bound = float(model.bound)
                f.pos_table, f.pos_offsets = ptr(pe.embeddings, torch.float32), ptr(pe.offsets, torch.int32)
                f.amb_table, f.amb_offsets = ptr(ae.embeddings, torch.float32), ptr(ae.offsets, torch.int32)
                f.pos_S, f.amb_S, f.base_res, f.gridtype, f.interp = st.pos_S, st.amb_S, st.base_res, st.gridtype, st.interp
                f.head_pack, f.head_pack16, f.amb_bias = ptr(st.head_pack), pack16.data_ptr(), ptr(amb_bias, torch.float32)
                saves = GfFieldSaves(**{n: t.data_ptr() for n, t in {**sv, **masks}.items()})
                check(lib().gf_field_forward_train16(C.byref(f), ptr(x, torch.float32), ptr(d, torch.float32), M, ptr(col_bias, torch.float32, allow_none=True),
                                                     ptr(sigma), ptr(rgb), ptr(amb), C.byref(saves), current_stream(dev)))
            ctx.model = model
            ctx.has_code = ind_code is not None
            ctx.save_for_backward(x, d, cond_feat, ind_code if ind_code is not None else torch.zeros(0, device=dev), sigma, rgb, amb,
                                  wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2, *[sv[n] for n in ("f3", "ha1", "ha2", "f2", "hs1", "hs2", "geo", "hc1", "sh")],
                                  *[masks[n] for n in ("m_hc1", "m_hs2", "m_hs1", "m_ha2", "m_ha1")])
        return sigma, rgb, amb

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_sigma, g_rgb, g_amb):
        with torch.autocast("cuda", enabled=False):
            return _HeadFieldAMP._backward(ctx, g_sigma, g_rgb, g_amb)

    @staticmethod
    def _backward(ctx, g_sigma, g_rgb, g_amb):
        (x, d, cond_feat, ind_code, sigma, rgb, amb, wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2, f3, ha1, ha2, f2, hs1, hs2, geo, hc1, sh,
         m_hc1, m_hs2, m_hs1, m_ha2, m_ha1) = ctx.saved_tensors
        from . import fused
        model = ctx.model
        M, dev = x.shape[0], x.device
        f32, f16 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.float16, device=dev)
        z = lambda *shape: torch.zeros(*shape, **f32)
        g_sigma = g_sigma.float().contiguous() if g_sigma is not None else z(M)
        g_rgb = g_rgb.float().contiguous() if g_rgb is not None else z(M, 3)
        g_amb = g_amb.float().contiguous() if g_amb is not None else z(M, 2)
        cond = cond_feat.reshape(-1).float()
        st = fused.get_state(model)
        f16_chain = getattr(model, "amp_backward", "f16") == "f16"      # "f32": the fp32 dX chain writing binary16 rows (stage 1 of round 6)
        flat = torch.cat([z(1)] + [w.detach().reshape(-1).float() for w in (wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2)])
        if f16_chain:
            if getattr(st, "_bwd16_idx", None) is None:
                st._bwd16_idx = _bwd16_stream_index([tuple(w.shape) for w in (wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2)]).to(dev)
                assert st._bwd16_idx.numel() == lib().gf_field_bwd16_stream_halves()
            stream = flat.half()[st._bwd16_idx]
        else:
            if getattr(st, "_bwd_idx", None) is None:
                st._bwd_idx = _bwd_stream_index([tuple(w.shape) for w in (wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2)]).to(dev)
            stream = flat[st._bwd_idx]
        out = {n: torch.empty(M, 128, **f16) for n in ("g_hc1", "g_geo", "g_hs2", "g_hs1", "g_ha2", "g_ha1")}
        out.update({n: torch.empty(M, w, **f32) for n, w in (("g_zc", 3), ("g_za", 2), ("g_f3", 32), ("g_f2", 32))})
        out["g_h0"] = torch.empty(M, **f32)
        out["s_hc1"], out["s_ha1"] = z(128), z(128)
        level_max = torch.zeros(32, dtype=torch.int32, device=dev)
        if M > 0:
            f = fused.GfFrame()
            pe, ae = model.position_embedder, model.ambient_embedder
            f.bound = float(model.bound)
            f.pos_offsets = ptr(pe.offsets, torch.int32)
            f.amb_table, f.amb_offsets = ptr(ae.embeddings, torch.float32), ptr(ae.offsets, torch.int32)
            f.pos_S, f.amb_S, f.base_res, f.gridtype, f.interp = st.pos_S, st.amb_S, st.base_res, st.gridtype, st.interp
            f.head_pack = ptr(st.head_pack)
            g = GfFieldGrads(g_sigma=g_sigma.data_ptr(), g_rgb=g_rgb.data_ptr(), g_amb=g_amb.data_ptr(), sigma=sigma.data_ptr(), rgb=rgb.data_ptr(),
                             amb=amb.data_ptr(), m_hc1=m_hc1.data_ptr(), m_hs2=m_hs2.data_ptr(), m_hs1=m_hs1.data_ptr(), m_ha2=m_ha2.data_ptr(),
                             m_ha1=m_ha1.data_ptr(), level_max=level_max.data_ptr(), out16=2 if f16_chain else 1, **{n: t.data_ptr() for n, t in out.items()})
            check(lib().gf_field_backward(C.byref(f), ptr(stream), M, C.byref(g), current_stream(dev)))
        g_zc, g_h0, g_za = out["g_zc"], out["g_h0"], out["g_za"]
        g_hc1, g_geo, g_hs2, g_hs1, g_ha2, g_ha1 = (out[n] for n in ("g_hc1", "g_geo", "g_hs2", "g_hs1", "g_ha2", "g_ha1"))
        # ---- weight gradients: the eight tall products G^T X (half operands, fp32 accumulation) in one launch + one fixed-order reduction
        # (gf_field_wgrad16, csrc/field_wgrad.hip; `model.amp_wgrad = "gemm"` keeps round 6 stage 2's batched library products)
        s_hc1, s_ha1 = out["s_hc1"], out["s_ha1"]
        g_code = None
        if getattr(model, "amp_wgrad", "fused") == "fused":
            g_wc2, g_wc1, g_ws3, g_ws2, g_ws1, g_wa3, g_wa2, g_wa1 = _fused_wgrad(
                st, True, M, dev, (wc2, wc1, ws3, ws2, ws1, wa3, wa2, wa1),
                dict(f3=f3, ha1=ha1, ha2=ha2, f2=f2, hs1=hs1, hs2=hs2, geo=geo, hc1=hc1, sh=sh, g_hc1=g_hc1, g_geo=g_geo, g_hs2=g_hs2, g_hs1=g_hs1,
                     g_ha2=g_ha2, g_ha1=g_ha1, g_zc=g_zc, g_h0=g_h0, g_za=g_za))
            if ctx.has_code:
                g_wc1[:, 144:] = torch.outer(s_hc1, ind_code.reshape(-1).float())
            elif wc1.shape[1] > 144:
                g_wc1[:, 144:] = 0
            g_wa1[:, 32:] = torch.outer(s_ha1, cond)
        else:
            tn = lambda gg, xx: _tall_tn(gg.half() if gg.dtype != torch.float16 else gg, xx, out_dtype=torch.float32)
            g_wc2 = tn(g_zc, hc1)
            parts = [tn(g_hc1, sh), tn(g_hc1, geo)]
            if ctx.has_code:
                parts.append(torch.outer(s_hc1, ind_code.reshape(-1).float()))
            g_wc1 = torch.cat(parts, dim=1)
            g_ws3 = torch.cat([tn(g_h0.unsqueeze(1), hs2), tn(g_geo, hs2)], dim=0)
            g_ws2 = tn(g_hs2, hs1)
            g_ws1 = torch.cat([tn(g_hs1, f3), tn(g_hs1, f2)], dim=1)
            g_wa3 = tn(g_za, ha2)
            g_wa2 = tn(g_ha2, ha1)
            g_wa1 = torch.cat([tn(g_ha1, f3), torch.outer(s_ha1, cond)], dim=1)
        if ctx.has_code:
            g_code = (s_hc1 @ wc1[:, 144:].float()).view_as(ind_code)
        g_cond = (s_ha1 @ wa1[:, 32:].float()).view_as(cond_feat).to(cond_feat.dtype)
        lm = level_max if M > 0 else None
        g_amb_tab, _ = _grid_backward(model.ambient_embedder, (amb + 1) / 2, out["g_f2"], False, level_major=True, level_max=None if lm is None else lm[16:])
        g_pos_tab, _ = _grid_backward(model.position_embedder, (x + model.bound) / (2 * model.bound), out["g_f3"], False, level_major=True,
                                      level_max=None if lm is None else lm[:16])
        return (None, None, None, g_cond, g_code, g_pos_tab, g_amb_tab, g_wa1, g_wa2, g_wa3, g_ws1, g_ws2, g_ws3, g_wc1, g_wc2)


_MAX_POINTS_F32, _MAX_POINTS_AMP = (1 << 23) - 4096, (1 << 24) - 4096


def head_field(model, position, direction, cond_feat, individual_code):
    """sigma [M], color [M,3], ambient [M,2] of RADNeRF.forward with gradients to the model's tables, weights, cond_feat and code.
    Under torch.autocast(float16) with `model.amp_field == "f16"` (the default) the node runs on the f16 tier (_HeadFieldAMP), as the
    reference's autocast step runs its Linear layers in half; `amp_field = "f32"` keeps the exact-fp32 node under autocast (round 5)."""
    a, s, c = model.ambient_net.net, model.sigma_net.net, model.color_net.net
    node = _HeadField
    if torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.float16 and getattr(model, "amp_field", "f16") == "f16":
        node = _HeadFieldAMP
    model._last_field_node = "amp_f16" if node is _HeadFieldAMP else "f32"      # which arithmetic the last call ran in (tests assert the tier)
    # the training kernels address their [M,128] rows with 32-bit byte offsets (M < 2^24 binary16, 2^23 fp32: gf_field_forward_train*): a
    # longer point list -- 65 536 rays at > 128 samples each -- goes through the node in slabs (each slab its own saves; autograd adds the
    # parameter gradients)
    limit = _MAX_POINTS_AMP if node is _HeadFieldAMP else _MAX_POINTS_F32
    if position.shape[0] > limit:
        parts = [head_field(model, position[i:i + limit], direction[i:i + limit], cond_feat, individual_code) for i in range(0, position.shape[0], limit)]
        return tuple(torch.cat([p[k] for p in parts], dim=0) for k in range(3))
    return node.apply(model, position, direction, cond_feat, individual_code, model.position_embedder.embeddings,
                            model.ambient_embedder.embeddings, a[0].weight, a[1].weight, a[2].weight, s[0].weight, s[1].weight, s[2].weight,
                            c[0].weight, c[1].weight)
