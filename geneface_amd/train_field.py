"""Training-time field of the RAD-NeRF head: RADNeRF.forward (modules/radnerfs/radnerf.py:73-105) as ONE autograd node.

The reference -- and this package's op-by-op route -- builds the graph out of ~40 torch ops per call: two grid encodes, SH, 8 GEMMs,
cats of per-point copies of the condition vector and the identity code, ReLUs, tanh / exp / sigmoid.  On a 1 M-point training batch the
GEMMs are 8 ms and the glue between them 5.6 ms of an 18 ms step (profiles/round2/r2t_train_kernel_stats.csv).

Forward: one launch of the renderer's own field core over the point list (gf_field_forward_train, csrc/frame_head.hip), which also leaves
every layer's activations as [M, width] matrices.  Backward: the chain rule written out once, by hand, over those matrices:
  * per-point constants (condition vector, identity code) enter as bias vectors, never as [M, 64] / [M, 4] copies; concatenated inputs
    become sums of products with column blocks of the weight, so no `cat` is ever materialised;
  * ReLU masks come from the saved post-activation values (aten.threshold_backward, one fused pass per layer);
  * weight gradients use the split reduction of cond_encoder._linear_tall (a [128 x 128] result over a 10^6-long reduction);
  * the two grid tables and the 2-D lookup's input gradient go through the library's own backward kernels (gf_grid_encode_backward).
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
    _fields_ = [(n, _vp) for n in ("f3", "ha1", "ha2", "f2", "hs1", "hs2", "geo", "hc1")]


def _tall_tn(g, x):
    """g^T @ x for g [M,O], x [M,I] with M ~ 10^6: batched partial products + a sum (see cond_encoder._linear_tall)."""
    B, O, I = x.shape[0], g.shape[1], x.shape[1]
    S = max(1, B // 4096)
    rows = B // S
    main = S * rows
    gw = torch.bmm(g[:main].view(S, rows, O).transpose(1, 2), x[:main].view(S, rows, I)).sum(0)
    if main < B:
        gw = gw + g[main:].t() @ x[main:]
    return gw


def _grid_backward(enc, x01, grad, want_input_grad):
    """Table gradient (and d/d x01 through a freshly evaluated dy_dx) of GridEncoder `enc` at inputs x01 [M,D] for an output gradient
    `grad` [M, L*C]: the library's backward kernels, without re-entering autograd."""
    L_ = lib()
    dev = x01.device
    B, D = x01.shape
    L, Cc = enc.num_levels, enc.level_dim
    S = float(np.log2(enc.per_level_scale))
    dy_dx = gin = None
    if want_input_grad:
        dy_dx = torch.empty(B, L * D * Cc, device=dev, dtype=torch.float32)
        scratch = torch.empty(B, L * Cc, device=dev, dtype=torch.float32)
        check(L_.gf_grid_encode_forward_blc(ptr(x01, torch.float32), ptr(enc.embeddings, torch.float32), ptr(enc.offsets, torch.int32), ptr(scratch),
                                            B, D, Cc, L, S, int(enc.base_resolution), ptr(dy_dx), enc.gridtype_id, int(bool(enc.align_corners)),
                                            enc.interp_id, current_stream(dev)))
        gin = torch.zeros_like(x01)
    g_lbc = grad.view(B, L, Cc).permute(1, 0, 2).contiguous()
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
                                                      ("geo", 128), ("hc1", 128))}
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
            saves = GfFieldSaves(**{n: t.data_ptr() for n, t in sv.items()})
            check(lib().gf_field_forward_train(C.byref(f), ptr(x, torch.float32), ptr(d, torch.float32), M, ptr(col_bias, torch.float32, allow_none=True),
                                               ptr(sigma), ptr(rgb), ptr(amb), C.byref(saves), current_stream(dev)))
        ctx.model = model
        ctx.has_code = ind_code is not None
        ctx.save_for_backward(x, d, cond_feat, ind_code if ind_code is not None else torch.zeros(0, device=dev), sigma, rgb, amb,
                              wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2, *[sv[n] for n in ("f3", "ha1", "ha2", "f2", "hs1", "hs2", "geo", "hc1")])
        return sigma, rgb, amb

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g_sigma, g_rgb, g_amb):
        (x, d, cond_feat, ind_code, sigma, rgb, amb, wa1, wa2, wa3, ws1, ws2, ws3, wc1, wc2, f3, ha1, ha2, f2, hs1, hs2, geo, hc1) = ctx.saved_tensors
        model = ctx.model
        M = x.shape[0]
        tb = torch.ops.aten.threshold_backward
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=x.device)
        g_sigma = g_sigma.float().contiguous() if g_sigma is not None else z(M)
        g_rgb = g_rgb.float().contiguous() if g_rgb is not None else z(M, 3)
        g_amb = g_amb.float().contiguous() if g_amb is not None else z(M, 2)
        cond = cond_feat.reshape(-1).float()
        # ---- colour net: rgb = sigmoid(W_c2 relu(W_c1 [sh | geo | code]))
        g_zc = g_rgb * rgb * (1 - rgb)                                            # [M,3]
        g_wc2 = _tall_tn(g_zc, hc1)
        g_hc1 = tb(g_zc @ wc2, hc1, 0)                                            # [M,128]
        sh = model.direction_embedder(d)                                          # [M,16] (no gradient: directions are data)
        s_hc1 = g_hc1.sum(0)
        parts = [_tall_tn(g_hc1, sh), _tall_tn(g_hc1, geo)]
        g_code = None
        if ctx.has_code:
            parts.append(torch.outer(s_hc1, ind_code.reshape(-1).float()))
            g_code = (s_hc1 @ wc1[:, 144:]).view_as(ind_code)
        g_wc1 = torch.cat(parts, dim=1)
        g_geo = g_hc1 @ wc1[:, 16:144]                                            # [M,128]
        # ---- sigma net: [log sigma | geo] = W_s3 relu(W_s2 relu(W_s1 [f3 | f2])), sigma = trunc_exp(.)
        g_h0 = g_sigma * sigma.clamp(min=float(np.exp(-15.0)), max=float(np.exp(15.0)))   # trunc_exp backward (utils.py:44-49)
        g_ws3 = torch.cat([_tall_tn(g_h0.unsqueeze(1), hs2), _tall_tn(g_geo, hs2)], dim=0)
        g_hs2 = torch.addmm(torch.outer(g_h0, ws3[0]), g_geo, ws3[1:])
        g_hs2 = tb(g_hs2, hs2, 0)
        g_ws2 = _tall_tn(g_hs2, hs1)
        g_hs1 = tb(g_hs2 @ ws2, hs1, 0)
        g_ws1 = torch.cat([_tall_tn(g_hs1, f3), _tall_tn(g_hs1, f2)], dim=1)
        g_f3 = g_hs1 @ ws1[:, :32]
        g_f2 = g_hs1 @ ws1[:, 32:]
        # ---- 2-D grid at (ambient + 1) / 2: table gradient and d/d ambient
        g_amb_tab, g_x2 = _grid_backward(model.ambient_embedder, ((amb + 1) / 2).contiguous(), g_f2.contiguous(), True)
        # ---- ambient net: ambient = tanh(W_a3 relu(W_a2 relu(W_a1 [f3 | cond])))
        g_za = (g_amb + 0.5 * g_x2) * (1 - amb * amb)                             # [M,2]
        g_wa3 = _tall_tn(g_za, ha2)
        g_ha2 = tb(g_za @ wa3, ha2, 0)
        g_wa2 = _tall_tn(g_ha2, ha1)
        g_ha1 = tb(g_ha2 @ wa2, ha1, 0)
        s_ha1 = g_ha1.sum(0)
        g_wa1 = torch.cat([_tall_tn(g_ha1, f3), torch.outer(s_ha1, cond)], dim=1)
        g_cond = (s_ha1 @ wa1[:, 32:]).view_as(cond_feat)
        g_f3 = g_f3.addmm_(g_ha1, wa1[:, :32])
        # ---- 3-D grid table
        g_pos_tab, _ = _grid_backward(model.position_embedder, ((x + model.bound) / (2 * model.bound)).contiguous(), g_f3, False)
        return (None, None, None, g_cond, g_code, g_pos_tab, g_amb_tab, g_wa1, g_wa2, g_wa3, g_ws1, g_ws2, g_ws3, g_wc1, g_wc2)


def head_field(model, position, direction, cond_feat, individual_code):
    """sigma [M], color [M,3], ambient [M,2] of RADNeRF.forward with gradients to the model's tables, weights, cond_feat and code."""
    a, s, c = model.ambient_net.net, model.sigma_net.net, model.color_net.net
    return _HeadField.apply(model, position, direction, cond_feat, individual_code, model.position_embedder.embeddings,
                            model.ambient_embedder.embeddings, a[0].weight, a[1].weight, a[2].weight, s[0].weight, s[1].weight, s[2].weight,
                            c[0].weight, c[1].weight)
