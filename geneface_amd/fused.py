"""Host side of the fused frame path (gf_render_head / gf_render_torso of include/geneface_hip.h).

`render_head_fused` / `render_torso_fused` are what `NeRFRenderer.render` / `RADNeRFTorso.render` dispatch to for
`render_impl="fused"`: same inputs, same result dict, but a frame is a handful of kernel launches on the current
stream with no host synchronisation (the reference's loop, renderer.py:316-351, syncs once per march iteration).
`render_frame_fused` is the frame-loop step used by FramePipeline (rays generated in-kernel from the pose, uint8 out).

Per-model state (packed weights, fold matrices, workspace) is built once and cached on the module.
"""
import ctypes as C
import itertools
import random

import numpy as np
import torch

from .lib import check, current_stream, lib, ptr

_u32, _f32, _vp = C.c_uint32, C.c_float, C.c_void_p


class GfFrame(C.Structure):
    """ctypes mirror of gf_frame_t (include/geneface_hip.h); size is checked against gf_frame_sizeof()."""
    _fields_ = [
        ("n_rays", _u32), ("img_h", _u32), ("img_w", _u32), ("_pad0", _u32),
        ("rays_o", _vp), ("rays_d", _vp),
        ("pose", _f32 * 12), ("intrinsics", _f32 * 4),
        ("aabb", _vp), ("bitfield", _vp),
        ("min_near", _f32), ("bound", _f32), ("dt_gamma", _f32), ("T_thresh", _f32),
        ("max_steps", _u32), ("cascade", _u32), ("grid_size", _u32), ("has_occ_aabb", _u32),
        ("occ_aabb", _f32 * 6), ("_pad1", _f32 * 2),
        ("pos_table", _vp), ("pos_offsets", _vp), ("amb_table", _vp), ("amb_offsets", _vp),
        ("pos_S", _f32), ("amb_S", _f32),
        ("base_res", _u32), ("gridtype", _u32), ("interp", _u32), ("precision", _u32),
        ("head_pack", _vp), ("head_pack16", _vp), ("head_pack_split", _vp), ("amb_bias", _vp),
        ("torso_pack", _vp), ("torso_bias", _vp), ("torso_table", _vp), ("torso_offsets", _vp), ("torso_occ", _vp), ("bg_coords", _vp),
        ("torso_S", _f32), ("torso_thresh", _f32), ("torso_shrink", _f32), ("_pad3", _f32),
        ("bg_color", _vp), ("out_rgb", _vp), ("out_depth", _vp), ("out_rgb8", _vp), ("out_torso_alpha", _vp),
        ("out_torso_rgb", _vp), ("out_deform", _vp),
        ("workspace", _vp),
        ("perturb_noise", _vp), ("torso_ha_pack", _vp), ("torso_ha_ws", _vp), ("torso_ha_branch", _u32), ("_pad4", _u32),
        ("torso_mask_list", _vp), ("torso_mask_dense_of", _vp), ("torso_mask_count", _vp),
    ]


class GfCond(C.Structure):
    """ctypes mirror of gf_cond_t (include/geneface_hip.h); size is checked against gf_cond_sizeof()."""
    _fields_ = [
        ("cond", _vp), ("S", _u32), ("T", _u32), ("C", _u32), ("dim_aud", _u32),
        ("conv_w", _vp * 4), ("conv_b", _vp * 4), ("conv_stride", _u32 * 4), ("conv_ch", _u32 * 5),
        ("fc1_w", _vp), ("fc1_b", _vp), ("fc2_w", _vp), ("fc2_b", _vp),
        ("att_w", _vp * 5), ("att_b", _vp * 5), ("att_lin_w", _vp), ("att_lin_b", _vp),
        ("cond_feat", _vp), ("W_cond", _vp), ("amb_bias", _vp),
        ("pose6", _vp), ("torso_code", _vp), ("code_dim", _u32), ("_pad", _u32),
        ("W_tconst", _vp), ("torso_bias", _vp),
    ]


def _np(t):
    return np.ascontiguousarray(t.detach().float().cpu().numpy())


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


def _pack_index(pack, shapes, n_out):
    """Where every float of a packed layout comes from: run the HOST packer on weights whose values are their own (1-based) flat indices
    and read the layout back as an index map (0 = a slot the packer fills with zero / a derived value).  With it a weight update becomes
    one device-side gather instead of a device -> host -> pack -> device round trip per training step."""
    arrays, off = [], 1
    for shp in shapes:
        n = int(np.prod(shp))
        arrays.append(np.arange(off, off + n, dtype=np.float32).reshape(shp))
        off += n
    assert off < (1 << 24), "flat indices must be exact in fp32"
    out = np.zeros(n_out, dtype=np.float32)
    pack(arrays, out)
    idx = out.astype(np.int64)
    assert np.array_equal(idx.astype(np.float32), out) and idx.min() >= 0 and idx.max() < off
    return torch.from_numpy(idx)


class FusedState:
    """Everything the kernels need that depends only on the model (built once, device resident)."""

    def __init__(self, model):
        L = lib()
        assert C.sizeof(GfFrame) == L.gf_frame_sizeof(), "gf_frame_t layout mismatch between fused.py and geneface_hip.h"
        self.check_architecture(model)
        dev = model.density_bitfield.device
        if dev.type != "cuda":
            raise RuntimeError("the fused render path needs the model on a HIP device (there is no CPU path)")
        self.device = dev
        self.has_torso = hasattr(model, "torso_deform_net")

        perm = (C.c_uint32 * 128)()
        check(L.gf_clayout_perm(perm))
        self.perm = torch.tensor(list(perm), dtype=torch.long)

        a, s, c = model.ambient_net.net, model.sigma_net.net, model.color_net.net
        ind = _np(model.individual_embeddings[0]) if model.individual_embedding_dim > 0 else None
        pack = np.empty(L.gf_head_pack_floats(), dtype=np.float32)
        check(L.gf_head_pack(_hp(_np(a[0].weight)), _hp(_np(a[1].weight)), _hp(_np(a[2].weight)), _hp(_np(s[0].weight)),
                             _hp(_np(s[1].weight)), _hp(_np(s[2].weight)), _hp(_np(c[0].weight)), _hp(_np(c[1].weight)),
                             _hp(ind) if ind is not None else None, _hp(pack)))
        self.head_pack = torch.from_numpy(pack).to(dev)
        self._head_pack16 = None            # fast path: packed on first use (pack16)
        self._head_pack_split = None        # split path: packed on first use (pack_split)
        # ambient L1's cond_feat columns, rows in accumulator-layout order: amb_bias = W_cond @ cond_feat per frame
        self.W_cond = a[0].weight.detach()[self.perm.to(dev), 32:].contiguous()
        # colour L1's identity-code columns, same row order: col_bias = W_ind @ individual_code (field_forward with a per-call code)
        self.W_ind = c[0].weight.detach()[self.perm.to(dev), 144:].contiguous() if model.individual_embedding_dim > 0 else None
        if self.W_ind is not None:   # the folded identity-code bias is always this device product (refresh_weights computes the same one)
            off = int(L.gf_head_pack_colbias_offset())
            self.head_pack[off:off + 128] = torch.mv(self.W_ind, model.individual_embeddings[0].detach().float())

        pe, ae = model.position_embedder, model.ambient_embedder
        self.pos_S, self.amb_S = float(np.log2(pe.per_level_scale)), float(np.log2(ae.per_level_scale))
        self.gridtype, self.interp, self.base_res = pe.gridtype_id, pe.interp_id, int(pe.base_resolution)
        for enc, D, S in ((pe, 3, self.pos_S), (ae, 2, self.amb_S)):
            off = np.ascontiguousarray(enc.offsets.detach().cpu().numpy().astype(np.int32))
            if L.gf_grid_levels_fusable(_hp(off), enc.num_levels, D, S, self.base_res) != 0:
                raise NotImplementedError(L.gf_last_error().decode() + " (use render_impl='ops')")
        # box around the occupied cells of the density bitfield: the marcher stops at a ray's exit from it
        bits = np.ascontiguousarray(model.density_bitfield.detach().cpu().numpy().astype(np.uint8))
        box = np.zeros(6, dtype=np.float32)
        check(L.gf_occupancy_aabb(_hp(bits), int(model.cascade), int(model.grid_size), float(model.bound), _hp(box)))
        self.occ_aabb = [float(v) for v in box]

        self.head_aware = bool(getattr(model, "torso_head_aware", False)) and self.has_torso
        if self.has_torso:
            d, cn = model.torso_deform_net.net, model.torso_canonicial_net.net
            tpack = np.empty(L.gf_torso_pack_floats(), dtype=np.float32)
            if self.head_aware:   # radnerf_torso.py:36-46: 16 encoder columns appended to both first layers
                hpack = np.empty(L.gf_torso_ha_pack_floats(), dtype=np.float32)
                check(L.gf_torso_pack_ha(*[_hp(_np(w)) for w in self._torso_weights(model)], _hp(tpack), _hp(hpack)))
                self.torso_ha_pack = torch.from_numpy(hpack).to(dev)
            else:
                check(L.gf_torso_pack(_hp(_np(d[0].weight)), _hp(_np(d[1].weight)), _hp(_np(d[2].weight)), _hp(_np(cn[0].weight)),
                                      _hp(_np(cn[1].weight)), _hp(_np(cn[2].weight)), _hp(tpack)))
            self.torso_pack = torch.from_numpy(tpack).to(dev)
            p64, p32 = self.perm[:64].to(dev), self.perm[:32].to(dev)
            # per-frame constants [enc(pose) 54 | code 8] fold into the first-layer biases
            self.W_tconst = torch.cat([d[0].weight.detach()[p64, 42:], cn[0].weight.detach()[p32, 74:]], dim=0).contiguous()  # [96, 62]
            self.torso_S = float(np.log2(model.torso_embedder.per_level_scale))
        self._ws = {}
        self.cond = self._build_cond(model)
        self._head_idx = self._torso_idx = None      # index maps of the packed layouts (refresh_weights), built on first use
        # everything above is a COPY of (or a raw pointer into) model state: remember what it was built from (see get_state)
        self._watch = _watched_tensors(model)
        self.stamp = _stamp(self._watch)

    # ---- in-place weight updates (a training step, load_state_dict, a broadcast): repack on the device, no host round trip
    @staticmethod
    def _head_weights(model):
        a, s, c = model.ambient_net.net, model.sigma_net.net, model.color_net.net
        return [a[0].weight, a[1].weight, a[2].weight, s[0].weight, s[1].weight, s[2].weight, c[0].weight, c[1].weight]

    @staticmethod
    def _torso_weights(model):
        d, cn = model.torso_deform_net.net, model.torso_canonicial_net.net
        ws = [d[0].weight, d[1].weight, d[2].weight, cn[0].weight, cn[1].weight, cn[2].weight]
        if getattr(model, "torso_head_aware", False):
            e = model.head_color_weights_encoder
            ws += [e[0].weight, e[0].bias, e[2].weight, e[2].bias, e[4].weight, e[4].bias]
        return ws

    def _torso_codes(self, model):
        """The per-frame constant vector the cond kernel folds into torso_bias behind enc(pose): the identity code and, for head-aware
        models, the encoder's 16 outputs -- for the coin's 'no head' outcome the encoding of a black, transparent head (a constant,
        radnerf_torso.py:69-71), for the other outcome zeros (the per-pixel encoding enters through the extra MFMA steps instead)."""
        code = model.torso_individual_codes[0].detach().float() if model.torso_individual_embedding_dim > 0 else torch.zeros(0, device=self.device)
        if not self.head_aware:
            return (code.contiguous() if code.numel() else None), None
        with torch.no_grad():
            e0 = model.head_color_weights_encoder(torch.zeros(1, 4, device=self.device)).reshape(-1).float()
        return torch.cat([code, e0]).contiguous(), torch.cat([code, torch.zeros_like(e0)]).contiguous()

    def refresh_weights(self, model):
        """The packed copies follow the model's current weights: per pack one cat + one gather on the device."""
        L = lib()
        dev = self.device
        if self._head_idx is None:
            ws = self._head_weights(model)
            self._head_idx = _pack_index(lambda arr, out: check(L.gf_head_pack(*[_hp(a) for a in arr], None, _hp(out))),
                                         [tuple(w.shape) for w in ws], L.gf_head_pack_floats()).to(dev)
            self._colbias_off = int(L.gf_head_pack_colbias_offset())
        zero = torch.zeros(1, dtype=torch.float32, device=dev)
        flat = torch.cat([zero] + [w.detach().reshape(-1).float() for w in self._head_weights(model)])
        self.head_pack = flat[self._head_idx]
        a, c = model.ambient_net.net, model.color_net.net
        perm = self.perm.to(dev)
        self.W_cond = a[0].weight.detach()[perm, 32:].contiguous()
        if model.individual_embedding_dim > 0:
            self.W_ind = c[0].weight.detach()[perm, 144:].contiguous()
            self.head_pack[self._colbias_off:self._colbias_off + 128] = torch.mv(self.W_ind, model.individual_embeddings[0].detach().float())
        self._head_pack16 = self._head_pack_split = None
        if self.has_torso:
            n_main = L.gf_torso_pack_floats()
            if self._torso_idx is None:
                ws = self._torso_weights(model)
                if self.head_aware:
                    packer = lambda arr, out: check(L.gf_torso_pack_ha(*[_hp(a) for a in arr], _hp(out[:n_main]), _hp(out[n_main:])))
                    n_out = n_main + L.gf_torso_ha_pack_floats()
                else:
                    packer, n_out = (lambda arr, out: check(L.gf_torso_pack(*[_hp(a) for a in arr], _hp(out)))), n_main
                self._torso_idx = _pack_index(packer, [tuple(w.shape) for w in ws], n_out).to(dev)
            flat = torch.cat([zero] + [w.detach().reshape(-1).float() for w in self._torso_weights(model)])
            both = flat[self._torso_idx]
            self.torso_pack = both[:n_main].contiguous()
            if self.head_aware:
                self.torso_ha_pack = both[n_main:].contiguous()
            d, cn = model.torso_deform_net.net, model.torso_canonicial_net.net
            self.W_tconst = torch.cat([d[0].weight.detach()[perm[:64], 42:], cn[0].weight.detach()[perm[:32], 74:]], dim=0).contiguous()
            self._torso_code, self._torso_code_ha = self._torso_codes(model)
        if self.cond is not None:          # the cached gf_cond_t points at W_cond / W_tconst / the torso code
            self.cond.W_cond = ptr(self.W_cond)
            if self.has_torso:
                self.cond.W_tconst = ptr(self.W_tconst)
                self.cond.torso_code = ptr(self._torso_code) if self._torso_code is not None else None

    def refresh_occupancy(self, model):
        bits = np.ascontiguousarray(model.density_bitfield.detach().cpu().numpy().astype(np.uint8))
        box = np.zeros(6, dtype=np.float32)
        check(lib().gf_occupancy_aabb(_hp(bits), int(model.cascade), int(model.grid_size), float(model.bound), _hp(box)))
        self.occ_aabb = [float(v) for v in box]

    def pack16(self, model):
        """f16 A-operand streams of the head's six MFMA layers (gf_frame_t.precision = 1, the "fast" parity tier)."""
        if self._head_pack16 is None:
            L = lib()
            a, s, c = model.ambient_net.net, model.sigma_net.net, model.color_net.net
            out = np.empty(L.gf_head_pack16_halves(), dtype=np.uint16)
            check(L.gf_head_pack16(_hp(_np(a[0].weight)), _hp(_np(a[1].weight)), _hp(_np(s[0].weight)), _hp(_np(s[1].weight)),
                                   _hp(_np(s[2].weight)), _hp(_np(c[0].weight)), _hp(out)))
            self._head_pack16 = torch.from_numpy(out.view(np.int16)).to(self.device)
        return self._head_pack16

    def pack_split(self, model):
        """Two-term f16 splits of the head's six MFMA layers (gf_frame_t.precision = 2, the "split" tier: fp32 values on the f16 matrix pipe)."""
        if self._head_pack_split is None:
            L = lib()
            a, s, c = model.ambient_net.net, model.sigma_net.net, model.color_net.net
            out = np.empty(L.gf_head_pack_split_halves(), dtype=np.uint16)
            check(L.gf_head_pack_split(_hp(_np(a[0].weight)), _hp(_np(a[1].weight)), _hp(_np(s[0].weight)), _hp(_np(s[1].weight)),
                                       _hp(_np(s[2].weight)), _hp(_np(c[0].weight)), _hp(out)))
            self._head_pack_split = torch.from_numpy(out.view(np.int16)).to(self.device)
        return self._head_pack_split

    def _build_cond(self, model):
        """gf_cond_t with every weight pointer filled in (None when the encoder is not the AudioNet + AudioAttNet pair the
        HIP kernel implements: the torch modules then run instead, still on the GPU)."""
        pre, att = getattr(model, "cond_prenet", None), getattr(model, "cond_att_net", None)
        if pre is None or att is None or not getattr(model, "with_att", False):
            return None
        L = lib()
        assert C.sizeof(GfCond) == L.gf_cond_sizeof(), "gf_cond_t layout mismatch between fused.py and geneface_hip.h"
        convs = [m for m in pre.encoder_conv if isinstance(m, torch.nn.Conv1d)]
        aconvs = [m for m in att.attentionConvNet if isinstance(m, torch.nn.Conv1d)]
        fcs = [m for m in pre.encoder_fc1 if isinstance(m, torch.nn.Linear)]
        lin = att.attentionNet[0]
        if len(convs) != 4 or len(aconvs) != 5 or len(fcs) != 2 or [c.out_channels for c in aconvs] != [16, 8, 4, 2, 1]:
            return None
        c = GfCond()
        f32 = torch.float32
        for i, m in enumerate(convs):
            c.conv_w[i], c.conv_b[i], c.conv_stride[i] = ptr(m.weight, f32), ptr(m.bias, f32), int(m.stride[0])
            c.conv_ch[i] = int(m.in_channels)
        c.conv_ch[4] = int(convs[3].out_channels)
        c.fc1_w, c.fc1_b, c.fc2_w, c.fc2_b = ptr(fcs[0].weight, f32), ptr(fcs[0].bias, f32), ptr(fcs[1].weight, f32), ptr(fcs[1].bias, f32)
        for i, m in enumerate(aconvs):
            c.att_w[i], c.att_b[i] = ptr(m.weight, f32), ptr(m.bias, f32)
        c.att_lin_w, c.att_lin_b = ptr(lin.weight, f32), ptr(lin.bias, f32)
        c.S, c.T, c.C, c.dim_aud = int(att.seq_len), int(pre.win_size), int(convs[0].in_channels), int(pre.dim_aud)
        c.W_cond = ptr(self.W_cond)
        if self.has_torso:
            c.W_tconst = ptr(self.W_tconst)
            self._torso_code, self._torso_code_ha = self._torso_codes(model)
            c.code_dim = int(self._torso_code.numel()) if self._torso_code is not None else 0
            c.torso_code = ptr(self._torso_code) if self._torso_code is not None else None
        if L.gf_cond_check(C.byref(c)) != 0:
            return None   # window / encoder outside the kernel's limits: the torch modules serve it
        return c

    @staticmethod
    def check_architecture(model):
        """The fused kernels are specialised for the GeneFace RAD-NeRF architecture (base.yaml:85-102)."""
        want = {"position_embedder.embeddings": 2, "ambient_embedder.embeddings": 2}
        sd = dict(model.named_parameters())
        for k, cdim in want.items():
            if sd[k].shape[1] != cdim:
                raise NotImplementedError(f"fused path: {k} must have level_dim {cdim}")
        shapes = {"ambient_net.net.0.weight": (128, 96), "ambient_net.net.1.weight": (128, 128), "ambient_net.net.2.weight": (2, 128),
                  "sigma_net.net.0.weight": (128, 64), "sigma_net.net.1.weight": (128, 128), "sigma_net.net.2.weight": (129, 128),
                  "color_net.net.1.weight": (3, 128)}
        for k, shp in shapes.items():
            if k not in sd or tuple(sd[k].shape) != shp:
                raise NotImplementedError(f"fused path: {k} must have shape {shp} (use render_impl='ops' for other architectures)")
        if sd["color_net.net.0.weight"].shape != (128, 144 + model.individual_embedding_dim) or model.individual_embedding_dim not in (0, 4):
            raise NotImplementedError("fused path: color_net.net.0 must be [128, 16+128+4]")
        if model.position_embedder.num_levels != 16 or model.ambient_embedder.num_levels != 16:
            raise NotImplementedError("fused path: grids must have 16 levels")
        if hasattr(model, "torso_deform_net"):
            ha = 16 if getattr(model, "torso_head_aware", False) else 0     # radnerf_torso.py:36-46: the head-colour encoding widens both first layers
            tshapes = {"torso_deform_net.net.0.weight": (64, 104 + ha), "torso_deform_net.net.1.weight": (64, 64), "torso_deform_net.net.2.weight": (2, 64),
                       "torso_canonicial_net.net.0.weight": (32, 136 + ha), "torso_canonicial_net.net.1.weight": (32, 32),
                       "torso_canonicial_net.net.2.weight": (4, 32)}
            for k, shp in tshapes.items():
                if tuple(sd[k].shape) != shp:
                    raise NotImplementedError(f"fused path: {k} must have shape {shp}")

    def workspace(self, n_rays, slot=0):
        """(buffer, control-block view) of frame slot `slot`: frames in flight on different streams use different slots."""
        key = (n_rays, slot)
        if key not in self._ws:
            nbytes = lib().gf_frame_workspace_bytes(n_rays)
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            off = lib().gf_frame_ctrl_offset(n_rays)
            ctrl = buf[off:off + 4 * lib().gf_frame_ctrl_words()].view(torch.int32)
            # a viewer with dynamic resolution asks for many ray counts: keep the most recent few (frames in flight use <= 4 slots of
            # ONE size, each allocated and used on its own stream, so a dropped buffer is only recycled behind its last launch)
            while len(self._ws) >= 12:
                self._ws.pop(next(iter(self._ws)))
            self._ws[key] = (buf, ctrl)
        else:
            self._ws[key] = self._ws.pop(key)      # most recently used last
        return self._ws[key]


def _watched_tensors(model):
    """Every tensor a FusedState copies from or points into: all parameters (MLP weights are repacked, the condition encoder's and the
    tables' device pointers are handed to the kernels) and the occupancy buffers (the march box is derived from the bitfield)."""
    t = list(model.parameters())
    for name in ("density_bitfield", "aabb_infer", "density_grid_torso"):
        if getattr(model, name, None) is not None:
            t.append(getattr(model, name))
    return t


_untracked = itertools.count(-1, -1)


def _ver(t) -> int:
    """`t._version` -- or, for a tensor made under torch.inference_mode() (no version counter), a number no earlier call returned: whatever
    is keyed on it is then rebuilt every time instead of being trusted across a change nobody can see."""
    try:
        return t._version
    except RuntimeError:
        return next(_untracked)


def _stamp(tensors):
    # in-place updates (optimizer steps, load_state_dict, broadcast, update_extra_state) bump _version; .half() / .to() / `p.data = ...`
    # change data_ptr.  ~10 us for the ~45 tensors of a head+torso model.
    return tuple((_ver(t), t.data_ptr()) for t in tensors)


def get_state(model) -> FusedState:
    """The model's FusedState, brought up to date whenever a tensor it was packed from has changed since (training steps, train ->
    validate loops, load_state_dict, broadcast_model_, dtype / device moves): the packed copies can never go stale silently.  Values
    changed in place are re-gathered on the device (refresh_weights); reallocated tensors or a new device rebuild everything.
    (Re)packing is fp32 arithmetic whatever the caller's autocast state: the folded bias rows are torch.mv products.)"""
    if torch.is_autocast_enabled():
        with torch.autocast("cuda", enabled=False):
            return get_state(model)
    st = getattr(model, "_fused_state", None)
    if st is None or st.device != model.density_bitfield.device:
        st = FusedState(model)
        object.__setattr__(model, "_fused_state", st)
        return st
    stamp = _stamp(st._watch)
    if stamp != st.stamp:
        if any(a[1] != b[1] for a, b in zip(stamp, st.stamp)):          # a data_ptr moved
            st = FusedState(model)
            object.__setattr__(model, "_fused_state", st)
            return st
        n_param = len(list(model.parameters()))
        if stamp[n_param:] != st.stamp[n_param:]:                       # an occupancy buffer changed in place
            st.refresh_occupancy(model)
        if stamp[:n_param] != st.stamp[:n_param]:
            st.refresh_weights(model)
        st.stamp = stamp
    return st


def invalidate(model):
    """Drop the packed copies now (they are rebuilt on the next render).  Not required for correctness -- get_state notices changed
    tensors by itself -- but frees the workspaces and packs early."""
    if hasattr(model, "_fused_state"):
        object.__delattr__(model, "_fused_state")


def _bg_tensor(bg_color, N, device):
    if bg_color is None:
        bg_color = 1
    if not torch.is_tensor(bg_color):
        return torch.full((N, 3), float(bg_color), dtype=torch.float32, device=device)
    bg = bg_color.to(device=device, dtype=torch.float32)
    if bg.numel() == 3:
        bg = bg.reshape(1, 3).expand(N, 3)
    return bg.reshape(N, 3).contiguous()


def _fill_common(f: GfFrame, model, st: FusedState, N, dt_gamma, max_steps, T_thresh, amb_bias, bg, out_rgb, out_depth, out_rgb8=None,
                 slot=0):
    f.n_rays = N
    f.aabb, f.bitfield = ptr(model.aabb_infer, torch.float32), ptr(model.density_bitfield, torch.uint8)
    f.min_near, f.bound, f.dt_gamma, f.T_thresh = float(model.min_near), float(model.bound), float(dt_gamma), float(T_thresh)
    f.max_steps, f.cascade, f.grid_size = int(max_steps), int(model.cascade), int(model.grid_size)
    f.has_occ_aabb = 1
    for k in range(6):
        f.occ_aabb[k] = st.occ_aabb[k]
    pe, ae = model.position_embedder, model.ambient_embedder
    f.pos_table, f.pos_offsets = ptr(pe.embeddings, torch.float32), ptr(pe.offsets, torch.int32)
    f.amb_table, f.amb_offsets = ptr(ae.embeddings, torch.float32), ptr(ae.offsets, torch.int32)
    f.pos_S, f.amb_S = st.pos_S, st.amb_S
    f.base_res, f.gridtype, f.interp = st.base_res, st.gridtype, st.interp
    f.head_pack, f.amb_bias = ptr(st.head_pack), ptr(amb_bias, torch.float32)
    precision = getattr(model, "render_precision", "fp32")
    if precision not in ("fp32", "fast", "split"):
        raise ValueError(f"render_precision must be 'fp32', 'fast' or 'split', got {precision!r}")
    f.precision = {"fp32": 0, "fast": 1, "split": 2}[precision]
    f.head_pack16 = st.pack16(model).data_ptr() if precision == "fast" else None
    f.head_pack_split = st.pack_split(model).data_ptr() if precision == "split" else None
    f.bg_color = ptr(bg, torch.float32)
    f.out_rgb, f.out_depth = ptr(out_rgb), ptr(out_depth)
    f.out_rgb8 = ptr(out_rgb8, torch.uint8) if out_rgb8 is not None else None
    f.workspace = st.workspace(N, slot)[0].data_ptr()


def torso_mask_list(model, st: FusedState, bg_coords):
    """(list, dense_of, count) device tensors of gf_torso_mask_list for these pixel coordinates: which pixels the torso mask selects
    (radnerf_torso.py:166-172), as the dense list the torso field kernel runs over.  A property of (bg_coords, density_grid_torso, threshold):
    built once and kept while none of them changes -- a frame loop's bg_coords are one tensor for the whole shard, so its frames skip the
    per-frame mask launch (9 us of a 512 x 512 frame: 1 024 same-address atomics).  A caller that passes fresh coordinates every frame (the
    module API) gets a fresh list every frame, as before."""
    grid = model.density_grid_torso
    thresh = float(min(model.density_thresh_torso, model.mean_density_torso))
    key = (_ver(grid), grid.data_ptr(), thresh, _ver(bg_coords), bg_coords.data_ptr(), tuple(bg_coords.shape))
    hit = getattr(st, "_mask_list", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    N, dev = bg_coords.numel() // 2, bg_coords.device
    lst = torch.empty(N, dtype=torch.int32, device=dev)
    dense_of = torch.empty(N, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().gf_torso_mask_list(ptr(bg_coords, torch.float32), ptr(grid, torch.float32), N, int(model.grid_size), thresh, lst.data_ptr(),
                                   dense_of.data_ptr(), count.data_ptr(), current_stream(dev)))
    ev = torch.cuda.Event()
    ev.record()
    # the entry keeps `bg_coords` and the grid alive: an address in the key cannot be handed to another tensor while the entry exists
    st._mask_list = (key, (lst, dense_of, count, ev, {torch.cuda.current_stream(dev).cuda_stream}), (bg_coords, grid))
    return st._mask_list[1]


def _fill_torso(f: GfFrame, model, st: FusedState, bg_coords, torso_bias, out_alpha, out_trgb, out_deform, ha_branch=False, slot=0, cache_mask=False):
    te = model.torso_embedder
    if cache_mask:      # frame loop: the mask's dense list is built once per (coordinates, occupancy, threshold); other streams wait for it once
        lst, dense_of, count, ev, waited = torso_mask_list(model, st, bg_coords)
        cur = torch.cuda.current_stream(bg_coords.device)
        if cur.cuda_stream not in waited:
            cur.wait_event(ev)
            waited.add(cur.cuda_stream)
        f.torso_mask_list, f.torso_mask_dense_of, f.torso_mask_count = lst.data_ptr(), dense_of.data_ptr(), count.data_ptr()
    else:
        f.torso_mask_list = f.torso_mask_dense_of = f.torso_mask_count = None
    if st.head_aware and ha_branch:     # the torso sees the rendered head (radnerf_torso.py:175-177): encoder outputs per pixel, 8 extra MFMA steps
        N = int(f.n_rays)
        if not hasattr(st, "_ha_ws"):
            st._ha_ws = {}
        key = (N, slot)
        if key not in st._ha_ws:
            if len(st._ha_ws) >= 8:
                st._ha_ws.pop(next(iter(st._ha_ws)))
            st._ha_ws[key] = torch.empty(N, 16, dtype=torch.float32, device=st.device)
        f.torso_ha_pack, f.torso_ha_ws, f.torso_ha_branch = ptr(st.torso_ha_pack), st._ha_ws[key].data_ptr(), 1
    else:
        f.torso_ha_pack, f.torso_ha_ws, f.torso_ha_branch = None, None, 0
    f.torso_pack, f.torso_bias = ptr(st.torso_pack), ptr(torso_bias, torch.float32)
    f.torso_table, f.torso_offsets = ptr(te.embeddings, torch.float32), ptr(te.offsets, torch.int32)
    f.torso_occ, f.bg_coords = ptr(model.density_grid_torso, torch.float32), ptr(bg_coords, torch.float32)
    f.torso_S = st.torso_S
    f.torso_thresh = float(min(model.density_thresh_torso, model.mean_density_torso))
    f.torso_shrink = float(model.torso_shrink)
    f.out_torso_alpha = ptr(out_alpha) if out_alpha is not None else None
    f.out_torso_rgb = ptr(out_trgb) if out_trgb is not None else None
    f.out_deform = ptr(out_deform) if out_deform is not None else None


def head_aware_coin(model, bg_coords=None) -> bool:
    """radnerf_torso.py:175-179: with torso_head_aware the reference flips a coin PER FRAME, at inference too -- heads: the torso field sees
    the head's colour and opacity at each pixel; tails: zeros.  Same draw (random.random() < 0.5) so a seeded run decides alike.  The
    reference draws inside `if mask.any():` (:174), i.e. only when the torso mask selects at least one pixel, with
    mask = grid_sample(density_grid_torso, bg_coords) > min(density_thresh_torso, mean_density_torso) (:166-172).  A model whose mask is empty
    draws nothing, which keeps a seeded run's random stream aligned with the reference's.  The mask is evaluated by the model's own
    `torso_mask` (the same grid_sample) once per (occupancy grid, bg_coords, thresholds) and its `any()` cached: a lone cell barely above
    the threshold can leave every bilinear sample below it, so the grid alone does not decide."""
    if not bool(getattr(model, "torso_head_aware", False)):
        return False
    grid = getattr(model, "density_grid_torso", None)
    if grid is None:
        return False
    thresh = float(min(model.density_thresh_torso, model.mean_density_torso))
    key = (_ver(grid), grid.data_ptr(), thresh) + ((_ver(bg_coords), bg_coords.data_ptr(), tuple(bg_coords.shape)) if bg_coords is not None else ())
    st = getattr(model, "_torso_occ_any", None)
    if st is None or st[0] != key:
        with torch.no_grad():
            if bg_coords is not None:
                any_px = bool(model.torso_mask(bg_coords.reshape(-1, 2).to(grid.device)).any())
            else:       # no pixel coordinates at hand: nothing is selected by a grid that is nowhere above the threshold
                any_px = bool((grid > thresh).any())
        # the entry holds the tensors its key was taken from (as torso_mask_list does): while it lives the allocator cannot hand their address
        # to different coordinates whose fresh `_version` 0 would then match a stale key
        st = (key, any_px, grid, bg_coords)
        object.__setattr__(model, "_torso_occ_any", st)
    return st[1] and random.random() < 0.5


def cond_encode_batch_accepts(st, cond_wins) -> bool:
    """The ONE predicate for "gf_cond_encode_batch takes these windows": the encoder is the AudioNet + AudioAttNet pair the kernel implements
    and cond_wins is [n, S, T, C] fp32 contiguous (rows of a contiguous block are contiguous).  FramePipeline.prepare asks it before it draws
    a head-aware pass's coins, cond_encode_batch before it launches -- so a refusal can never follow a draw."""
    c = st.cond
    return c is not None and cond_wins.dim() == 4 and tuple(cond_wins.shape[1:]) == (c.S, c.T, c.C) and cond_wins.dtype == torch.float32 \
        and cond_wins.is_contiguous()


def cond_encode_batch(model, st, cond_wins, poses6=None, ha_branch=False):
    """gf_cond_encode_batch: the condition encoder + both bias folds for n frames in ONE launch on the current stream.
    cond_wins [n, S, T, C] fp32 contiguous, poses6 [n, 6] or None -> (cond_feat [n, A], amb_bias [n, 128], torso_bias [n, 96] or None);
    row k is bit-identical to the single-frame launch on frame k.  None when the encoder is not the AudioNet + AudioAttNet pair the kernel
    implements (or the window is outside its limits): the caller then uses the torch modules."""
    c = st.cond
    if not cond_encode_batch_accepts(st, cond_wins):
        return None
    n, dev = cond_wins.shape[0], cond_wins.device
    cond_feat = torch.empty(n, c.dim_aud, dtype=torch.float32, device=dev)
    amb_bias = torch.empty(n, 128, dtype=torch.float32, device=dev)
    torso_bias = torch.empty(n, 96, dtype=torch.float32, device=dev) if poses6 is not None else None
    call = GfCond.from_buffer_copy(c)
    call.cond, call.cond_feat, call.amb_bias = ptr(cond_wins), ptr(cond_feat), ptr(amb_bias)
    if poses6 is not None:
        p6 = poses6.reshape(n, 6).float().contiguous()
        call.pose6, call.torso_bias = ptr(p6), ptr(torso_bias)
        if st.head_aware:     # which constant rides behind the identity code: enc(black, transparent head), or zeros (per-pixel encoding elsewhere)
            call.torso_code = ptr(st._torso_code_ha if ha_branch else st._torso_code)
    else:
        call.torso_bias = None
    check(lib().gf_cond_encode_batch(C.byref(call), n, current_stream(dev)))
    return cond_feat, amb_bias, torso_bias


def _per_frame_vectors(model, st, cond, poses6=None, ha_branch=False):
    """cond encoder + the per-frame bias folds: one HIP launch (gf_cond_encode) on the current stream; the torch modules only
    when the encoder is not the AudioNet + AudioAttNet pair (or the window shape is outside the kernel's limits)."""
    if cond.dim() == 3:
        r = cond_encode_batch(model, st, cond[None], poses6, ha_branch)
        if r is not None:
            return r[0][0], r[1][0], (r[2][0] if r[2] is not None else None)
    cond_feat = model.cal_cond_feat(cond).reshape(-1).float()
    with torch.autocast("cuda", enabled=False):      # (a caller under autocast must still get fp32 bias vectors)
        amb_bias = torch.mv(st.W_cond, cond_feat)
    torso_bias = None
    if poses6 is not None:
        v = [model.torso_pose_embedder(poses6.reshape(1, 6).float()).reshape(-1)]
        if model.torso_individual_embedding_dim > 0:
            v.append(model.torso_individual_codes[0].detach())
        if st.head_aware:
            e0 = model.head_color_weights_encoder(torch.zeros(1, 4, device=st.device)).reshape(-1).float()
            v.append(torch.zeros_like(e0) if ha_branch else e0)
        with torch.autocast("cuda", enabled=False):
            torso_bias = torch.mv(st.W_tconst, torch.cat(v).float())
    return cond_feat, amb_bias, torso_bias


def _perturb_noise(perturb, perturb_noise, N, dev):
    """perturb=True at inference (renderer.py:338-342): U[0,1) per ray for the first march iteration -- the caller's draws, or fresh ones."""
    if not perturb:
        return None
    if perturb_noise is None:
        return torch.rand(N, dtype=torch.float32, device=dev)
    noise = perturb_noise.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    if noise.numel() != N:
        raise ValueError(f"perturb_noise: {noise.numel()} draws for {N} rays")
    return noise


def render_head_fused(model, rays_o, rays_d, cond, bg_coords, poses, dt_gamma, bg_color, perturb, max_steps, T_thresh, perturb_noise=None):
    """NeRFRenderer.render (renderer.py:263-367, inference) on the fused path."""
    with torch.no_grad():
        st = get_state(model)
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        _, amb_bias, _ = _per_frame_vectors(model, st, cond)
        bg = _bg_tensor(bg_color, N, dev)
        out_rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_depth = torch.empty(N, dtype=torch.float32, device=dev)
        f = GfFrame()
        _fill_common(f, model, st, N, dt_gamma, max_steps, T_thresh, amb_bias, bg, out_rgb, out_depth)
        f.rays_o, f.rays_d = ptr(rays_o), ptr(rays_d)
        noise = _perturb_noise(perturb, perturb_noise, N, dev)
        f.perturb_noise = ptr(noise) if noise is not None else None
        check(lib().gf_render_head(C.byref(f), current_stream(dev)))
        model.last_ctrl = st.workspace(N)[1]
        return {"depth_map": out_depth.view(*prefix), "rgb_map": out_rgb.view(*prefix, 3)}


def render_torso_fused(model, rays_o, rays_d, cond, bg_coords, poses, dt_gamma, bg_color, perturb, max_steps, T_thresh,
                       return_deform=True, perturb_noise=None):
    """RADNeRFTorso.render (radnerf_torso.py:86-198, inference) on the fused path."""
    with torch.no_grad():
        st = get_state(model)
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        bg_coords = bg_coords.contiguous().view(-1, 2).float()
        N, dev = rays_o.shape[0], rays_o.device
        ha_branch = head_aware_coin(model, bg_coords)
        _, amb_bias, torso_bias = _per_frame_vectors(model, st, cond, poses, ha_branch)
        bg = _bg_tensor(bg_color, N, dev)
        out_rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_depth = torch.empty(N, dtype=torch.float32, device=dev)
        out_alpha = torch.empty(N, 1, dtype=torch.float32, device=dev)
        out_trgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_deform = torch.zeros(N, 2, dtype=torch.float32, device=dev) if return_deform else None
        f = GfFrame()
        _fill_common(f, model, st, N, dt_gamma, max_steps, T_thresh, amb_bias, bg, out_rgb, out_depth)
        f.rays_o, f.rays_d = ptr(rays_o), ptr(rays_d)
        noise = _perturb_noise(perturb, perturb_noise, N, dev)
        f.perturb_noise = ptr(noise) if noise is not None else None
        _fill_torso(f, model, st, bg_coords, torso_bias, out_alpha, out_trgb, out_deform, ha_branch)
        s = current_stream(dev)
        check(lib().gf_render_head(C.byref(f), s))
        check(lib().gf_render_torso(C.byref(f), s))
        model.last_ctrl = st.workspace(N)[1]
        results = {"torso_alpha_map": out_alpha, "torso_rgb_map": out_trgb.view(*prefix, 3) if len(prefix) > 1 else out_trgb,
                   "rgb_map": out_rgb.view(*prefix, 3), "depth_map": out_depth.view(*prefix)}
        if return_deform:
            mask = out_alpha.view(-1) > 0  # sigmoid() > 0 on every masked pixel, exactly 0 elsewhere
            if bool(mask.any()):
                results["deform"] = out_deform[mask]
        return results


def field_forward(model, position, direction, cond_feat, individual_code):
    """RADNeRF.forward (radnerf.py:73-105) for a dense point list in one launch (gf_field_forward): sigma [M], color [M,3], ambient [M,2].
    Inference arithmetic, no autograd graph: callers are in eval / no_grad context (viewer, frozen head of torso training, op-by-op
    render loop).  fp32 whatever the autocast state of the caller (the torso task under the Trainer's autocast renders its frozen head through here:
    torch.mv would hand back half bias vectors)."""
    with torch.no_grad(), torch.autocast("cuda", enabled=False):
        st = get_state(model)
        dev = position.device
        x = position.detach().reshape(-1, 3).float().contiguous()
        d = direction.detach().reshape(-1, 3).float().contiguous()
        M = x.shape[0]
        sigma = torch.empty(M, dtype=torch.float32, device=dev)
        rgb = torch.empty(M, 3, dtype=torch.float32, device=dev)
        amb = torch.empty(M, 2, dtype=torch.float32, device=dev)
        if M == 0:
            return sigma, rgb, amb
        amb_bias = torch.mv(st.W_cond, cond_feat.detach().reshape(-1).float())
        col_bias = None
        if st.W_ind is not None:
            code = individual_code.detach().reshape(-1).float() if individual_code is not None else torch.zeros(st.W_ind.shape[1], device=dev)
            col_bias = torch.mv(st.W_ind, code)
        f = GfFrame()
        pe, ae = model.position_embedder, model.ambient_embedder
        f.bound = float(model.bound)
        f.pos_table, f.pos_offsets = ptr(pe.embeddings, torch.float32), ptr(pe.offsets, torch.int32)
        f.amb_table, f.amb_offsets = ptr(ae.embeddings, torch.float32), ptr(ae.offsets, torch.int32)
        f.pos_S, f.amb_S, f.base_res, f.gridtype, f.interp = st.pos_S, st.amb_S, st.base_res, st.gridtype, st.interp
        f.head_pack, f.amb_bias = ptr(st.head_pack), ptr(amb_bias, torch.float32)
        check(lib().gf_field_forward(C.byref(f), ptr(x, torch.float32), ptr(d, torch.float32), M, ptr(col_bias, torch.float32, allow_none=True),
                                     ptr(sigma), ptr(rgb), ptr(amb), current_stream(dev)))
        return sigma, rgb, amb


def pinhole_rays(pose, intrinsics, H: int, W: int, device=None):
    """get_rays(poses, intrinsics, H, W, -1) of utils.py:282-363 for ONE pose in one launch (gf_pinhole_rays): rays_o, rays_d [1, H*W, 3].
    Bit for bit the rays a pose-mode frame generates inside k_frame_init (one device function serves both), so the module API fed with
    these tensors and the frame loop render the same frame; torch's get_rays can differ from them in the last ulp (its rotation is a
    BLAS matmul).  `pose` [4,4] / [1,4,4] / [3,4] cam2world, ngp axes."""
    pose = torch.as_tensor(pose)
    dev = torch.device(device) if device is not None else pose.device
    if dev.type != "cuda":
        raise RuntimeError("pinhole_rays needs a HIP device (there is no CPU path)")
    p = np.ascontiguousarray(pose.detach().float().cpu().numpy().reshape(-1, 4)[:3].reshape(12))
    k = np.ascontiguousarray(np.asarray([float(v) for v in intrinsics], dtype=np.float32))
    with torch.cuda.device(dev):
        rays_o = torch.empty(1, H * W, 3, dtype=torch.float32, device=dev)
        rays_d = torch.empty(1, H * W, 3, dtype=torch.float32, device=dev)
        check(lib().gf_pinhole_rays(_hp(p), _hp(k), int(H), int(W), ptr(rays_o), ptr(rays_d), current_stream(dev)))
    return rays_o, rays_d


def frame_stats(ctrl: torch.Tensor, N: int, max_steps: int) -> dict:
    """Decode the device control block of the last frame (forces a sync).  `schedule` is the reference's per-iteration
    (n_alive, n_step) list (renderer.py:330-351) replayed from the terminal-index histogram, `budget` the total number of
    samples a never-terminating ray receives; `budget_device` is the same number as the phase-1 kernel computed it."""
    c = ctrl.cpu().numpy().astype(np.int64)
    hist = c[64:64 + max_steps + 2]          # control-block layout: geneface_amd/csrc/frame.hpp (kCtrl*)
    sched, cum, dead, d = [], 0, 0, 0
    while cum < max_steps:
        while d < cum:
            d += 1
            dead += int(hist[d])
        n_alive = N - dead
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        sched.append((n_alive, n_step))
        cum += n_step
    return {"schedule": sched, "budget": cum, "budget_device": int(c[10]), "n_hit": int(c[1]), "n_survivors": int(c[2]),
            "samples": (int(c[32]), int(c[34])), "tiles": (int(c[33]), int(c[35])), "rounds": (int(c[36]), int(c[38])),
            "composited": (int(c[37]), int(c[39]))}


# --------------------------------------------------------------------------------------------- frame-loop step
class _PipeBuffers:
    def __init__(self, pipe):
        dev, N = pipe.device, pipe.H * pipe.W
        slots = max(2, getattr(pipe, "max_in_flight", getattr(pipe, "in_flight", 2)))
        self.rgb = [torch.empty(N, 3, dtype=torch.float32, device=dev) for _ in range(slots)]
        self.depth = [torch.empty(N, dtype=torch.float32, device=dev) for _ in range(slots)]
        self.rgb8 = [torch.empty(pipe.H, pipe.W, 3, dtype=torch.uint8, device=dev) for _ in range(slots)]
        self.bg = pipe.bg.reshape(N, 3).contiguous()
        self.bg_coords = pipe.bg_coords.reshape(N, 2).contiguous()
        self.poses_host = pipe.poses.cpu().numpy()


def _fill_pose_frame(pipe, i, f, rgb8, slot=0):
    """Describe frame i of the pipeline (frame slot `slot`): rays come from pose + intrinsics inside the kernel."""
    model, hp = pipe.model, pipe.hp
    st = get_state(model)
    bufs = getattr(pipe, "_fused_bufs", None)
    if bufs is None:
        bufs = pipe._fused_bufs = _PipeBuffers(pipe)
    N = pipe.H * pipe.W
    torso = st.has_torso
    pre = pipe.prepared(i) if hasattr(pipe, "prepared") else None
    coin = pipe.prepared_coin(i) if (pre is not None and st.head_aware) else None
    if st.head_aware and coin is None:
        pre = None               # no coin was drawn for this frame by a batch: draw it now and encode the frame by itself
    ha_branch = (coin if coin is not None else head_aware_coin(model, bufs.bg_coords)) if torso else False
    if pre is not None:          # the pass's batched launch (FramePipeline.prepare) already holds this frame's vectors
        amb_bias, torso_bias = pre
    else:
        _, amb_bias, torso_bias = _per_frame_vectors(model, st, pipe.cond_wins[i], pipe.pose6[i:i + 1] if torso else None, ha_branch)
    _fill_common(f, model, st, N, hp["dt_gamma"], hp["max_steps"], hp.get("T_thresh", 1e-4), amb_bias, bufs.bg, bufs.rgb[slot], bufs.depth[slot],
                 rgb8, slot)   # the reference forwards **hparams to render(): a T_thresh key overrides the 1e-4 default (renderer.py:263)
    f.img_h, f.img_w = pipe.H, pipe.W
    f.rays_o = f.rays_d = None
    p = bufs.poses_host[i]
    for r in range(3):
        for c in range(4):
            f.pose[r * 4 + c] = float(p[r, c])
    for k in range(4):
        f.intrinsics[k] = float(pipe.intrinsics[k])
    if torso:
        _fill_torso(f, model, st, bufs.bg_coords, torso_bias, None, None, None, ha_branch, slot, cache_mask=True)
    return st, bufs, (amb_bias, torso_bias)


def render_frame_fused(pipe, i, slot=0):
    """One frame of FramePipeline on the fused path -> device uint8 [H,W,3], enqueued on the CURRENT stream.  Frame slots
    0/1 own separate workspaces and output buffers, so two consecutive frames may be in flight on two streams (the next
    frame's kernels fill the CUs the previous frame's draining persistent grid leaves idle) and the async D2H copy of
    frame i may still be reading one uint8 buffer while frame i+1 is rendered into the other."""
    with torch.no_grad():
        f = GfFrame()
        st, bufs, keep = _fill_pose_frame(pipe, i, f, None, slot)
        rgb8 = bufs.rgb8[slot]
        f.out_rgb8 = rgb8.data_ptr()
        s = current_stream(pipe.device)
        check(lib().gf_render_head(C.byref(f), s))
        if st.has_torso:
            check(lib().gf_render_torso(C.byref(f), s))
        return rgb8


def profile_frames(pipe, first, n_frames, flop_per_sample, peak_tflops):
    """bench.py roofline leg: time every launch of the dominant kernel (k_head_phase: march + field + composite; two launches
    per frame) with HIP events on the stream it runs on, and divide the algorithmic FLOPs of the samples it evaluated by
    that time."""
    phase_ms = (C.c_float * 4)()
    n_ph = C.c_uint32(0)
    tot_ms, tot_samples, tot_comp, launches, per_frame = 0.0, 0, 0, 0, []
    init_ms, hit_rays = 0.0, 0
    N = pipe.H * pipe.W
    with torch.no_grad():
        for i in range(first, first + n_frames):
            f = GfFrame()
            st, bufs, keep = _fill_pose_frame(pipe, i, f, None)
            check(lib().gf_render_head_timed(C.byref(f), current_stream(pipe.device), phase_ms, C.byref(n_ph)))
            fs = frame_stats(st.workspace(N)[1], N, pipe.hp["max_steps"])
            ms = phase_ms[0] + phase_ms[1]
            samples = fs["samples"][0] + fs["samples"][1]
            tot_ms += ms
            tot_samples += samples
            tot_comp += fs["composited"][0] + fs["composited"][1]
            init_ms += phase_ms[2]
            hit_rays += fs["n_hit"]
            launches += 2
            per_frame.append({"phase_ms": [round(phase_ms[0], 4), round(phase_ms[1], 4)], "init_ms": round(phase_ms[2], 4), "samples": list(fs["samples"]),
                              "composited": list(fs["composited"]), "tiles": list(fs["tiles"]), "rounds": list(fs["rounds"]), "budget": fs["budget_device"],
                              "reference_schedule": fs["schedule"], "n_hit": fs["n_hit"], "n_survivors": fs["n_survivors"]})
    achieved = tot_samples * flop_per_sample / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0
    tiles = sum(sum(p["tiles"]) for p in per_frame)
    nf = max(n_frames, 1)
    return {"bound": "mfma", "achieved": achieved, "peak": peak_tflops, "unit": "TFLOP/s", "frac": achieved / peak_tflops,
            "traffic": None, "kernel": "k_head_phase (march + field + composite; 2 launches per frame)",
            "launches": launches, "avg_launch_ms": tot_ms / max(launches, 1), "kernel_ms_per_frame": tot_ms / nf,
            "samples_per_frame": tot_samples / nf, "samples_composited_per_frame": tot_comp / nf, "tile_fill": tot_samples / max(32 * tiles, 1),
            "flop_per_sample": flop_per_sample, "frames_profiled": n_frames, "example_frame": per_frame[0] if per_frame else None,
            "marcher": {"kernel": "k_frame_init (ray generation, slab tests, the march through empty space to the first occupied sample; 1 launch per frame)",
                        "ms": init_ms / nf, "rays": N, "hit_rays": hit_rays / nf}}
