"""Occupancy-grid ray-march renderer (inference).

Drop-in for `NeRFRenderer` of modules/radnerfs/renderer.py:63-367 on the inference branch: same
constructor hparams, same buffers / parameters (state-dict compatible, :78-99), same
`render(...)` signature and result dict (:263-367).  Two execution strategies share every kernel's
arithmetic (geneface_amd/csrc):

  impl="ops"    the reference's own loop structure (march -> field -> composite -> compact, one
                host sync per iteration, renderer.py:316-351) over the stand-alone HIP operators;
  impl="fused"  (default once built) the whole frame enqueued without host synchronisation by
                geneface_amd/fused.py.

`self.training == True` takes the reference's training branch (:296-313) over the training-tier ops (march_rays_train,
composite_rays_train, encoder backward passes); density-grid maintenance is `update_extra_state` / `mark_untrained_grid`.
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn

from . import raymarching


def _rand_like(x, generator=None):
    """U[0,1) noise shaped like x; with a generator the numbers are drawn on the generator's device (a CPU generator gives
    the same jitter to a CPU restatement and to the GPU run) and moved to x's."""
    if generator is None:
        return torch.rand_like(x)
    return torch.rand(x.shape, generator=generator, device=generator.device, dtype=x.dtype).to(x.device)


def _mark_written(*tensors):
    """A kernel of the library wrote these tensors through raw pointers: bump their version counters, as an in-place torch op would, so
    that anything keyed on `_version` (geneface_amd.fused.get_state's staleness check of the packed copies and of the occupancy box) sees it."""
    for t in tensors:
        try:
            torch.autograd.graph.increment_version(t)
        except AttributeError:      # older torch: an in-place no-op does the same
            t.add_(0)


class NeRFRenderer(nn.Module):
    #: execution strategy of render(): "fused", "ops", or "auto" = fused whenever the call is inside what the fused
    #: kernels cover (max_steps <= 1024 -- the reference's default and its viewer's maximum --, the GeneFace layer shapes), op-by-op otherwise.  Both run on
    #: the GPU through libgeneface_hip.so; neither has a CPU fallback.
    render_impl = "auto"
    #: arithmetic of the fused head field: "fp32" (strict parity: everything in fp32, the offline inference path of the reference) or
    #: "fast" (f16 MFMA operands and activations, fp32 accumulation -- what the reference computes under autocast / model.half(), its
    #: training and viewer paths; BASELINE.md section 4 "fast": PSNR >= 40 dB, <= 1 LSB on >= 99.9 % of uint8 pixels) or
    #: "split" (fp32 values carried as two-term f16 splits on the f16 matrix pipe, fp32 accumulation: fp32-level accuracy -- the strict
    #: tolerance max|d rgb| <= 1e-4 holds -- at about twice the frame rate; not fp32 bit patterns, hence opt-in).
    render_precision = "fp32"

    def __init__(self, hparams):
        super().__init__()
        self.bound = hparams["bound"]
        self.cascade = 1 + math.ceil(math.log2(hparams["bound"]))
        self.grid_size = hparams["grid_size"]
        self.density_scale = 1
        self.min_near = hparams["min_near"]
        self.density_thresh = hparams["density_thresh"]
        self.cuda_ray = hparams["cuda_ray"]

        b = float(self.bound)
        aabb = torch.tensor([-b, -b / 2, -b, b, b / 2, b], dtype=torch.float32)  # flat in y (the depth axis)
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())

        self.individual_embedding_num = hparams["individual_embedding_num"]
        self.individual_embedding_dim = hparams["individual_embedding_dim"]
        if self.individual_embedding_dim > 0:
            self.individual_embeddings = nn.Parameter(torch.randn(self.individual_embedding_num, self.individual_embedding_dim) * 0.1)

        self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
        self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
        self.mean_density = 0
        self.iter_density = 0
        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
        self.mean_count = 0
        self.local_step = 0

    def __getstate__(self):
        """copy.deepcopy / pickle / torch.save of a model that has already rendered: the packed device copies and caches (fused.FusedState:
        ctypes structures, workspaces, streams; the torso-mask cache) belong to THIS object on ITS device -- they are left behind and the copy
        builds its own on first use (fused.get_state)."""
        state = self.__dict__.copy()
        for k in ("_fused_state", "_torso_occ_any"):
            state.pop(k, None)
        return state

    def _apply(self, fn, *args, **kwargs):
        """`.half()` / `.to(torch.float16)` on the model or on a module that holds it (inference/nerfs/radnerf_gui.py:604-605: `nerf_task.half()`
        when `amp` is set): the reference then keeps half parameters and computes in half.  Here a half model IS the f16 tier
        (`render_precision = "fast"`: f16 MFMA operands packed from the masters, fp32 accumulation) and the masters stay fp32 -- the C ABI takes
        fp32 tables and weights, and `state_dict()` keeps the precision the checkpoint had.  `.float()` afterwards restores the tier the model
        had.  Device moves inside the same call are applied; other dtype conversions have no tier and are refused."""
        here = next((b.device for b in self.buffers()), torch.device("cpu"))
        probe = fn(torch.zeros(1, dtype=torch.float32, device=here))            # what `fn` does to an fp32 tensor that lives where the model lives
        if torch.is_tensor(probe) and probe.dtype == torch.float16:
            if self.render_precision != "fast":
                self._precision_before_half = self.render_precision
                self.render_precision = "fast"
            return super()._apply(lambda t: t.to(probe.device), *args, **kwargs)    # the device move of the same call, if there is one
        if torch.is_tensor(probe) and probe.dtype != torch.float32:
            raise NotImplementedError(f"NeRFRenderer has no {probe.dtype} tier: render_precision is one of 'fp32', 'split', 'fast' (= .half())")
        if getattr(self, "_precision_before_half", None) is not None and fn(torch.zeros(1, dtype=torch.float16)).dtype == torch.float32:
            self.render_precision, self._precision_before_half = self._precision_before_half, None       # .float() after .half()
        return super()._apply(fn, *args, **kwargs)

    # --- the field interface subclasses provide (renderer.py:103-114) ---
    def cal_cond_feat(self, cond):
        raise NotImplementedError()

    def forward(self, x, d, cond_feat, individual_code):
        raise NotImplementedError()

    def density(self, x, cond_feat):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """renderer.py:129-196: cells no training camera sees get density -1 (never marched, never updated).
        poses [B,4,4] c2w (ngp axes), intrinsic (fx, fy, cx, cy).  One launch (gf_mark_untrained_grid: one lane per cascade cell walks
        the cameras and stops at the first frustum that contains the cell) instead of the reference's S^3-block x cascade x camera-batch
        Python loops; `S` only sized those blocks and is accepted for signature parity."""
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        fx, fy, cx, cy = (float(v) for v in intrinsic)
        dev = self.density_bitfield.device
        p44 = poses.to(dev).float().contiguous()
        from .lib import check, current_stream, lib, ptr
        check(lib().gf_mark_untrained_grid(ptr(p44, torch.float32), int(p44.shape[0]), fx, fy, cx, cy, int(self.cascade), int(self.grid_size),
                                           float(self.bound), ptr(self.density_grid, torch.float32), current_stream(dev)))
        _mark_written(self.density_grid)

    def _cell_jitter(self, S, generator):
        """U[0,1) jitter for every cascade cell, [cascade, G^3, 3] in meshgrid order.  With a generator the numbers are drawn block by
        block and cascade by cascade exactly as the reference's loop consumes them (renderer.py:222-243), so a CPU restatement seeded the
        same way sees the same jitter; without one a single device draw serves (the stream is unobservable)."""
        G, C, dev = self.grid_size, self.cascade, self.density_bitfield.device
        if generator is None:
            return torch.rand(C, G ** 3, 3, device=dev)
        noise = torch.empty(C, G, G, G, 3, dtype=torch.float32)
        X = torch.arange(G).split(S)
        for xs in X:
            for ys in X:
                for zs in X:
                    n = len(xs) * len(ys) * len(zs)
                    for cas in range(C):
                        blk = torch.rand((n, 3), generator=generator, device=generator.device, dtype=torch.float32).cpu()
                        noise[cas, xs[0]:xs[-1] + 1, ys[0]:ys[-1] + 1, zs[0]:zs[-1] + 1] = blk.view(len(xs), len(ys), len(zs), 3)
        return noise.view(C, G ** 3, 3).to(dev)

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, cond=None, generator=None):
        """renderer.py:199-260: re-sample the density field on the jittered cell centres of every cascade, dilate in Morton
        space, EMA-max into density_grid, re-derive mean_density and the packed density_bitfield.  The reference draws the
        condition window at random from `self.conds` (set by the task); pass `cond` ([smo_win, cond_win, C]) to fix it.
        `generator` seeds the cell jitter.
        Three launches on the device (geneface_hip.h: gf_grid_density = every cell of every cascade through the density head on the
        matrix pipe, gf_grid_update = dilation + EMA-max + mean, then bit packing) and one host read (mean_density is a Python float in
        the reference's API); models the fused field does not cover take the op-by-op route."""
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        if cond is None:
            if not hasattr(self, "conds"):
                raise RuntimeError("update_extra_state: give `cond` or set model.conds (the task does, tasks/radnerfs/radnerf.py:44-47)")
            from .utils import get_audio_features
            rand_idx = random.randint(0, self.conds.shape[0] - 1)
            cond = get_audio_features(self.conds, 2, rand_idx, self.smo_win_size)
        noise = self._cell_jitter(S, generator)
        if self._pick_impl("auto", False, 1) == "fused":
            tmp_grid = self._density_grid_fused(cond.to(dev), noise)
        else:
            tmp_grid = self._density_grid_ops(cond.to(dev), noise)
        from .lib import check, current_stream, lib, ptr
        L = lib()
        C, G = int(self.cascade), int(self.grid_size)
        ws = torch.empty(L.gf_grid_update_ws_bytes(C, G), dtype=torch.uint8, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        check(L.gf_grid_update(ptr(self.density_grid, torch.float32), ptr(tmp_grid, torch.float32), C, G, float(decay), float(self.density_thresh),
                               ptr(self.density_bitfield, torch.uint8), ws.data_ptr(), ptr(stats), current_stream(dev)))
        _mark_written(self.density_grid, self.density_bitfield)   # raw-pointer writes: tell autograd's version counters (fused.get_state watches them)
        self.mean_density = float(stats[0].item())
        self.iter_density += 1
        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0
        from .fused import invalidate
        invalidate(self)   # the fused path caches the occupancy bounding box of the bitfield

    def _density_grid_fused(self, cond, noise):
        """tmp_grid [C, G^3] (Morton order) = density * density_scale of every jittered cell, one launch."""
        import ctypes as C_
        from . import fused
        from .lib import check, current_stream, lib, ptr
        dev = self.density_bitfield.device
        st = fused.get_state(self)
        _, amb_bias, _ = fused._per_frame_vectors(self, st, cond.float().contiguous())
        f = fused.GfFrame()
        pe, ae = self.position_embedder, self.ambient_embedder
        f.bound, f.cascade, f.grid_size = float(self.bound), int(self.cascade), int(self.grid_size)
        f.pos_table, f.pos_offsets = ptr(pe.embeddings, torch.float32), ptr(pe.offsets, torch.int32)
        f.amb_table, f.amb_offsets = ptr(ae.embeddings, torch.float32), ptr(ae.offsets, torch.int32)
        f.pos_S, f.amb_S, f.base_res, f.gridtype, f.interp = st.pos_S, st.amb_S, st.base_res, st.gridtype, st.interp
        f.head_pack, f.amb_bias = ptr(st.head_pack), ptr(amb_bias, torch.float32)
        tmp_grid = torch.empty_like(self.density_grid)
        check(lib().gf_grid_density(C_.byref(f), ptr(noise, torch.float32), float(self.density_scale), ptr(tmp_grid, torch.float32), current_stream(dev)))
        return tmp_grid

    def _density_grid_ops(self, cond, noise):
        """The same tmp_grid through the stand-alone ops (any architecture `density()` serves): one field query per cascade."""
        dev = self.density_bitfield.device
        G = self.grid_size
        enc_a = self.cal_cond_feat(cond)
        ax = torch.arange(G, dtype=torch.int32, device=dev)
        coords = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(-1, 3)
        indices = raymarching.morton3D(coords).long()
        xyzs = 2 * coords.float() / (G - 1) - 1
        tmp_grid = torch.zeros_like(self.density_grid)
        for cas in range(self.cascade):
            bound = min(2 ** cas, self.bound)
            half_grid_size = bound / G
            pts = xyzs * (bound - half_grid_size) + (noise[cas] * 2 - 1) * half_grid_size
            tmp_grid[cas, indices] = self.density(pts, enc_a)["sigma"].reshape(-1).float() * self.density_scale
        return tmp_grid

    def _pick_impl(self, impl, perturb, max_steps):
        if impl != "auto":
            return impl
        if max_steps > 1024:      # beyond the reference's own default and viewer maximum (renderer.py:263, radnerf_gui.py:466-471)
            return "ops"
        ok = getattr(self, "_fused_arch_ok", None)
        if ok is None:
            from .fused import FusedState
            try:
                FusedState.check_architecture(self)
                ok = True
            except NotImplementedError:
                ok = False
            object.__setattr__(self, "_fused_arch_ok", ok)
        return "fused" if ok else "ops"

    # --- shared pieces of render() ---
    def _ind_code(self):
        return self.individual_embeddings[0] if self.individual_embedding_dim > 0 else None

    def _march_head_ops(self, rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma, perturb, max_steps, T_thresh, perturb_noise=None):
        """renderer.py:316-351: wavefront loop with a host-visible alive list."""
        N, device = rays_o.shape[0], rays_o.device
        weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
        depth = torch.zeros(N, dtype=torch.float32, device=device)
        image = torch.zeros(N, 3, dtype=torch.float32, device=device)
        rays_alive = torch.arange(N, dtype=torch.int32, device=device)
        rays_t = nears.clone()
        step = 0
        self.last_schedule = []
        while step < max_steps:
            n_alive = rays_alive.shape[0]
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            self.last_schedule.append((n_alive, n_step))
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                        self.density_bitfield, self.cascade, self.grid_size, nears, fars, 128,
                                                        perturb if step == 0 else False, dt_gamma, max_steps,
                                                        noises=perturb_noise if step == 0 else None)
            sigmas, rgbs, _ = self(xyzs, dirs, cond_feat, ind_code)
            sigmas = self.density_scale * sigmas
            raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
            rays_alive = rays_alive[rays_alive >= 0]
            step += n_step
        return weights_sum, depth, image

    def _march_head_train(self, rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma, perturb, force_all_rays, max_steps):
        """renderer.py:296-313: one pass over all rays (count / prefix / write marcher), the field through torch autograd (grid, SH
        encoders and trunc_exp carry custom backward passes), differentiable compositing."""
        counter = self.step_counter[self.local_step % 16]
        counter.zero_()
        self.local_step += 1
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size,
                                                                nears, fars, counter, self.mean_count, perturb, 128, force_all_rays, dt_gamma,
                                                                max_steps)
        sigmas, rgbs, ambient = self(xyzs, dirs, cond_feat, ind_code)
        sigmas = self.density_scale * sigmas
        weights_sum, ambient_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, ambient.abs().sum(-1), deltas, rays)
        return weights_sum, ambient_sum, depth, image

    def _render_train(self, rays_o, rays_d, cond, index, dt_gamma, bg_color, perturb, force_all_rays, max_steps):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train, self.min_near)
        nears, fars = nears.detach(), fars.detach()
        cond_feat = self.cal_cond_feat(cond)
        ind_code = self.individual_embeddings[index] if self.individual_embedding_dim > 0 else None
        weights_sum, ambient_sum, depth, image = self._march_head_train(rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma, perturb,
                                                                        force_all_rays, max_steps)
        results = {"weights_sum": weights_sum, "ambient": ambient_sum}
        if bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        results["rgb_map"] = image.view(*prefix, 3).clamp(0, 1)
        results["depth_map"] = (torch.clamp(depth - nears, min=0) / (fars - nears)).view(*prefix)
        return results

    def render(self, rays_o, rays_d, cond, bg_coords, poses, index=0, dt_gamma=0, bg_color=None, perturb=False,
               force_all_rays=False, max_steps=1024, T_thresh=1e-4, **kwargs):
        if self.training:
            return self._render_train(rays_o, rays_d, cond, index, dt_gamma, bg_color, perturb, force_all_rays, max_steps)
        impl = self._pick_impl(kwargs.get("render_impl", self.render_impl), perturb, max_steps)
        if impl == "fused":
            from .fused import render_head_fused
            return render_head_fused(self, rays_o, rays_d, cond, bg_coords, poses, dt_gamma, bg_color, perturb, max_steps, T_thresh,
                                     perturb_noise=kwargs.get("perturb_noise"))
        with torch.no_grad():
            prefix = rays_o.shape[:-1]
            rays_o = rays_o.contiguous().view(-1, 3)
            rays_d = rays_d.contiguous().view(-1, 3)
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
            cond_feat = self.cal_cond_feat(cond)
            weights_sum, depth, image = self._march_head_ops(rays_o, rays_d, nears, fars, cond_feat, self._ind_code(),
                                                             dt_gamma, perturb, max_steps, T_thresh, kwargs.get("perturb_noise"))
            if bg_color is None:
                bg_color = 1
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            image = image.view(*prefix, 3).clamp(0, 1)
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
            return {"depth_map": depth.view(*prefix), "rgb_map": image}
