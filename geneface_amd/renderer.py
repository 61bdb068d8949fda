"""Occupancy-grid ray-march renderer (inference).

Drop-in for `NeRFRenderer` of modules/radnerfs/renderer.py:63-367 on the inference branch: same
constructor hparams, same buffers / parameters (state-dict compatible, :78-99), same
`render(...)` signature and result dict (:263-367).  Two execution strategies share every kernel's
arithmetic (geneface_amd/csrc):

  impl="ops"    the reference's own loop structure (march -> field -> composite -> compact, one
                host sync per iteration, renderer.py:316-351) over the stand-alone HIP operators;
  impl="fused"  (default once built) the whole frame enqueued without host synchronisation by
                geneface_amd/fused.py.

Training (`self.training == True`) and density-grid maintenance are outside this round's scope.
"""
import math

import torch
import torch.nn as nn

from . import raymarching


class NeRFRenderer(nn.Module):
    #: execution strategy of render(): "fused", "ops", or "auto" = fused whenever the call is inside what the fused
    #: kernels cover (no perturbation, max_steps <= 64, the GeneFace layer shapes), op-by-op otherwise.  Both run on
    #: the GPU through libgeneface_hip.so; neither has a CPU fallback.
    render_impl = "auto"

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

    def mark_untrained_grid(self, poses, intrinsic, S=64):
        raise NotImplementedError("density-grid maintenance is the next scope row (SURVEY.md 8f-1)")

    def update_extra_state(self, decay=0.95, S=128):
        raise NotImplementedError("density-grid maintenance is the next scope row (SURVEY.md 8f-1)")

    def _pick_impl(self, impl, perturb, max_steps):
        if impl != "auto":
            return impl
        if perturb or max_steps > 64:
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

    def _march_head_ops(self, rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma, perturb, max_steps, T_thresh):
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
                                                        perturb if step == 0 else False, dt_gamma, max_steps)
            sigmas, rgbs, _ = self(xyzs, dirs, cond_feat, ind_code)
            sigmas = self.density_scale * sigmas
            raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh)
            rays_alive = rays_alive[rays_alive >= 0]
            step += n_step
        return weights_sum, depth, image

    def render(self, rays_o, rays_d, cond, bg_coords, poses, index=0, dt_gamma=0, bg_color=None, perturb=False,
               force_all_rays=False, max_steps=1024, T_thresh=1e-4, **kwargs):
        if self.training:
            raise NotImplementedError("NeRFRenderer.render: the training branch is outside this round's scope (SURVEY.md 8f-2)")
        impl = self._pick_impl(kwargs.get("render_impl", self.render_impl), perturb, max_steps)
        if impl == "fused":
            from .fused import render_head_fused
            return render_head_fused(self, rays_o, rays_d, cond, bg_coords, poses, dt_gamma, bg_color, perturb, max_steps, T_thresh)
        with torch.no_grad():
            prefix = rays_o.shape[:-1]
            rays_o = rays_o.contiguous().view(-1, 3)
            rays_d = rays_d.contiguous().view(-1, 3)
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
            cond_feat = self.cal_cond_feat(cond)
            weights_sum, depth, image = self._march_head_ops(rays_o, rays_d, nears, fars, cond_feat, self._ind_code(),
                                                             dt_gamma, perturb, max_steps, T_thresh)
            if bg_color is None:
                bg_color = 1
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            image = image.view(*prefix, 3).clamp(0, 1)
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
            return {"depth_map": depth.view(*prefix), "rgb_map": image}
