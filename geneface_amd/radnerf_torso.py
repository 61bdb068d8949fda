"""Head + 2-D deformable torso renderer.

Drop-in for `RADNeRFTorso` of modules/radnerfs/radnerf_torso.py:17-241 on the inference branch:
`forward_torso` :51-84 and `render` :86-198 keep their signatures and result keys
(`rgb_map, depth_map, torso_alpha_map, torso_rgb_map, deform`).  Unlike the reference, `torso_shrink`
and `torso_head_aware` are read from the hparams given to the constructor rather than from a
process-global dict.  `torso_head_aware=True` (:36-46, :68-74, :175-179: a small encoder of the head's colour / opacity at the
pixel feeds both torso MLPs, on a coin flip per frame) runs on the fused path too since round 5 (`k_torso_field<true>`: the encoder's
outputs are eight extra MFMA steps of both first layers; `_pick_impl` says "fused" for it and tests/test_gpu_sweep.py asserts that);
the op-by-op path stays as `render_impl="ops"`.
"""
import random

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import raymarching
from .cond_encoder import MLP
from .encoders import get_encoder
from .radnerf import RADNeRF
from .renderer import _rand_like


class RADNeRFTorso(RADNeRF):
    def __init__(self, hparams):
        super().__init__(hparams)
        self.register_buffer("density_grid_torso", torch.zeros([self.grid_size ** 2]))
        self.mean_density_torso = 0  # not in the state_dict: a freshly loaded model thresholds at 0
        self.density_thresh_torso = hparams["density_thresh_torso"]
        self.torso_shrink = hparams["torso_shrink"]
        self.torso_head_aware = bool(hparams.get("torso_head_aware", False))

        self.torso_individual_embedding_num = hparams["individual_embedding_num"]
        self.torso_individual_embedding_dim = hparams["torso_individual_embedding_dim"]
        if self.torso_individual_embedding_dim > 0:
            self.torso_individual_codes = nn.Parameter(
                torch.randn(self.torso_individual_embedding_num, self.torso_individual_embedding_dim) * 0.1)

        self.torso_pose_embedder, self.pose_embedding_dim = get_encoder("frequency", input_dim=6, multires=4)
        self.torso_deform_pos_embedder, self.torso_deform_pos_dim = get_encoder("frequency", input_dim=2, multires=10)
        self.torso_embedder, self.torso_in_dim = get_encoder("tiledgrid", input_dim=2, num_levels=16, level_dim=2,
                                                             base_resolution=16, log2_hashmap_size=16, desired_resolution=2048)
        deform_in = self.torso_deform_pos_dim + self.pose_embedding_dim + self.torso_individual_embedding_dim
        if self.torso_head_aware:   # radnerf_torso.py:36-46
            self.head_color_weights_encoder = nn.Sequential(nn.Linear(3 + 1, 16, bias=True), nn.LeakyReLU(0.02, True),
                                                            nn.Linear(16, 32, bias=True), nn.LeakyReLU(0.02, True),
                                                            nn.Linear(32, 16, bias=True))
            deform_in += 16
        self.torso_deform_net = MLP(deform_in, 2, 64, 3)
        self.torso_canonicial_net = MLP(self.torso_in_dim + deform_in, 4, 32, 3)

    #: training branch: True = the fused torso field on every sampled pixel, masked afterwards -- no compaction, no host sync per step (round 6);
    #: False = the reference's boolean-mask gather / scatter (one compaction and one sync per step)
    torso_train_dense = True
    #: with torso_train_dense: "fused" = mask, torso-over-background and head-over-torso blends and the clamp as one autograd node (one launch
    #: forward, one backward); "ops" = the torch expressions of radnerf_torso.py:181-192
    torso_blend_impl = "fused"

    def _torso_code(self):
        return self.torso_individual_codes[0] if self.torso_individual_embedding_dim > 0 else None

    def forward_torso(self, x, poses, c=None, image=None, weights_sum=None):
        """x [m,2] in [-1,1], poses [1,6], c [8] -> alpha [m,1], colour [m,3], dx [m,2]."""
        if self._fused_torso_train_ok(x, c, image):
            # training (round 6): the whole field as ONE autograd node -- fused forward with saves, one-launch input-gradient chain
            # (train_torso.py); ~180 launches of the op graph below become 2 + the six weight-gradient products
            from .train_torso import forward_torso_fused
            return forward_torso_fused(self, x, poses, c)
        m = x.shape[0]
        x = x * self.torso_shrink
        parts = [self.torso_deform_pos_embedder(x), self.torso_pose_embedder(poses).reshape(1, -1).expand(m, -1)]
        if c is not None:
            parts.append(c.reshape(1, -1).expand(m, -1))
        if self.torso_head_aware:   # radnerf_torso.py:68-74: no head given -> the encoding of a black, transparent head
            if image is None:
                image = torch.zeros([m, 3], dtype=parts[0].dtype, device=x.device)
                weights_sum = torch.zeros([m, 1], dtype=parts[0].dtype, device=x.device)
            parts.append(self.head_color_weights_encoder(torch.cat([image, weights_sum], dim=-1)))
        h = torch.cat(parts, dim=-1)
        dx = self.torso_deform_net(h)
        xc = (x + dx).clamp(-1, 1).float()
        h = self.torso_canonicial_net(torch.cat([self.torso_embedder(xc, bound=1), h], dim=-1))
        return torch.sigmoid(h[..., :1]), torch.sigmoid(h[..., 1:]), dx

    def _fused_torso_train_ok(self, x, c, image):
        """Under autograd, on the GPU, default architecture, and only when the model has not been pinned to the op graph (`field_impl` /
        `render_impl`, like RADNeRF.forward's fused head field)."""
        if not (torch.is_grad_enabled() and x.is_cuda and c is not None and image is None and self.field_impl == "auto"
                and self.render_impl in ("auto", "fused") and x.shape[0] > 0):
            return False
        if self._pick_impl("auto", False, 1) != "fused":
            return False
        from .train_torso import supported
        return supported(self)

    def _cond_feat_no_grad(self, cond):
        """cal_cond_feat for a FROZEN head (the torso task: tasks/radnerfs/radnerf_torso.py:40-42): one HIP launch (gf_cond_encode) instead of
        the ~30 launches of the conv / attention graph; the torch modules whenever the kernel does not cover the encoder or the window."""
        if cond.is_cuda and cond.dim() == 3 and self.field_impl == "auto" and self.render_impl in ("auto", "fused") \
                and self._pick_impl("auto", False, 1) == "fused":
            from .fused import cond_encode_batch, get_state
            r = cond_encode_batch(self, get_state(self), cond[None].float().contiguous())
            if r is not None:
                return r[0][0]
        return self.cal_cond_feat(cond)

    def torso_mask(self, bg_coords):
        thresh = min(self.density_thresh_torso, self.mean_density_torso)
        occ = F.grid_sample(self.density_grid_torso.view(1, 1, self.grid_size, self.grid_size), bg_coords.view(1, -1, 1, 2),
                            align_corners=True).view(-1)
        return occ > thresh

    def _render_train_torso(self, rays_o, rays_d, cond, bg_coords, poses, index, dt_gamma, bg_color, perturb, force_all_rays, max_steps):
        """radnerf_torso.py:86-198 with self.training: the (frozen) head is rendered under no_grad by the training marcher, only the
        torso field receives gradients."""
        with torch.no_grad():
            prefix = rays_o.shape[:-1]
            rays_o = rays_o.contiguous().view(-1, 3)
            rays_d = rays_d.contiguous().view(-1, 3)
            bg_coords = bg_coords.contiguous().view(-1, 2)
            N, device = rays_o.shape[0], rays_o.device
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train, self.min_near)
            cond_feat = self._cond_feat_no_grad(cond)
            ind_code = self.individual_embeddings[index] if self.individual_embedding_dim > 0 else None
            weights_sum, ambient_sum, depth, image = self._march_head_train(rays_o, rays_d, nears, fars, cond_feat, ind_code, dt_gamma, perturb,
                                                                            force_all_rays, max_steps)
            results = {"weights_sum": weights_sum, "ambient": ambient_sum}
            if bg_color is None:
                bg_color = 1
        code = self.torso_individual_codes[index] if self.torso_individual_embedding_dim > 0 else None
        mask = self.torso_mask(bg_coords)
        dense = self.torso_train_dense and code is not None and self._fused_torso_train_ok(bg_coords, code, None)
        if dense and self.torso_blend_impl == "fused":
            # the field on every sampled pixel (see below) and the whole tail -- mask, both blends, clamp -- as one autograd node over one launch
            # each way (train_torso.torso_blend_train: the torch expressions' operations in their order, each rounded on its own)
            from .train_torso import torso_blend_train
            a, c, deform = self.forward_torso(bg_coords, poses, code)
            m = mask.to(torch.float32)
            torso_alpha, torso_rgb, rgb = torso_blend_train(a, c, m, bg_color, image, weights_sum)
            results["deform"] = deform * m.unsqueeze(-1)
            results["torso_alpha_map"] = torso_alpha
            results["torso_rgb_map"] = torso_rgb
            results["rgb_map"] = rgb.view(*prefix, 3)
            results["depth_map"] = (torch.clamp(depth - nears, min=0) / (fars - nears)).view(*prefix)
            return results
        torso_alpha = torch.zeros([N, 1], device=device)
        torso_color = torch.zeros([N, 3], device=device)
        if dense:
            # Round 6: NO compaction and no host sync.  The fused field costs 38 us for 65 536 pixels, the boolean-mask statements of the
            # reference (radnerf_torso.py:174-184) cost a nonzero() -- a device-to-host sync in the middle of every step, so the CPU can never
            # run ahead of the GPU and the step ran at the launch rate (3.2 ms for 1.85 ms of kernels).  The field is evaluated on EVERY
            # sampled pixel and its outputs are multiplied by the mask: a pixel's field does not depend on its neighbours in the list (same
            # values for the masked ones), unmasked pixels contribute alpha = 0 and exact zero gradients.  `deform` comes back for all N
            # pixels here, zero where unmasked (the reference returns the masked rows; nothing in its tasks reads it).
            a, c, deform = self.forward_torso(bg_coords, poses, code)
            m = mask.unsqueeze(-1).to(torch.float32)
            torso_alpha, torso_color = a.float() * m, c.float() * m
            results["deform"] = deform * m
            sel = None
        else:
            # the masked pixels' indices ONCE (one compaction, one host sync): `t[mask]` / `t[mask] = v` each run their own nonzero() -- three
            # compactions and three syncs per step for the reference's three boolean-mask statements (radnerf_torso.py:174-184); same values
            sel = mask.nonzero(as_tuple=True)[0]
        if sel is not None and sel.numel() > 0:
            if self.torso_head_aware and random.random() < 0.5:
                a, c, deform = self.forward_torso(bg_coords[sel], poses, code, image[sel], weights_sum.unsqueeze(-1)[sel])
            else:
                a, c, deform = self.forward_torso(bg_coords[sel], poses, code)
            torso_alpha = torso_alpha.index_copy(0, sel, a.float())
            torso_color = torso_color.index_copy(0, sel, c.float())
            results["deform"] = deform
        bg_color = torso_color * torso_alpha + bg_color * (1 - torso_alpha)
        results["torso_alpha_map"] = torso_alpha
        results["torso_rgb_map"] = bg_color
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        results["rgb_map"] = image.view(*prefix, 3).clamp(0, 1)
        results["depth_map"] = (torch.clamp(depth - nears, min=0) / (fars - nears)).view(*prefix)
        return results

    def render(self, rays_o, rays_d, cond, bg_coords, poses, index=0, dt_gamma=0, bg_color=None, perturb=False,
               force_all_rays=False, max_steps=1024, T_thresh=1e-4, **kwargs):
        if self.training:
            return self._render_train_torso(rays_o, rays_d, cond, bg_coords, poses, index, dt_gamma, bg_color, perturb, force_all_rays, max_steps)
        impl = self._pick_impl(kwargs.get("render_impl", self.render_impl), perturb, max_steps)
        if impl == "fused":
            from .fused import render_torso_fused
            return render_torso_fused(self, rays_o, rays_d, cond, bg_coords, poses, dt_gamma, bg_color, perturb, max_steps, T_thresh,
                                      perturb_noise=kwargs.get("perturb_noise"))
        with torch.no_grad():
            prefix = rays_o.shape[:-1]
            rays_o = rays_o.contiguous().view(-1, 3)
            rays_d = rays_d.contiguous().view(-1, 3)
            bg_coords = bg_coords.contiguous().view(-1, 2)
            N, device = rays_o.shape[0], rays_o.device
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
            cond_feat = self.cal_cond_feat(cond)
            weights_sum, depth, image = self._march_head_ops(rays_o, rays_d, nears, fars, cond_feat, self._ind_code(),
                                                             dt_gamma, perturb, max_steps, T_thresh, kwargs.get("perturb_noise"))
            if bg_color is None:
                bg_color = 1
            results = {}
            mask = self.torso_mask(bg_coords)
            torso_alpha = torch.zeros([N, 1], device=device)
            torso_color = torch.zeros([N, 3], device=device)
            if mask.any():
                if self.torso_head_aware and random.random() < 0.5:   # the reference flips this coin at inference too (:175-179)
                    a, c, deform = self.forward_torso(bg_coords[mask], poses, self._torso_code(), image[mask], weights_sum.unsqueeze(-1)[mask])
                else:
                    a, c, deform = self.forward_torso(bg_coords[mask], poses, self._torso_code())
                torso_alpha[mask] = a.float()
                torso_color[mask] = c.float()
                results["deform"] = deform
            bg_color = torso_color * torso_alpha + bg_color * (1 - torso_alpha)
            results["torso_alpha_map"] = torso_alpha
            results["torso_rgb_map"] = bg_color
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            results["rgb_map"] = image.view(*prefix, 3).clamp(0, 1)
            results["depth_map"] = (torch.clamp(depth - nears, min=0) / (fars - nears)).view(*prefix)
            return results

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, pose6=None, ind_code=None, generator=None):
        """radnerf_torso.py:200-241: only the 2-D torso occupancy is refreshed (the head grid is frozen while the torso trains):
        alpha of the torso field on the jittered cell centres, 5x5 max-pool dilation, EMA-max, new mean_density_torso.
        The reference draws pose and identity code at random from `self.poses`; pass `pose6` [1,6] to fix them."""
        dev = self.density_bitfield.device
        if pose6 is None:
            if not hasattr(self, "poses"):
                raise RuntimeError("RADNeRFTorso.update_extra_state: give `pose6` or set model.poses (tasks/radnerfs/radnerf_torso.py:44-46)")
            from .utils import convert_poses
            rand_idx = random.randint(0, self.poses.shape[0] - 1)
            pose6 = convert_poses(self.poses[[rand_idx]]).to(dev)
            if ind_code is None and self.torso_individual_embedding_dim > 0:
                ind_code = self.torso_individual_codes[rand_idx]
        elif ind_code is None and self.torso_individual_embedding_dim > 0:
            ind_code = self.torso_individual_codes[0]
        G = self.grid_size
        fused_query = False
        if dev.type == "cuda" and ind_code is not None and self.field_impl == "auto" and self.render_impl in ("auto", "fused") \
                and self._pick_impl("auto", False, 1) == "fused":
            from .train_torso import supported, torso_field_no_grad
            fused_query = supported(self)
        tmp = torch.zeros_like(self.density_grid_torso)
        X = torch.arange(G, dtype=torch.int32, device=dev).split(S)
        half_grid_size = 1 / G
        for xs in X:
            for ys in X:
                xx, yy = torch.meshgrid(xs, ys, indexing="ij")
                coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1)], dim=-1)
                indices = (coords[:, 1] * G + coords[:, 0]).long()   # xy transposed, as in the reference
                xys = (2 * coords.float() / (G - 1) - 1) * (1 - half_grid_size)
                noise = _rand_like(xys, generator)
                xys = xys + (noise * 2 - 1) * half_grid_size
                if fused_query:       # one launch (train_torso.torso_field_no_grad) instead of the ~60 of the op graph
                    alphas, _, _ = torso_field_no_grad(self, xys, pose6.to(dev), ind_code)
                else:
                    alphas, _, _ = self.forward_torso(xys, pose6.to(dev), ind_code)
                tmp[indices] = alphas.squeeze(1).float()
        tmp = torch.nn.functional.max_pool2d(tmp.view(1, 1, G, G), kernel_size=5, stride=1, padding=2).view(-1)
        self.density_grid_torso = torch.maximum(self.density_grid_torso * decay, tmp)
        self.mean_density_torso = torch.mean(self.density_grid_torso).item()
