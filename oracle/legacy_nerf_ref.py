"""Baseline B2 (TEST INFRASTRUCTURE / cpu_baseline leg ONLY): the reference's pure-PyTorch vanilla NeRF render path on CPU.

north_star asks for "the reference's pure-PyTorch CPU NeRF path timed on the host cores" beside the GPU number.  RAD-NeRF has no
CPU path (its kernels are CUDA-only); the one renderer in the reference that is device-agnostic torch is the legacy AD-NeRF-style
model it replaced: `Lm3dNeRF` (modules/nerfs/lm3d_nerf/lm3d_nerf.py:13-58) rendered by `render_dynamic_face`
(modules/nerfs/commons/volume_rendering.py:234-282).  It is a different model from the hot path (64 stratified + 128
importance samples per ray through two 8x256 MLPs, no occupancy grid), so it is reported as context next to the like-for-like
oracle port, not as the parity oracle.  The reference's Python cannot travel to the GPU box, hence this functional restatement;
tests/test_vs_reference.py (-m reference) checks it against the reference's own classes on identical weights and random draws.

Lines followed: embedders.py:5-45 (FreqEmbedder, log bands, include_input), adnerf/backbone.py:82-137 (NeRFBackbone, skip at
layer 4, colour branch 128-wide), volume_rendering.py:9-59 (raw2outputs), :62-95 (sample_pdf), :98-210 (render_rays, perturb=1
as run_model leaves it at inference), :213-231 (chunking, 2048), ray_samplers.py:11-45 (get_rays); config egs/egs_bases/nerf/base.yaml:
near 0.3, far 0.9, 64 + 128 samples, hidden 256, cond 64.
"""
import math

import torch
import torch.nn.functional as F

POS_RES, VIEW_RES, HID, COND = 10, 4, 256, 64
POS_DIM, VIEW_DIM = 3 + 2 * 3 * POS_RES, 3 + 2 * 3 * VIEW_RES
N_COARSE, N_FINE, NEAR, FAR, CHUNK = 64, 128, 0.3, 0.9, 2048


def freq_embed(x, multires):
    out = [x]
    for f in 2.0 ** torch.linspace(0.0, multires - 1, steps=multires):
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, dim=-1)


def backbone_shapes():
    din = POS_DIM + COND
    dens = [(HID, din)] + [(HID, HID + din) if i == 4 else (HID, HID) for i in range(7)]     # Linear i+1 takes the skip when i == 4
    col = [(HID // 2, VIEW_DIM + HID), (HID // 2, HID // 2), (HID // 2, HID // 2)]
    return {"density_linears": dens, "density_out_linear": [(1, HID)], "color_linears": col, "color_out_linear": [(3, HID // 2)]}


def make_weights(seed=0):
    """nn.Linear default init (uniform +-1/sqrt(fan_in) for weight and bias) for the coarse and the fine backbone."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for net in ("model_coarse", "model_fine"):
        for group, shapes in backbone_shapes().items():
            for i, (o, k) in enumerate(shapes):
                b = 1.0 / math.sqrt(k)
                name = f"{net}.{group}.{i}" if len(shapes) > 1 else f"{net}.{group}"
                w[name + ".weight"] = (torch.rand(o, k, generator=g) * 2 - 1) * b
                w[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * b
    return w


def backbone(w, net, pos, cond, view):
    """pos [R,S,63], cond [64], view [R,27] -> [R,S,4] (rgb, sigma)."""
    R, S, _ = pos.shape
    x = torch.cat([pos, cond.reshape(1, 1, -1).expand(R, S, COND)], dim=-1)
    h = x
    for i in range(8):
        h = F.relu(F.linear(h, w[f"{net}.density_linears.{i}.weight"], w[f"{net}.density_linears.{i}.bias"]))
        if i == 4:
            h = torch.cat([x, h], dim=-1)
    sigma = F.linear(h, w[f"{net}.density_out_linear.weight"], w[f"{net}.density_out_linear.bias"])
    h = torch.cat([h, view[:, None, :].expand(R, S, VIEW_DIM)], dim=-1)
    for i in range(3):
        h = F.relu(F.linear(h, w[f"{net}.color_linears.{i}.weight"], w[f"{net}.color_linears.{i}.bias"]))
    rgb = F.linear(h, w[f"{net}.color_out_linear.weight"], w[f"{net}.color_out_linear.bias"])
    return torch.cat([rgb, sigma], dim=-1)


def raw2outputs(raw, z_vals, rays_d, bc_rgb):
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], dim=-1) * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    rgb = torch.cat((rgb[:, :-1, :], bc_rgb.unsqueeze(1)), dim=1)             # the last sample is the background colour
    alpha = 1.0 - torch.exp(-(F.relu(raw[..., 3]) + 1e-6) * dists)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], dim=-1), dim=-1)[:, :-1]
    weights = alpha * T
    return torch.sum(weights[..., None] * rgb, dim=-2), weights


def sample_pdf(bins, weights, n):
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = torch.rand(list(cdf.shape[:-1]) + [n]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below, above = (inds - 1).clamp(min=0), inds.clamp(max=cdf.shape[-1] - 1)
    inds_g = torch.stack([below, above], -1)
    shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def render_rays(w, rays_o, rays_d, viewdirs, bc_rgb, cond):
    R = rays_o.shape[0]
    t = torch.linspace(0.0, 1.0, steps=N_COARSE)
    z = (NEAR * (1.0 - t) + FAR * t).expand(R, N_COARSE)
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper, lower = torch.cat([mids, z[..., -1:]], -1), torch.cat([z[..., :1], mids], -1)
    t_rand = torch.rand(z.shape)
    t_rand[..., -1] = 1.0
    z = lower + (upper - lower) * t_rand
    view = freq_embed(viewdirs, VIEW_RES)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    raw = backbone(w, "model_coarse", freq_embed(pts, POS_RES), cond, view)
    _, weights = raw2outputs(raw, z, rays_d, bc_rgb)
    z_mid = 0.5 * (z[..., 1:] + z[..., :-1])
    z_fine = sample_pdf(z_mid, weights[..., 1:-1], N_FINE)
    z, _ = torch.sort(torch.cat([z, z_fine], -1), -1)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * z[..., :, None]
    raw = backbone(w, "model_fine", freq_embed(pts, POS_RES), cond, view)
    rgb, _ = raw2outputs(raw, z, rays_d, bc_rgb)
    return rgb


def get_rays(H, W, focal, c2w, cx, cy):
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i, j = i.t(), j.t()
    d = torch.stack([(i - cx) / focal, -(j - cy) / focal, -torch.ones_like(i)], dim=-1)
    rays_d = torch.sum(d[..., None, :] * c2w[:3, :3], dim=-1)
    return c2w[:3, -1].expand(rays_d.shape), rays_d


def render(w, H, W, focal, cx, cy, c2w, bg_img, cond, max_rays=None):
    """One frame (or its first `max_rays` rays, for a bounded timing sample) -> rgb [n,3]."""
    with torch.no_grad():
        rays_o, rays_d = get_rays(H, W, focal, c2w, cx, cy)
        rays_o, rays_d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
        view = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
        bc = bg_img.reshape(-1, 3)
        n = rays_o.shape[0] if max_rays is None else min(max_rays, rays_o.shape[0])
        out = [render_rays(w, rays_o[s:s + CHUNK], rays_d[s:s + CHUNK], view[s:s + CHUNK], bc[s:s + CHUNK], cond) for s in range(0, n, CHUNK)]
        return torch.cat(out, 0)[:n]
