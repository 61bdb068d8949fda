"""Model-level oracle (TEST INFRASTRUCTURE ONLY): the RAD-NeRF head / head+torso frame on CPU.

A functional (state_dict in, tensors out) torch-fp32 restatement of the reference's Python layers
on top of the C kernels in radnerf_kernels.c.  It needs neither /root/reference nor a GPU, so it
is what the `-m gpu` parity tests, smoke() and bench.py's cpu_baseline leg check against on the
GPU box.  It is itself pinned, in the build container, against the reference's own unmodified
Python (oracle/refshim.py) by tests/test_oracle_vs_reference.py and the committed golden vectors.

Reference lines followed (relative to /root/reference):
  cond encoder    modules/radnerfs/cond_encoder.py:44-52 (AudioNet), :79-89 (AudioAttNet), :106-111 (MLP)
  head field      modules/radnerfs/radnerf.py:61-105
  march loop      modules/radnerfs/renderer.py:263-367 (inference branch :314-351)
  torso           modules/radnerfs/radnerf_torso.py:51-84, :156-198
  op wrappers     raymarching/raymarching.py:18-48, :347-420; gridencoder/grid.py:27-63, :145-161;
                  shencoder/sphere_harmonics.py:14-40; freqencoder/freq.py:15-36
  rays / poses    modules/radnerfs/utils.py:263-363
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

import contextlib

from . import kernels as K

RM, GE, SH, FQ = K.raymarching_face, K.gridencoder, K.shencoder, K.freqencoder


@contextlib.contextmanager
def kernel_backend(modules):
    """Runs the restatement over another set of the four extension modules -- oracle/ref_kernels.py's build of the reference's
    own kernels for gfx950 (then the tensors live on the GPU) -- instead of the C restatement.  Same positional signatures."""
    global RM, GE, SH, FQ
    saved = (RM, GE, SH, FQ)
    RM, GE, SH, FQ = modules
    try:
        yield
    finally:
        RM, GE, SH, FQ = saved


# ----------------------------------------------------------------------------- op wrappers
def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
    N = rays_o.shape[0]
    nears, fars = torch.empty(N, device=rays_o.device), torch.empty(N, device=rays_o.device)
    RM.near_far_from_aabb(rays_o.contiguous(), rays_d.contiguous(), aabb.contiguous(), N, min_near, nears, fars)
    return nears, fars


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars, align, dt_gamma, max_steps, noises=None):
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    dev = rays_o.device
    xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
    if noises is None:
        noises = torch.zeros(n_alive, device=dev)  # perturb=False; with perturb=True the reference draws torch.rand(n_alive) here (raymarching.py:395-398)
    RM.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, bitfield, nears, fars,
                  xyzs, dirs, deltas, noises)
    return xyzs, dirs, deltas


class _GridEncode(torch.autograd.Function):
    """gridencoder/grid.py:24-90 over the C kernels: [B, L*C] out, table gradient by scatter-add, input gradient through dy_dx."""

    @staticmethod
    def forward(ctx, x01, embeddings, offsets, S, base_resolution, gridtype, align_corners, interp, need_dx):
        x01, embeddings = x01.detach().contiguous(), embeddings.detach().contiguous()
        B, D = x01.shape
        L, C = offsets.shape[0] - 1, embeddings.shape[1]
        out = torch.empty(L, B, C, device=x01.device)
        dy_dx = torch.empty(B, L * D * C, device=x01.device) if need_dx else None
        GE.grid_encode_forward(x01, embeddings, offsets, out, B, D, C, L, S, base_resolution, dy_dx, gridtype, align_corners, interp)
        ctx.save_for_backward(x01, embeddings, offsets, dy_dx)
        ctx.cfg = (B, D, C, L, S, base_resolution, gridtype, align_corners, interp)
        return out.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        x01, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, align_corners, interp = ctx.cfg
        g = grad.contiguous().view(B, L, C).permute(1, 0, 2).contiguous()
        g_emb = torch.zeros_like(embeddings)
        g_in = torch.zeros(B, D, device=grad.device) if dy_dx is not None else None
        GE.grid_encode_backward(g, x01, embeddings, offsets, g_emb, B, D, C, L, S, H, dy_dx, g_in, gridtype, align_corners, interp)
        return g_in, g_emb, None, None, None, None, None, None, None


def grid_encode(x01, embeddings, offsets, per_level_scale, base_resolution, gridtype, align_corners, interp):
    need_graph = torch.is_grad_enabled() and (x01.requires_grad or embeddings.requires_grad)
    return _GridEncode.apply(x01, embeddings, offsets, float(np.log2(per_level_scale)), base_resolution, gridtype, align_corners, interp,
                             need_graph and x01.requires_grad)


class _TruncExp(torch.autograd.Function):
    """modules/radnerfs/utils.py:36-49"""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(ctx.saved_tensors[0].clamp(-15, 15))


class _FreqEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, degree):
        x = x.detach().contiguous()
        B, D = x.shape
        C = D + 2 * D * degree
        out = torch.empty(B, C, device=x.device)
        FQ.freq_encode_forward(x, B, D, degree, C, out)
        ctx.save_for_backward(out)
        ctx.cfg = (B, D, degree, C)
        return out

    @staticmethod
    def backward(ctx, grad):
        B, D, degree, C = ctx.cfg
        g_in = torch.zeros(B, D, device=grad.device)
        FQ.freq_encode_backward(grad.contiguous(), ctx.saved_tensors[0], B, D, degree, C, g_in)
        return g_in, None


class _CompositeTrain(torch.autograd.Function):
    """raymarching.py:286-342 over the C kernels (T_thresh 1e-4)."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, ambient, deltas, rays, T_thresh):
        sigmas, rgbs, ambient = sigmas.detach().contiguous(), rgbs.detach().contiguous(), ambient.detach().contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        ws, amb, dep, img = torch.empty(N), torch.empty(N), torch.empty(N), torch.empty(N, 3)
        RM.composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh, ws, amb, dep, img)
        ctx.save_for_backward(sigmas, rgbs, ambient, deltas, rays, ws, amb, img)
        ctx.cfg = (M, N, T_thresh)
        return ws, amb, dep, img

    @staticmethod
    def backward(ctx, g_ws, g_amb, g_dep, g_img):
        sigmas, rgbs, ambient, deltas, rays, ws, amb, img = ctx.saved_tensors
        M, N, T_thresh = ctx.cfg
        gs, gc, ga = torch.zeros(M), torch.zeros(M, 3), torch.zeros(M)
        RM.composite_rays_train_backward(g_ws.contiguous(), g_amb.contiguous(), g_img.contiguous(), sigmas, rgbs, ambient, deltas, rays, ws, amb, img,
                                         M, N, T_thresh, gs, gc, ga)
        return gs, gc, ga, None, None, None


def sh_encode(d, degree=4):
    d = d.contiguous()
    out = torch.empty(d.shape[0], degree * degree, device=d.device)
    SH.sh_encode_forward(d, out, d.shape[0], 3, degree, None)
    return out


def freq_encode(x, degree):
    return _FreqEncode.apply(x, degree)


# ----------------------------------------------------------------------------- small networks
def mlp(sd, prefix, x, num_layers):
    for l in range(num_layers):
        x = F.linear(x, sd[f"{prefix}.net.{l}.weight"])
        if l != num_layers - 1:
            x = F.relu(x)
    return x


def cal_cond_feat(sd, hp, cond):
    """cond [smo_win, cond_win, C] -> [cond_out_dim]"""
    strides = {1: (1, 1, 1, 1), 2: (2, 1, 1, 1), 3: (2, 2, 1, 1), 4: (2, 2, 1, 1), 16: (2, 2, 2, 2)}[hp["cond_win_size"]]
    x = cond.permute(0, 2, 1)
    for i, s in enumerate(strides):
        x = F.leaky_relu(F.conv1d(x, sd[f"cond_prenet.encoder_conv.{2 * i}.weight"], sd[f"cond_prenet.encoder_conv.{2 * i}.bias"],
                                  stride=s, padding=1), 0.02)
    x = x.squeeze(-1)
    x = F.leaky_relu(F.linear(x, sd["cond_prenet.encoder_fc1.0.weight"], sd["cond_prenet.encoder_fc1.0.bias"]), 0.02)
    x = F.linear(x, sd["cond_prenet.encoder_fc1.2.weight"], sd["cond_prenet.encoder_fc1.2.bias"]).squeeze()
    if not hp["with_att"]:
        return x
    seq = hp["smo_win_size"]
    y = x[:, :hp["cond_out_dim"]].permute(1, 0).unsqueeze(0)
    for i in range(5):
        y = F.leaky_relu(F.conv1d(y, sd[f"cond_att_net.attentionConvNet.{2 * i}.weight"], sd[f"cond_att_net.attentionConvNet.{2 * i}.bias"],
                                  stride=1, padding=1), 0.02)
    y = F.linear(y.view(1, seq), sd["cond_att_net.attentionNet.0.weight"], sd["cond_att_net.attentionNet.0.bias"])
    y = torch.softmax(y, dim=1).view(seq, 1)
    return torch.sum(y * x, dim=0)


def _grid_args(hp, desired_resolution):
    pls = np.exp2(np.log2(desired_resolution / 16) / (16 - 1))
    gridtype = {"hashgrid": 0, "tiledgrid": 1}[hp["grid_type"]]
    interp = {"linear": 0, "smoothstep": 1}[hp["grid_interpolation_type"]]
    return pls, gridtype, interp


def head_field(sd, hp, position, direction, cond_feat, ind_code):
    """RADNeRF.forward: -> sigma [M], color [M,3], ambient_pos [M,2]"""
    M = position.shape[0]
    bound = hp["bound"]
    pls3, gt, ip = _grid_args(hp, hp["desired_resolution"] * bound)
    pls2, _, _ = _grid_args(hp, hp["desired_resolution"])
    pos_feat = grid_encode((position + bound) / (2 * bound), sd["position_embedder.embeddings"], sd["position_embedder.offsets"],
                           pls3, 16, gt, False, ip)
    ambient_in = torch.cat([pos_feat, cond_feat.reshape(1, -1).repeat(M, 1)], dim=1)
    ambient_pos = torch.tanh(mlp(sd, "ambient_net", ambient_in, hp["num_layers_ambient"]).float())
    ambient_feat = grid_encode((ambient_pos + 1) / 2, sd["ambient_embedder.embeddings"], sd["ambient_embedder.offsets"], pls2, 16,
                               gt, False, ip)
    h = mlp(sd, "sigma_net", torch.cat([pos_feat, ambient_feat], dim=-1), hp["num_layers_sigma"])
    sigma = _TruncExp.apply(h[..., 0])
    parts = [sh_encode(direction), h[..., 1:]]
    if ind_code is not None:
        parts.append(ind_code.reshape(1, -1).repeat(M, 1))
    color = torch.sigmoid(mlp(sd, "color_net", torch.cat(parts, dim=-1), hp["num_layers_color"]))
    return sigma, color, ambient_pos


def torso_field(sd, hp, x, poses6, code, image=None, weights_sum=None):
    """RADNeRFTorso.forward_torso (radnerf_torso.py:51-84): x [m,2], poses6 [1,6], code [8] -> alpha [m,1], color [m,3], dx [m,2];
    with hp['torso_head_aware'] the head's colour / opacity at the pixel (or zeros) goes through head_color_weights_encoder (:68-74)."""
    m = x.shape[0]
    x = x * hp["torso_shrink"]
    parts = [freq_encode(x, 10), freq_encode(poses6.reshape(1, 6), 4).repeat(m, 1)]
    if code is not None:
        parts.append(code.reshape(1, -1).repeat(m, 1))
    if hp.get("torso_head_aware", False):
        if image is None:
            image, weights_sum = torch.zeros(m, 3, device=x.device), torch.zeros(m, 1, device=x.device)   # (on the GPU when the reference's kernels are the backend)
        e = torch.cat([image, weights_sum], dim=-1)
        for i in (0, 2, 4):
            e = F.linear(e, sd[f"head_color_weights_encoder.{i}.weight"], sd[f"head_color_weights_encoder.{i}.bias"])
            if i < 4:
                e = F.leaky_relu(e, 0.02)
        parts.append(e)
    h = torch.cat(parts, dim=-1)
    dx = mlp(sd, "torso_deform_net", h, 3)
    xc = (x + dx).clamp(-1, 1).float()
    pls = np.exp2(np.log2(2048 / 16) / 15)
    feat = grid_encode((xc + 1) / 2, sd["torso_embedder.embeddings"], sd["torso_embedder.offsets"], pls, 16, 1, False, 0)
    h = mlp(sd, "torso_canonicial_net", torch.cat([feat, h], dim=-1), 3)
    return torch.sigmoid(h[..., :1]), torch.sigmoid(h[..., 1:]), dx


# ----------------------------------------------------------------------------- frame
def _count_composited(n_alive, n_step, T_thresh, ws0, sigmas, deltas):
    """How many of an iteration's marched samples the compositor actually consumes (it stops at the first empty slot and
    after the sample at which T < T_thresh, raymarching.cu:977,1004) -- the lower bound on field evaluations any
    schedule needs; `n_valid` (everything marched) is what the reference's schedule evaluates."""
    M = n_alive * n_step
    s, dt = sigmas[:M].view(n_alive, n_step), deltas[:M, 0].view(n_alive, n_step)
    ws, live, cnt = ws0.clone(), torch.ones(n_alive, dtype=torch.bool, device=ws0.device), 0
    for k in range(n_step):
        valid = live & (dt[:, k] != 0)
        T = 1 - ws
        ws = torch.where(valid, ws + (1 - torch.exp(-s[:, k] * dt[:, k])) * T, ws)
        cnt += int(valid.sum())
        live = valid & ~(T < T_thresh)
    return cnt


def march_head(sd, hp, rays_o, rays_d, cond_feat, dt_gamma, max_steps, T_thresh, trace=None, perturb_noise=None):
    """renderer.py:316-351.  perturb_noise [N]: the U[0,1) draws of perturb=True, used by the FIRST iteration only (renderer.py:338-342:
    `perturb if step == 0 else False`; all N rays are alive then, so draw n belongs to ray n)."""
    N = rays_o.shape[0]
    cascade = 1 + math.ceil(math.log2(hp["bound"]))
    nears, fars = near_far_from_aabb(rays_o, rays_d, sd["aabb_infer"], hp["min_near"])
    ind_code = sd["individual_embeddings"][0] if hp["individual_embedding_dim"] > 0 else None
    dev = rays_o.device
    weights_sum, depth, image = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
    rays_alive = torch.arange(N, dtype=torch.int32, device=dev)
    rays_t = nears.clone()
    step = 0
    while step < max_steps:
        n_alive = rays_alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, deltas = march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, float(hp["bound"]), sd["density_bitfield"],
                                        cascade, hp["grid_size"], nears, fars, 128, dt_gamma, max_steps,
                                        noises=perturb_noise.contiguous().float() if (perturb_noise is not None and step == 0) else None)
        sigmas, rgbs, _ = head_field(sd, hp, xyzs, dirs, cond_feat, ind_code)
        if trace is not None:
            trace.append({"n_alive": n_alive, "n_step": n_step, "n_valid": int((deltas[:, 0] > 0).sum()),
                          "n_composited": _count_composited(n_alive, n_step, T_thresh, weights_sum[rays_alive.long()], sigmas, deltas)})
        RM.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.contiguous(), rgbs.contiguous(), deltas, weights_sum,
                          depth, image)
        rays_alive = rays_alive[rays_alive >= 0].contiguous()
        step += n_step
    return weights_sum, depth, image, nears, fars


def render(sd, hp, rays_o, rays_d, cond, bg_coords, poses6, bg_color, torso, dt_gamma=None, max_steps=None, T_thresh=1e-4, trace=None,
           head_aware_branch=False, perturb_noise=None):
    """One frame: the dict `NeRFRenderer.render` / `RADNeRFTorso.render` return at inference.  `head_aware_branch` fixes the coin the
    reference flips per frame when torso_head_aware is set (radnerf_torso.py:175-179): True = the torso sees the rendered head."""
    dt_gamma = hp["dt_gamma"] if dt_gamma is None else dt_gamma
    max_steps = hp["max_steps"] if max_steps is None else max_steps
    with torch.no_grad():
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        cond_feat = cal_cond_feat(sd, hp, cond)
        weights_sum, depth, image, nears, fars = march_head(sd, hp, rays_o, rays_d, cond_feat, dt_gamma, max_steps, T_thresh, trace, perturb_noise)
        if bg_color is None:
            bg_color = 1
        out = {}
        if torso:
            bg_coords = bg_coords.contiguous().view(-1, 2)
            G = hp["grid_size"]
            occ = F.grid_sample(sd["density_grid_torso"].view(1, 1, G, G), bg_coords.view(1, -1, 1, 2), align_corners=True).view(-1)
            mask = occ > min(hp["density_thresh_torso"], 0)  # mean_density_torso is 0 after a fresh load
            torso_alpha, torso_color = torch.zeros(N, 1, device=rays_o.device), torch.zeros(N, 3, device=rays_o.device)
            if mask.any():
                code = sd["torso_individual_codes"][0] if hp["torso_individual_embedding_dim"] > 0 else None
                if hp.get("torso_head_aware", False) and head_aware_branch:
                    a, c, deform = torso_field(sd, hp, bg_coords[mask], poses6, code, image[mask], weights_sum.unsqueeze(-1)[mask])
                else:
                    a, c, deform = torso_field(sd, hp, bg_coords[mask], poses6, code)
                torso_alpha[mask], torso_color[mask] = a, c
                out["deform"] = deform
            bg_color = torso_color * torso_alpha + bg_color * (1 - torso_alpha)
            out["torso_alpha_map"], out["torso_rgb_map"] = torso_alpha, bg_color
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        out["rgb_map"] = image.view(*prefix, 3).clamp(0, 1)
        out["depth_map"] = (torch.clamp(depth - nears, min=0) / (fars - nears)).view(*prefix)
        out["weights_sum"] = weights_sum
        return out


# ----------------------------------------------------------------------------- rays / poses
def get_rays(pose44, intrinsics, H, W):
    fx, fy, cx, cy = intrinsics
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i = i.t().reshape(1, H * W) + 0.5
    j = j.t().reshape(1, H * W) + 0.5
    zs = torch.ones_like(i)
    directions = torch.stack(((i - cx) / fx * zs, (j - cy) / fy * zs, zs), dim=-1)
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    rays_d = directions @ pose44[:, :3, :3].transpose(-1, -2)
    rays_o = pose44[..., :3, 3][..., None, :].expand_as(rays_d)
    return rays_o, rays_d


def get_bg_coords(H, W):
    X = torch.arange(H) / (H - 1) * 2 - 1
    Y = torch.arange(W) / (W - 1) * 2 - 1
    xs, ys = torch.meshgrid(X, Y, indexing="ij")
    return torch.cat([xs.reshape(-1, 1), ys.reshape(-1, 1)], dim=-1).unsqueeze(0)


def convert_poses(pose44):
    m = pose44[:, :3, :3]
    out = torch.empty(pose44.shape[0], 6)
    out[:, 0] = torch.atan2(-m[:, 1, 2], m[:, 2, 2])
    out[:, 1] = torch.asin(m[:, 0, 2])
    out[:, 2] = torch.atan2(-m[:, 0, 1], m[:, 0, 0])
    out[:, 3:] = pose44[:, :3, 3]
    return out


# ----------------------------------------------------------------------------- occupancy-grid maintenance
def _cell_coords(G, S):
    X = torch.arange(G, dtype=torch.int32).split(S)
    for xs in X:
        for ys in X:
            for zs in X:
                xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                yield torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1).contiguous()


def _morton(coords):
    idx = torch.empty(coords.shape[0], dtype=torch.int32)
    RM.morton3D(coords, coords.shape[0], idx)
    return idx.long()


def update_density_grid(sd, hp, density_grid, cond, generator, decay=0.95, S=128, density_scale=1.0):
    """NeRFRenderer.update_extra_state (modules/radnerfs/renderer.py:199-260) for a fixed condition window and jitter
    generator: returns (new density_grid [C, G^3], mean_density, density_bitfield u8 [C*G^3/8])."""
    G, bound = hp["grid_size"], hp["bound"]
    cascade = 1 + math.ceil(math.log2(bound))
    enc_a = cal_cond_feat(sd, hp, cond)
    tmp = torch.zeros_like(density_grid)
    dummy_ind = sd["individual_embeddings"][0] if hp["individual_embedding_dim"] > 0 else None
    for coords in _cell_coords(G, S):
        indices = _morton(coords)
        xyzs = 2 * coords.float() / (G - 1) - 1
        for cas in range(cascade):
            b = min(2 ** cas, bound)
            hgs = b / G
            cas_xyzs = xyzs * (b - hgs)
            cas_xyzs = cas_xyzs + (torch.rand(cas_xyzs.shape, generator=generator) * 2 - 1) * hgs
            d = torch.zeros_like(cas_xyzs)
            d[:, 2] = 1.0
            sigma, _, _ = head_field(sd, hp, cas_xyzs, d, enc_a, dummy_ind)
            tmp[cas, indices] = sigma.reshape(-1) * density_scale
    dil = torch.empty_like(tmp)
    RM.morton3D_dilation(tmp.contiguous(), cascade, G, dil)
    grid = density_grid.clone()
    valid = (grid >= 0) & (dil >= 0)
    grid[valid] = torch.maximum(grid[valid] * decay, dil[valid])
    mean_density = torch.mean(grid.clamp(min=0)).item()
    thresh = min(mean_density, hp["density_thresh"])
    bits = torch.zeros(cascade * G ** 3 // 8, dtype=torch.uint8)
    RM.packbits(grid.contiguous(), bits.numel(), thresh, bits)
    return grid, mean_density, bits


def update_density_grid_torso(sd, hp, density_grid_torso, pose6, code, generator, decay=0.95, S=128):
    """RADNeRFTorso.update_extra_state (modules/radnerfs/radnerf_torso.py:200-241) for a fixed pose / identity code."""
    G = hp["grid_size"]
    tmp = torch.zeros_like(density_grid_torso)
    X = torch.arange(G, dtype=torch.int32).split(S)
    hgs = 1 / G
    for xs in X:
        for ys in X:
            xx, yy = torch.meshgrid(xs, ys, indexing="ij")
            coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1)], dim=-1)
            indices = (coords[:, 1] * G + coords[:, 0]).long()
            xys = (2 * coords.float() / (G - 1) - 1) * (1 - hgs)
            xys = xys + (torch.rand(xys.shape, generator=generator) * 2 - 1) * hgs
            alphas, _, _ = torso_field(sd, hp, xys, pose6, code)
            tmp[indices] = alphas.squeeze(1).float()
    tmp = F.max_pool2d(tmp.view(1, 1, G, G), kernel_size=5, stride=1, padding=2).view(-1)
    grid = torch.maximum(density_grid_torso * decay, tmp)
    return grid, torch.mean(grid).item()


def mark_untrained_grid(hp, density_grid, poses, intrinsic, S=64):
    """NeRFRenderer.mark_untrained_grid (modules/radnerfs/renderer.py:129-196): -1 where no camera sees the cell."""
    G, bound = hp["grid_size"], hp["bound"]
    cascade = 1 + math.ceil(math.log2(bound))
    fx, fy, cx, cy = intrinsic
    count = torch.zeros_like(density_grid)
    B = poses.shape[0]
    for coords in _cell_coords(G, S):
        indices = _morton(coords)
        world = (2 * coords.float() / (G - 1) - 1).unsqueeze(0)
        for cas in range(cascade):
            b = min(2 ** cas, bound)
            hgs = b / G
            cw = world * (b - hgs)
            head = 0
            while head < B:
                tail = min(head + S, B)
                cam = (cw - poses[head:tail, :3, 3].unsqueeze(1)) @ poses[head:tail, :3, :3]
                m = (cam[:, :, 2] > 0) & (cam[:, :, 0].abs() < cx / fx * cam[:, :, 2] + hgs * 2) & (cam[:, :, 1].abs() < cy / fy * cam[:, :, 2] + hgs * 2)
                count[cas, indices] += m.sum(0).reshape(-1)
                head += S
    out = density_grid.clone()
    out[count == 0] = -1
    return out


# ----------------------------------------------------------------------------- training branch
def render_train(sd, hp, rays_o, rays_d, cond, bg_coords, poses6, bg_color, torso, index=0, mean_count=-1, noises=None, force_all_rays=True):
    """NeRFRenderer.render / RADNeRFTorso.render with self.training (renderer.py:296-313, radnerf_torso.py:93-198): differentiable
    w.r.t. every tensor of `sd` that requires grad.  Returns the result dict (+ the marcher's `rays` and point count)."""
    dt_gamma, max_steps = hp["dt_gamma"], hp["max_steps"]
    prefix = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3).float()
    rays_d = rays_d.contiguous().view(-1, 3).float()
    N = rays_o.shape[0]
    cascade = 1 + math.ceil(math.log2(hp["bound"]))
    nears, fars = near_far_from_aabb(rays_o, rays_d, sd["aabb_train"], hp["min_near"])

    def head():
        cond_feat = cal_cond_feat(sd, hp, cond)
        ind_code = sd["individual_embeddings"][index] if hp["individual_embedding_dim"] > 0 else None
        M = N * max_steps if (force_all_rays or mean_count <= 0) else mean_count + (128 - mean_count % 128)
        xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        rays, counter = torch.empty(N, 3, dtype=torch.int32), torch.zeros(2, dtype=torch.int32)
        RM.march_rays_train(rays_o, rays_d, sd["density_bitfield"], float(hp["bound"]), dt_gamma, max_steps, N, cascade, hp["grid_size"], M,
                            nears, fars, xyzs, dirs, deltas, rays, counter, torch.zeros(N) if noises is None else noises)
        if force_all_rays or mean_count <= 0:
            m = counter[0].item()
            m += 128 - m % 128
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        sigmas, rgbs, ambient = head_field(sd, hp, xyzs, dirs, cond_feat, ind_code)
        ws, amb, dep, img = _CompositeTrain.apply(sigmas, rgbs, ambient.abs().sum(-1), deltas.contiguous(), rays, 1e-4)
        return ws, amb, dep, img, rays, counter

    if torso:
        with torch.no_grad():
            weights_sum, ambient_sum, depth, image, rays, counter = head()
    else:
        weights_sum, ambient_sum, depth, image, rays, counter = head()
    out = {"weights_sum": weights_sum, "ambient": ambient_sum, "rays": rays, "n_points": int(counter[0])}
    if bg_color is None:
        bg_color = 1
    if torso:
        bg_coords = bg_coords.contiguous().view(-1, 2)
        G = hp["grid_size"]
        occ = F.grid_sample(sd["density_grid_torso"].view(1, 1, G, G), bg_coords.view(1, -1, 1, 2), align_corners=True).view(-1)
        mask = occ > min(hp["density_thresh_torso"], 0)
        torso_alpha, torso_color = torch.zeros(N, 1), torch.zeros(N, 3)
        if mask.any():
            code = sd["torso_individual_codes"][index] if hp["torso_individual_embedding_dim"] > 0 else None
            a, c, deform = torso_field(sd, hp, bg_coords[mask], poses6, code)
            torso_alpha = torso_alpha.masked_scatter(mask.unsqueeze(-1), a)
            torso_color = torso_color.masked_scatter(mask.unsqueeze(-1).expand(-1, 3), c)
            out["deform"] = deform
        bg_color = torso_color * torso_alpha + bg_color * (1 - torso_alpha)
        out["torso_alpha_map"], out["torso_rgb_map"] = torso_alpha, bg_color
    image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
    out["rgb_map"] = image.view(*prefix, 3).clamp(0, 1)
    out["depth_map"] = (torch.clamp(depth - nears, min=0) / (fars - nears)).view(*prefix)
    return out
