"""Camera / pose helpers of the render path (torch, tiny, once per frame or per sequence).

Behaviour follows modules/radnerfs/utils.py of the reference: trunc_exp :36-49,
nerf_matrix_to_ngp :53-60, matrix_to_euler_angles :160-199 ('XYZ'), convert_poses :263-269,
get_bg_coords :273-278, get_rays :282-363 (full-image branch, N = -1), and
tasks/radnerfs/dataset_utils.py:16-36 smooth_camera_path.
"""
import numpy as np
import torch


class _trunc_exp(torch.autograd.Function):
    """modules/radnerfs/utils.py:36-49: exp in the forward, gradient clamped to exp(clamp(x, -15, 15)) in the backward."""

    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply


def nerf_matrix_to_ngp(pose, scale=4, offset=(0, 0, 0)):
    p = np.asarray(pose)
    return np.array([
        [p[1, 0], -p[1, 1], -p[1, 2], p[1, 3] * scale + offset[0]],
        [p[2, 0], -p[2, 1], -p[2, 2], p[2, 3] * scale + offset[1]],
        [p[0, 0], -p[0, 1], -p[0, 2], p[0, 3] * scale + offset[2]],
        [0, 0, 0, 1]], dtype=np.float32)


def matrix_to_euler_angles_xyz(m: torch.Tensor) -> torch.Tensor:
    """XYZ Tait-Bryan angles of rotation matrices [...,3,3] (pytorch3d convention used by the reference)."""
    central = torch.asin(m[..., 0, 2])
    a0 = torch.atan2(-m[..., 1, 2], m[..., 2, 2])
    a2 = torch.atan2(-m[..., 0, 1], m[..., 0, 0])
    return torch.stack((a0, central, a2), -1)


@torch.autocast("cuda", enabled=False)      # as the reference's (modules/radnerfs/utils.py:262-281): geometry stays fp32 under the Trainer's autocast
def convert_poses(poses: torch.Tensor) -> torch.Tensor:
    """[B,4,4] cam2world -> [B,6] (XYZ euler, translation): the torso network's pose input."""
    out = torch.empty(poses.shape[0], 6, dtype=torch.float32, device=poses.device)
    out[:, :3] = matrix_to_euler_angles_xyz(poses[:, :3, :3].float())
    out[:, 3:] = poses[:, :3, 3]
    return out


@torch.autocast("cuda", enabled=False)      # as the reference's (modules/radnerfs/utils.py:262-281): geometry stays fp32 under the Trainer's autocast
def get_bg_coords(H, W, device):
    X = torch.arange(H, device=device) / (H - 1) * 2 - 1
    Y = torch.arange(W, device=device) / (W - 1) * 2 - 1
    xs, ys = torch.meshgrid(X, Y, indexing="ij")
    return torch.cat([xs.reshape(-1, 1), ys.reshape(-1, 1)], dim=-1).unsqueeze(0)


@torch.autocast("cuda", enabled=False)      # as the reference's (modules/radnerfs/utils.py:262-281): geometry stays fp32 under the Trainer's autocast
def get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, rect=None):
    """Pinhole rays (modules/radnerfs/utils.py:282-363): pixel centres at +0.5, unit directions rotated by pose[:3,:3].
    N = -1: every pixel, row-major.  Training modes, same draws from torch's global generator as the reference makes: N > 0 random pixels
    (duplicates possible), patch_size > 1: N // patch_size^2 random patches (top-left corners in [0, H - p) x [0, W - p)), rect =
    (row0, row1, col0, col1): every pixel of that rectangle (the lip-finetune crop).  Returns i, j (pixel centres), inds, rays_o, rays_d,
    all [B, N, ...]."""
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsics
    if rect is not None:
        r0, r1, c0, c1 = rect
        N = (r1 - r0) * (c1 - c0)
    if N > 0:
        N = min(N, H * W)
        if patch_size > 1:
            num_patch = N // (patch_size ** 2)
            rows = torch.randint(0, H - patch_size, size=[num_patch], device=device)
            cols = torch.randint(0, W - patch_size, size=[num_patch], device=device)
            p = torch.arange(patch_size, device=device)
            rr = (rows[:, None, None] + p[None, :, None]).expand(num_patch, patch_size, patch_size)
            cc = (cols[:, None, None] + p[None, None, :]).expand(num_patch, patch_size, patch_size)
            inds = (rr * W + cc).reshape(-1)
            inds = inds.expand([B, inds.numel()])
        elif rect is not None:
            r0, r1, c0, c1 = rect
            rr = torch.arange(max(r0, 0), min(r1, H), device=device)
            cc = torch.arange(max(c0, 0), min(c1, W), device=device)
            inds = (rr[:, None] * W + cc[None, :]).reshape(1, -1)
        else:
            inds = torch.randint(0, H * W, size=[N], device=device)
            inds = inds.expand([B, N])
    else:
        inds = torch.arange(H * W, device=device).expand([B, H * W])
    i = (inds % W).float() + 0.5      # the reference gathers these from a [H, W] meshgrid: small integers, exact in fp32 either way
    j = torch.div(inds, W, rounding_mode="floor").float() + 0.5
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    directions = torch.stack((xs, ys, zs), dim=-1)
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    return {"i": i, "j": j, "inds": inds, "rays_o": rays_o, "rays_d": rays_d}


def smooth_camera_path(poses: np.ndarray, kernel_size=7) -> np.ndarray:
    """Box-filter translations and chordal-mean rotations over a centred window (in place, like the reference)."""
    from scipy.spatial.transform import Rotation
    N, K = poses.shape[0], kernel_size // 2
    trans, rots = poses[:, :3, 3].copy(), poses[:, :3, :3].copy()
    for i in range(N):
        s, e = max(0, i - K), min(N, i + K + 1)
        poses[i, :3, 3] = trans[s:e].mean(0)
        try:
            poses[i, :3, :3] = Rotation.from_matrix(rots[s:e]).mean().as_matrix()
        except Exception:
            poses[i, :3, :3] = rots[i] if i == 0 else poses[i - 1, :3, :3]
    return poses


def get_audio_features(features, att_mode, index, smo_win_size):
    """modules/radnerfs/utils.py:71-103: the window of condition frames around `index` (zero padded at the ends)."""
    if att_mode == 0:
        return features[[index]]
    if att_mode == 1:
        left = index - smo_win_size
        pad_left = max(0, -left)
        auds = features[max(left, 0):index]
        if pad_left > 0:
            auds = torch.cat([torch.zeros(pad_left, *auds.shape[1:], device=auds.device, dtype=auds.dtype), auds], dim=0)
        return auds
    if att_mode == 2:
        left, right = index - smo_win_size // 2, index + (smo_win_size - smo_win_size // 2)
        pad_left, pad_right = max(0, -left), max(0, right - features.shape[0])
        auds = features[max(left, 0):min(right, features.shape[0])]
        if pad_left > 0:
            auds = torch.cat([torch.zeros(pad_left, *auds.shape[1:], device=auds.device, dtype=auds.dtype), auds], dim=0)
        if pad_right > 0:
            auds = torch.cat([auds, torch.zeros(pad_right, *auds.shape[1:], device=auds.device, dtype=auds.dtype)], dim=0)
        return auds
    raise NotImplementedError(f"wrong att_mode: {att_mode}")
