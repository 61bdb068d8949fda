"""Host-side ray-marching operators (inference + occupancy-grid maintenance).

Python-visible behaviour mirrors modules/radnerfs/raymarching/raymarching.py of the reference
(`near_far_from_aabb` :18-50, `morton3D` :85-110, `morton3D_invert` :112-135, `packbits` :137-163,
`morton3D_dilation` :165-183, `march_rays` :347-398, `composite_rays` :401-423): same argument
order, same allocation/zero-fill/padding rules, forward only (the reference wraps them in
autograd.Functions whose backward is None for these ops).  The arithmetic runs in
libgeneface_hip.so; there is no CPU path -- tensors that are not on a HIP device raise.
"""
import torch

from .compat import _raymarching_face as _backend


def _f32(x):
    return x if x.dtype == torch.float32 else x.float()  # custom_fwd(cast_inputs=torch.float32)


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o = _f32(rays_o).contiguous().view(-1, 3)
    rays_d = _f32(rays_d).contiguous().view(-1, 3)
    N = rays_o.shape[0]
    nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
    fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
    _backend.near_far_from_aabb(rays_o, rays_d, _f32(aabb).contiguous(), N, min_near, nears, fars)
    return nears, fars


def morton3D(coords):
    coords = coords.int().contiguous()
    N = coords.shape[0]
    indices = torch.empty(N, dtype=torch.int32, device=coords.device)
    _backend.morton3D(coords, N, indices)
    return indices


def morton3D_invert(indices):
    indices = indices.int().contiguous()
    N = indices.shape[0]
    coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    _backend.morton3D_invert(indices, N, coords)
    return coords


def packbits(grid, thresh, bitfield=None):
    grid = _f32(grid).contiguous()
    C, H3 = grid.shape[0], grid.shape[1]
    N = C * H3 // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    _backend.packbits(grid, N, thresh, bitfield)
    return bitfield


def morton3D_dilation(grid):
    grid = _f32(grid).contiguous()
    C, H3 = grid.shape[0], grid.shape[1]
    H = int(round(H3 ** (1 / 3)))
    out = torch.empty_like(grid)
    _backend.morton3D_dilation(grid, C, H, out)
    return out


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
               perturb=False, dt_gamma=0, max_steps=1024):
    rays_o = _f32(rays_o).contiguous().view(-1, 3)
    rays_d = _f32(rays_d).contiguous().view(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)  # note: pads a full `align` when M is already a multiple, as the reference does
    dev = rays_o.device
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    noises = torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb else torch.zeros(n_alive, dtype=torch.float32, device=dev)
    _backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                        density_bitfield, near, far, xyzs, dirs, deltas, noises)
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    _backend.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, _f32(sigmas).contiguous(), _f32(rgbs).contiguous(),
                            deltas, weights_sum, depth, image)
    return tuple()
