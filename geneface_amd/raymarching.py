"""Host-side ray-marching operators (inference + occupancy-grid maintenance).

Python-visible behaviour mirrors modules/radnerfs/raymarching/raymarching.py of the reference
(`near_far_from_aabb` :18-50, `morton3D` :85-110, `morton3D_invert` :112-135, `packbits` :137-163,
`morton3D_dilation` :165-183, `march_rays` :347-398, `composite_rays` :401-423): same argument
order, same allocation/zero-fill/padding rules, inference ops forward only (the reference wraps them in
autograd.Functions whose backward is None for these ops); the training pair `march_rays_train` :185-283 /
`composite_rays_train` :286-342 are autograd Functions with the reference's backward semantics.  The arithmetic runs in
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
               perturb=False, dt_gamma=0, max_steps=1024, noises=None):
    rays_o = _f32(rays_o).contiguous().view(-1, 3)
    rays_d = _f32(rays_d).contiguous().view(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)  # note: pads a full `align` when M is already a multiple, as the reference does
    dev = rays_o.device
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    if noises is not None and perturb:      # caller-supplied draws (extension; the reference always draws here, raymarching.py:395-398)
        noises = noises.to(device=dev, dtype=torch.float32).contiguous()
    else:
        noises = torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb else torch.zeros(n_alive, dtype=torch.float32, device=dev)
    _backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                        density_bitfield, near, far, xyzs, dirs, deltas, noises)
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    _backend.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, _f32(sigmas).contiguous(), _f32(rgbs).contiguous(),
                            deltas, weights_sum, depth, image)
    return tuple()


# ----------------------------------------------------------------------------------------------------------------------
# training functions (raymarching.py:185-342)
# ----------------------------------------------------------------------------------------------------------------------
class _march_rays_train(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        """-> xyzs [M,3], dirs [M,3], deltas [M,2] (dt, t), rays i32 [N,3] (ray, point offset, point count); see raymarching.py:189-208.
        Point offsets follow ray order (deterministic; the reference's atomics give an arbitrary order).  When `mean_count` underestimates
        the batch, the rays that lose their samples are the tail of that order; under perturb=True the order starts at a block chosen by
        the call's own jitter, so the victims move from step to step as in the reference instead of always being the last rows of a
        raster-ordered batch."""
        rays_o = _f32(rays_o).contiguous().view(-1, 3)
        rays_d = _f32(rays_d).contiguous().view(-1, 3)
        density_bitfield = density_bitfield.contiguous()
        N = rays_o.shape[0]
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count
        dev = rays_o.device
        xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
        noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else torch.zeros(N, dtype=torch.float32, device=dev)
        _backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M, _f32(nears).contiguous(),
                                  _f32(fars).contiguous(), xyzs, dirs, deltas, rays, step_counter, noises)
        if force_all_rays or mean_count <= 0:
            m = int(step_counter[0].item())  # D2H copy, as in the reference
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        ctx.save_for_backward(rays, deltas)
        ctx.mark_non_differentiable(rays)
        return xyzs, dirs, deltas, rays

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_xyzs, grad_dirs, grad_deltas, grad_rays):
        rays, deltas = ctx.saved_tensors
        N, M = rays.shape[0], grad_xyzs.shape[0]
        grad_rays_o = torch.zeros(N, 3, dtype=torch.float32, device=rays.device)
        grad_rays_d = torch.zeros(N, 3, dtype=torch.float32, device=rays.device)
        _backend.march_rays_train_backward(_f32(grad_xyzs).contiguous(), _f32(grad_dirs).contiguous(), rays, deltas.contiguous(), N, M,
                                           grad_rays_o, grad_rays_d)
        return (grad_rays_o, grad_rays_d) + (None,) * 13


march_rays_train = _march_rays_train.apply


class _composite_rays_train(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, ambient, deltas, rays, T_thresh=1e-4):
        """-> weights_sum [N], ambient_sum [N], depth [N], image [N,3] (raymarching.py:289-302)."""
        sigmas, rgbs, ambient = _f32(sigmas).contiguous(), _f32(rgbs).contiguous(), _f32(ambient).contiguous()
        deltas = _f32(deltas).contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        ambient_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        _backend.composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh, weights_sum, ambient_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        return weights_sum, ambient_sum, depth, image

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights_sum, grad_ambient_sum, grad_depth, grad_image):
        # grad_depth is not propagated, as in the reference (raymarching.py:322)
        sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        grad_sigmas, grad_rgbs, grad_ambient = torch.zeros_like(sigmas), torch.zeros_like(rgbs), torch.zeros_like(ambient)
        _backend.composite_rays_train_backward(_f32(grad_weights_sum).contiguous(), _f32(grad_ambient_sum).contiguous(),
                                               _f32(grad_image).contiguous(), sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, image,
                                               M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient)
        return grad_sigmas, grad_rgbs, grad_ambient, None, None, None


composite_rays_train = _composite_rays_train.apply
