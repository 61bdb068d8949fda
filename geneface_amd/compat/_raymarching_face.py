"""Drop-in for the reference's pybind module `_raymarching_face`
(modules/radnerfs/raymarching/src/bindings.cpp:5-21): same function names, same positional
arguments (at::Tensor -> torch.Tensor, outputs pre-allocated by the caller, in-place, returns None),
executed by libgeneface_hip.so on the current HIP stream."""
import torch

from ..lib import check, current_stream, lib, ptr

_F, _I, _U8 = torch.float32, torch.int32, torch.uint8


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    check(lib().gf_near_far_from_aabb(ptr(rays_o, _F), ptr(rays_d, _F), ptr(aabb, _F), N, min_near, ptr(nears, _F),
                                      ptr(fars, _F), current_stream(rays_o.device)))


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars,
               xyzs, dirs, deltas, noises):
    check(lib().gf_march_rays(n_alive, n_step, ptr(rays_alive, _I), ptr(rays_t, _F), ptr(rays_o, _F), ptr(rays_d, _F),
                              bound, dt_gamma, max_steps, C, H, ptr(grid, _U8), ptr(nears, _F), ptr(fars, _F),
                              ptr(xyzs, _F), ptr(dirs, _F), ptr(deltas, _F), ptr(noises, _F),
                              current_stream(rays_o.device)))


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    check(lib().gf_composite_rays(n_alive, n_step, T_thresh, ptr(rays_alive, _I), ptr(rays_t, _F), ptr(sigmas, _F),
                                  ptr(rgbs, _F), ptr(deltas, _F), ptr(weights_sum, _F), ptr(depth, _F), ptr(image, _F),
                                  current_stream(image.device)))


def packbits(grid, N, density_thresh, bitfield):
    check(lib().gf_packbits(ptr(grid, _F), N, density_thresh, ptr(bitfield, _U8), current_stream(grid.device)))


def morton3D(coords, N, indices):
    check(lib().gf_morton3D(ptr(coords, _I), N, ptr(indices, _I), current_stream(coords.device)))


def morton3D_invert(indices, N, coords):
    check(lib().gf_morton3D_invert(ptr(indices, _I), N, ptr(coords, _I), current_stream(indices.device)))


def morton3D_dilation(grid, C, H, grid_dilation):
    check(lib().gf_morton3D_dilation(ptr(grid, _F), C, H, ptr(grid_dilation, _F), current_stream(grid.device)))


_ws_cache = {}


def _train_ws(N, device):
    """scratch for the three-pass march_rays_train (per-ray counts, prefixes, block totals)"""
    key = (device, N)
    if key not in _ws_cache:
        _ws_cache.clear()
        _ws_cache[key] = torch.empty(lib().gf_march_rays_train_workspace_bytes(N), dtype=torch.uint8, device=device)
    return _ws_cache[key]


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
    check(lib().gf_march_rays_train(ptr(rays_o, _F), ptr(rays_d, _F), ptr(grid, _U8), bound, dt_gamma, max_steps, N, C, H, M, ptr(nears, _F),
                                    ptr(fars, _F), ptr(xyzs, _F), ptr(dirs, _F), ptr(deltas, _F), ptr(rays, _I), ptr(counter, _I), ptr(noises, _F),
                                    _train_ws(N, rays_o.device).data_ptr(), current_stream(rays_o.device)))


def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas, N, M, grad_rays_o, grad_rays_d):
    check(lib().gf_march_rays_train_backward(ptr(grad_xyzs, _F), ptr(grad_dirs, _F), ptr(rays, _I), ptr(deltas, _F), N, M, ptr(grad_rays_o, _F),
                                             ptr(grad_rays_d, _F), current_stream(rays.device)))


def composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh, weights_sum, ambient_sum, depth, image):
    check(lib().gf_composite_rays_train_forward(ptr(sigmas, _F), ptr(rgbs, _F), ptr(ambient, _F), ptr(deltas, _F), ptr(rays, _I), M, N, T_thresh,
                                                ptr(weights_sum, _F), ptr(ambient_sum, _F), ptr(depth, _F), ptr(image, _F),
                                                current_stream(sigmas.device)))


def composite_rays_train_backward(grad_weights_sum, grad_ambient_sum, grad_image, sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum,
                                  image, M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient):
    check(lib().gf_composite_rays_train_backward(ptr(grad_weights_sum, _F), ptr(grad_ambient_sum, _F), ptr(grad_image, _F), ptr(sigmas, _F),
                                                 ptr(rgbs, _F), ptr(ambient, _F), ptr(deltas, _F), ptr(rays, _I), ptr(weights_sum, _F),
                                                 ptr(ambient_sum, _F), ptr(image, _F), M, N, T_thresh, ptr(grad_sigmas, _F), ptr(grad_rgbs, _F),
                                                 ptr(grad_ambient, _F), current_stream(sigmas.device)))


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    check(lib().gf_sph_from_ray(ptr(rays_o, _F), ptr(rays_d, _F), radius, N, ptr(coords, _F), current_stream(rays_o.device)))
