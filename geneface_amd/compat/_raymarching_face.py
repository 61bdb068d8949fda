"""Drop-in for the reference's pybind module `_raymarching_face`
(modules/radnerfs/raymarching/src/bindings.cpp:5-21): same function names, same positional
arguments (at::Tensor -> torch.Tensor, outputs pre-allocated by the caller, in-place, returns None),
executed by libgeneface_hip.so on the current HIP stream.  Training-only exports raise."""
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


def _training_only(name):
    def f(*a, **k):
        raise NotImplementedError(f"_raymarching_face.{name}: training path, outside this round's scope (SURVEY.md 8f-2)")
    f.__name__ = name
    return f


sph_from_ray = _training_only("sph_from_ray")  # no call site in GeneFace
march_rays_train = _training_only("march_rays_train")
march_rays_train_backward = _training_only("march_rays_train_backward")
composite_rays_train_forward = _training_only("composite_rays_train_forward")
composite_rays_train_backward = _training_only("composite_rays_train_backward")
