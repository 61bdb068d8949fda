"""ctypes face of the C oracle (TEST INFRASTRUCTURE ONLY -- never imported by geneface_amd/).

Each class below exposes the *pybind signatures* of one of the reference's four CUDA
extension modules (modules/radnerfs/raymarching/src/bindings.cpp:5-21,
encoders/gridencoder/src/bindings.cpp:5-9, encoders/shencoder/src/bindings.cpp,
encoders/freqencoder/src/bindings.cpp): at::Tensor arguments, outputs pre-allocated by the
caller, in-place writes, void return.  Tensors are CPU, contiguous.
"""
import ctypes as C
import os

import torch

from . import build as _build

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = _build.OUT if os.path.exists(_build.OUT) and os.path.getmtime(_build.OUT) >= os.path.getmtime(_build.SRC) \
            else _build.build()
        _LIB = C.CDLL(path)
        _LIB.orc_grid_encode_forward.restype = C.c_int
        _LIB.orc_sh_encode_forward.restype = C.c_int
    return _LIB


def _p(t, dtype=None):
    if t is None:
        return C.c_void_p(0)
    assert t.device.type == "cpu", "oracle kernels are CPU only"
    assert t.is_contiguous(), "oracle kernels need contiguous tensors"
    if dtype is not None:
        assert t.dtype == dtype, f"expected {dtype}, got {t.dtype}"
    return C.c_void_p(t.data_ptr())


def _f(x):
    return C.c_float(float(x))


def _u(x):
    return C.c_uint32(int(x))


class raymarching_face:
    """`_raymarching_face` (raymarching/src/raymarching.h:7-20), inference + grid-maintenance subset."""

    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        lib().orc_near_far_from_aabb(_p(rays_o, torch.float32), _p(rays_d, torch.float32), _p(aabb, torch.float32),
                                     _u(N), _f(min_near), _p(nears, torch.float32), _p(fars, torch.float32))

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, H, grid,
                   nears, fars, xyzs, dirs, deltas, noises):
        lib().orc_march_rays(_u(n_alive), _u(n_step), _p(rays_alive, torch.int32), _p(rays_t, torch.float32),
                             _p(rays_o, torch.float32), _p(rays_d, torch.float32), _f(bound), _f(dt_gamma),
                             _u(max_steps), _u(Cc), _u(H), _p(grid, torch.uint8), _p(nears, torch.float32),
                             _p(fars, torch.float32), _p(xyzs, torch.float32), _p(dirs, torch.float32),
                             _p(deltas, torch.float32), _p(noises, torch.float32))

    @staticmethod
    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        lib().orc_composite_rays(_u(n_alive), _u(n_step), _f(T_thresh), _p(rays_alive, torch.int32),
                                 _p(rays_t, torch.float32), _p(sigmas, torch.float32), _p(rgbs, torch.float32),
                                 _p(deltas, torch.float32), _p(weights_sum, torch.float32), _p(depth, torch.float32),
                                 _p(image, torch.float32))

    @staticmethod
    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        lib().orc_sph_from_ray(_p(rays_o, torch.float32), _p(rays_d, torch.float32), _f(radius), _u(N), _p(coords, torch.float32))

    @staticmethod
    def packbits(grid, N, density_thresh, bitfield):
        lib().orc_packbits(_p(grid, torch.float32), _u(N), _f(density_thresh), _p(bitfield, torch.uint8))

    @staticmethod
    def morton3D(coords, N, indices):
        lib().orc_morton3D(_p(coords, torch.int32), _u(N), _p(indices, torch.int32))

    @staticmethod
    def morton3D_invert(indices, N, coords):
        lib().orc_morton3D_invert(_p(indices, torch.int32), _u(N), _p(coords, torch.int32))

    @staticmethod
    def morton3D_dilation(grid, Cc, H, grid_dilation):
        lib().orc_morton3D_dilation(_p(grid, torch.float32), _u(Cc), _u(H), _p(grid_dilation, torch.float32))

    # ---- training tier (raymarching.h:13-18)
    @staticmethod
    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, Cc, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
        lib().orc_march_rays_train(_p(rays_o, torch.float32), _p(rays_d, torch.float32), _p(grid, torch.uint8), _f(bound), _f(dt_gamma),
                                   _u(max_steps), _u(N), _u(Cc), _u(H), _u(M), _p(nears, torch.float32), _p(fars, torch.float32),
                                   _p(xyzs, torch.float32), _p(dirs, torch.float32), _p(deltas, torch.float32), _p(rays, torch.int32),
                                   _p(counter, torch.int32), _p(noises, torch.float32))

    @staticmethod
    def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas, N, M, grad_rays_o, grad_rays_d):
        lib().orc_march_rays_train_backward(_p(grad_xyzs, torch.float32), _p(grad_dirs, torch.float32), _p(rays, torch.int32),
                                            _p(deltas, torch.float32), _u(N), _u(M), _p(grad_rays_o, torch.float32), _p(grad_rays_d, torch.float32))

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh, weights_sum, ambient_sum, depth, image):
        lib().orc_composite_rays_train_forward(_p(sigmas, torch.float32), _p(rgbs, torch.float32), _p(ambient, torch.float32),
                                               _p(deltas, torch.float32), _p(rays, torch.int32), _u(M), _u(N), _f(T_thresh),
                                               _p(weights_sum, torch.float32), _p(ambient_sum, torch.float32), _p(depth, torch.float32),
                                               _p(image, torch.float32))

    @staticmethod
    def composite_rays_train_backward(grad_weights_sum, grad_ambient_sum, grad_image, sigmas, rgbs, ambient, deltas, rays, weights_sum,
                                      ambient_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient):
        lib().orc_composite_rays_train_backward(_p(grad_weights_sum, torch.float32), _p(grad_ambient_sum, torch.float32),
                                                _p(grad_image, torch.float32), _p(sigmas, torch.float32), _p(rgbs, torch.float32),
                                                _p(ambient, torch.float32), _p(deltas, torch.float32), _p(rays, torch.int32),
                                                _p(weights_sum, torch.float32), _p(ambient_sum, torch.float32), _p(image, torch.float32),
                                                _u(M), _u(N), _f(T_thresh), _p(grad_sigmas, torch.float32), _p(grad_rgbs, torch.float32),
                                                _p(grad_ambient, torch.float32))


class gridencoder:
    """`_gridencoder` forward (encoders/gridencoder/src/gridencoder.h:11)."""

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, dy_dx, gridtype, align_corners, interp):
        rc = lib().orc_grid_encode_forward(_p(inputs, torch.float32), _p(embeddings, torch.float32),
                                           _p(offsets, torch.int32), _p(outputs, torch.float32), _u(B), _u(D), _u(Cc),
                                           _u(L), _f(S), _u(H), _p(dy_dx, torch.float32) if dy_dx is not None else _p(None),
                                           _u(gridtype), C.c_int(int(bool(align_corners))), _u(interp))
        if rc == -1:
            raise RuntimeError("GridEncoding: D must be 2..5")  # gridencoder.cu:398
        if rc == -2:
            raise RuntimeError("GridEncoding: C must be 1, 2, 4, or 8.")  # gridencoder.cu:381

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):
        rc = lib().orc_grid_encode_backward(_p(grad, torch.float32), _p(inputs, torch.float32), _p(embeddings, torch.float32), _p(offsets, torch.int32),
                                            _p(grad_embeddings, torch.float32), _u(B), _u(D), _u(Cc), _u(L), _f(S), _u(H),
                                            _p(dy_dx, torch.float32) if dy_dx is not None else _p(None),
                                            _p(grad_inputs, torch.float32) if grad_inputs is not None else _p(None),
                                            _u(gridtype), C.c_int(int(bool(align_corners))), _u(interp))
        if rc != 0:
            raise RuntimeError("GridEncoding backward: unsupported D / C")

    @staticmethod
    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, Cc, L, S, H, gridtype, align_corners):
        rc = lib().orc_grad_total_variation(_p(inputs, torch.float32), _p(embeddings, torch.float32), _p(grad, torch.float32),
                                            _p(offsets, torch.int32), _f(weight), _u(B), _u(D), _u(Cc), _u(L), _f(S), _u(H), _u(gridtype),
                                            C.c_int(int(bool(align_corners))))
        if rc != 0:
            raise RuntimeError("GridEncoding total variation: unsupported D / C")


class shencoder:
    """`_shencoder` forward (encoders/shencoder/src/shencoder.h:9)."""

    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, Cc, dy_dx):
        rc = lib().orc_sh_encode_forward(_p(inputs, torch.float32), _p(outputs, torch.float32), _u(B), _u(D), _u(Cc))
        if rc != 0:
            raise RuntimeError("SH oracle: D must be 3 and degree in [1,8]")
        if dy_dx is not None:
            lib().orc_sh_encode_dy_dx(_p(inputs, torch.float32), _p(dy_dx, torch.float32), _u(B), _u(Cc))

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, Cc, dy_dx, grad_inputs):
        lib().orc_sh_encode_backward(_p(grad, torch.float32), _p(dy_dx, torch.float32), _u(B), _u(Cc), _p(grad_inputs, torch.float32))


class freqencoder:
    """`_freqencoder` forward (encoders/freqencoder/src/freqencoder.h:7)."""

    @staticmethod
    def freq_encode_forward(inputs, B, D, deg, Cc, outputs):
        lib().orc_freq_encode_forward(_p(inputs, torch.float32), _u(B), _u(D), _u(deg), _u(Cc), _p(outputs, torch.float32))

    @staticmethod
    def freq_encode_backward(grad, outputs, B, D, deg, Cc, grad_inputs):
        lib().orc_freq_encode_backward(_p(grad, torch.float32), _p(outputs, torch.float32), _u(B), _u(D), _u(deg), _u(Cc), _p(grad_inputs, torch.float32))


def grid_level_meta(L, S, H):
    scale = torch.empty(L, dtype=torch.float32)
    res = torch.empty(L, dtype=torch.int32)
    lib().orc_grid_level_meta(_u(L), _f(S), _u(H), _p(scale), _p(res))
    return scale, res
