"""Drop-in for the reference's pybind module `_gridencoder`
(modules/radnerfs/encoders/gridencoder/src/bindings.cpp:5-9)."""
import torch

from ..lib import check, current_stream, lib, ptr


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    if embeddings.dtype != torch.float32:
        raise RuntimeError("grid_encode_forward: only float32 embeddings are built (reference inference is fp32)")
    check(lib().gf_grid_encode_forward(ptr(inputs, torch.float32), ptr(embeddings, torch.float32), ptr(offsets, torch.int32),
                                       ptr(outputs, torch.float32), B, D, C, L, float(S), H,
                                       ptr(dy_dx, torch.float32, allow_none=True), gridtype, int(bool(align_corners)), interp,
                                       current_stream(inputs.device)))


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):
    check(lib().gf_grid_encode_backward(ptr(grad, torch.float32), ptr(inputs, torch.float32), ptr(embeddings, torch.float32), ptr(offsets, torch.int32),
                                        ptr(grad_embeddings, torch.float32), B, D, C, L, float(S), H, ptr(dy_dx, torch.float32, allow_none=True),
                                        ptr(grad_inputs, torch.float32, allow_none=True), gridtype, int(bool(align_corners)), interp,
                                        current_stream(grad.device)))


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
    check(lib().gf_grad_total_variation(ptr(inputs, torch.float32), ptr(embeddings, torch.float32), ptr(grad, torch.float32),
                                        ptr(offsets, torch.int32), float(weight), B, D, C, L, float(S), H, gridtype, int(align_corners),
                                        current_stream(inputs.device)))
