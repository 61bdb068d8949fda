"""Drop-in for the reference's pybind module `_gridencoder`
(modules/radnerfs/encoders/gridencoder/src/bindings.cpp:5-9).

dtype: the reference dispatches on the table's scalar type (AT_DISPATCH_FLOATING_TYPES_AND_HALF) and its wrapper hands the backend HALF
tables, outputs, dy_dx and gradients whenever autocast is on (grid.py:41-44, the May config's `amp: true` on the training and viewer
paths); inputs stay float.  The library computes in fp32: a half TABLE is read as it is (gf_grid_encode_forward_f16 widens each row on load;
the backward never reads the table), the B-sized half outputs / dy_dx / gradients are converted at this seam: same call, same in-place
semantics, fp32 arithmetic inside (at least as accurate as the half kernels; the table gradient in particular is accumulated in fp32
instead of with half atomics, then added into the caller's half buffer once).
"""
import torch

from ..lib import check, current_stream, lib, ptr

_F = torch.float32


def _f32(t):
    return t if t is None or t.dtype == _F else t.float()


def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
    if embeddings.dtype not in (_F, torch.float16):
        raise RuntimeError(f"grid_encode_forward: embeddings must be float32 or float16, got {embeddings.dtype}")
    half = embeddings.dtype != _F
    out = torch.empty(outputs.shape, dtype=_F, device=outputs.device) if half else outputs
    dx = (torch.empty(dy_dx.shape, dtype=_F, device=dy_dx.device) if half else dy_dx) if dy_dx is not None else None
    x = _f32(inputs).contiguous()     # named: a converted copy must outlive the launch
    if half and C in (2, 4, 8):
        # the half table is read as it is (gf_grid_encode_forward_f16: rows widened on load) -- until round 4 the whole table (6.9 MB for the
        # position grid) was converted to fp32 on every call, on top of the reference's own fp32 -> half cast of it (grid.py:43)
        check(lib().gf_grid_encode_forward_f16(ptr(x, _F), ptr(embeddings, torch.float16), ptr(offsets, torch.int32), ptr(out, _F),
                                               B, D, C, L, float(S), H, ptr(dx, _F, allow_none=True), gridtype, int(bool(align_corners)), interp,
                                               current_stream(inputs.device)))
    else:
        e = _f32(embeddings)
        check(lib().gf_grid_encode_forward(ptr(x, _F), ptr(e, _F), ptr(offsets, torch.int32), ptr(out, _F),
                                           B, D, C, L, float(S), H, ptr(dx, _F, allow_none=True), gridtype, int(bool(align_corners)), interp,
                                           current_stream(inputs.device)))
    if half:
        outputs.copy_(out)
        if dy_dx is not None:
            dy_dx.copy_(dx)


def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners, interp):
    half = grad_embeddings.dtype != _F
    g_emb = torch.zeros(grad_embeddings.shape, dtype=_F, device=grad_embeddings.device) if half else grad_embeddings
    g_in = (torch.zeros(grad_inputs.shape, dtype=_F, device=grad_inputs.device) if grad_inputs.dtype != _F else grad_inputs) \
        if grad_inputs is not None else None
    g, x, dx = _f32(grad).contiguous(), _f32(inputs).contiguous(), _f32(dy_dx)
    # (the table gradient does not depend on the table's values -- gridencoder.cu:248-341 never reads `grid` -- so the half table the
    # reference saved is not converted, or even passed: the C entry ignores that argument)
    check(lib().gf_grid_encode_backward(ptr(g, _F), ptr(x, _F), None,
                                        ptr(offsets, torch.int32), ptr(g_emb, _F), B, D, C, L, float(S), H, ptr(dx, _F, allow_none=True),
                                        ptr(g_in, _F, allow_none=True), gridtype, int(bool(align_corners)), interp, current_stream(grad.device)))
    if half:
        grad_embeddings.add_(g_emb)           # the reference accumulates into the caller's (zero-filled) buffer
    if grad_inputs is not None and g_in is not grad_inputs:
        grad_inputs.copy_(g_in)


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
    half = grad.dtype != _F
    g = torch.zeros(grad.shape, dtype=_F, device=grad.device) if half else grad
    x, e = _f32(inputs).contiguous(), _f32(embeddings)
    check(lib().gf_grad_total_variation(ptr(x, _F), ptr(e, _F), ptr(g, _F), ptr(offsets, torch.int32),
                                        float(weight), B, D, C, L, float(S), H, gridtype, int(align_corners), current_stream(inputs.device)))
    if half:
        grad.add_(g)
