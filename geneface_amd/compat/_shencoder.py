"""Drop-in for the reference's pybind module `_shencoder`
(modules/radnerfs/encoders/shencoder/src/bindings.cpp)."""
import torch

from ..lib import check, current_stream, lib, ptr


def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
    check(lib().gf_sh_encode_forward(ptr(inputs, torch.float32), ptr(outputs, torch.float32), B, D, C,
                                     ptr(dy_dx, torch.float32, allow_none=True), current_stream(inputs.device)))


def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
    check(lib().gf_sh_encode_backward(ptr(grad, torch.float32), ptr(inputs, torch.float32), B, D, C, ptr(dy_dx, torch.float32),
                                      ptr(grad_inputs, torch.float32), current_stream(grad.device)))
