"""Drop-in for the reference's pybind module `_freqencoder`
(modules/radnerfs/encoders/freqencoder/src/bindings.cpp)."""
import torch

from ..lib import check, current_stream, lib, ptr


def freq_encode_forward(inputs, B, D, deg, C, outputs):
    check(lib().gf_freq_encode_forward(ptr(inputs, torch.float32), B, D, deg, C, ptr(outputs, torch.float32),
                                       current_stream(inputs.device)))


def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
    check(lib().gf_freq_encode_backward(ptr(grad, torch.float32), ptr(outputs, torch.float32), B, D, deg, C, ptr(grad_inputs, torch.float32),
                                        current_stream(grad.device)))
