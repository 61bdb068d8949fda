"""Spherical-harmonics direction encoder (mirrors encoders/shencoder/sphere_harmonics.py:62-86)."""
import torch
import torch.nn as nn

from ..compat import _shencoder as _backend


def sh_encode(inputs, degree, calc_grad_inputs=False):
    if calc_grad_inputs:
        raise NotImplementedError("sh_encode: input gradients belong to the training path (SURVEY.md 8f-2)")
    inputs = inputs.float().contiguous()
    B, input_dim = inputs.shape
    outputs = torch.empty(B, degree ** 2, dtype=torch.float32, device=inputs.device)
    _backend.sh_encode_forward(inputs, outputs, B, input_dim, degree, None)
    return outputs


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        assert input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < degree <= 8, "SH encoder only supports degree in [1, 8]"
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix = list(inputs.shape[:-1])
        out = sh_encode(inputs.reshape(-1, self.input_dim), self.degree, False)
        return out.reshape(prefix + [self.output_dim])
