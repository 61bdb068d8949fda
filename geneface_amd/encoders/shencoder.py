"""Spherical-harmonics direction encoder (mirrors encoders/shencoder/sphere_harmonics.py:62-86)."""
import torch
import torch.nn as nn

from ..compat import _shencoder as _backend


class _sh_encode(torch.autograd.Function):
    """sphere_harmonics.py:14-58: forward (+ dy_dx when the directions need a gradient), backward through dy_dx."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.float().contiguous()
        B, input_dim = inputs.shape
        outputs = torch.empty(B, degree ** 2, dtype=torch.float32, device=inputs.device)
        dy_dx = torch.empty(B, input_dim * degree ** 2, dtype=torch.float32, device=inputs.device) if calc_grad_inputs else None
        _backend.sh_encode_forward(inputs, outputs, B, input_dim, degree, dy_dx)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = [B, input_dim, degree]
        return outputs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        B, input_dim, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _backend.sh_encode_backward(grad.float().contiguous(), inputs, B, input_dim, degree, dy_dx, grad_inputs)
        return grad_inputs, None, None


sh_encode = _sh_encode.apply


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
        out = sh_encode(inputs.reshape(-1, self.input_dim), self.degree, inputs.requires_grad)
        return out.reshape(prefix + [self.output_dim])
