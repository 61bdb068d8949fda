"""NeRF frequency encoder (mirrors encoders/freqencoder/freq.py:58-76)."""
import torch
import torch.nn as nn

from ..compat import _freqencoder as _backend


class _freq_encode(torch.autograd.Function):
    """freq.py:15-53: the backward reads sin / cos back from the forward outputs."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, output_dim):
        inputs = inputs.float().contiguous()
        B, input_dim = inputs.shape
        outputs = torch.empty(B, output_dim, dtype=torch.float32, device=inputs.device)
        _backend.freq_encode_forward(inputs, B, input_dim, degree, output_dim, outputs)
        ctx.save_for_backward(inputs, outputs)
        ctx.dims = [B, input_dim, degree, output_dim]
        return outputs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, outputs = ctx.saved_tensors
        B, input_dim, degree, output_dim = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _backend.freq_encode_backward(grad.float().contiguous(), outputs, B, input_dim, degree, output_dim, grad_inputs)
        return grad_inputs, None, None


freq_encode = _freq_encode.apply


class FreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def __repr__(self):
        return f"FreqEncoder: input_dim={self.input_dim} degree={self.degree} output_dim={self.output_dim}"

    def forward(self, inputs, **kwargs):
        prefix = list(inputs.shape[:-1])
        out = freq_encode(inputs.reshape(-1, self.input_dim), self.degree, self.output_dim)
        return out.reshape(prefix + [self.output_dim])
