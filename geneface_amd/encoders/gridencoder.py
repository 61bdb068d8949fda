"""Multi-resolution tiled / hashed grid encoder module.

Mirrors modules/radnerfs/encoders/gridencoder/grid.py of the reference: constructor arguments,
parameter/buffer names (`embeddings`, `offsets`), table sizing (:113-131), init (:138-140) and the
[-bound, bound] -> [0, 1] input remap (:149) are the same, so reference checkpoints load.
The lookup and its backward (table scatter + input gradient) run in libgeneface_hip.so; the forward writes [B, L*C] directly.
"""
import numpy as np
import torch
import torch.nn as nn

from ..lib import check, current_stream, lib, ptr

_gridtype_to_id = {"hash": 0, "tiled": 1}
_interp_to_id = {"linear": 0, "smoothstep": 1}


def per_level_scale_for(desired_resolution, base_resolution, num_levels):
    return np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))


def grid_offsets(input_dim, num_levels=16, base_resolution=16, log2_hashmap_size=19, desired_resolution=None,
                 per_level_scale=2, align_corners=False) -> np.ndarray:
    """int32 [L+1] row offsets of each level's table (grid.py:113-131)."""
    if desired_resolution is not None:
        per_level_scale = per_level_scale_for(desired_resolution, base_resolution, num_levels)
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


class _grid_encode(torch.autograd.Function):
    """grid.py:24-90.  Forward: [B, L*C] written directly (the reference writes [L,B,C] and permutes); backward: the table gradient
    is scattered with f32 atomics, the input gradient goes through dy_dx when the inputs require one."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # the reference keeps fp16 tables under autocast (grid.py:43); fp32 here
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False,
                interpolation=0):
        inputs = inputs.float().contiguous()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        outputs = torch.empty(B, L * C, device=inputs.device, dtype=torch.float32)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=torch.float32) if calc_grad_inputs else None
        check(lib().gf_grid_encode_forward_blc(ptr(inputs, torch.float32), ptr(embeddings, torch.float32), ptr(offsets, torch.int32),
                                               ptr(outputs), B, D, C, L, S, int(base_resolution), ptr(dy_dx, torch.float32, allow_none=True),
                                               gridtype, int(bool(align_corners)), interpolation, current_stream(inputs.device)))
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, int(base_resolution), gridtype, interpolation]
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation = ctx.dims
        grad = grad.float().view(B, L, C).permute(1, 0, 2).contiguous()   # [L, B, C], what the kernel indexes (grid.py:75)
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs) if dy_dx is not None else None
        check(lib().gf_grid_encode_backward(ptr(grad, torch.float32), ptr(inputs, torch.float32), ptr(embeddings, torch.float32),
                                            ptr(offsets, torch.int32), ptr(grad_embeddings, torch.float32), B, D, C, L, S, H,
                                            ptr(dy_dx, torch.float32, allow_none=True), ptr(grad_inputs, torch.float32, allow_none=True), gridtype,
                                            int(bool(ctx.align_corners)), interpolation, current_stream(grad.device)))
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0):
    """inputs [B,D] in [0,1] -> [B, L*C] (the value `_grid_encode.apply` returns in the reference, grid.py:27-63)."""
    if not (torch.is_grad_enabled() and (embeddings.requires_grad or inputs.requires_grad)):
        calc_grad_inputs = False
    return _grid_encode.apply(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype, align_corners, interpolation)


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False, interpolation="linear"):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = per_level_scale_for(desired_resolution, base_resolution, num_levels)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.log2_hashmap_size, self.base_resolution = per_level_scale, log2_hashmap_size, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype, self.gridtype_id = gridtype, _gridtype_to_id[gridtype]
        self.interpolation, self.interp_id = interpolation, _interp_to_id[interpolation]
        self.align_corners = align_corners
        offsets = grid_offsets(input_dim, num_levels, base_resolution, log2_hashmap_size, None, per_level_scale, align_corners)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.n_params = int(offsets[-1]) * level_dim
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners} interpolation={self.interpolation}")

    def forward(self, inputs, bound=1):
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        out = grid_encode(inputs.view(-1, self.input_dim), self.embeddings, self.offsets, self.per_level_scale,
                          self.base_resolution, inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id)
        return out.view(prefix + [self.output_dim])
