"""Test infrastructure (GPU box): the training-step rate of tools/bench_train.py with THE REFERENCE'S OWN KERNELS (oracle/_ref)
swapped in under the same host code, so the product's training tier has a same-GPU baseline beside the one published figure.
Patched seams: geneface_amd.raymarching / shencoder / freqencoder `_backend` (the reference's pybind signatures), and the grid
encoder's autograd Function (the reference writes [L,B,C] and permutes, grid.py:24-90).  Usage:
    python tests/train_rate_reference.py [--steps K --warmup W]     -> one JSON line, like tools/bench_train.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]

from oracle import ref_kernels   # noqa: E402

RM, GE, SH, FQ = ref_kernels.load("fast")


class _RefGridEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False,
                interpolation=0):
        inputs = inputs.float().contiguous()
        B, D = inputs.shape
        L, C, S = offsets.shape[0] - 1, embeddings.shape[1], float(np.log2(per_level_scale))
        outputs = torch.empty(L, B, C, device=inputs.device)
        dy_dx = torch.empty(B, L * D * C, device=inputs.device) if calc_grad_inputs else None
        GE.grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, int(base_resolution), dy_dx, gridtype, align_corners, interpolation)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = [B, D, C, L, S, int(base_resolution), gridtype, interpolation, align_corners]
        return outputs.permute(1, 0, 2).reshape(B, L * C)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype, interpolation, align_corners = ctx.dims
        grad = grad.float().view(B, L, C).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs) if dy_dx is not None else None
        GE.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, gridtype, align_corners,
                                interpolation)
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None


def main():
    import geneface_amd.encoders.freqencoder as fe
    import geneface_amd.encoders.gridencoder as ge
    import geneface_amd.encoders.shencoder as she
    import geneface_amd.raymarching as rmod
    rmod._backend, she._backend, fe._backend = RM, SH, FQ
    ge._grid_encode = _RefGridEncode
    # the reference's structure: the field is a torch op graph over its encoders (since round 3 the product's training field is ONE fused
    # autograd node that never reaches the patched seams -- with it this script measured the product against itself), and the density-grid
    # refresh queries that graph block by block
    import geneface_amd.radnerf as rn
    import geneface_amd.renderer as rr
    rn.RADNeRF.field_impl = "ops"
    rr.NeRFRenderer._pick_impl = lambda self, impl, perturb, max_steps: "ops"
    import bench_train
    bench_train.main()


if __name__ == "__main__":
    main()
