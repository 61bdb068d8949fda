"""Stand-alone timing of gf_grid_encode_backward on ray-like points (the training batch shape): python tools/bench_grid_backward.py"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geneface_amd.compat import _gridencoder as GE
from geneface_amd.encoders.gridencoder import grid_offsets

dev = "cuda:0"
B = 1 << 20
g = torch.Generator().manual_seed(1)
for D in (3, 2):
    off = torch.from_numpy(grid_offsets(D, 16, 16, 16, 2048)).to(dev)
    rays, per = B // 16, 16
    o = torch.rand(rays, 1, D, generator=g) * 0.6 + 0.2
    d = torch.randn(rays, 1, D, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    t = (torch.arange(per).view(1, per, 1) * 0.0137)
    x = (o + d * t).clamp(0, 1).reshape(B, D).contiguous().to(dev)
    grad = torch.randn(16, B, 2, generator=g).to(dev)
    emb = torch.zeros(int(off[-1]), 2, device=dev)
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    for it in range(3):
        ge = torch.zeros_like(emb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        GE.grid_encode_backward(grad, x, emb, off, ge, B, D, 2, 16, S, 16, None, None, 1, False, 0)
        e1.record()
        torch.cuda.synchronize()
    print(f"D={D} lib={os.environ.get('GF_HIP_LIB','base')[-16:]}: {e0.elapsed_time(e1):.3f} ms  sum={float(ge.double().sum()):.3f}")
