"""Where the table scatter's time goes, level by level: gf_grid_encode_backward on a training-shaped batch (1 M ray-like points) with the
gradient of ONE level non-zero at a time (a level whose gradient is all zero returns at once), 3-D and 2-D tables of the May config.

    [GF_HIP_LIB=.../libgeneface_hip_<variant>.so] python tools/grid_backward_levels.py"""
import hashlib
import json
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
out = {"lib": os.path.basename(os.environ.get("GF_HIP_LIB", "libgeneface_hip.so")), "B": B}
for D in (3, 2):
    off_h = grid_offsets(D, 16, 16, 16, 2048)
    off = torch.from_numpy(off_h).to(dev)
    rays, per = B // 16, 16
    o = torch.rand(rays, 1, D, generator=g) * 0.6 + 0.2
    d = torch.randn(rays, 1, D, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    t = (torch.arange(per).view(1, per, 1) * 0.0137)
    x = (o + d * t).clamp(0, 1).reshape(B, D).contiguous().to(dev)
    full = torch.randn(16, B, 2, generator=g).to(dev)
    emb = torch.zeros(int(off[-1]), 2, device=dev)
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))

    def timed(grad):
        best = 1e9
        for it in range(4):
            ge = torch.zeros_like(emb)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            GE.grid_encode_backward(grad, x, emb, off, ge, B, D, 2, 16, S, 16, None, None, 1, False, 0)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best, hashlib.md5(ge.cpu().numpy().tobytes()).hexdigest()[:12] + f" {float(ge.double().sum()):.6f}"
    rec = {"rows_per_level": [int(off_h[l + 1] - off_h[l]) for l in range(16)]}
    rec["all_levels_ms"], rec["checksum"] = timed(full)
    rec["no_level_ms"], _ = timed(torch.zeros_like(full))          # the max pass + sixteen empty launches' worth of workgroups
    per_level = []
    for l in range(16):
        gr = torch.zeros_like(full)
        gr[l] = full[l]
        per_level.append(round(timed(gr)[0], 4))
    rec["one_level_ms"] = per_level
    out[f"D{D}"] = rec
print(json.dumps(out))
