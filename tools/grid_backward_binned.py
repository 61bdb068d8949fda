"""The table scatter as the training node calls it (gf_grid_encode_backward_binned, per-level maxima supplied) against the unbinned entry, on a
training-shaped batch: ms per call, 3-D and 2-D tables of the May config.  [GF_HIP_LIB=...] python tools/grid_backward_binned.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geneface_amd.encoders.gridencoder import grid_offsets
from geneface_amd.lib import check, current_stream, lib

dev = torch.device("cuda:0")
L_ = lib()
B = 1 << 20
g = torch.Generator().manual_seed(1)
out = {"lib": os.path.basename(os.environ.get("GF_HIP_LIB", "libgeneface_hip.so")), "B": B,
       "env": {k: v for k, v in os.environ.items() if k.startswith("GF_GB_")}}
for D in (3, 2):
    off_h = grid_offsets(D, 16, 16, 16, 2048)
    off = torch.from_numpy(off_h).to(dev)
    rays, per = B // 16, 16
    o = torch.rand(rays, 1, D, generator=g) * 0.6 + 0.2
    d = torch.randn(rays, 1, D, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    t = (torch.arange(per).view(1, per, 1) * 0.0137)
    x = (o + d * t).clamp(0, 1).reshape(B, D).contiguous().to(dev)
    grad = torch.randn(16, B, 2, generator=g).to(dev)
    lmax = grad.abs().amax(dim=(1, 2)).contiguous().view(torch.int32)
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    ws = torch.empty(L_.gf_grid_backward_ws_bytes(B, 16), dtype=torch.uint8, device=dev)
    st = current_stream(dev)
    rec = {}
    for name in ("binned", "unbinned"):
        best = 1e9
        for it in range(5):
            tab = torch.zeros(int(off_h[-1]), 2, device=dev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == "binned":
                check(L_.gf_grid_encode_backward_binned(grad.data_ptr(), x.data_ptr(), off.data_ptr(), tab.data_ptr(), B, D, 2, 16, S, 16, 1, 0, 0,
                                                        lmax.data_ptr(), ws.data_ptr(), st))
            else:
                check(L_.gf_grid_encode_backward_scaled(grad.data_ptr(), x.data_ptr(), off.data_ptr(), tab.data_ptr(), B, D, 2, 16, S, 16, 1, 0, 0,
                                                        lmax.data_ptr(), st))
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        rec[name + "_ms"] = round(best, 4)
        rec[name + "_sum"] = float(tab.double().sum())
    out[f"D{D}"] = rec
print(json.dumps(out))
