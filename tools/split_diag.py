#!/usr/bin/env python
"""Where does the split tier deviate from the oracle by more than the strict tolerance?  For a few bench frames: the oracle's iteration trace
(n_alive, n_step), the replayed schedule / budget of the fp32 and split tiers, the pixels beyond 1e-4 and what kind of rays they are."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from geneface_amd import hparams as HP, synthetic as S
from geneface_amd.fused import frame_stats
from geneface_amd.radnerf_torso import RADNeRFTorso
from oracle import radnerf_ref as R

frames = [int(a) for a in sys.argv[1:]] or [1, 37]
hp = HP.may_hparams(True)
sd = S.make_state_dict(hp, True)
seq = S.make_sequence(110, 512, 512, hp)
dev = "cuda:0"
models = {}
for prec in ("fp32", "split"):
    m = RADNeRFTorso(hp); m.load_state_dict(sd, strict=True); m.render_impl, m.render_precision = "fused", prec
    models[prec] = m.to(dev).eval()
H = W = 512
bgc = R.get_bg_coords(H, W); bg = torch.from_numpy(seq["bg_img"]).view(1, -1, 3)
for i in frames:
    pose = torch.from_numpy(seq["poses"][i:i + 1])
    ro, rd = R.get_rays(pose, seq["intrinsics"], H, W)
    cond, p6 = torch.from_numpy(seq["cond_wins"][i]), R.convert_poses(pose)
    trace = []
    ref = R.render(sd, hp, ro, rd, cond, bgc, p6, bg, torso=True, trace=trace)
    print(f"frame {i}: oracle schedule", [(t["n_alive"], t["n_step"]) for t in trace])
    outs = {}
    for prec, m in models.items():
        with torch.no_grad():
            out = m.render(ro.to(dev), rd.to(dev), cond.to(dev), bgc.to(dev), p6.to(dev), index=0, staged=False, bg_color=bg.to(dev), perturb=False, force_all_rays=True, **hp)
        fs = frame_stats(m.last_ctrl, H * W, hp["max_steps"])
        err = (out["rgb_map"].cpu().reshape(-1, 3) - ref["rgb_map"].reshape(-1, 3)).abs().max(dim=1).values
        bad = (err > 1e-4).nonzero().flatten()
        outs[prec] = out
        print(f"  {prec}: budget {fs['budget']} device {fs['budget_device']} schedule {fs['schedule']} max err {float(err.max()):.3e} bad pixels {len(bad)}")
        if len(bad):
            ws = ref["weights_sum"].reshape(-1) if "weights_sum" in ref else None
            d = (out["depth_map"].cpu().reshape(-1) - ref["depth_map"].reshape(-1)).abs()
            print("    bad idx", bad[:12].tolist(), "err", [f"{float(e):.2e}" for e in err[bad[:12]]], "depth err", [f"{float(e):.2e}" for e in d[bad[:12]]])
    e2 = (outs["fp32"]["rgb_map"] - outs["split"]["rgb_map"]).abs().reshape(-1, 3).max(dim=1).values
    print(f"  fp32 vs split: max {float(e2.max()):.3e}, pixels > 1e-4: {int((e2 > 1e-4).sum())}, > 1e-5: {int((e2 > 1e-5).sum())}")
