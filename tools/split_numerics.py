#!/usr/bin/env python
"""Numerics study for the split tier's next step (NOTES.md "what comes next"): which activation rows could be stored hi-ONLY (one f16 per
value instead of the hi + lo' pair) without leaving the strict tolerance?  Hi-only rows halve the write-back conversions of their layer and
drop one of the three MFMAs of every product term set that reads them.

CPU only, no kernel involved: the head field in float64 on the bench fixture's weights, at sample positions of the fixture's head region,
with ONE activation at a time rounded to f16 (round to nearest, as v_cvt_f16_f32 does) and everything else exact.  What comes out per
variant: the largest change of a sample's colour, the largest relative change of its density, and the bound on a pixel that follows --
a pixel is a convex combination of its samples' colours (weights sum <= 1), and a relative density error d moves a sample's alpha by at
most 0.37 d (max of x exp(-x)), so |d rgb| <= max|d colour| + 0.37 max|d sigma / sigma| -- against the strict tolerance 1e-4.

    python tools/split_numerics.py [n_points]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

from geneface_amd import hparams as HP, synthetic as S
from oracle import radnerf_ref as R        # the checker's grid / SH encoders (this script is a study, not a product path)

torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
hp = HP.may_hparams(True)
sd = S.make_state_dict(hp, True)
seq = S.make_sequence(4, 64, 64, hp)
cond_feat = R.cal_cond_feat(sd, hp, torch.from_numpy(seq["cond_wins"][1])).double()
ind = sd["individual_embeddings"][0].double() if "individual_embeddings" in sd else None
bound = hp["bound"]
pos = (torch.rand(N, 3) - 0.5) * 0.9          # the synthetic head sits inside |x| < 0.45
d = F.normalize(torch.randn(N, 3), dim=1)
pls3, gt, ip = R._grid_args(hp, hp["desired_resolution"] * bound)
pls2, _, _ = R._grid_args(hp, hp["desired_resolution"])
W = {k: v.double() for k, v in sd.items() if k.endswith(".weight")}


def q16(x):
    return x.float().half().double()


def field(quant):
    """quant: set of activation names rounded to f16"""
    def maybe(name, x):
        return q16(x) if name in quant else x
    pf = R.grid_encode((pos + bound) / (2 * bound), sd["position_embedder.embeddings"], sd["position_embedder.offsets"], pls3, 16, gt, False, ip).double()
    pf = maybe("grid3d", pf)
    h = torch.cat([pf, cond_feat.reshape(1, -1).repeat(N, 1)], dim=1)
    h = maybe("amb1", F.relu(F.linear(h, W["ambient_net.net.0.weight"])))
    h = maybe("amb2", F.relu(F.linear(h, W["ambient_net.net.1.weight"])))
    amb = torch.tanh(F.linear(h, W["ambient_net.net.2.weight"]))
    af = R.grid_encode(((amb + 1) / 2).float(), sd["ambient_embedder.embeddings"], sd["ambient_embedder.offsets"], pls2, 16, gt, False, ip).double()
    af = maybe("grid2d", af)
    h = torch.cat([pf, af], dim=1)
    h = maybe("sig1", F.relu(F.linear(h, W["sigma_net.net.0.weight"])))
    h = maybe("sig2", F.relu(F.linear(h, W["sigma_net.net.1.weight"])))
    h = F.linear(h, W["sigma_net.net.2.weight"])
    sigma = torch.exp(h[:, 0].clamp(max=15))
    geo = maybe("geo", h[:, 1:])
    sh = maybe("sh", R.sh_encode(d).double())
    parts = [sh, geo] + ([ind.reshape(1, -1).repeat(N, 1)] if ind is not None else [])
    h = maybe("col1", F.relu(F.linear(torch.cat(parts, dim=1), W["color_net.net.0.weight"])))
    color = torch.sigmoid(F.linear(h, W["color_net.net.1.weight"]))
    return sigma, color


with torch.no_grad():
    s0, c0 = field(set())
    live = s0 > 1e-3                      # densities that can matter to a pixel at the fixture's step sizes
    print(f"{N} points, {int(live.sum())} with sigma > 1e-3; sigma median {s0[live].median():.3g}, max {s0.max():.3g}")
    print(f"{'hi-only activation':22s} {'max|d colour|':>14s} {'max|d sigma/sigma|':>19s} {'pixel bound':>12s}   within 1e-4?")
    names = ["grid3d", "amb1", "amb2", "grid2d", "sig1", "sig2", "geo", "sh", "col1"]
    rows = []
    for nm in names + ["ALL"]:
        s1, c1 = field(set(names) if nm == "ALL" else {nm})
        dc = (c1 - c0).abs().max().item()
        ds = ((s1 - s0).abs() / s0)[live].max().item()
        b = dc + 0.37 * ds
        rows.append((nm, dc, ds, b))
        print(f"{nm:22s} {dc:14.3e} {ds:19.3e} {b:12.3e}   {'yes' if b <= 1e-4 else 'no'}")
