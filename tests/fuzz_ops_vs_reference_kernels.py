#!/usr/bin/env python
"""Random configurations of the three encoder seams against the RUNNING reference kernels (oracle/_ref, built from the CUDA sources where they
lie under /root/reference: test infrastructure, GPU only) and, for the grid encoder, against the CPU oracle as well.  tests/test_gpu_vs_ref_kernels.py compares fixed cases; this script draws them:
dimension, channel count, level count, base resolution, table size, index rule, interpolation, align_corners, batch sizes from 1 to ~100 K,
inputs on and beyond the unit cube.  Not collected by pytest (a soak, minutes): `python tests/fuzz_ops_vs_reference_kernels.py --cases 400
--out gpurun_out/fuzz.json`; exit code 1 and the offending configuration on the first disagreement.

Bars against the oracle: forward outputs and dy_dx 1e-5 * scale (a few fma groupings apart: the bar of test_grid_encoder_large_random_vs_oracle),
table gradients 2e-5 * scale (fixed-point scatter against a float one), input gradients 1e-5 * scale; against the reference kernels the grid bars grow
with the finest resolution (their level scale comes from the DEVICE's exp2f and may sit one ulp from the host's); SH / frequency encodings 2e-6."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kernels as K  # noqa: E402  (test infrastructure: the CPU restatement)
DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    scale = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / scale


def rel_q(a, b, q=0.999):
    """The q-quantile of |a - b| over max |b|: for quantities that are DISCONTINUOUS in the level scale (dy_dx of a linear lookup is piecewise
    constant: a point within an ulp of a cell face lands in the other cell under the reference's device-side exp2f) a handful of elements may
    differ by O(1) without either side being wrong."""
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    scale = max(float(b.abs().max()), 1e-30)
    d = (a - b).abs()
    k = min(d.numel() - 1, int(q * d.numel()))
    return float(d.kthvalue(k + 1).values) / scale


def grid_case(rng, ref, prod):
    from geneface_amd.encoders.gridencoder import GridEncoder
    D = int(rng.choice([2, 3]))
    C = int(rng.choice([1, 2, 4, 8]))
    L = int(rng.integers(2, 17))      # (one level + desired_resolution divides by zero in the reference's own constructor, grid.py:104-106)
    H = int(rng.choice([4, 8, 16, 32]))
    log2T = int(rng.integers(8, 20))
    desired = int(rng.choice([64, 256, 1024, 2048, 4096]))
    gridtype = str(rng.choice(["hash", "tiled"]))
    interp = str(rng.choice(["linear", "smoothstep"]))
    align = bool(rng.integers(0, 2))
    B = int(rng.choice([1, 2, 63, 64, 65, 127, 1000, 4097, int(rng.integers(1, 100_000))]))
    cfg = dict(op="grid", D=D, C=C, L=L, H=H, log2T=log2T, desired=desired, gridtype=gridtype, interp=interp, align_corners=align, B=B)
    if desired <= H:
        desired = H * 2
        cfg["desired"] = desired
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=H, log2_hashmap_size=log2T, desired_resolution=desired,
                      gridtype=gridtype, align_corners=align, interpolation=interp)
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    table = (torch.rand(enc.embeddings.shape, generator=g) * 2 - 1).to(DEV)
    off = enc.offsets.to(DEV)
    x = torch.rand(B, D, generator=g)
    if B > 8:                     # a few points on the faces and beyond the cube, as test_fused_lookup_far_out_of_range... does
        x[0], x[1], x[2], x[3] = 0.0, 1.0, -0.25, 1.5
    x = x.to(DEV)
    grad = torch.randn(L, B, C, generator=g).to(DEV)
    S = float(np.log2(enc.per_level_scale))
    gt, ip = enc.gridtype_id, enc.interp_id
    res = {}
    for name, m, dev in (("ref", ref[1], DEV), ("product", prod[1], DEV), ("oracle", K.gridencoder, "cpu")):
        out = torch.empty(L, B, C, device=dev)
        dy = torch.empty(B, L * D * C, device=dev)
        m.grid_encode_forward(x.to(dev), table.to(dev), off.to(dev), out, B, D, C, L, S, H, dy, gt, align, ip)
        g_emb, g_in = torch.zeros(table.shape, device=dev), torch.zeros(B, D, device=dev)
        m.grid_encode_backward(grad.to(dev), x.to(dev), table.to(dev), off.to(dev), g_emb, B, D, C, L, S, H, dy, g_in, gt, align, ip)
        if dev != "cpu":
            torch.cuda.synchronize()
        res[name] = dict(out=out, dy_dx=dy, g_emb=g_emb, g_in=g_in)
    # Against the CPU oracle (the same host-derived level scales: grid_core.hpp derives scale = exp2f(l S) H - 1 ONCE on the host, as the oracle does):
    # rounding of a few fma groupings.  Against the running reference kernels the level scale itself may sit one ulp away (their kernel evaluates exp2f
    # on the device): positions move by ulp(scale), i.e. the bar scales with the finest resolution.
    fine = float(enc.base_resolution * enc.per_level_scale ** (L - 1))
    loose = max(1e-5, 8 * fine * 2.0 ** -23)
    bars = {"oracle.out": 1e-5, "oracle.dy_dx": 1e-5, "oracle.g_emb": 2e-5, "oracle.g_in": 1e-5,
            "ref.out": loose, "ref.dy_dx": 4 * loose, "ref.g_emb": 4 * loose, "ref.g_in": 4 * loose}
    errs = {k: (rel if k.startswith("oracle") else rel_q)(res["product"][k.split(".")[1]], res[k.split(".")[0]][k.split(".")[1]]) for k in bars}
    return cfg, errs, bars


def sh_case(rng, ref, prod):
    degree = int(rng.integers(1, 9))
    B = int(rng.choice([1, 63, 64, 65, 1000, int(rng.integers(1, 200_000))]))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    d = torch.randn(B, 3, generator=g)
    if rng.integers(0, 2):
        d = d / d.norm(dim=-1, keepdim=True)
    n = degree * degree
    res = {}
    for name, m, dev in (("ref", ref[2], DEV), ("product", prod[2], DEV), ("oracle", K.shencoder, "cpu")):
        out, dy = torch.empty(B, n, device=dev), torch.empty(B, 3 * n, device=dev)
        m.sh_encode_forward(d.to(dev), out, B, 3, degree, dy)            # the extension's "C" is the degree (shencoder.cu:400-412)
        if dev != "cpu":
            torch.cuda.synchronize()
        res[name] = dict(out=out, dy_dx=dy)
    # (free-length directions: degree-7 polynomials of components up to ~4 cancel; on unit directions the agreement is ~1e-7)
    bars = {"oracle.out": 1e-5, "oracle.dy_dx": 2e-5, "ref.out": 1e-5, "ref.dy_dx": 2e-5}
    errs = {k: rel(res["product"][k.split(".")[1]], res[k.split(".")[0]][k.split(".")[1]]) for k in bars}
    return dict(op="sh", degree=degree, B=B), errs, bars


def freq_case(rng, ref, prod):
    D = int(rng.integers(1, 7))
    deg = int(rng.integers(1, 13))
    B = int(rng.choice([1, 63, 64, 65, 1000, int(rng.integers(1, 200_000))]))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    x = (torch.rand(B, D, generator=g) * 2 - 1)
    C = D + D * deg * 2
    res = {}
    for name, m, dev in (("ref", ref[3], DEV), ("product", prod[3], DEV), ("oracle", K.freqencoder, "cpu")):
        out = torch.empty(B, C, device=dev)
        m.freq_encode_forward(x.to(dev), B, D, deg, C, out)
        if dev != "cpu":
            torch.cuda.synchronize()
        res[name] = out
    # The reference calls the hardware's fast sine (__sinf, freqencoder.cu:52): its absolute error grows with the argument (about an ulp of
    # 2^(deg-1) x); oracle and product evaluate the sine itself (sh_core.hpp::sin_reduced, 9.3e-8 up to |x| = 512).
    bars = {"oracle.out": 4e-7, "ref.out": max(2e-6, 4 * 2.0 ** (deg - 1) * 2.0 ** -23)}
    errs = {"oracle.out": rel(res["product"], res["oracle"]), "ref.out": rel(res["product"], res["ref"])}
    return dict(op="freq", D=D, deg=deg, B=B), errs, bars


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from geneface_amd.compat import _freqencoder, _gridencoder, _raymarching_face, _shencoder
    from oracle import ref_kernels
    ref = ref_kernels.load("off")      # the reference's sources with -ffp-contract=off: the arithmetic the CPU oracle restates
    prod = (_raymarching_face, _gridencoder, _shencoder, _freqencoder)
    rng = np.random.default_rng(args.seed)
    worst, t0, n = {}, time.time(), {"grid": 0, "sh": 0, "freq": 0}
    for i in range(args.cases):
        fn = (grid_case, grid_case, grid_case, sh_case, freq_case)[i % 5]
        try:
            cfg, errs, bars = fn(rng, ref, prod)
        except RuntimeError as e:
            print(json.dumps({"case": i, "error": str(e)[:300]}))
            raise
        n[cfg["op"]] += 1
        for k, e in errs.items():
            key = f"{cfg['op']}.{k}"
            if e > worst.get(key, (0.0, None))[0]:
                worst[key] = (e, cfg)
            if not e <= bars[k]:
                print(json.dumps({"case": i, "config": cfg, "quantity": k, "relative_error": e, "bar": bars[k]}))
                return 1
    record = {"cases": n, "seed": args.seed, "seconds": round(time.time() - t0, 1), "reference_build": "oracle/_ref, -ffp-contract=off",
              "worst_relative_error": {k: {"error": v[0], "config": v[1]} for k, v in sorted(worst.items())}}
    print(json.dumps(record))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(record, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
