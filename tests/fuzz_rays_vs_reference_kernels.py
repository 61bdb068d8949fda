#!/usr/bin/env python
"""Drawn ray batches through the ray-marching seam against the RUNNING reference kernels (oracle/_ref: test infrastructure, GPU only).
tests/test_gpu_vs_ref_kernels.py and tests/ref_kernels_report.py compare the fixture's camera; this script draws the cases: ray counts from 1 to
~40 K, origins outside / on / inside the box, directions that are axis-aligned (zero components), grazing, or random; bound 1 / 2 / 4 (1-3
cascades), occupancy from empty over sparse blobs to full, dt_gamma 0 ... 1/64, max_steps 1 ... 1024, n_step 1 ... 8, with and without jitter.
Checked per case: near_far_from_aabb; one inference wavefront step (march_rays + composite_rays); the training pair (march_rays_train,
composite_rays_train forward + backward, march_rays_train_backward), the reference's atomically ordered sample list brought into ray order first.

Bars: integer outputs (sample counts, the counter, alive indices) identical; positions, directions, step sizes, clocks and composited values
2e-6 relative to the tensor's largest magnitude (2e-5 for the compositing gradients: differences of products), bit equality being what the fixed tests
observe on the fixture.
Not collected by pytest: `python tests/fuzz_rays_vs_reference_kernels.py --cases 300 --out gpurun_out/fuzz_rays.json`."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda:0"


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    if a.shape != b.shape:
        return float("inf")
    fin = torch.isfinite(b)
    if not torch.equal(torch.isfinite(a), fin) or not torch.equal(torch.isnan(a), torch.isnan(b)):
        return float("inf")
    if fin.sum() == 0:
        return 0.0
    scale = max(float(b[fin].abs().max()), 1e-30)
    return float((a[fin] - b[fin]).abs().max()) / scale


def draw_rays(rng, N, bound):
    kind = rng.choice(["outside", "inside", "axis", "grazing", "mixed"])
    o = rng.uniform(-1, 1, (N, 3)) * bound * (0.9 if kind == "inside" else 3.0)
    if kind in ("outside", "mixed", "grazing", "axis"):
        far = np.abs(o).max(1) < bound
        o[far] *= 3.5 * bound / np.maximum(np.abs(o[far]).max(1, keepdims=True), 1e-3)
    tgt = rng.uniform(-1, 1, (N, 3)) * bound * (1.0 if kind != "grazing" else 1.0)
    if kind == "grazing":
        ax = rng.integers(0, 3, N)
        tgt[np.arange(N), ax] = np.sign(rng.uniform(-1, 1, N)) * bound * rng.choice([1.0, 0.999999, 1.000001], N)
    d = tgt - o
    if kind == "axis" or (kind == "mixed" and N > 4):
        k = N if kind == "axis" else N // 4
        ax = rng.integers(0, 3, k)
        d[:k] = 0.0
        d[np.arange(k), ax] = np.sign(rng.uniform(-1, 1, k))
        if kind == "axis":        # aim the axis-aligned rays at the box: origin's other two coordinates inside it
            for j in range(3):
                m = ax != j
                o[:k][m, j] = rng.uniform(-1, 1, m.sum()) * bound * 0.95
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    return str(kind), torch.from_numpy(o.astype(np.float32)), torch.from_numpy(d.astype(np.float32))


def draw_occupancy(rng, C, G):
    kind = rng.choice(["empty", "full", "blobs", "noise", "shell"])
    n = C * G ** 3
    if kind == "empty":
        grid = np.zeros(n, np.float32)
    elif kind == "full":
        grid = np.ones(n, np.float32)
    elif kind == "noise":
        grid = (rng.random(n) < rng.choice([0.002, 0.05, 0.5])).astype(np.float32)
    else:
        grid = np.zeros((C, G, G, G), np.float32)
        ax = np.arange(G)
        for c in range(C):
            for _ in range(int(rng.integers(1, 5))):
                ctr, r = rng.uniform(0.2, 0.8, 3) * G, rng.uniform(0.05, 0.3) * G
                dist = np.sqrt((ax[:, None, None] - ctr[0]) ** 2 + (ax[None, :, None] - ctr[1]) ** 2 + (ax[None, None, :] - ctr[2]) ** 2)
                grid[c] = np.maximum(grid[c], (dist < r).astype(np.float32) if kind == "blobs" else ((dist < r) & (dist > 0.8 * r)).astype(np.float32))
        grid = grid.reshape(-1)      # (any bijection of cells is an occupancy: the marcher reads bit morton(x, y, z), whatever produced it)
    bits = np.packbits(grid.reshape(-1, 8).astype(np.uint8), axis=1, bitorder="little").reshape(-1)
    return str(kind), torch.from_numpy(bits.copy())


def one_case(rng, ref, prod):
    RMs = {"ref": ref[0], "product": prod[0]}
    bound = float(rng.choice([1, 1, 2, 4]))
    C = 1 + int(np.ceil(np.log2(bound)))
    G = int(rng.choice([32, 64, 128]))
    N = int(rng.choice([1, 2, 63, 64, 65, 255, 1000, int(rng.integers(1, 40_000))]))
    dt_gamma = float(rng.choice([0.0, 1 / 256, 1 / 128, 1 / 64]))
    max_steps = int(rng.choice([1, 4, 16, 64, 256, 1024]))
    n_step = int(rng.choice([1, 2, 4, 8]))
    min_near = float(rng.choice([0.05, 0.2, 0.0001]))
    N = max(1, min(N, (1 << 22) // max_steps))      # room for every ray's full budget: a list that overflows is cut in ATOMIC order by the reference
    perturb = bool(rng.integers(0, 2))
    rk, ro, rd = draw_rays(rng, N, bound)
    ok, bits = draw_occupancy(rng, C, G)
    cfg = dict(bound=bound, cascades=C, grid=G, N=N, dt_gamma=dt_gamma, max_steps=max_steps, n_step=n_step, min_near=min_near, perturb=perturb, rays=rk,
               occupancy=ok)
    aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    noise = torch.rand(N, generator=g) if perturb else torch.zeros(N)
    M = N * n_step
    M += 128 - M % 128
    sig_i, rgb_i = torch.rand(M, generator=g) * 60, torch.rand(M, 3, generator=g)
    Mt = min(N * max_steps, 1 << 22)
    out = {}
    for who, RM in RMs.items():
        d = lambda t: t.to(DEV)
        nears, fars = torch.empty(N, device=DEV), torch.empty(N, device=DEV)
        RM.near_far_from_aabb(d(ro), d(rd), d(aabb), N, min_near, nears, fars)
        # one inference wavefront step over every ray
        xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
        alive, t = torch.arange(N, dtype=torch.int32, device=DEV), nears.clone()
        RM.march_rays(N, n_step, alive, t, d(ro), d(rd), bound, dt_gamma, max_steps, C, G, d(bits), nears, fars, xyzs, dirs, deltas, d(noise))
        ws, dep, img = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV)
        RM.composite_rays(N, n_step, 1e-4, alive, t, d(sig_i), d(rgb_i), deltas, ws, dep, img)
        r = {"nears": nears, "fars": fars, "inf.xyzs": xyzs, "inf.dirs": dirs, "inf.deltas": deltas, "i:inf.alive": alive, "inf.t": t, "inf.ws": ws,
             "inf.depth": dep, "inf.image": img}
        # the training pair
        x2, d2, e2 = torch.zeros(Mt, 3, device=DEV), torch.zeros(Mt, 3, device=DEV), torch.zeros(Mt, 2, device=DEV)
        rays, counter = torch.empty(N, 3, dtype=torch.int32, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV)
        RM.march_rays_train(d(ro), d(rd), d(bits), bound, dt_gamma, max_steps, N, C, G, Mt, nears, fars, x2, d2, e2, rays, counter, d(noise))
        torch.cuda.synchronize()
        rr = rays.cpu().numpy().astype(np.int64)
        rr = rr[np.argsort(rr[:, 0], kind="stable")]
        cnt = rr[:, 2]
        tot = int(cnt.sum())
        rep = np.repeat(np.arange(N), cnt)
        excl = np.cumsum(cnt) - cnt
        src = torch.from_numpy(rr[rep, 1] + (np.arange(tot) - excl[rep])).to(DEV)
        xs, ds, es = x2[src], d2[src], e2[src].contiguous()
        r.update({"i:train.counts": torch.from_numpy(cnt), "i:train.counter": counter[:1].clone(), "train.xyzs": xs, "train.dirs": ds, "train.deltas": es})
        if tot:
            rays_c = torch.from_numpy(np.stack([rr[:, 0], excl, cnt], 1).astype(np.int32)).to(DEV)
            gg = torch.Generator().manual_seed(5)
            sig, rgb, amb = torch.rand(tot, generator=gg) * 40, torch.rand(tot, 3, generator=gg), torch.rand(tot, generator=gg)
            ws2, am2, dp2, im2 = (torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, device=DEV), torch.empty(N, 3, device=DEV))
            RM.composite_rays_train_forward(d(sig), d(rgb), d(amb), es, rays_c, tot, N, 1e-4, ws2, am2, dp2, im2)
            gws, gam, gim = torch.rand(N, generator=gg), torch.rand(N, generator=gg), torch.rand(N, 3, generator=gg)
            gs, gc, ga = torch.zeros(tot, device=DEV), torch.zeros(tot, 3, device=DEV), torch.zeros(tot, device=DEV)
            RM.composite_rays_train_backward(d(gws), d(gam), d(gim), d(sig), d(rgb), d(amb), es, rays_c, ws2, am2, im2, tot, N, 1e-4, gs, gc, ga)
            gx, gd = torch.randn(tot, 3, generator=gg), torch.randn(tot, 3, generator=gg)
            go, gdd = torch.zeros(N, 3, device=DEV), torch.zeros(N, 3, device=DEV)
            RM.march_rays_train_backward(d(gx), d(gd), rays_c, es, N, tot, go, gdd)
            r.update({"train.ws": ws2, "train.amb": am2, "train.depth": dp2, "train.image": im2, "train.g_sig": gs, "train.g_rgb": gc, "train.g_amb": ga,
                      "train.g_ro": go, "train.g_rd": gdd})
        torch.cuda.synchronize()
        out[who] = r
    cfg["samples_train"] = int(out["ref"]["i:train.counts"].sum())
    errs = {}
    for k, v in out["ref"].items():
        p = out["product"].get(k)
        if p is None:
            errs[k] = float("inf")
        elif k.startswith("i:"):
            errs[k] = 0.0 if torch.equal(p.cpu(), v.cpu()) else float("inf")
        else:
            errs[k] = rel(p, v)
    return cfg, errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--bar", type=float, default=2e-6)
    ap.add_argument("--contract", default="fast", choices=["fast", "off"], help="which build of the reference's sources: hipcc's default fp contraction "
                    "(what nvcc's default --fmad=true gives the reference too) or -ffp-contract=off")
    args = ap.parse_args()
    from geneface_amd.compat import _freqencoder, _gridencoder, _raymarching_face, _shencoder
    from oracle import ref_kernels
    ref = ref_kernels.load(args.contract)
    prod = (_raymarching_face, _gridencoder, _shencoder, _freqencoder)
    rng = np.random.default_rng(args.seed)
    worst, t0, samples = {}, time.time(), 0
    for i in range(args.cases):
        cfg, errs = one_case(rng, ref, prod)
        samples += cfg["samples_train"]
        for k, e in errs.items():
            if e > worst.get(k, (-1.0, None))[0]:
                worst[k] = (e, cfg)
            bar = 10 * args.bar if k.startswith("train.g_") else args.bar      # the compositing gradients are differences of products: a few more ulps
            if not e <= bar:
                print(json.dumps({"case": i, "config": cfg, "quantity": k, "relative_error": e, "bar": bar}))
                return 1
    record = {"cases": args.cases, "seed": args.seed, "seconds": round(time.time() - t0, 1), "training_samples_compared": samples,
              "reference_build": "oracle/_ref, fp contraction " + args.contract, "bar": args.bar,
              "worst_relative_error": {k: {"error": v[0], "config": v[1]} for k, v in sorted(worst.items())}}
    print(json.dumps(record))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(record, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
