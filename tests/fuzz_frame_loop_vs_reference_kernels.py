#!/usr/bin/env python
"""Drawn shards of the FRAME LOOP (geneface_amd.infer.FramePipeline: rays generated inside k_frame_init from pose + intrinsics, frames in flight on
side streams, uint8 frames through pinned memory) against (a) the module API fed with gf_pinhole_rays of the same pose -- the same bytes, by
construction -- and (b) the reference pipeline over the reference's own kernels (oracle/_ref, test infrastructure) on those rays.  The soaks of
tools/parity_hunt.py walk 512 x 512 on the fixture's path; this script draws frame sizes (1 x 1 ... 300 x 300, non-square), cameras from inside the
head to past it, intrinsics, the arithmetic tier and the number of frames in flight, three to six frames per shard.

Bars: uint8 frames of the loop == uint8 of the module API on the kernel's own rays (byte equality); against the reference's kernels the strict tiers
within 1 LSB everywhere and max |d rgb| <= 1e-4 on the float picture; fast: PSNR >= 40 dB (the fraction of bytes within 1 LSB is recorded).
Not collected by pytest: `python tests/fuzz_frame_loop_vs_reference_kernels.py --cases 200 --out gpurun_out/x.json`."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_frames_vs_reference_kernels import look_at  # noqa: E402
from helpers import model_fixture, sequence  # noqa: E402
from oracle import radnerf_ref as R  # noqa: E402
from oracle import ref_kernels  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from geneface_amd.infer import FramePipeline
    from geneface_amd.radnerf_torso import RADNeRFTorso
    rng = np.random.default_rng(args.seed)
    mods = ref_kernels.load("fast")
    models = {}
    for ident in (0, 1000):
        hp, sd = model_fixture(True, ident)
        m = RADNeRFTorso(hp)
        m.load_state_dict(sd, strict=True)
        models[ident] = (hp, {k: v.to(DEV) for k, v in sd.items()}, m.to(DEV).eval())
    base = sequence(8, 64, 64)
    worst, frames, t0 = {}, 0, time.time()
    fast_min_within, fast_min_db = 1.0, 99.0
    for i in range(args.cases):
        ident = int(rng.choice([0, 1000]))
        hp, sd_g, model = models[ident]
        H, W = (int(rng.integers(1, 300)), int(rng.integers(1, 300))) if rng.random() < 0.8 else (int(rng.choice([1, 2, 64, 256])),) * 2
        T = int(rng.integers(3, 7))
        poses = []
        for _ in range(T):
            radius = float(rng.choice([0.25, 1.0, 2.0, 3.35, 3.35, 6.0]) * rng.uniform(0.9, 1.1))
            v = rng.normal(size=3)
            poses.append(look_at((v / np.linalg.norm(v) * radius).astype(np.float64), float(rng.uniform(-math.pi, math.pi))))
        fov = math.radians(float(rng.choice([10, 21.24, 60]) * rng.uniform(0.9, 1.1)))
        fy = 0.5 * H / math.tan(0.5 * fov)
        intr = [fy * float(rng.choice([1.0, 1.0, 0.85, 1.2])), fy, W / 2 + float(rng.uniform(-0.15, 0.15)) * W, H / 2 + float(rng.uniform(-0.15, 0.15)) * H]
        precision = str(rng.choice(["fp32", "fp32", "split", "fast"]))
        in_flight = int(rng.choice([1, 2, 3, 4]))
        seq = {"cond_wins": base["cond_wins"][:T], "poses": np.stack(poses).astype(np.float32), "intrinsics": np.asarray(intr, np.float32),
               "bg_img": np.random.default_rng(i).random((H * W, 3), dtype=np.float32), "H": H, "W": W}
        cfg = dict(identity=ident, H=H, W=W, T=T, precision=precision, in_flight=in_flight, fov_deg=round(math.degrees(fov), 1))
        model.render_impl, model.render_precision = "fused", precision
        with torch.no_grad():
            pipe = FramePipeline(model, hp, seq, DEV, in_flight=in_flight)
            got = {k: torch.from_numpy(frame.copy()) for k, frame in pipe.stream(range(T))}      # the loop as the entry point drives it: frames in flight, slots rotating
            for k in range(T):
                smp = pipe.kernel_sample(k)          # rays_o / rays_d = gf_pinhole_rays of the pose: the device function k_frame_init runs
                ro, rd = smp["rays_o"], smp["rays_d"]
                out = model.render(ro, rd, smp["cond_wins"], smp["bg_coords"], smp["pose"], index=0, staged=False, bg_color=smp["bg_img"], perturb=False,
                                   force_all_rays=True, **hp)
                u8 = (out["rgb_map"].reshape(H, W, 3) * 255).to(torch.uint8).cpu()
                if not torch.equal(u8, got[k]):
                    print(json.dumps({"case": i, "frame": k, "config": cfg, "bytes_differing_between_the_loop_and_the_module_api": int((u8 != got[k]).sum())}))
                    return 1
                with R.kernel_backend(mods):
                    ref = R.render(sd_g, hp, ro, rd, smp["cond_wins"], smp["bg_coords"], smp["pose"], smp["bg_img"], True)
                rgb_ref = ref["rgb_map"].reshape(H, W, 3).float().cpu()
                ref8 = (rgb_ref * 255).to(torch.uint8)
                lsb = (got[k].int() - ref8.int()).abs()
                err = float((out["rgb_map"].reshape(H, W, 3).float().cpu() - rgb_ref).abs().max())
                frames += 1
                if err > worst.get(precision, (-1.0,))[0]:
                    worst[precision] = (err, int(lsb.max()), cfg)
                within = float((lsb <= 1).float().mean())
                mse = float(((out["rgb_map"].reshape(H, W, 3).float().cpu() - rgb_ref) ** 2).mean())
                db = 99.0 if mse == 0 else 10 * math.log10(1.0 / mse)
                if precision == "fast":
                    fast_min_within, fast_min_db = min(fast_min_within, within), min(fast_min_db, db)
                # fast: PSNR >= 40 dB; its second clause, <= 1 LSB on >= 99.9 % of the bytes, is a statement about the benchmark's
                # camera -- with the camera inside the head half-precision densities move the early-termination step at more pixels (recorded, not gated)
                ok = (db >= 40.0) if precision == "fast" else (err <= 1e-4 and int(lsb.max()) <= 1)
                if not ok:
                    print(json.dumps({"case": i, "frame": k, "config": cfg, "max_abs_rgb": err, "max_lsb": int(lsb.max()), "psnr": db, "bytes_within_1_lsb": within}))
                    return 1
    record = {"cases": args.cases, "frames": frames, "seed": args.seed, "seconds": round(time.time() - t0, 1),
              "loop_vs_module_api": "byte-identical on every frame", "fast_tier_min_psnr_db": round(fast_min_db, 2),
              "fast_tier_min_fraction_of_bytes_within_1_lsb": round(fast_min_within, 5),
              "worst_vs_reference_kernels": {k: {"max_abs_rgb": v[0], "max_lsb": v[1], "config": v[2]} for k, v in sorted(worst.items())}}
    print(json.dumps(record))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(record, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
