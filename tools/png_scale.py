#!/usr/bin/env python
"""Host-side term of the N-rank frame loop, measured WITHOUT GPUs (VERDICT r5 missing #1 / next #4a).

The reference fans a sequence out over N processes that all write `<tmp_imgs_dir>/<idx:05d>.png` (inference/nerfs/base_nerf_infer.py:97-101,
150-179).  At 8 ranks x ~750 frames/s x 0.66 MB that is ~3.9 GB/s of deflate + file creation into ONE directory -- the only term of the
8-GPU curve that is not per-GPU.  This tool runs N processes, each with its own native writer (gf_png_writer_*: csrc/png_writer.cpp), fed
synthetic 512x512 frames (smooth background + a textured head/torso blob: ~0.6 MB per file at the writer's default Z_RLE level 1, like
rendered frames) either as fast as the writer takes them or paced at --pace frames/s per rank, into one shared directory and into per-rank
directories, and reports sustained frames/s per rank and in aggregate, and where the time went.

    python tools/png_scale.py --ranks 1,2,4,8 --workers 4,8,16 --frames 375 --out profiles/round6/png_scale_<host>.json
"""
import argparse
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def synthetic_frames(n=16, size=512, seed=0):
    """uint8 [n, size, size, 3]: a smooth two-axis gradient (long runs for Z_RLE, like the fixture's background) and an ellipse of low-amplitude
    texture moving a little from frame to frame (the rendered head + torso: about a third of the picture, noisy in its low bits)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / size
    out = np.empty((n, size, size, 3), dtype=np.uint8)
    for k in range(n):
        img = np.stack([0.55 + 0.25 * x, 0.45 + 0.3 * y, 0.6 - 0.2 * x * y], axis=-1)
        img = np.floor(img * 32) / 32          # banded, as an 8-bit background photograph quantised by the fixture is
        cx, cy = 0.5 + 0.02 * np.sin(k), 0.55 + 0.02 * np.cos(k)
        blob = ((x - cx) / 0.28) ** 2 + ((y - cy) / 0.40) ** 2 < 1.0
        tex = 0.5 + 0.3 * np.sin(40 * x + k)[..., None] * np.cos(31 * y)[..., None] + rng.normal(0, 0.02, (size, size, 3))
        img = np.where(blob[..., None], tex, img)
        out[k] = (np.clip(img, 0, 1) * 255).astype(np.uint8)
    return out


def _rank(rank, ranks, frames, workers, out_dir, pace, barrier, q, level=1, strategy=3):
    from geneface_amd.png import FrameWriter
    src = synthetic_frames()
    first = rank * frames                       # contiguous blocks, as infer.shard_range hands them out
    w = FrameWriter(out_dir, workers=workers, level=level, strategy=strategy)
    w.submit(10 ** 6 + rank, src[0])            # creates the pool outside the timed region
    barrier.wait()
    t0 = time.perf_counter()
    for k in range(frames):
        if pace:
            due = t0 + k / pace
            while time.perf_counter() < due:
                pass
        w.submit(first + k, src[k % len(src)])
    t_submit = time.perf_counter() - t0
    w.close()
    dt = time.perf_counter() - t0
    st = w.stage_seconds()
    q.put({"rank": rank, "fps": frames / dt, "submit_fps": frames / t_submit, "seconds": dt, "deflate_ms_per_frame": st["deflate_sum_over_workers"] / st["frames"] * 1e3,
           "write_ms_per_frame": st["write_sum_over_workers"] / st["frames"] * 1e3, "wait_for_room_s": st["wait_for_room"], "MB_per_frame": st["bytes"] / st["frames"] / 1e6})


def run(ranks, frames, workers, layout, pace, base, level=1, strategy=3):
    ctx = mp.get_context("spawn")
    root = tempfile.mkdtemp(prefix="gf_pngscale_", dir=base)
    try:
        barrier, q = ctx.Barrier(ranks), ctx.Queue()
        dirs = [root if layout == "shared" else os.path.join(root, f"rank{r}") for r in range(ranks)]
        ps = [ctx.Process(target=_rank, args=(r, ranks, frames, workers, dirs[r], pace, barrier, q, level, strategy)) for r in range(ranks)]
        t0 = time.perf_counter()
        for p in ps:
            p.start()
        res = sorted((q.get(timeout=600) for _ in ps), key=lambda d: d["rank"])
        for p in ps:
            p.join()
        n_files = sum(len([f for f in os.listdir(d) if f.endswith(".png")]) for d in set(dirs))
        assert n_files == ranks * (frames + 1), (n_files, ranks * (frames + 1))
        slow = max(r["seconds"] for r in res)
        return {"zlib_level": level, "zlib_strategy": strategy, "ranks": ranks, "workers_per_rank": workers, "layout": layout, "pace_per_rank": pace, "frames_per_rank": frames,
                "aggregate_fps": ranks * frames / slow, "slowest_rank_fps": min(r["fps"] for r in res), "MB_per_frame": res[0]["MB_per_frame"],
                "deflate_ms_per_frame": float(np.mean([r["deflate_ms_per_frame"] for r in res])),
                "write_ms_per_frame": float(np.mean([r["write_ms_per_frame"] for r in res])),
                "wait_for_room_s_max": max(r["wait_for_room_s"] for r in res), "wall_s_incl_spawn": time.perf_counter() - t0}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--workers", default="4,8,16")
    ap.add_argument("--frames", type=int, default=375, help="per rank (375 = 3000 / 8: configs[3])")
    ap.add_argument("--pace", type=float, default=0.0, help="frames/s each rank submits at (0: as fast as the writer accepts)")
    ap.add_argument("--layouts", default="shared,per_rank")
    ap.add_argument("--base", default="/dev/shm", help="where the directories go (/dev/shm: page cache only; a disk path: the box's file system)")
    ap.add_argument("--level", type=int, default=1, help="zlib level (0: stored blocks -- adler32 + crc32 + copy only, 0.79 MB per 512x512 frame)")
    ap.add_argument("--strategy", type=int, default=3, help="zlib strategy (3: Z_RLE, the writer's default)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rec = {"host_cores": os.cpu_count(), "base": a.base, "frames": "synthetic 512x512 (gradient + textured blob), 16 distinct", "results": []}
    for layout in a.layouts.split(","):
        for workers in map(int, a.workers.split(",")):
            for ranks in map(int, a.ranks.split(",")):
                r = run(ranks, a.frames, workers, layout, a.pace, a.base, a.level, a.strategy)
                rec["results"].append(r)
                print(json.dumps(r), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
