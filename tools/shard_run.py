#!/usr/bin/env python
"""BASELINE.json configs[3] at its real per-GPU shape, on ONE GPU: the block one rank of an 8-GPU job renders of a 3000-frame sequence
(375 frames, base_nerf_infer.py:150-155) through the entry point -- LM3d_RADNeRFInfer.infer_once: landmark file in, smoothed poses from
the dataset dict, the frame loop, one `%05d.png` per frame (base_nerf_infer.py:97-101) -- with fps and peak HBM recorded.

    python tools/shard_run.py [--frames 3000] [--ranks 8] [--rank 3] [--size 512] [--json profiles/round3/shard_375_of_3000.json]

Synthetic assets (no checkpoint / dataset exists offline): the seeded fixture of geneface_amd/synthetic.py at the May configuration.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--json", default=None)
    ap.add_argument("--no-png", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "split", "fast"])
    ap.add_argument("--files-only", action="store_true", help="inp['return_frames'] = False: the PNG files are the output, no stacked result (the reference's contract)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.infer import shard_range
    from geneface_amd.lm3d_radnerf_infer import LM3d_RADNeRFInfer, RADNeRFPoseSource
    from geneface_amd.png import decode_rgb8
    from geneface_amd.radnerf_torso import RADNeRFTorso

    hp = HP.may_hparams(True)
    model = RADNeRFTorso(hp)
    model.load_state_dict(S.make_state_dict(hp, True), strict=True)
    model.render_precision = args.precision
    dd, _ = S.make_dataset_dict(T=args.frames, H=args.size, W=args.size)
    inf = LM3d_RADNeRFInfer(hp, model=model, dataset=RADNeRFPoseSource(dd, hp), device="cuda:0")
    work = tempfile.mkdtemp(prefix="gf_shard_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        cond = os.path.join(work, "pred_lm3d.npy")
        np.save(cond, S.make_landmarks(args.frames).astype(np.float32)[None])      # [1, T, 204] as PostNet writes it
        imgs = None if args.no_png else os.path.join(work, "imgs")
        inp = {"cond_name": cond, "out_video_name": "", "audio_source_name": "", "tmp_imgs_dir": imgs, "shard": (args.rank, args.ranks),
               "return_frames": not args.files_only}
        lo, hi = shard_range(args.frames, args.rank, args.ranks)
        # one small warm-up block (library load, packing, first-launch costs), then the measured shard
        inf.infer_once(dict(inp, tmp_imgs_dir=None, shard=(0, max(args.ranks, args.frames // 8))))
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        frames = inf.infer_once(inp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        free, total = torch.cuda.mem_get_info()
        assert args.files_only or frames.shape == (hi - lo, args.size, args.size, 3)
        res = {"render_precision": args.precision,
               "workload": f"frames [{lo}, {hi}) of a {args.frames}-frame sequence = rank {args.rank} of {args.ranks} (BASELINE.json configs[3]), "
                           f"May head+torso {args.size}x{args.size}, through LM3d_RADNeRFInfer.infer_once" + ("" if args.no_png else " with one PNG per frame"),
               "frames": hi - lo, "seconds": dt, "fps": (hi - lo) / dt, "includes": "landmark normalisation + windows, pose lookup, H2D of the shard's inputs, one batched "
               "cond-encoder launch, render, D2H, PNG encode + write (tmpfs), the stacked uint8 result",
               "peak_torch_allocated_MB": torch.cuda.max_memory_allocated() / 1e6, "peak_torch_reserved_MB": torch.cuda.max_memory_reserved() / 1e6,
               "device_used_MB_after": (total - free) / 1e6, "device_total_MB": total / 1e6, "frames_in_flight": None}
        if imgs:
            names = sorted(os.listdir(imgs))
            assert names == [f"{i:05d}.png" for i in range(lo, hi)], (names[:3], lo, hi)
            res["png_MB_per_frame"] = sum(os.path.getsize(os.path.join(imgs, n)) for n in names) / len(names) / 1e6
            for k in (0, len(names) // 2, len(names) - 1):      # the files decode, and to the returned frames
                img = decode_rgb8(open(os.path.join(imgs, names[k]), "rb").read())
                assert img.shape == (args.size, args.size, 3) and (args.files_only or np.array_equal(img, frames[k]))
            res["returns"] = "image directory only" if args.files_only else "stacked uint8 frames + image directory"
        print(json.dumps(res))
        if args.json:
            os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
            json.dump(res, open(args.json, "w"), indent=1)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
