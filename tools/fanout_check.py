#!/usr/bin/env python
"""The entry point's N-rank fan-out on a box with ONE GPU (VERDICT r5 next #4b): LM3d_RADNeRFInfer.forward_system(world_size=N) --
mp.spawn, one replica per rank filled by the broadcast from rank 0, contiguous blocks (base_nerf_infer.py:131-193), every rank writing
`<tmp_imgs_dir>/<idx:05d>.png` into the SAME directory -- with the ranks time-sharing cuda:0 (inp["ranks_share_gpu"]: gloo carries the
collectives), against the same sequence rendered by one rank.  Checked: every index written exactly once, every file byte-identical to the
1-rank run's (the writer's deflate is deterministic, so equal files <=> equal frames).  No scaling meaning: the ranks share one GPU.

    python tools/fanout_check.py --frames 3000 --ranks 8 --json profiles/round6/r6_fanout_8_ranks_one_gpu.json
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def digest_dir(d):
    out = {}
    for n in sorted(os.listdir(d)):
        with open(os.path.join(d, n), "rb") as f:
            out[n] = hashlib.sha256(f.read()).hexdigest()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "split"])
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.lm3d_radnerf_infer import LM3d_RADNeRFInfer, RADNeRFPoseSource
    from geneface_amd.radnerf_torso import RADNeRFTorso

    hp = dict(HP.may_hparams(True), render_precision=args.precision)
    model = RADNeRFTorso(hp)
    model.load_state_dict(S.make_state_dict(hp, True), strict=True)
    model.render_precision = args.precision
    dd, _ = S.make_dataset_dict(T=args.frames, H=args.size, W=args.size)
    inf = LM3d_RADNeRFInfer(hp, model=model, dataset=RADNeRFPoseSource(dd, hp), device="cuda:0")
    work = tempfile.mkdtemp(prefix="gf_fanout_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    os.environ["MASTER_PORT"] = str(29500 + os.getpid() % 2000)
    try:
        cond = os.path.join(work, "pred_lm3d.npy")
        np.save(cond, S.make_landmarks(args.frames).astype(np.float32)[None])
        res = {"workload": f"May head+torso {args.size}x{args.size}, {args.frames} frames (BASELINE.json configs[3]) through LM3d_RADNeRFInfer: "
                           f"forward_system(world_size={args.ranks}) with the ranks SHARING cuda:0 (gloo) vs one rank; one PNG per frame into one directory",
               "render_precision": args.precision, "host_cores": os.cpu_count()}
        digests = {}
        for tag, world in (("one_rank", 1), ("fanout", args.ranks)):
            imgs = os.path.join(work, tag)
            inf.inp = {"cond_name": cond, "out_video_name": "", "audio_source_name": "", "tmp_imgs_dir": imgs, "return_frames": False,
                       "ranks_share_gpu": world > 1}
            samples = inf.get_pose_from_ds(inf.get_cond_from_input(inf.inp))
            t0 = time.perf_counter()
            out = inf.forward_system(samples, world_size=world, collect=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert out == imgs
            names = sorted(os.listdir(imgs))
            assert names == [f"{i:05d}.png" for i in range(args.frames)], (len(names), names[:3])       # every index, exactly once
            digests[tag] = digest_dir(imgs)
            res[tag] = {"ranks": world, "seconds_incl_process_start": dt, "files": len(names),
                        "MB_per_frame": sum(os.path.getsize(os.path.join(imgs, n)) for n in names) / len(names) / 1e6}
        diff = [n for n in digests["one_rank"] if digests["one_rank"][n] != digests["fanout"][n]]
        res["files_identical_to_the_one_rank_run"] = len(digests["one_rank"]) - len(diff)
        res["files_differing"] = diff[:16]
        res["digest_of_digests"] = hashlib.sha256("".join(digests["fanout"][n] for n in sorted(digests["fanout"])).encode()).hexdigest()
        print(json.dumps(res))
        if args.json:
            os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
            json.dump(res, open(args.json, "w"), indent=1)
        assert not diff, f"{len(diff)} files differ between the {args.ranks}-rank fan-out and the 1-rank run: {diff[:8]}"
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
