#!/usr/bin/env python
"""bench.py -- rendered 512x512 fps (head+torso) of the RAD-NeRF frame renderer on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]                     (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one frame: cond encoder + ray generation + occupancy-grid march + per-sample field + composite for
the head, then the torso pass and the final blend, uint8 conversion and the async D2H copy -- for the May
`lm3d_radnerf` + `lm3d_radnerf_torso` configuration (BASELINE.json configs[2]) on the seeded synthetic fixture
(random-init weights of that architecture, analytic head occupancy; there are no offline checkpoints).  Inputs
(landmark windows, poses, background, weights) are resident in HBM before the timed region.  Frames shard
across ranks with no data-path collective (weak scaling: every rank renders K frames); the one collective is
the weight broadcast before the loop.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts: one hardware queue per frame in flight (geneface_amd/__init__.py)

FLOP_PER_HEAD_SAMPLE = 178_688   # 2*(96*128+128*128+128*2 + 64*128+128*128+128*129 + 148*128+128*3)
FLOP_PER_TORSO_PIXEL = 32_768    # 2*(104*64+64*64+64*2 + 136*32+32*32+32*4)
BYTES_PER_HEAD_SAMPLE = 1_536    # fp32 table gathers: 16 levels * (8 + 4 corners) * 8 B
BYTES_PER_TORSO_PIXEL = 512
BYTES_PER_RAY = 56
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense MFMA peak for f32 inputs
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--impl", default=None, choices=[None, "ops", "fused"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--profile-frames", type=int, default=8)
    ap.add_argument("--in-flight", type=int, default=0, help="frames enqueued concurrently on separate streams (fused path); 0 = the pipeline's default")
    ap.add_argument("--no-overlap", action="store_true", help="one stream: frames do not overlap (per-kernel profiling runs)")
    ap.add_argument("--png-frames", type=int, default=48, help="frames of the extra leg that also writes every frame as PNG (0 = skip)")
    ap.add_argument("--fast", action="store_true", help="secondary line: the 'fast' parity tier of BASELINE.md section 4 (f16 MFMA operands and "
                                                        "activations, fp32 accumulate); the default line is fp32")
    ap.add_argument("--head-only", action="store_true", help="BASELINE.json configs[1]: May lm3d_radnerf head-only (default: configs[2], head+torso)")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-step timed loop until this much time has been measured; the line reports the median repetition")
    ap.add_argument("--repeats", type=int, default=0, help="fixed number of repetitions of the K-step loop (0 = from --min-seconds)")
    ap.add_argument("--no-stress", action="store_true", help="skip the sensitivity leg (thin-density fixture: every hit ray spends its whole sample budget)")
    return ap.parse_args()


def cpu_baseline(hp, sd, seq, n_frames, torso=True):
    """The oracle (CPU port of the reference's render path: torch-fp32 layers over the C kernels) timed on the host
    cores of this box, on a bounded sample of the same workload."""
    import torch
    from oracle import radnerf_ref as R
    H, W = seq["H"], seq["W"]
    bgc = R.get_bg_coords(H, W)
    bg = torch.from_numpy(seq["bg_img"]).view(1, -1, 3)

    def one(i):
        pose = torch.from_numpy(seq["poses"][i:i + 1])
        ro, rd = R.get_rays(pose, seq["intrinsics"], H, W)
        return R.render(sd, hp, ro, rd, torch.from_numpy(seq["cond_wins"][i]), bgc, R.convert_poses(pose), bg, torso=torso)
    one(0)  # warm-up (thread pools, page faults)
    frames = {}
    t0 = time.perf_counter()
    for i in range(1, 1 + n_frames):
        frames[i] = one(i)
    dt = time.perf_counter() - t0
    out = {"value": n_frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n_frames} {'head+torso' if torso else 'head-only'} {H}x{W} frames after 1 warm-up, oracle/radnerf_ref.render (torch fp32 + OpenMP C kernels)"}
    out["legacy_nerf"] = legacy_nerf_baseline(seq)
    return out, frames


def parity_vs_oracle(pipe, oracle_frames):
    """BASELINE.json's "PSNR vs reference": the frames the cpu_baseline leg rendered with the oracle against the same frames from the
    product (module API -> fp32 rgb_map, and the frame loop's uint8 output)."""
    import numpy as np
    import torch
    psnrs, max_abs, lsb = [], 0.0, 1.0
    for i, ref in sorted(oracle_frames.items()):
        rgb_ref = ref["rgb_map"].reshape(-1, 3).double()
        with torch.no_grad():
            out = pipe.run_model(pipe.sample(i))["rgb_map"].reshape(-1, 3).double().cpu()
            u8 = pipe.render_frame(i)
            pipe.wait()
        mse = float(((out - rgb_ref) ** 2).mean())
        psnrs.append(99.0 if mse == 0 else -10.0 * np.log10(mse))
        max_abs = max(max_abs, float((out - rgb_ref).abs().max()))
        ref8 = (rgb_ref.float() * 255).to(torch.uint8).reshape(u8.shape).int()
        lsb = min(lsb, float(((u8.int() - ref8).abs() <= 1).float().mean()))
    return {"psnr_db": min(psnrs), "max_abs_rgb": max_abs, "uint8_within_1_lsb": lsb, "frames": len(psnrs),
            "reference": "oracle/radnerf_ref.render (CPU restatement, pinned against the reference's own kernels) on the same inputs",
            "tolerance": "BASELINE.md section 4: max|d rgb| <= 1e-4 strict, PSNR >= 40 dB fast tier"}


def legacy_nerf_baseline(seq, rays=4096):
    """Baseline B2: the reference's only pure-PyTorch renderer, the vanilla Lm3dNeRF it replaced
    (64 + 128 samples per ray, two 8x256 MLPs, chunk 2048), restated in oracle/legacy_nerf_ref.py, random weights; a bounded
    sample of rays of one 512x512 frame, extrapolated to the frame.  Context only: a different model from the hot path."""
    import torch
    from oracle import legacy_nerf_ref as LN
    H, W = seq["H"], seq["W"]
    fx, _, cx, cy = (float(v) for v in seq["intrinsics"])
    w = LN.make_weights(0)
    c2w = torch.tensor([[1, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 0.6]], dtype=torch.float32)
    bg, cond = torch.from_numpy(seq["bg_img"]).view(H, W, 3), torch.zeros(64)
    # 2048-ray chunks of 256-wide layers do not scale to a whole two-socket host: time one chunk at a few thread counts, keep the best
    all_threads = torch.get_num_threads()
    best_t, best_dt = all_threads, None
    for t in sorted({all_threads, min(all_threads, 32), min(all_threads, 16)}, reverse=True):
        torch.set_num_threads(t)
        LN.render(w, H, W, fx, cx, cy, c2w, bg, cond, max_rays=LN.CHUNK)        # warm-up at this thread count
        t0 = time.perf_counter()
        LN.render(w, H, W, fx, cx, cy, c2w, bg, cond, max_rays=LN.CHUNK)
        d = time.perf_counter() - t0
        if best_dt is None or d < best_dt:
            best_t, best_dt = t, d
    torch.set_num_threads(best_t)
    t0 = time.perf_counter()
    LN.render(w, H, W, fx, cx, cy, c2w, bg, cond, max_rays=rays)
    dt = time.perf_counter() - t0
    torch.set_num_threads(all_threads)
    s_per_frame = dt / rays * H * W
    return {"value": 1.0 / s_per_frame, "unit": "frames/s", "s_per_frame": s_per_frame, "cores": best_t, "kind": "port",
            "sample": f"{rays} of {H * W} rays of one frame (2 chunks of 2048), extrapolated; published anchor ~28.8 s/frame on an RTX 2080 Ti"}


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON): whatever libraries print while the process runs (RCCL's version banner at communicator
    # creation, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torchrun (RANK set) the process group is created even for one rank, so a 1-GPU launch exercises the same RCCL init, broadcast,
    # barrier and all-reduce calls as the 2/4/8-GPU runs
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.infer import FramePipeline, broadcast_model_, shard_range
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.radnerf_torso import RADNeRFTorso

    impl = args.impl
    if impl is None:
        try:
            import geneface_amd.fused  # noqa: F401
            impl = "fused"
        except ImportError:
            impl = "ops"

    torso = not args.head_only
    hp = HP.may_hparams(torso)
    K, Wm = args.steps, args.warmup
    per_rank = K + Wm
    seq = S.make_sequence(per_rank * world, args.size, args.size, hp)
    sd = S.make_state_dict(hp, torso)
    model = (RADNeRFTorso if torso else RADNeRF)(hp)
    if rank == 0:
        model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    if args.fast:
        model.render_precision = "fast"
    broadcast_model_(model, src=0)  # the only collective (RCCL): one flattened weight buffer
    pipe = FramePipeline(model, hp, seq, dev, frames=shard_range(per_rank * world, rank, world), impl=impl, overlap=not args.no_overlap, in_flight=args.in_flight or None)

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    def timed_pass(p):
        """EXACTLY K steps between two barrier + synchronize pairs; the max over ranks."""
        barrier()
        t0 = time.perf_counter()
        for i in range(Wm, Wm + K):
            p.render_frame(i)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_loop(p):
        """The K-step pass repeated until --min-seconds of it have been measured (the driver's --steps 20 is 30 ms of GPU work: mostly
        pipeline fill and drain, and invisible to a utilisation sampler); every rank derives the same repeat count from the reduced time
        of the first pass.  Reports the median pass."""
        dts = [timed_pass(p)]
        reps = args.repeats or int(min(400, max(1, -(-args.min_seconds // dts[0]))))
        while len(dts) < reps:
            dts.append(timed_pass(p))
        return sorted(dts)[len(dts) // 2], dts

    # CPU baseline FIRST (rank 0, N = 1): the GPU legs then run back to back at the end of the process, where a utilisation sampler sees them
    cpu, parity = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, oracle_frames = cpu_baseline(hp, sd, seq, args.cpu_frames, torso)
        parity = parity_vs_oracle(pipe, oracle_frames)

    with torch.no_grad():
        for i in range(Wm):
            pipe.render_frame(i)
        dt, dts = timed_loop(pipe)

        roofline = None
        if rank == 0:
            roofline = measure_roofline(pipe, impl, Wm, min(args.profile_frames, K), PEAK_F32_MFMA_TFLOPS, fast=args.fast)

    if rank == 0:
        line = {
            "metric": "rendered 512x512 fps (head+torso)" if torso else "rendered 512x512 fps (head only)", "value": world * K / dt, "unit": "frames/s", "n_gpus": world,
            "steps": K, "warmup": Wm, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "repeats": len(dts), "timed_region_s": sum(dts), "ms_per_step_min_max": [min(dts) / K * 1e3, max(dts) / K * 1e3],
            "vs_baseline": None, "dtype": "f16 operands / f32 accumulate (fast tier)" if args.fast else "f32", "data": "synthetic",
            "config": {"workload": (f"May lm3d_radnerf + lm3d_radnerf_torso head+torso {args.size}x{args.size}, {K} frames per GPU "
                                    f"(BASELINE.json configs[2])" if torso else
                                    f"May lm3d_radnerf head-only {args.size}x{args.size}, {K} frames per GPU (BASELINE.json configs[1])")
                                   + f"; frame-sharded over {world} GPU(s)",
                       "impl": impl, "frames_total": world * K, "rays_per_frame": args.size * args.size,
                       "max_steps": hp["max_steps"], "parallelism": f"frame-shard x{world}",
                       "frames_in_flight": pipe.in_flight if impl == "fused" else 1, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "repeats": len(dts), "timing": "median of `repeats` passes of exactly `steps` frames, each between barrier + synchronize pairs"},
            "roofline": roofline,
            "parity": parity,
        }
        if roofline and not args.fast and roofline.get("samples_per_frame") and world == 1:
            # The same algorithmic FLOPs priced against the WHOLE frame time of the timed region (several frames in flight: the uneven end of one
            # launch -- 12 % of the kernel alone, DESIGN.md 4.2 -- is filled by the next frame's workgroups, but the frame also pays for the
            # three small kernels).  A lower bound of what the head kernel sustains in the pipelined product configuration.
            tf = roofline["samples_per_frame"] * FLOP_PER_HEAD_SAMPLE * (K / dt) / 1e12
            roofline["pipelined"] = {"achieved": tf, "frac": tf / roofline["peak"], "unit": roofline["unit"],
                                     "note": "algorithmic FLOPs per frame x measured fps; `achieved` / `frac` above are the kernel alone, one frame in flight"}
        if args.png_frames > 0 and world == 1:
            line["with_png"] = png_leg(pipe, Wm, min(args.png_frames, K))
        if not args.no_stress and world == 1 and impl == "fused":
            line["stress_fixture"] = stress_leg(args, hp, torso, seq, dev, impl, timed_loop)
        line["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


def stress_leg(args, hp, torso, seq, dev, impl, timed_loop):
    """Sensitivity of `value` to the fixture: the same frames through a model whose density head is scaled down until no ray saturates,
    so every ray that hits the occupancy grid spends its whole sample budget (the worst case a trained, thinner-than-synthetic May model
    can approach).  fps falls with the sample count; the kernel's roofline fraction should not."""
    import torch
    from geneface_amd import synthetic as S
    from geneface_amd.infer import FramePipeline
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.radnerf_torso import RADNeRFTorso
    sd = S.make_state_dict(hp, torso, sigma_row_scale=0.02)
    model = (RADNeRFTorso if torso else RADNeRF)(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    if args.fast:
        model.render_precision = "fast"
    pipe = FramePipeline(model, hp, seq, dev, frames=(0, args.steps + args.warmup), impl=impl, overlap=not args.no_overlap, in_flight=args.in_flight or None)
    with torch.no_grad():
        for i in range(args.warmup):
            pipe.render_frame(i)
        dt, dts = timed_loop(pipe)
        r = measure_roofline(pipe, impl, args.warmup, min(4, args.steps), PEAK_F32_MFMA_TFLOPS, fast=args.fast)
    return {"value": args.steps / dt, "unit": "frames/s", "repeats": len(dts), "samples_per_frame": r.get("samples_per_frame"),
            "roofline_frac": r.get("frac"), "kernel_ms_per_frame": r.get("kernel_ms_per_frame"), "tile_fill": r.get("tile_fill"),
            "example_frame": r.get("example_frame"),
            "fixture": "density row of sigma_net scaled by 0.02 (sigma ~ 1): no ray terminates early, every hit ray marches its full budget"}


def png_leg(pipe, first, n):
    """the rate with the PNG files of base_nerf_infer.py:97-101 written as well (worker threads, zlib level 1, off the
    render thread; a tmpfs directory).  Reported beside `value`, never inside it."""
    import shutil
    import tempfile
    import torch
    from geneface_amd.png import FrameWriter
    out_dir = tempfile.mkdtemp(prefix="gf_png_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    workers = min(32, os.cpu_count() or 4)
    try:
        writer = FrameWriter(out_dir, workers=workers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for i, frame in pipe.stream(range(first, first + n)):
                writer.submit(i, frame)
        writer.close()
        dt = time.perf_counter() - t0
        nbytes = sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir))
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    return {"value": n / dt, "unit": "frames/s", "frames": n, "png_workers": workers, "png_MB_per_frame": nbytes / n / 1e6,
            "note": "render + D2H + PNG encode/write on worker threads (FramePipeline.stream keeps the pipeline full)"}


def pmc_traffic():
    """HBM bytes per k_head_phase launch from the newest committed PMC summary (separate `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` passes of this same command, tools/gpu_round.sh + tools/pmc_summary.py; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be collected inside the timed run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*pmc_summary.json")))
    for path in reversed(files):
        try:
            d = json.load(open(path)).get("k_head_phase")
            if d and "fetch_MB_x2" in d and "write_MB_raw" in d:
                return (d["fetch_MB_x2"] + d["write_MB_raw"]) * 1e6, os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def measure_roofline(pipe, impl, first, n_frames, peak=None, fast=False):
    """Dominant-kernel roofline from live HIP-event timing of that kernel's launches (outside the fps region)."""
    import torch
    if impl == "fused":
        from geneface_amd.fused import profile_frames
        r = profile_frames(pipe, first, n_frames, FLOP_PER_HEAD_SAMPLE, peak or PEAK_F32_MFMA_TFLOPS)
        r["algorithmic_bytes_per_launch"] = r["samples_per_frame"] * BYTES_PER_HEAD_SAMPLE / 2 if r.get("samples_per_frame") else None
        if fast:
            # k_head_phase<true> keeps the matrix pipe busy for ~5 % of a round: it is bound by the table gathers (TA issue + L2 latency;
            # the tables are L2 / Infinity-Cache resident, so neither the MFMA nor the HBM peak prices it).  Report the algorithmic
            # gather rate; the guide gives no L2 gather peak to divide by, so no fraction is claimed for this secondary line.
            ms = r["kernel_ms_per_frame"]
            r.update({"bound": "l2-gather", "unit": "GB/s", "peak": None, "frac": None, "mfma_tflops": r["achieved"],
                      "achieved": r["samples_per_frame"] * BYTES_PER_HEAD_SAMPLE / (ms * 1e-3) / 1e9 if ms else None,
                      "traffic": None, "note": "fast tier: gather bound; algorithmic table bytes per second, no peak claimed"})
            return r
        r["traffic"], r["traffic_source"] = pmc_traffic()
        return r
    # impl == "ops": the dominant kernel is whichever rocBLAS SGEMM torch dispatches; it is not ours to time per launch.
    return {"bound": "mfma", "achieved": None, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
            "note": "impl=ops runs the MLPs through rocBLAS; per-kernel roofline is reported for impl=fused only"}


if __name__ == "__main__":
    main()
