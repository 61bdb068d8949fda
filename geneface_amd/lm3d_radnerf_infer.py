"""Entry point of the landmark-driven RAD-NeRF stage: the host side of inference/nerfs/lm3d_radnerf_infer.py.

Keeps the reference's call surface -- `LM3d_RADNeRFInfer(hparams, ...)`, `.get_cond_from_input(inp)`, `.get_pose_from_ds(samples)`,
`.forward_system(batches)`, `.infer_once(inp)`, classmethod `.example_run(inp)` and the `inp` keys (`cond_name`,
`out_video_name`, `audio_source_name`) of inference/nerfs/base_nerf_infer.py:261-300 -- on top of the MI355X frame pipeline:

  * landmarks: `np.load(cond_name)[0]` ([T,204], PostNet's `pred_lm3d/*.npy`) -> normalise by the dataset statistics, clamp,
    sequential EMA, windows (lm3d_radnerf_infer.py:45-85)  -> geneface_amd/lm3d.py, once per sequence on the host;
  * poses / intrinsics / background: the inference-relevant part of RADNeRFDataset.__init__
    (tasks/radnerfs/dataset_utils.py:40-90): AD-NeRF c2w -> ngp axes, 7-tap smoothing, focal/cx/cy, bg image.  The reference
    materialises rays for every frame up front (lm3d_radnerf_infer.py:18-32, 6 MB per frame on the GPU); here only the 3x4
    poses go to the device and rays are generated inside the kernel;
  * model: `build_model` = RADNeRF(Torso)Task.build_model + BaseNeRFInfer.build_nerf_task: for the torso task the head
    checkpoint of `head_model_dir` is loaded non-strictly first (tasks/radnerfs/radnerf_torso.py:30-38), then the newest
    `model_ckpt_steps_*.ckpt` of `work_dir` strictly (base_nerf_infer.py:72-79) -- legacy-pickle files with optimizer state
    and numpy scalars, as the reference's Trainer writes them (geneface_amd/ckpt_utils.py);
  * frames: `forward_system` fans out like base_nerf_infer.py:131-193: the GPU ids of CUDA_VISIBLE_DEVICES, one spawned process
    per GPU (torch.multiprocessing.spawn, NCCL == RCCL, 127.0.0.1), contiguous block per rank (:150-155), one weight
    broadcast instead of the DDP constructor's, every rank writes its own `<tmp_imgs_dir>/<idx:05d>.png` block, barrier,
    rank 0 muxes.  Under torchrun (process group already initialised) the caller's rank renders its block instead;
  * output: uint8 RGB frames, returned; with `tmp_imgs_dir` in `inp` every frame is also written as `<dir>/<idx:05d>.png`
    (the files base_nerf_infer.py:97-101 produces) by worker threads, off the render thread; `out_video_name` ending in .npy
    stores the stack.  The ffmpeg mux (:307) runs only when an ffmpeg binary exists (this image has none).
"""
import os

import numpy as np
import torch

from . import lm3d, utils
from .ckpt_utils import load_ckpt
from .infer import FramePipeline, broadcast_model_, shard_range


def load_background_image(path: str, H: int, W: int) -> np.ndarray:
    """`infer_bg_img_fname` naming a file (tasks/radnerfs/dataset_utils.py:69-74): the picture as float32 RGB [H, W, 3] in [0, 1], resized to the
    dataset's resolution when it differs (the reference: cv2.imread + INTER_AREA; here Pillow's box filter when shrinking -- the same area
    average for integer factors -- and bilinear when enlarging).  Without Pillow only PNGs of exactly H x W written as 8-bit RGB load."""
    try:
        from PIL import Image
    except ImportError:
        from .png import decode_rgb8
        with open(path, "rb") as fh:
            img = decode_rgb8(fh.read())
        if img.shape[:2] != (H, W):
            raise ValueError(f"{path}: {img.shape[1]}x{img.shape[0]}, the dataset is {W}x{H} (resizing needs Pillow)")
        return img.astype(np.float32) / 255.0
    with Image.open(path) as im:
        im = im.convert("RGB")
        if im.size != (W, H):
            shrink = im.size[0] >= W and im.size[1] >= H
            im = im.resize((W, H), Image.BOX if shrink else Image.BILINEAR)
        return np.asarray(im, dtype=np.float32) / 255.0


class RADNeRFPoseSource:
    """What inference needs from `RADNeRFDataset('trainval', training=False)`: smoothed ngp poses, intrinsics, background,
    landmark statistics.  `ds_dict` is the dict stored in data/binary/videos/<id>/trainval_dataset.npy
    (data_gen/nerf/binarizer.py:175-199,255): train_samples / val_samples (each with a 4x4 `c2w`), H, W, focal, cx, cy,
    bg_img (uint8 [H,W,3]), idexp_lm3d_mean, idexp_lm3d_std."""

    def __init__(self, ds_dict: dict, hparams: dict):
        samples = list(ds_dict["train_samples"]) + list(ds_dict["val_samples"])   # prefix 'trainval' (dataset_utils.py:50-51)
        self.H, self.W = int(ds_dict["H"]), int(ds_dict["W"])
        self.intrinsics = np.array([ds_dict["focal"], ds_dict["focal"], ds_dict["cx"], ds_dict["cy"]], dtype=np.float32)
        name = hparams.get("infer_bg_img_fname", "")
        if name == "":
            bg = np.asarray(ds_dict["bg_img"], dtype=np.float32) / 255.0
        elif name == "white":
            bg = np.ones((self.H, self.W, 3), dtype=np.float32)
        elif name == "black":
            bg = np.zeros((self.H, self.W, 3), dtype=np.float32)
        else:
            bg = load_background_image(name, self.H, self.W)
        self.bg_img = bg.reshape(-1, 3).astype(np.float32)
        self.idexp_lm3d_mean = np.asarray(ds_dict["idexp_lm3d_mean"], dtype=np.float32)
        self.idexp_lm3d_std = np.asarray(ds_dict["idexp_lm3d_std"], dtype=np.float32)
        poses = np.stack([utils.nerf_matrix_to_ngp(np.asarray(s["c2w"], dtype=np.float32), scale=hparams["camera_scale"],
                                                   offset=hparams["camera_offset"]) for s in samples]).astype(np.float32)
        if np.isnan(poses).any():
            raise ValueError("Found NaN in transform_matrix, please check the face_tracker process!")
        if hparams.get("infer_smooth_camera_path", True):
            poses = utils.smooth_camera_path(poses, kernel_size=hparams["infer_smooth_camera_path_kernel_size"])
        self.poses = poses.astype(np.float32)

    @classmethod
    def from_file(cls, path: str, hparams: dict):
        return cls(np.load(path, allow_pickle=True).tolist(), hparams)

    def __len__(self):
        return len(self.poses)


def _is_torso(hparams) -> bool:
    return "torso" in str(hparams.get("task_cls", "")).lower()


def _new_model(hparams):
    from .radnerf import RADNeRF
    from .radnerf_torso import RADNeRFTorso
    m = (RADNeRFTorso if _is_torso(hparams) else RADNeRF)(hparams)
    if hparams.get("render_precision"):      # not a reference key: this renderer's arithmetic tier ("fp32" default, "split", "fast"), so that spawned ranks follow the parent
        m.render_precision = hparams["render_precision"]
    return m


def _render_block(model, hparams, dataset, batches, device, rank, world_size, tmp_imgs_dir, pipeline_cls, collect=True):
    """One rank's share of base_nerf_infer.py:81-106 / :131-181: identical weights everywhere (one broadcast), the rank's contiguous block
    of frames, `<tmp_imgs_dir>/<global idx:05d>.png` per frame.  Returns (first global index, uint8 [n,H,W,3] or None)."""
    if float(hparams.get("infer_scale_factor", 1.0)) != 1.0:
        raise NotImplementedError("infer_scale_factor != 1.0: the frame loop renders at the dataset's resolution")
    T = len(batches)
    seq = {"cond_wins": np.stack([b["cond_wins"] for b in batches]).astype(np.float32),
           "poses": np.stack([b["pose44"] for b in batches]).astype(np.float32),
           "intrinsics": dataset.intrinsics, "bg_img": dataset.bg_img, "H": dataset.H, "W": dataset.W}
    broadcast_model_(model, src=0)
    lo, hi = shard_range(T, rank, world_size)
    writer = None
    if tmp_imgs_dir:
        from .png import FrameWriter
        os.makedirs(tmp_imgs_dir, exist_ok=True)
        # ~5 ms of deflate per 512x512 frame: four workers would cap the loop near 830 fps, below the split and fast tiers.  This rank's share
        # of the cores the process may really use (cgroup quota and affinity, not os.cpu_count(): png.effective_cpus), at most 32; stored
        # blocks instead of Z_RLE when the share is too small for a rank's frame rate (png.plan_writer; bench.py's PNG leg plans the same way)
        from .png import plan_writer
        workers, level = plan_writer(world_size, hparams.get("infer_png_zlib_level"))
        writer = FrameWriter(tmp_imgs_dir, workers=workers, level=level)
    pipe = pipeline_cls(model, hparams, seq, device, frames=(lo, hi), impl="fused" if torch.device(device).type == "cuda" else None)
    out = np.empty((hi - lo, dataset.H, dataset.W, 3), dtype=np.uint8) if collect else None
    try:
        with torch.no_grad():
            for k, frame in pipe.stream(range(hi - lo)):
                if out is not None:
                    out[k] = frame
                if writer is not None:
                    writer.submit(lo + k, frame)      # FrameWriter.submit copies the (reused) pinned buffer
    finally:
        if writer is not None:
            writer.close()      # joins the native threads and raises the writer's first error, also when the loop above failed
    return lo, out


def _spawned_rank(rank, world_size, hparams, dataset, state_dict, batches, tmp_imgs_dir, pipeline_cls, use_cuda, port, block_dir, share_gpu=False):
    """Process `rank` of forward_system's fan-out (base_nerf_infer.py:131-181): own GPU, own replica, own block of frames.
    share_gpu (inp["ranks_share_gpu"], a TEST mode for a box with one GPU, like bench.py --ranks-share-gpu): every rank uses cuda:0 and gloo
    carries the collectives (RCCL refuses two ranks on one device) -- the same replicas, broadcast, blocks and files, no scaling meaning."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"                     # init_ddp_connection, :108-113
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if use_cuda and share_gpu:
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
    elif use_cuda:
        torch.cuda.set_device(rank)
        device = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        model = _new_model(hparams)
        if rank == 0:
            model.load_state_dict(state_dict, strict=True)      # the other replicas receive the weights over RCCL (broadcast_model_)
        model = model.to(device).eval()
        dist.barrier()
        lo, out = _render_block(model, hparams, dataset, batches, device, rank, world_size, tmp_imgs_dir, pipeline_cls, collect=bool(block_dir))
        if block_dir:
            np.save(os.path.join(block_dir, f"block_{lo:07d}.npy"), out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


class LM3d_RADNeRFInfer:
    pipeline_cls = FramePipeline    # what renders a rank's block (tests substitute a CPU stand-in)

    def __init__(self, hparams: dict, model: torch.nn.Module = None, dataset: RADNeRFPoseSource = None, device=None):
        self.hparams = hparams
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        if dataset is None:
            path = os.path.join(hparams["binary_data_dir"], hparams["video_id"], "trainval_dataset.npy")
            dataset = RADNeRFPoseSource.from_file(path, hparams)
        self.dataset = dataset
        # base_nerf_infer.py:63-69: the GPUs to use are the ids listed in CUDA_VISIBLE_DEVICES; more than one -> one process per GPU
        self.all_gpu_ids = [int(x) for x in os.environ.get("CUDA_VISIBLE_DEVICES", "").split(",") if x != ""]
        self.num_gpus = len(self.all_gpu_ids)
        self.use_ddp = self.num_gpus > 1
        self.proc_rank = 0
        self.global_step = None
        if model is None:
            model = self.build_model()
        self.model = model.to(self.device).eval()

    # ------------------------------------------------------------------ model (radnerf_torso.py:30-38 + base_nerf_infer.py:72-79)
    def build_model(self):
        """RADNeRF(Torso) from the hparams and the reference's checkpoints.  Torso task: a RADNeRF is filled from the newest checkpoint
        of `head_model_dir` and copied in with strict=False (the torso keys stay at their init), exactly as RADNeRFTorsoTask.build_model;
        then, for both tasks, the newest `model_ckpt_steps_*.ckpt` of `work_dir` is loaded strictly (build_nerf_task)."""
        model = _new_model(self.hparams)
        if _is_torso(self.hparams):
            from .radnerf import RADNeRF
            head_model = RADNeRF(self.hparams)
            load_ckpt(head_model, self.hparams["head_model_dir"])
            print(f"Loaded Head Model from {self.hparams['head_model_dir']}")
            model.load_state_dict(head_model.state_dict(), strict=False)
            del head_model
        ckpt = load_ckpt(model, self.hparams["work_dir"], "model")
        self.global_step = ckpt["global_step"]
        return model

    # ------------------------------------------------------------------ condition (lm3d_radnerf_infer.py:34-86)
    def get_cond_from_input(self, inp: dict):
        assert inp["cond_name"].endswith(".npy")
        lm3d_arr = np.load(inp["cond_name"])[0]                     # [T, 204]
        norm = lm3d.normalize_and_smooth(lm3d_arr, self.dataset.idexp_lm3d_mean, self.dataset.idexp_lm3d_std,
                                         self.hparams["infer_lm3d_clamp_std"])
        wins = lm3d.cond_windows(norm, self.hparams["cond_win_size"], self.hparams["smo_win_size"])   # [T, smo, win, 204]
        return [{"cond": norm[i:i + 1], "cond_wins": wins[i]} for i in range(norm.shape[0])]

    # ------------------------------------------------------------------ poses (lm3d_radnerf_infer.py:18-32)
    def get_pose_from_ds(self, samples):
        if len(samples) > len(self.dataset):
            raise IndexError(f"{len(samples)} landmark frames but the pose source has {len(self.dataset)} (the reference indexes dataset[i])")
        for i, s in enumerate(samples):
            s["pose44"], s["idx"], s["H"], s["W"] = self.dataset.poses[i], i, self.dataset.H, self.dataset.W
        return samples

    # ------------------------------------------------------------------ frame loop (base_nerf_infer.py:81-193)
    def forward_system(self, batches, world_size: int = None, tmp_imgs_dir: str = None, collect: bool = True):
        """base_nerf_infer.py:182-193.  Three ways in, one block partition (:150-155):
          * a process group already exists (torchrun): this process renders ITS block; rank 0 returns the whole sequence (the blocks are
            gathered through per-rank files, as in the spawn path), the other ranks their own block;
          * `world_size` (default: the number of GPU ids in CUDA_VISIBLE_DEVICES) > 1: spawn one process per GPU, each renders and
            writes its block, the parent returns the concatenated frames (collect=True) or the image directory, like the reference;
          * otherwise: this process renders everything.
        Returns uint8 [n, H, W, 3] (host) of the frames this call is responsible for, or `tmp_imgs_dir` when collect=False."""
        import torch.distributed as dist
        tmp_imgs_dir = tmp_imgs_dir if tmp_imgs_dir is not None else getattr(self, "inp", {}).get("tmp_imgs_dir")
        shard = getattr(self, "inp", {}).get("shard")
        if shard is not None:
            # inp["shard"] = (rank, world_size): this process renders exactly the block rank `rank` of a `world_size`-GPU job would (:150-155) and
            # nothing else -- for launchers that start every rank as an independent process (no process group; every replica loads the
            # same checkpoint), and for measuring one GPU's share of a sharded sequence on one GPU.
            rank, world = int(shard[0]), int(shard[1])
            if not 0 <= rank < world:
                raise ValueError(f"shard {shard}: need 0 <= rank < world_size")
            _, out = _render_block(self.model, self.hparams, self.dataset, batches, self.device, rank, world, tmp_imgs_dir, self.pipeline_cls, collect)
            return out if collect else tmp_imgs_dir
        if dist.is_available() and dist.is_initialized():
            self.proc_rank, world = dist.get_rank(), dist.get_world_size()
            lo, out = _render_block(self.model, self.hparams, self.dataset, batches, self.device, self.proc_rank, world,
                                    tmp_imgs_dir, self.pipeline_cls, collect)
            if not collect or world == 1:
                dist.barrier()
                return out if collect else tmp_imgs_dir
            # Same contract as the spawn path: rank 0 gets the WHOLE sequence back (it is the rank that post-processes, base_nerf_infer.py:267).
            # One node by design, so the blocks travel as .npy files in a directory rank 0 names (no pickling of 300 MB blocks through the
            # collective layer); the other ranks return their own block.
            import shutil
            import socket
            import tempfile
            hosts = [None] * world
            dist.all_gather_object(hosts, socket.gethostname())
            if len(set(hosts)) != 1:
                raise RuntimeError(f"infer_once under torchrun gathers the ranks' blocks through a local directory: one node only, got hosts {sorted(set(hosts))} "
                                   "(render with inp['shard'] per node, or collect=False and read the PNGs)")
            box = [None, None]
            if self.proc_rank == 0:
                try:
                    box[0] = tempfile.mkdtemp(prefix="gf_blocks_")
                except OSError as e:          # the other ranks must not wait at a barrier for a directory that never came
                    box[1] = repr(e)
            dist.broadcast_object_list(box, src=0)
            if box[1] is not None:
                raise RuntimeError(f"rank 0 could not create the block directory: {box[1]}")
            err = None
            try:
                np.save(os.path.join(box[0], f"block_{lo:07d}.npy"), out)
            except OSError as e:
                err = repr(e)
            errs = [None] * world
            dist.all_gather_object(errs, err)     # (also the barrier behind the writes: every rank learns of any rank's failure instead of hanging)
            try:
                if any(errs):
                    raise RuntimeError(f"writing the frame blocks failed: {[e for e in errs if e]}")
                if self.proc_rank == 0:
                    out = np.concatenate([np.load(os.path.join(box[0], f)) for f in sorted(os.listdir(box[0]))], axis=0)
            finally:
                dist.barrier()
                if self.proc_rank == 0:
                    shutil.rmtree(box[0], ignore_errors=True)
            return out
        world = int(world_size) if world_size is not None else (self.num_gpus if self.use_ddp else 1)
        if world <= 1:
            _, out = _render_block(self.model, self.hparams, self.dataset, batches, self.device, 0, 1, tmp_imgs_dir, self.pipeline_cls, collect)
            return out if collect else tmp_imgs_dir
        import shutil
        import tempfile
        import torch.multiprocessing as mp
        use_cuda = self.device.type == "cuda"
        share_gpu = bool(getattr(self, "inp", {}).get("ranks_share_gpu", False))
        if use_cuda and not share_gpu and torch.cuda.device_count() < world:
            raise RuntimeError(f"forward_system: {world} ranks requested but {torch.cuda.device_count()} GPUs are visible")
        block_dir = tempfile.mkdtemp(prefix="gf_blocks_") if collect else None
        state_dict = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        port = int(os.environ.get("MASTER_PORT", "12345"))          # the reference hard-codes 12345 (:112)
        try:
            mp.spawn(_spawned_rank, nprocs=world, join=True,
                     args=(world, self.hparams, self.dataset, state_dict, batches, tmp_imgs_dir, self.pipeline_cls, use_cuda, port, block_dir, share_gpu))
            if not collect:
                return tmp_imgs_dir
            blocks = [np.load(os.path.join(block_dir, f)) for f in sorted(os.listdir(block_dir))]
            return np.concatenate(blocks, axis=0)
        finally:
            if block_dir:
                shutil.rmtree(block_dir, ignore_errors=True)

    def infer_once(self, inp: dict):
        self.inp = inp
        samples = self.get_pose_from_ds(self.get_cond_from_input(inp))
        # inp["return_frames"] = False: the PNG files are the output (what the reference's loop produces, base_nerf_infer.py:97-101) and the
        # stacked uint8 result -- 2.4 GB of first-touch host memory for a 3000-frame sequence -- is not assembled
        collect = bool(inp.get("return_frames", True)) or not inp.get("tmp_imgs_dir")
        frames = self.forward_system(samples, collect=collect)
        if self.proc_rank != 0:          # base_nerf_infer.py:267: only rank 0 post-processes
            return frames
        name = inp.get("out_video_name", "")
        if name.endswith(".npy") and not isinstance(frames, np.ndarray):
            raise ValueError("out_video_name ends in .npy but return_frames is False: there is no frame stack to store")
        if name.endswith(".npy"):
            os.makedirs(os.path.dirname(name) or ".", exist_ok=True)
            np.save(name, frames)
        elif name:
            out = self.postprocess_output(frames)
            if out:
                print(f"The synthesized video is saved at {out}")
        return frames

    # ------------------------------------------------------------------ IO (base_nerf_infer.py:255-259, 303-317)
    def save_wav16k(self, inp: dict):
        """base_nerf_infer.py:309-317: the audio source resampled to a 16 kHz wav beside it (ffmpeg), remembered as `self.wav16k_name`.
        Returns the name, or None when there is no audio source or no ffmpeg binary (this image ships none; the frames are still rendered)."""
        import shutil
        import subprocess
        source_name = inp.get("audio_source_name") or ""
        self.wav16k_name = None
        if not source_name:
            return None
        supported_types = (".wav", ".mp3", ".mp4", ".avi")
        assert source_name.endswith(supported_types), f"Now we only support {','.join(supported_types)} as audio source!"
        if not shutil.which("ffmpeg") or not os.path.exists(source_name):
            return None
        wav16k_name = source_name[:-4] + "_16k.wav"
        subprocess.run(["ffmpeg", "-i", source_name, "-v", "quiet", "-f", "wav", "-ar", "16000", wav16k_name, "-y"], check=True)
        print(f"Saved 16khz wav file to {wav16k_name}.")
        self.wav16k_name = wav16k_name
        return wav16k_name

    @classmethod
    def save_mp4(cls, img_dir: str, wav_name, out_name: str):
        """base_nerf_infer.py:305-306, the same ffmpeg invocation (25 fps, libx264, yuv420p, 2000k, -shortest against the audio track)."""
        import subprocess
        cmd = ["ffmpeg", "-i", os.path.join(img_dir, "%5d.png")]
        if wav_name:
            cmd += ["-i", wav_name, "-shortest"]
        cmd += ["-v", "quiet", "-c:v", "libx264", "-pix_fmt", "yuv420p", "-b:v", "2000k", "-r", "25", "-strict", "-2", "-y", out_name]
        os.makedirs(os.path.dirname(out_name) or ".", exist_ok=True)
        subprocess.run(cmd, check=True)

    def postprocess_output(self, output):
        """base_nerf_infer.py:255-259: mux `<tmp_imgs_dir>/%5d.png` (+ the 16 kHz audio) into `out_video_name`.  Without an ffmpeg binary, or
        without an image directory, the PNG frames / the returned stack are the output and None is returned."""
        import shutil
        tmp_imgs_dir, name = self.inp.get("tmp_imgs_dir"), self.inp.get("out_video_name", "")
        if not tmp_imgs_dir or not shutil.which("ffmpeg"):
            print(f"| {name}: no ffmpeg binary (or no tmp_imgs_dir): the frames are returned / written as PNG only")
            return None
        if getattr(self, "wav16k_name", None) is None:
            self.save_wav16k(self.inp)
        self.save_mp4(tmp_imgs_dir, self.wav16k_name, name)
        return name

    @classmethod
    def example_run(cls, inp: dict, hparams: dict = None, config: str = None, hparams_str: str = "", config_root: str = None, **kw):
        """base_nerf_infer.py:271-300.  `config` is one of the reference's experiment files (e.g.
        egs/datasets/videos/May/lm3d_radnerf_torso.yaml, resolved through its `base_config` chain with `--hparams`-style overrides in
        `hparams_str`); without one the May values built into geneface_amd.hparams are used."""
        from .hparams import load_config, may_hparams
        if hparams is None:
            hparams = load_config(config, hparams_str, root=config_root) if config else may_hparams(True)
        inp = dict(inp)
        for key in ("cond_name", "audio_source_name", "out_video_name"):          # base_nerf_infer.py:284-291: hparams override the inputs
            if hparams.get("infer_" + key, "") != "":
                inp[key] = hparams["infer_" + key]
        name = inp.get("out_video_name", "")
        if name and not name.endswith(".npy") and not inp.get("tmp_imgs_dir"):    # :292-298: <out_dir>/tmp_imgs/<video name>
            inp["tmp_imgs_dir"] = os.path.join(os.path.dirname(name), "tmp_imgs", os.path.basename(name)[:-4])
        return cls(hparams, **kw).infer_once(inp)
