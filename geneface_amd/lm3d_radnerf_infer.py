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
  * frames: contiguous block per rank (base_nerf_infer.py:150-155), one weight broadcast, FramePipeline per rank;
  * output: uint8 RGB frames, returned; with `tmp_imgs_dir` in `inp` every frame is also written as `<dir>/<idx:05d>.png`
    (the files base_nerf_infer.py:97-101 produces) by worker threads, off the render thread; `out_video_name` ending in .npy
    stores the stack.  The ffmpeg mux (:307) runs only when an ffmpeg binary exists (this image has none).
"""
import os

import numpy as np
import torch

from . import lm3d, utils
from .infer import FramePipeline, broadcast_model_, shard_range


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
            raise NotImplementedError("infer_bg_img_fname: image files need cv2, which this image does not ship")
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


class LM3d_RADNeRFInfer:
    def __init__(self, hparams: dict, model: torch.nn.Module = None, dataset: RADNeRFPoseSource = None, device=None):
        self.hparams = hparams
        self.device = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
        if dataset is None:
            path = os.path.join(hparams["binary_data_dir"], hparams["video_id"], "trainval_dataset.npy")
            dataset = RADNeRFPoseSource.from_file(path, hparams)
        self.dataset = dataset
        if model is None:
            model = self.build_model()
        self.model = model.to(self.device).eval()

    # ------------------------------------------------------------------ model (base_nerf_infer.py:72-79)
    def build_model(self):
        """RADNeRF(Torso) from the hparams + the newest `model_ckpt_steps_*.ckpt` of hparams['work_dir'] (ckpt_utils.py:7-66)."""
        from .radnerf import RADNeRF
        from .radnerf_torso import RADNeRFTorso
        torso = "torso" in str(self.hparams.get("task_cls", "")).lower()
        model = (RADNeRFTorso if torso else RADNeRF)(self.hparams)
        work_dir = self.hparams.get("work_dir", "")
        ckpts = sorted((f for f in os.listdir(work_dir) if f.startswith("model_ckpt_steps_") and f.endswith(".ckpt")),
                       key=lambda f: int(f[len("model_ckpt_steps_"):-len(".ckpt")])) if os.path.isdir(work_dir) else []
        if not ckpts:
            raise FileNotFoundError(f"no model_ckpt_steps_*.ckpt under work_dir={work_dir!r}")
        ck = torch.load(os.path.join(work_dir, ckpts[-1]), map_location="cpu")
        sd = ck["state_dict"]["model"] if "state_dict" in ck and "model" in ck["state_dict"] else ck
        model.load_state_dict(sd, strict=True)
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
    def forward_system(self, batches, rank: int = 0, world_size: int = 1, writer=None):
        """Renders this rank's contiguous block of frames -> uint8 [n, H, W, 3] (host).  With torch.distributed initialised the
        caller passes its rank / world size; weights are made identical to rank 0's with one broadcast."""
        T = len(batches)
        seq = {"cond_wins": np.stack([b["cond_wins"] for b in batches]).astype(np.float32),
               "poses": np.stack([b["pose44"] for b in batches]).astype(np.float32),
               "intrinsics": self.dataset.intrinsics, "bg_img": self.dataset.bg_img, "H": self.dataset.H, "W": self.dataset.W}
        broadcast_model_(self.model, src=0)
        lo, hi = shard_range(T, rank, world_size)
        pipe = FramePipeline(self.model, self.hparams, seq, self.device, frames=(lo, hi), impl="fused" if self.device.type == "cuda" else None)
        out = np.empty((hi - lo, self.dataset.H, self.dataset.W, 3), dtype=np.uint8)
        with torch.no_grad():
            for k, frame in pipe.stream(range(hi - lo)):
                out[k] = frame
                if writer is not None:
                    writer.submit(lo + k, out[k])
        return out

    def infer_once(self, inp: dict):
        samples = self.get_pose_from_ds(self.get_cond_from_input(inp))
        writer = None
        if inp.get("tmp_imgs_dir"):
            from .png import FrameWriter
            writer = FrameWriter(inp["tmp_imgs_dir"])
        frames = self.forward_system(samples, writer=writer)
        if writer is not None:
            writer.close()
        name = inp.get("out_video_name", "")
        if name.endswith(".npy"):
            os.makedirs(os.path.dirname(name) or ".", exist_ok=True)
            np.save(name, frames)
        elif name:
            import shutil
            import subprocess
            if shutil.which("ffmpeg") and writer is not None:   # base_nerf_infer.py:307 (video only; the wav is muxed when given)
                wav = inp.get("audio_source_name") or None
                cmd = ["ffmpeg", "-y", "-loglevel", "error", "-r", "25", "-i", os.path.join(inp["tmp_imgs_dir"], "%05d.png")]
                cmd += (["-i", wav] if wav and os.path.exists(wav) else []) + ["-c:v", "libx264", "-pix_fmt", "yuv420p", "-r", "25", name]
                subprocess.run(cmd, check=True)
            else:
                print(f"| {name}: no ffmpeg binary (or no tmp_imgs_dir): the frames are returned / written as PNG only")
        return frames

    @classmethod
    def example_run(cls, inp: dict, hparams: dict = None, config: str = None, hparams_str: str = "", config_root: str = None, **kw):
        """base_nerf_infer.py:271-300.  `config` is one of the reference's experiment files (e.g.
        egs/datasets/videos/May/lm3d_radnerf_torso.yaml, resolved through its `base_config` chain with `--hparams`-style overrides in
        `hparams_str`); without one the May values built into geneface_amd.hparams are used."""
        from .hparams import load_config, may_hparams
        if hparams is None:
            hparams = load_config(config, hparams_str, root=config_root) if config else may_hparams(True)
        return cls(hparams, **kw).infer_once(inp)
