"""Shared test plumbing: synthetic model + frame inputs as CPU tensors (oracle side)."""
import functools

import numpy as np
import torch

from geneface_amd import hparams as HP
from geneface_amd import synthetic as S
from oracle import radnerf_ref as R


@functools.lru_cache(maxsize=4)
def model_fixture(torso: bool, seed: int = 0):
    hp = HP.may_hparams(torso)
    return hp, S.make_state_dict(hp, torso, seed)


@functools.lru_cache(maxsize=8)
def sequence(T: int, H: int, W: int, seed: int = 0):
    return S.make_sequence(T, H, W, HP.may_hparams(True), seed)


def frame_inputs(seq, idx):
    """CPU tensors in the layout `run_model` hands to render()."""
    H, W = seq["H"], seq["W"]
    pose = torch.from_numpy(seq["poses"][idx:idx + 1])
    rays_o, rays_d = R.get_rays(pose, seq["intrinsics"], H, W)
    return dict(rays_o=rays_o.contiguous(), rays_d=rays_d.contiguous(), bg_coords=R.get_bg_coords(H, W),
                cond=torch.from_numpy(seq["cond_wins"][idx]), pose6=R.convert_poses(pose), pose44=pose,
                bg=torch.from_numpy(seq["bg_img"]).view(1, -1, 3))


def pipe_inputs(pipe, idx, kernel_rays=True):
    """The bits a FramePipeline feeds frame `idx` with, copied to the host in frame_inputs' layout -- rays (kernel_rays: the ones the frame
    loop generates for itself, gf_pinhole_rays = the device function k_frame_init runs; else torch's get_rays on the GPU), background
    coordinates, euler pose, window, background: what the oracle must be fed to arbitrate a pose-mode pixel.  EVERY input comes from the
    device: torch on the GPU divides by a scalar as a multiplication by its reciprocal, so even get_bg_coords differs from the CPU's in the last
    ulp -- and the torso mask `occ > 0` (radnerf_torso.py:167-172) is a step function of those coordinates wherever a bilinear sample lands on
    a node of the 128 x 128 occupancy grid (every pixel of a 128 x 128 frame).  GPU tests only."""
    s = pipe.kernel_sample(idx) if kernel_rays else pipe.sample(idx)
    c = lambda t: t.detach().cpu().contiguous()
    return dict(rays_o=c(s["rays_o"]), rays_d=c(s["rays_d"]), bg_coords=c(s["bg_coords"]), cond=c(s["cond_wins"]), pose6=c(s["pose"]), bg=c(s["bg_img"]))


def oracle_u8(sd, hp, fi, torso=True, **kw):
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso, **kw)
    return (ref["rgb_map"].reshape(-1, 3) * 255).to(torch.uint8)


def oracle_threads(limit=16):
    """The oracle's 128-row torch layers and OpenMP loops do not scale to a two-socket host (6 s per 512x512 frame at 128 threads, ~1.5 s
    at 16): cap the pools for the tests that render many oracle frames."""
    import ctypes
    import os
    t = max(1, min(limit, os.cpu_count() or limit))
    torch.set_num_threads(t)
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(t)
    except OSError:
        pass
    return t


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else -10 * np.log10(mse)


class StubPipeline:
    """CPU stand-in for geneface_amd.infer.FramePipeline in world-size-2 gloo tests of the entry point's fan-out: same constructor and
    `stream`, but a "frame" is a flat image whose bytes encode (global frame index, a checksum of the replica's weights, the rank's first
    cond value), so the test can tell which rank rendered what from which weights.  Importable from spawned processes."""

    def __init__(self, model, hp, seq, device, frames=None, impl=None, **kw):
        self.H, self.W = int(seq["H"]), int(seq["W"])
        self.lo, self.hi = frames
        self.cond = np.asarray(seq["cond_wins"])[self.lo:self.hi]
        w = next(p for n, p in model.named_parameters() if n.endswith("sigma_net.net.0.weight"))
        self.wsum = int(abs(float(w.detach().double().sum())) * 1000) % 251
        self._buf = np.zeros((self.H, self.W, 3), dtype=np.uint8)      # ONE reused buffer, like the pinned slots of the real pipeline

    def stream(self, indices):
        for k in indices:
            g = self.lo + k
            self._buf[...] = 0
            self._buf[..., 0] = g % 256
            self._buf[..., 1] = self.wsum
            self._buf[..., 2] = int(abs(float(self.cond[k].reshape(-1)[0])) * 100) % 256
            yield k, self._buf
