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


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else -10 * np.log10(mse)
