"""The oracle against the committed golden vectors (made by the REFERENCE's own Python, see
tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as GI
from helpers import frame_inputs, model_fixture, psnr, sequence
from oracle import radnerf_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLD, "ops.npz"))


@pytest.mark.parametrize("D,enc,interp", GI.GRID_CASES)
def test_grid_encoder_golden(ops, oracle_lib, D, enc, interp):
    tag, x, table, off = GI.grid_case(D, enc, interp)
    pls = np.exp2(np.log2(2048 / 16) / 15)
    y = R.grid_encode((torch.from_numpy(x) + 1) / 2, torch.from_numpy(table), torch.from_numpy(off), pls, 16,
                      {"hashgrid": 0, "tiledgrid": 1}[enc], False, {"linear": 0, "smoothstep": 1}[interp])
    assert np.array_equal(y.numpy(), ops[tag + "_y"])  # same C kernels underneath: bit-exact
    assert np.count_nonzero(y[1].numpy()) > 0 and np.count_nonzero(y[0].numpy()) > 0  # +-1 are in range


def test_sh_freq_golden(ops, oracle_lib):
    assert np.array_equal(R.sh_encode(torch.from_numpy(GI.sh_dirs())).numpy(), ops["sh_y"])
    for dim, deg in ((6, 4), (2, 10)):
        assert np.array_equal(R.freq_encode(torch.from_numpy(GI.freq_case(dim, deg)), deg).numpy(), ops[f"freq_{dim}_{deg}_y"])


def test_march_composite_golden(ops, oracle_lib):
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(2, 32, 32), 0)
    ro, rd = fi["rays_o"].view(-1, 3), fi["rays_d"].view(-1, 3)
    # our get_rays restatement against the reference's get_rays output
    assert np.allclose(ro.numpy(), ops["rays_o"], atol=0) and np.allclose(rd.numpy(), ops["rays_d"], atol=1e-7)
    ro, rd = torch.from_numpy(ops["rays_o"]), torch.from_numpy(ops["rays_d"])
    nears, fars = R.near_far_from_aabb(ro, rd, sd["aabb_infer"], hp["min_near"])
    assert np.array_equal(nears.numpy(), ops["march_nears"]) and np.array_equal(fars.numpy(), ops["march_fars"])
    N = ro.shape[0]
    alive = torch.arange(N, dtype=torch.int32)
    rays_t = nears.clone()
    xyzs, dirs, deltas = R.march_rays(N, 3, alive, rays_t, ro, rd, 1.0, sd["density_bitfield"], 1, 128, nears, fars, 128,
                                      hp["dt_gamma"], hp["max_steps"])
    assert xyzs.shape[0] == ops["march_xyzs"].shape[0]  # padding rule of raymarching.py:379-381
    assert np.array_equal(xyzs.numpy(), ops["march_xyzs"]) and np.array_equal(deltas.numpy(), ops["march_deltas"])
    sig, rgb = (torch.from_numpy(a) for a in GI.composite_inputs(xyzs.shape[0]))
    ws, dep, img = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    R.RM.composite_rays(N, 3, 1e-4, alive, rays_t, sig, rgb, deltas, ws, dep, img)
    for got, key in ((alive, "comp_alive"), (rays_t, "comp_rays_t"), (ws, "comp_ws"), (dep, "comp_depth"), (img, "comp_image")):
        assert np.array_equal(got.numpy(), ops[key]), key


@pytest.mark.parametrize("torso", [False, True])
@pytest.mark.parametrize("size,idx", [(64, 1), (96, 3)])
def test_frame_golden(oracle_lib, torso, size, idx):
    """oracle/radnerf_ref.render == the reference's RADNeRF.render / RADNeRFTorso.render on the same inputs."""
    gold = np.load(os.path.join(GOLD, f"frame_{'torso' if torso else 'head'}_{size}.npz"))
    hp, sd = model_fixture(torso)
    fi = frame_inputs(sequence(4, size, size), idx)
    assert np.allclose(fi["pose6"].numpy(), gold["pose6"], atol=1e-6)
    assert abs(float(fi["rays_d"].double().abs().sum()) - float(gold["rays_d_checksum"])) < 1e-3
    cf = R.cal_cond_feat(sd, hp, fi["cond"])
    assert np.allclose(cf.numpy(), gold["cond_feat"], atol=1e-6)
    out = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso)
    # fp32 restatement of fp32 code: differences come only from sum order inside torch ops (amplified by exp())
    assert np.abs(out["rgb_map"].numpy() - gold["rgb_map"]).max() < 3e-4
    assert psnr(out["rgb_map"], torch.from_numpy(gold["rgb_map"])) > 70
    assert np.abs(out["depth_map"].numpy() - gold["depth_map"]).max() < 1e-3
    if torso:
        assert np.abs(out["torso_alpha_map"].numpy() - gold["torso_alpha_map"]).max() < 1e-5
        assert np.abs(out["torso_rgb_map"].numpy() - gold["torso_rgb_map"]).max() < 1e-5
        assert np.abs(out["deform"].numpy() - gold["deform"]).max() < 1e-5


VARIANT_GOLD = ("hash", "hash_smoothstep", "smoothstep", "head_aware_coin_heads", "head_aware_coin_tails", "audio")


def variant_case(tag, size=48, idx=2):
    """(hp, sd, frame inputs, head-aware coin outcome, golden arrays) of tests/golden/frame_variant_<tag>_48.npz -- one head+torso frame of
    another RAD-NeRF configuration the reference ships, rendered by the reference's OWN Python built with those hparams (make_golden.py::
    golden_variants).  Shared with the GPU test of the product (tests/test_gpu_render.py)."""
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    name = tag.split("_coin_")[0]
    hp = HP.variant_hparams(name, True)
    seed = 1000 if name == "audio" else 0
    sd = S.make_state_dict(hp, True, seed=seed)
    fi = frame_inputs(S.make_sequence(4, size, size, hp, seed=seed), idx)
    gold = np.load(os.path.join(GOLD, f"frame_variant_{tag}_{size}.npz"))
    assert np.allclose(fi["pose6"].numpy(), gold["pose6"], atol=1e-6)
    assert abs(float(fi["rays_d"].double().abs().sum()) - float(gold["rays_d_checksum"])) < 1e-3
    return hp, sd, fi, tag.endswith("_coin_heads"), gold


@pytest.mark.parametrize("tag", VARIANT_GOLD)
def test_variant_frame_golden(oracle_lib, tag):
    """The oracle's restatement of the branches only the other shipped configurations reach -- hash / smoothstep flags down to the grid
    kernels, the head-colour encoder of torso_head_aware with both coin outcomes (radnerf_torso.py:36-46,68-74,175-179), the strided-conv
    AudioNet of the audio-driven config (cond_encoder.py:14-41) -- against the reference's own Python on the same inputs."""
    hp, sd, fi, branch, gold = variant_case(tag)
    assert np.allclose(R.cal_cond_feat(sd, hp, fi["cond"]).numpy(), gold["cond_feat"], atol=1e-6)
    out = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True, head_aware_branch=branch)
    assert np.abs(out["rgb_map"].numpy() - gold["rgb_map"]).max() < 3e-4
    assert psnr(out["rgb_map"], torch.from_numpy(gold["rgb_map"])) > 70
    assert np.abs(out["depth_map"].numpy() - gold["depth_map"]).max() < 1e-3
    assert np.abs(out["torso_alpha_map"].numpy() - gold["torso_alpha_map"]).max() < 1e-5
    assert np.abs(out["torso_rgb_map"].numpy() - gold["torso_rgb_map"].reshape(out["torso_rgb_map"].shape)).max() < 1e-5
    assert np.abs(out["deform"].numpy() - gold["deform"]).max() < 1e-5
    if tag == "head_aware_coin_heads":       # the two outcomes of the coin are different pictures
        other = np.load(os.path.join(GOLD, "frame_variant_head_aware_coin_tails_48.npz"))
        assert np.abs(other["torso_alpha_map"] - gold["torso_alpha_map"]).max() > 1e-3
