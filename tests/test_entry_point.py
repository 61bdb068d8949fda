"""Host side of the landmark-driven RAD-NeRF entry point (geneface_amd/lm3d_radnerf_infer.py) -- the reference's
inference/nerfs/lm3d_radnerf_infer.py + base_nerf_infer.py frame loop.  CPU tests cover the wire formats either side of
the path (pred_lm3d .npy in, trainval_dataset.npy dict schema); the -m gpu test runs infer_once end to end."""
import os

import numpy as np
import pytest
import torch

from geneface_amd import hparams as HP
from geneface_amd import lm3d, utils
from geneface_amd import synthetic as S
from geneface_amd.lm3d_radnerf_infer import LM3d_RADNeRFInfer, RADNeRFPoseSource


def _ds_dict(T=9, H=64, W=64, seed=3):
    """A trainval_dataset.npy-shaped dict (data_gen/nerf/binarizer.py:175-199) with AD-NeRF convention c2w matrices."""
    rng = np.random.default_rng(seed)
    ngp = S.make_poses(T)                                 # ngp-axes poses of the synthetic orbit camera
    scale = 4.0
    c2w = []
    for p in ngp:                                         # invert nerf_matrix_to_ngp: rows (y,z,x) <- (x,y,z), t / scale
        m = np.eye(4, dtype=np.float32)
        m[0, :3], m[1, :3], m[2, :3] = [p[2, 0], -p[2, 1], -p[2, 2]], [p[0, 0], -p[0, 1], -p[0, 2]], [p[1, 0], -p[1, 1], -p[1, 2]]
        m[0, 3], m[1, 3], m[2, 3] = p[2, 3] / scale, p[0, 3] / scale, p[1, 3] / scale
        c2w.append(m)
    K = S.intrinsics(H, W)
    samples = [{"c2w": m, "idx": i} for i, m in enumerate(c2w)]
    return {"train_samples": samples[:T - 2], "val_samples": samples[T - 2:], "H": H, "W": W, "focal": float(K[0]), "cx": float(K[2]),
            "cy": float(K[3]), "bg_img": (S.make_bg_img(H, W).reshape(H, W, 3) * 255).astype(np.uint8),
            "idexp_lm3d_mean": rng.normal(size=(1, 68, 3)).astype(np.float32) * 0.1,
            "idexp_lm3d_std": (1 + 0.1 * rng.random(size=(1, 68, 3))).astype(np.float32)}, ngp


def test_pose_source_follows_dataset_init():
    hp = HP.may_hparams(True)
    dd, ngp = _ds_dict()
    src = RADNeRFPoseSource(dd, hp)
    assert len(src) == 9 and src.H == 64 and src.W == 64
    raw = np.stack([utils.nerf_matrix_to_ngp(s["c2w"], scale=hp["camera_scale"], offset=hp["camera_offset"])
                    for s in dd["train_samples"] + dd["val_samples"]])
    np.testing.assert_allclose(raw[:, :3, :], ngp[:, :3, :], atol=1e-5)          # the c2w round trip of the fixture
    want = utils.smooth_camera_path(raw.astype(np.float32), kernel_size=hp["infer_smooth_camera_path_kernel_size"])
    np.testing.assert_allclose(src.poses, want, atol=1e-6)
    np.testing.assert_allclose(src.intrinsics, [dd["focal"], dd["focal"], dd["cx"], dd["cy"]])
    assert src.bg_img.shape == (64 * 64, 3) and 0.0 <= src.bg_img.min() and src.bg_img.max() <= 1.0
    hp2 = dict(hp, infer_bg_img_fname="white", infer_smooth_camera_path=False)
    src2 = RADNeRFPoseSource(dd, hp2)
    assert float(src2.bg_img.min()) == 1.0
    np.testing.assert_allclose(src2.poses, raw, atol=1e-6)


def test_cond_from_input_and_pose_lookup(tmp_path):
    hp = HP.may_hparams(True)
    dd, _ = _ds_dict()
    src = RADNeRFPoseSource(dd, hp)
    inf = LM3d_RADNeRFInfer.__new__(LM3d_RADNeRFInfer)     # host logic only: no model, no device
    inf.hparams, inf.dataset = hp, src
    T = 7
    raw = S.make_landmarks(T).astype(np.float32)           # [T, 204]
    path = os.path.join(tmp_path, "zozo.npy")
    np.save(path, raw[None])                               # PostnetInfer writes [1, T, 204] (postnet_infer.py:87-99)
    samples = inf.get_cond_from_input({"cond_name": path})
    assert len(samples) == T
    norm = lm3d.normalize_and_smooth(raw, src.idexp_lm3d_mean, src.idexp_lm3d_std, hp["infer_lm3d_clamp_std"])
    wins = lm3d.cond_windows(norm, hp["cond_win_size"], hp["smo_win_size"])
    for i, s in enumerate(samples):
        assert s["cond_wins"].shape == (hp["smo_win_size"], hp["cond_win_size"], 204)
        np.testing.assert_array_equal(s["cond_wins"], wins[i])
        np.testing.assert_array_equal(s["cond_wins"][hp["smo_win_size"] // 2, 0], norm[i])   # centre of the window = the frame
    samples = inf.get_pose_from_ds(samples)
    np.testing.assert_array_equal(samples[3]["pose44"], src.poses[3])
    with pytest.raises(IndexError):
        inf.get_pose_from_ds([{} for _ in range(len(src) + 1)])
    with pytest.raises(AssertionError):
        inf.get_cond_from_input({"cond_name": "x.txt"})


@pytest.mark.gpu
def test_infer_once_end_to_end_vs_oracle(tmp_path):
    from helpers import psnr
    from geneface_amd.radnerf_torso import RADNeRFTorso
    from oracle import radnerf_ref as R
    hp = HP.may_hparams(True)
    sd = S.make_state_dict(hp, True)
    model = RADNeRFTorso(hp)
    model.load_state_dict(sd, strict=True)
    dd, _ = _ds_dict(T=6, H=64, W=64)
    inf = LM3d_RADNeRFInfer(hp, model=model, dataset=RADNeRFPoseSource(dd, hp), device="cuda:0")
    raw = S.make_landmarks(5).astype(np.float32)
    cond_path, out_path = os.path.join(tmp_path, "lm.npy"), os.path.join(tmp_path, "out", "frames.npy")
    np.save(cond_path, raw[None])
    frames = inf.infer_once({"cond_name": cond_path, "out_video_name": out_path, "audio_source_name": "", "tmp_imgs_dir": os.path.join(tmp_path, "imgs")})
    from geneface_amd.png import decode_rgb8
    for i in range(5):
        np.testing.assert_array_equal(decode_rgb8(open(os.path.join(tmp_path, "imgs", f"{i:05d}.png"), "rb").read()), frames[i])
    assert frames.shape == (5, 64, 64, 3) and frames.dtype == np.uint8
    np.testing.assert_array_equal(np.load(out_path), frames)
    samples = inf.get_pose_from_ds(inf.get_cond_from_input({"cond_name": cond_path}))
    bgc, bg = R.get_bg_coords(64, 64), torch.from_numpy(inf.dataset.bg_img).view(1, -1, 3)
    for i in (0, 2, 4):
        pose = torch.from_numpy(samples[i]["pose44"][None])
        ro, rd = R.get_rays(pose, inf.dataset.intrinsics, 64, 64)
        ref = R.render(sd, hp, ro, rd, torch.from_numpy(samples[i]["cond_wins"]), bgc, R.convert_poses(pose), bg, torso=True)
        ref8 = (ref["rgb_map"] * 255).view(64, 64, 3).to(torch.uint8)
        got = torch.from_numpy(frames[i])
        assert ((got.int() - ref8.int()).abs() <= 1).float().mean().item() > 0.995
        assert psnr(got.float() / 255, ref8.float() / 255) > 45


def test_png_writer_roundtrip(tmp_path):
    """The %05d.png frame files of base_nerf_infer.py:97-101, written without cv2: valid PNG signature/chunks/CRCs, lossless."""
    import struct
    import zlib
    from geneface_amd.png import FrameWriter, decode_rgb8, encode_rgb8
    rng = np.random.default_rng(0)
    for H, W in ((1, 1), (7, 5), (64, 48)):
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        data = encode_rgb8(img)
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        n, tag = struct.unpack(">I4s", data[8:16])
        assert tag == b"IHDR" and n == 13 and struct.unpack(">II", data[16:24]) == (W, H)
        assert struct.unpack(">I", data[29:33])[0] == zlib.crc32(data[12:29]) & 0xFFFFFFFF
        np.testing.assert_array_equal(decode_rgb8(data), img)
    w = FrameWriter(str(tmp_path / "imgs"), workers=2)
    frames = rng.integers(0, 256, size=(5, 16, 16, 3), dtype=np.uint8)
    for i, f in enumerate(frames):
        w.submit(i, f)
    w.close()
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(decode_rgb8(open(tmp_path / "imgs" / f"{i:05d}.png", "rb").read()), f)
    with pytest.raises(ValueError):
        encode_rgb8(np.zeros((4, 4), dtype=np.uint8))


def test_orbit_camera_pose_and_intrinsics():
    """radnerf_gui.py:21-82: pose = rot @ translate(-radius) - centre; intrinsics from fovy; update_pose inverts pose."""
    from geneface_amd.gui import OrbitCamera
    cam = OrbitCamera(512, 512, r=3.35, fovy=21.24)
    p = cam.pose
    assert np.allclose(p[:3, :3], [[0, -1, 0], [0, 0, -1], [1, 0, 0]]) and np.allclose(p[:3, 3], [0, 3.35, 0], atol=1e-6)
    f = cam.intrinsics
    assert abs(f[0] - 256 / np.tan(np.deg2rad(21.24) / 2)) < 1e-6 and f[2] == 256 and f[3] == 256
    cam.orbit(300, -120)
    cam.scale(2)
    q = cam.pose
    assert abs(np.linalg.norm(q[:3, 3]) - 3.35 * 1.1 ** -2) < 1e-5 and np.allclose(q[:3, :3] @ q[:3, :3].T, np.eye(3), atol=1e-6)
    other = OrbitCamera(512, 512)
    other.update_pose(q)
    assert np.allclose(other.pose, q, atol=1e-5)
    other.update_intrinsics([1365.3, 1365.3, 256, 256])
    assert abs(other.fovy - 21.24) < 0.01 and other.W == 512
