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


_ds_dict = S.make_dataset_dict      # a trainval_dataset.npy-shaped dict with AD-NeRF convention c2w matrices, and the ngp poses it encodes


def test_pose_source_follows_dataset_init(tmp_path):
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
    # a background picture from a file (dataset_utils.py:69-74): same size as the dataset, and twice as large (area-averaged down)
    from geneface_amd.png import encode_rgb8
    pic = np.random.default_rng(3).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    (tmp_path / "bg.png").write_bytes(encode_rgb8(pic))
    (tmp_path / "bg2x.png").write_bytes(encode_rgb8(np.repeat(np.repeat(pic, 2, axis=0), 2, axis=1)))
    for name in ("bg.png", "bg2x.png"):
        src3 = RADNeRFPoseSource(dd, dict(hp, infer_bg_img_fname=str(tmp_path / name)))
        assert src3.bg_img.dtype == np.float32 and np.array_equal(np.round(src3.bg_img * 255).astype(np.uint8).reshape(64, 64, 3), pic), name


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
        from test_gpu_render import check_u8

        def on_kernel_rays(i=i, pose=pose):      # every input as the device computes it (rays, background coordinates, euler pose)
            from geneface_amd import utils
            from geneface_amd.fused import pinhole_rays
            kro, krd = pinhole_rays(pose[0], inf.dataset.intrinsics, 64, 64, "cuda:0")
            r = R.render(sd, hp, kro.cpu(), krd.cpu(), torch.from_numpy(samples[i]["cond_wins"]), utils.get_bg_coords(64, 64, "cuda:0").cpu(),
                         utils.convert_poses(pose.to("cuda:0")).cpu(), bg, torso=True)
            return (r["rgb_map"] * 255).view(64, 64, 3).to(torch.uint8)
        check_u8(got, ref8, rerender=on_kernel_rays)      # >= 99.9 % of the bytes identical, <= 1 LSB, PSNR >= 55 dB; nothing excused (see check_u8)


def _save_reference_checkpoint(path, model_sd, step, extra_children=True):
    """A checkpoint exactly as the reference's Trainer writes it (utils/commons/trainer.py:454-473): per-child state dicts, optimizer
    states, a numpy scalar, LEGACY (non-zip) serialization."""
    params = [torch.nn.Parameter(torch.zeros(3))]
    opt = torch.optim.Adam(params, lr=1e-3)
    params[0].grad = torch.ones(3)
    opt.step()
    state = {"model": model_sd}
    if extra_children:
        state["criterion_lpips"] = {"net.lin0.weight": torch.zeros(4)}      # RADNeRFTask has an LPIPS child with parameters
    ck = {"epoch": 3, "global_step": step, "checkpoint_callback_best": np.float64(0.123), "optimizer_states": [opt.state_dict()],
          "state_dict": state}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(ck, path, _use_new_zipfile_serialization=False)


def test_build_model_from_reference_checkpoints(tmp_path):
    """f3: LM3d_RADNeRFInfer.build_model = RADNeRFTorsoTask.build_model (head checkpoint, strict=False) + build_nerf_task (work_dir,
    strict) over legacy-pickle checkpoints with optimizer states and a numpy scalar; newest step wins; the head-only task too."""
    import zipfile
    from geneface_amd import ckpt_utils
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp_t, hp_h = HP.may_hparams(True), HP.may_hparams(False)
    sd_head, sd_torso_old, sd_torso = S.make_state_dict(hp_h, False, seed=2), S.make_state_dict(hp_t, True, seed=3), S.make_state_dict(hp_t, True, seed=4)
    head_dir, work_dir = str(tmp_path / "ckpt" / "lm3d_radnerf"), str(tmp_path / "ckpt" / "lm3d_radnerf_torso")
    _save_reference_checkpoint(os.path.join(head_dir, "model_ckpt_steps_250000.ckpt"), sd_head, 250000)
    _save_reference_checkpoint(os.path.join(work_dir, "model_ckpt_steps_8000.ckpt"), sd_torso_old, 8000)
    _save_reference_checkpoint(os.path.join(work_dir, "model_ckpt_steps_250000.ckpt"), sd_torso, 250000)
    assert not zipfile.is_zipfile(os.path.join(work_dir, "model_ckpt_steps_250000.ckpt"))       # the legacy container, as the reference writes
    assert [os.path.basename(p) for p in ckpt_utils.get_all_ckpts(work_dir)] == ["model_ckpt_steps_250000.ckpt", "model_ckpt_steps_8000.ckpt"]
    ck, path = ckpt_utils.get_last_checkpoint(work_dir)
    assert path.endswith("250000.ckpt") and ck["global_step"] == 250000 and float(ck["checkpoint_callback_best"]) == 0.123
    assert ck["optimizer_states"][0]["state"][0]["exp_avg"].shape == (3,)

    dd, _ = _ds_dict()
    hp = dict(hp_t, work_dir=work_dir, head_model_dir=head_dir)
    inf = LM3d_RADNeRFInfer(hp, dataset=RADNeRFPoseSource(dd, hp), device="cpu")
    assert isinstance(inf.model, RADNeRFTorso) and inf.global_step == 250000
    got = inf.model.state_dict()
    for k, v in sd_torso.items():
        assert torch.equal(got[k], v), k
    # the two stages separately: after the head stage alone the head keys are the head checkpoint's, the torso keys untouched
    stage1 = RADNeRFTorso(hp)
    init = {k: v.clone() for k, v in stage1.state_dict().items()}
    head = RADNeRF(hp)
    ckpt_utils.load_ckpt(head, head_dir)
    stage1.load_state_dict(head.state_dict(), strict=False)
    for k, v in stage1.state_dict().items():
        assert torch.equal(v, sd_head[k] if k in sd_head else init[k]), k
    # head-only task: one strict load from work_dir
    hp1 = dict(hp_h, work_dir=head_dir, task_cls="tasks.radnerfs.radnerf.RADNeRFTask")
    inf1 = LM3d_RADNeRFInfer(hp1, dataset=RADNeRFPoseSource(dd, hp1), device="cpu")
    assert isinstance(inf1.model, RADNeRF) and not isinstance(inf1.model, RADNeRFTorso)
    assert all(torch.equal(inf1.model.state_dict()[k], v) for k, v in sd_head.items())
    # load_ckpt's other behaviours (ckpt_utils.py:27-66): a file path, `steps`, flat 'model.' prefixes, strict=False dropping a mismatched
    # shape, and the assert on a missing checkpoint
    m = RADNeRFTorso(hp)
    ckpt_utils.load_ckpt(m, work_dir, steps=8000)
    assert torch.equal(m.state_dict()["sigma_net.net.0.weight"], sd_torso_old["sigma_net.net.0.weight"])
    ckpt_utils.load_ckpt(m, os.path.join(work_dir, "model_ckpt_steps_250000.ckpt"))
    assert torch.equal(m.state_dict()["sigma_net.net.0.weight"], sd_torso["sigma_net.net.0.weight"])
    flat = {"state_dict": {f"model.{k}": v for k, v in sd_torso_old.items()}, "global_step": 1}
    torch.save(flat, str(tmp_path / "flat.ckpt"), _use_new_zipfile_serialization=False)
    ckpt_utils.load_ckpt(m, str(tmp_path / "flat.ckpt"))
    assert torch.equal(m.state_dict()["sigma_net.net.0.weight"], sd_torso_old["sigma_net.net.0.weight"])
    bad = dict(sd_torso, **{"color_net.net.1.weight": torch.zeros(5, 7)})
    _save_reference_checkpoint(str(tmp_path / "bad" / "model_ckpt_steps_1.ckpt"), bad, 1, extra_children=False)
    with pytest.raises(RuntimeError):
        ckpt_utils.load_ckpt(m, str(tmp_path / "bad"))
    before = m.state_dict()["color_net.net.1.weight"].clone()
    ckpt_utils.load_ckpt(m, str(tmp_path / "bad"), strict=False)
    assert torch.equal(m.state_dict()["color_net.net.1.weight"], before)
    with pytest.raises(AssertionError):
        ckpt_utils.load_ckpt(m, str(tmp_path / "nothing_here"))
    assert ckpt_utils.load_ckpt(m, str(tmp_path / "nothing_here"), force=False) is None
    with pytest.raises(AssertionError):                                   # a torso task whose head_model_dir is missing fails like the reference
        LM3d_RADNeRFInfer(dict(hp, head_model_dir=str(tmp_path / "nope")), dataset=RADNeRFPoseSource(dd, hp), device="cpu")


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
    # submit() must take a private copy: the frame loop hands in views of ONE reused (pinned) buffer and overwrites it right away
    w = FrameWriter(str(tmp_path / "reused"), workers=1, max_pending=3)
    buf = np.zeros((64, 64, 3), dtype=np.uint8)
    for i in range(12):
        buf[...] = i
        w.submit(i, buf)                                                   # blocks while 3 frames wait: the queue is bounded
    w.close()
    st = w.stage_seconds()
    assert st["frames"] == 12 and st["bytes"] > 0 and st["deflate_sum_over_workers"] > 0
    for i in range(12):
        assert (decode_rgb8(open(tmp_path / "reused" / f"{i:05d}.png", "rb").read()) == i).all()
    # the native encoder and the pure-Python one agree on the pixels, and a stock reader (Pillow, where installed) reads the files
    import ctypes as C
    from geneface_amd.lib import lib
    img = rng.integers(0, 256, size=(33, 17, 3), dtype=np.uint8)
    out, n = np.empty(1 << 16, dtype=np.uint8), C.c_uint64(0)
    for strategy in (0, 2, 3):
        assert lib().gf_png_encode_rgb8(img.ctypes.data, 33, 17, 1, strategy, out.ctypes.data, out.size, C.cast(C.byref(n), C.c_void_p)) == 0
        np.testing.assert_array_equal(decode_rgb8(out[:n.value].tobytes()), img)
    try:
        from PIL import Image
        np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "imgs" / "00003.png")), frames[3])
    except ImportError:
        pass
    with pytest.raises(RuntimeError):                                      # a directory that cannot be written: the error surfaces, loudly
        bad = FrameWriter(str(tmp_path / "imgs2"), workers=1)
        os.rmdir(tmp_path / "imgs2")
        (tmp_path / "imgs2").write_bytes(b"")                               # a FILE where the directory was
        bad.submit(0, buf)
        bad.close()
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


@pytest.mark.gpu
def test_infer_once_shard_is_the_ranks_block_of_the_whole_run(tmp_path):
    """inp["shard"] = (rank, world): one process renders exactly the frames rank `rank` of a `world`-GPU job would (base_nerf_infer.py:150-155)
    -- the per-GPU share of BASELINE.json configs[3] on one GPU (tools/shard_run.py measures it at 375 of 3000 frames, 512x512).  Same bytes as
    the same frames of the unsharded run, PNG names are GLOBAL frame indices."""
    from geneface_amd.png import decode_rgb8
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = HP.may_hparams(True)
    model = RADNeRFTorso(hp)
    model.load_state_dict(S.make_state_dict(hp, True), strict=True)
    T = 23
    dd, _ = _ds_dict(T=T, H=64, W=64)
    inf = LM3d_RADNeRFInfer(hp, model=model, dataset=RADNeRFPoseSource(dd, hp), device="cuda:0")
    cond_path = os.path.join(tmp_path, "lm.npy")
    np.save(cond_path, S.make_landmarks(T).astype(np.float32)[None])
    whole = inf.infer_once({"cond_name": cond_path, "out_video_name": "", "audio_source_name": ""})
    assert whole.shape == (T, 64, 64, 3)
    seen = []
    for rank in range(3):
        imgs = os.path.join(tmp_path, f"imgs{rank}")
        part = inf.infer_once({"cond_name": cond_path, "out_video_name": "", "audio_source_name": "", "tmp_imgs_dir": imgs, "shard": (rank, 3)})
        lo, hi = rank * (T // 3), ((rank + 1) * (T // 3) if rank < 2 else T)
        np.testing.assert_array_equal(part, whole[lo:hi])
        assert sorted(os.listdir(imgs)) == [f"{i:05d}.png" for i in range(lo, hi)]
        np.testing.assert_array_equal(decode_rgb8(open(os.path.join(imgs, f"{lo:05d}.png"), "rb").read()), whole[lo])
        seen.extend(range(lo, hi))
    assert seen == list(range(T))
    with pytest.raises(ValueError):
        inf.infer_once({"cond_name": cond_path, "out_video_name": "", "audio_source_name": "", "shard": (3, 3)})


def test_postprocess_runs_the_references_ffmpeg_commands(tmp_path, monkeypatch):
    """base_nerf_infer.py:255-259, 303-317: after the frames, rank 0 resamples the audio source to a 16 kHz wav and muxes
    `<tmp_imgs_dir>/%5d.png` + that wav into out_video_name.  No ffmpeg binary exists in this image, so a recording stand-in on PATH checks
    the two command lines (the reference's, argument for argument); without any ffmpeg the entry point says so and returns the frames."""
    import json
    import stat
    log = tmp_path / "ffmpeg_calls.jsonl"
    fake = tmp_path / "bin" / "ffmpeg"
    fake.parent.mkdir()
    fake.write_text(f"#!{os.sys.executable}\nimport json, sys\nopen({str(log)!r}, 'a').write(json.dumps(sys.argv[1:]) + '\\n')\n"
                    "out = [a for a in sys.argv[1:] if a.endswith(('.wav', '.mp4')) and not a.startswith('-')][-1]\nopen(out, 'wb').write(b'x')\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    hp = HP.may_hparams(True)
    dd, _ = _ds_dict(T=6, H=16, W=16)
    inf = LM3d_RADNeRFInfer.__new__(LM3d_RADNeRFInfer)      # host logic only
    inf.hparams, inf.dataset, inf.proc_rank = hp, RADNeRFPoseSource(dd, hp), 0
    frames = np.zeros((4, 16, 16, 3), dtype=np.uint8)
    inf.forward_system = lambda samples, collect=True: frames
    cond = tmp_path / "lm.npy"
    np.save(cond, S.make_landmarks(4).astype(np.float32)[None])
    wav = tmp_path / "zozo.wav"
    wav.write_bytes(b"RIFF")
    inp = {"cond_name": str(cond), "audio_source_name": str(wav), "out_video_name": str(tmp_path / "out" / "zozo.mp4"), "tmp_imgs_dir": str(tmp_path / "imgs")}
    # no ffmpeg anywhere: frames come back, nothing is muxed
    monkeypatch.setenv("PATH", str(tmp_path / "empty"))
    assert inf.infer_once(dict(inp)) is frames and not log.exists()
    # with the stand-in on PATH
    monkeypatch.setenv("PATH", str(fake.parent))
    assert inf.infer_once(dict(inp)) is frames
    calls = [json.loads(ln) for ln in log.read_text().splitlines()]
    wav16k = str(wav)[:-4] + "_16k.wav"
    assert calls[0] == ["-i", str(wav), "-v", "quiet", "-f", "wav", "-ar", "16000", wav16k, "-y"]
    assert calls[1] == ["-i", os.path.join(str(tmp_path / "imgs"), "%5d.png"), "-i", wav16k, "-shortest", "-v", "quiet", "-c:v", "libx264", "-pix_fmt",
                        "yuv420p", "-b:v", "2000k", "-r", "25", "-strict", "-2", "-y", inp["out_video_name"]]
    assert os.path.exists(inp["out_video_name"]) and inf.wav16k_name == wav16k
    with pytest.raises(AssertionError):
        inf.save_wav16k({"audio_source_name": str(tmp_path / "zozo.flac")})


def test_native_png_encoder_on_drawn_sizes_contents_levels_and_strategies():
    """gf_png_encode_rgb8 over 200 drawn pictures: 1 x 1 ... 200 x 200 (odd row lengths), noise / constant / ramps / sparse content, zlib levels 0
    (stored blocks: the small-pool mode of NOTES 10.2), 1, 6, 9 and all four strategies -- every file decodes to the picture; a buffer that is
    too small is refused with its size, not overrun."""
    import ctypes as C
    from geneface_amd.lib import lib
    from geneface_amd.png import decode_rgb8
    L, rng = lib(), np.random.default_rng(4)
    for _ in range(200):
        H, W = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        kind = rng.choice(["noise", "constant", "ramp", "sparse"])
        if kind == "noise":
            img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        elif kind == "constant":
            img = np.full((H, W, 3), int(rng.integers(0, 256)), np.uint8)
        elif kind == "ramp":
            img = (np.add.outer(np.arange(H), np.arange(W))[..., None] * np.array([1, 2, 3])).astype(np.uint8)
        else:
            img = np.zeros((H, W, 3), np.uint8)
            img[rng.random((H, W)) < 0.02] = 255
        img = np.ascontiguousarray(img)
        level, strategy = int(rng.choice([0, 1, 6, 9])), int(rng.choice([0, 1, 2, 3]))
        out, n = np.empty(H * (W * 3 + 1) + 8192, np.uint8), C.c_uint64(0)
        rc = L.gf_png_encode_rgb8(img.ctypes.data, H, W, level, strategy, out.ctypes.data, out.size, C.cast(C.byref(n), C.c_void_p))
        assert rc == 0, (L.gf_last_error(), H, W, level, strategy)
        np.testing.assert_array_equal(decode_rgb8(out[:n.value].tobytes()), img, err_msg=f"{H}x{W} {kind} level {level} strategy {strategy}")
    img = rng.integers(0, 256, (50, 50, 3), dtype=np.uint8)
    out, n = np.empty(100, np.uint8), C.c_uint64(0)
    assert L.gf_png_encode_rgb8(img.ctypes.data, 50, 50, 1, 0, out.ctypes.data, out.size, C.cast(C.byref(n), C.c_void_p)) == 1
    assert b"too small" in L.gf_last_error()
