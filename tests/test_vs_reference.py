"""Checks that need the read-only reference tree (build container only; skipped on the GPU box)."""
import numpy as np
import pytest
import torch

from helpers import model_fixture, sequence
from oracle import radnerf_ref as R
from oracle import refshim

pytestmark = pytest.mark.reference


def _reference_function(rel_path, name):
    """Execute ONE function of a reference file whose module-level imports cannot run here
    (data_gen/nerf/binarizer.py parses sys.argv at import): compile just that def from the file's AST."""
    import ast
    import os
    src = open(os.path.join(refshim.REFERENCE_ROOT, rel_path)).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np, "torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), rel_path, "exec"), ns)
    return ns[name]


def test_hparams_match_reference_yaml_chain():
    from geneface_amd.hparams import may_hparams
    for torso in (False, True):
        ours, ref = may_hparams(torso), refshim.reference_hparams(torso)
        for k, v in ours.items():
            if k in ref:
                assert ref[k] == v, (k, ref[k], v)
        for k in ("bound", "grid_size", "max_steps", "dt_gamma", "min_near", "density_thresh_torso", "torso_shrink", "cond_type",
                  "smo_win_size", "cond_win_size", "individual_embedding_dim", "torso_individual_embedding_dim"):
            assert k in ours and k in ref


def test_state_dict_is_loadable_by_the_reference_and_by_us():
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp, sd = model_fixture(True)
    ref_model, _ = refshim.build_reference_model(True)
    assert set(ref_model.state_dict().keys()) == set(sd.keys())
    ref_model.load_state_dict(sd, strict=True)
    ours = RADNeRFTorso(hp)
    assert list(ours.state_dict().keys()) == list(ref_model.state_dict().keys())
    for k, v in ours.state_dict().items():
        assert v.shape == ref_model.state_dict()[k].shape and v.dtype == ref_model.state_dict()[k].dtype, k


def test_field_and_cond_encoder_match_reference_modules():
    hp, sd = model_fixture(True)
    model, _ = refshim.build_reference_model(True)
    model.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(2000, 3, generator=g) * 2 - 1) * torch.tensor([0.5, 0.3, 0.5])
    d = torch.nn.functional.normalize(torch.randn(2000, 3, generator=g), dim=-1)
    cond = torch.randn(5, 1, 204, generator=g)
    with refshim.cpu_mode(), torch.no_grad():
        cf = model.cal_cond_feat(cond)
        s1, c1, a1 = model(x, d, cf, model.individual_embeddings[0])
        xy = torch.rand(500, 2, generator=g) * 2 - 1
        p6 = torch.tensor([[0.1, -0.05, 0.02, 0.0, 3.3, 0.0]])
        ta, tc, tdx = model.forward_torso(xy, p6, model.torso_individual_codes[0])
    assert torch.allclose(R.cal_cond_feat(sd, hp, cond), cf, atol=1e-6)
    s2, c2, a2 = R.head_field(sd, hp, x, d, cf, sd["individual_embeddings"][0])
    assert torch.equal(s1, s2) and torch.equal(c1, c2) and torch.equal(a1, a2)
    ra, rc, rdx = R.torso_field(sd, hp, xy, p6, sd["torso_individual_codes"][0])
    assert torch.equal(ta, ra) and torch.equal(tc, rc) and torch.equal(tdx, rdx)


def test_rays_poses_bgcoords_match_reference_utils():
    refshim.install()
    import modules.radnerfs.utils as U
    from geneface_amd import utils as ours
    seq = sequence(4, 64, 64)
    pose = torch.from_numpy(seq["poses"][2:3])
    ref = U.get_rays(pose, seq["intrinsics"], 64, 64, -1)
    mine = ours.get_rays(pose, seq["intrinsics"], 64, 64, -1)
    ro, rd = R.get_rays(pose, seq["intrinsics"], 64, 64)
    for a in (mine["rays_d"], rd):
        assert torch.equal(a, ref["rays_d"])
    assert torch.equal(mine["rays_o"], ref["rays_o"]) and torch.equal(ro, ref["rays_o"])
    # the training-time sampling modes draw the same pixels from the same generator state
    poses2 = torch.from_numpy(seq["poses"][:2])
    for kw in (dict(N=500), dict(N=4096, patch_size=8), dict(N=-1, rect=(10, 30, 5, 41)), dict(N=10 ** 6)):
        torch.manual_seed(11)
        ref = U.get_rays(poses2 if "rect" not in kw else pose, seq["intrinsics"], 64, 48, **kw)
        torch.manual_seed(11)
        mine = ours.get_rays(poses2 if "rect" not in kw else pose, seq["intrinsics"], 64, 48, **kw)
        for k in ("inds", "i", "j", "rays_o", "rays_d"):
            assert mine[k].shape == ref[k].shape and torch.equal(mine[k], ref[k]), (kw, k)
    assert torch.equal(ours.get_bg_coords(64, 64, "cpu"), U.get_bg_coords(64, 64, "cpu"))
    assert torch.equal(R.get_bg_coords(64, 64), U.get_bg_coords(64, 64, "cpu"))
    poses = torch.from_numpy(seq["poses"])
    assert torch.allclose(ours.convert_poses(poses), U.convert_poses(poses), atol=1e-7)
    assert torch.allclose(R.convert_poses(poses), U.convert_poses(poses), atol=1e-7)
    m = np.random.default_rng(0).standard_normal((4, 4)).astype(np.float32)
    assert np.array_equal(ours.nerf_matrix_to_ngp(m, 4.0, [0, 0, 0]), U.nerf_matrix_to_ngp(m, 4.0, [0, 0, 0]))


def test_camera_smoothing_and_window_helpers_match_reference():
    refshim.install()
    ref_win = _reference_function("data_gen/nerf/binarizer.py", "get_win_conds")
    from geneface_amd.lm3d import get_win_conds
    x = np.arange(7 * 3, dtype=np.float32).reshape(7, 3)
    for idx in (-2, 0, 1, 3, 6, 9):
        for w in (1, 2, 5, 8):
            for pad in ("zero", "edge"):
                assert np.array_equal(get_win_conds(x, idx, w, pad), ref_win(x, idx, w, pad)), (idx, w, pad)
    from geneface_amd.synthetic import make_poses
    from geneface_amd.utils import smooth_camera_path
    import tasks.radnerfs.dataset_utils as DU
    p = make_poses(20)
    assert np.allclose(smooth_camera_path(p.copy(), 7), DU.smooth_camera_path(p.copy(), 7), atol=1e-7)


def test_landmark_postprocess_matches_reference_entry_point():
    """Drive LM3d_RADNeRFInfer.get_cond_from_input (lm3d_radnerf_infer.py:34-86) with a stand-in `self`."""
    import os
    import tempfile
    import types
    refshim.install()
    from geneface_amd.lm3d import cond_windows, normalize_and_smooth
    from geneface_amd.synthetic import make_landmarks
    from utils.commons.hparams import hparams as ghp
    ghp.clear()
    ghp.update(refshim.reference_hparams(True))
    import sys
    # the method imports get_win_conds from a module that parses sys.argv at import time: pre-seed that module
    # with the reference's own function body (compiled from its file), nothing else
    fake_mod = types.ModuleType("data_gen.nerf.binarizer")
    fake_mod.get_win_conds = _reference_function("data_gen/nerf/binarizer.py", "get_win_conds")
    sys.modules["data_gen.nerf.binarizer"] = fake_mod
    from inference.nerfs.lm3d_radnerf_infer import LM3d_RADNeRFInfer
    lm = make_landmarks(12) * 1.7 + 0.3
    mean = torch.full((68, 3), 0.25)
    std = torch.full((68, 3), 1.5)
    fake = types.SimpleNamespace(dataset=types.SimpleNamespace(idexp_lm3d_mean=mean, idexp_lm3d_std=std), save_wav16k=lambda inp: None)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "lm.npy")
        np.save(path, lm[None])
        samples = LM3d_RADNeRFInfer.get_cond_from_input(fake, {"cond_name": path})
    norm = normalize_and_smooth(lm, mean.numpy(), std.numpy(), 2.5)
    wins = cond_windows(norm, 1, 5)
    for i, s in enumerate(samples):
        assert np.allclose(s["cond"].numpy()[0], norm[i], atol=1e-6)
        assert np.allclose(s["cond_wins"].numpy(), wins[i], atol=1e-6)


def test_legacy_nerf_baseline_matches_reference_pure_torch_path():
    """Baseline B2: oracle/legacy_nerf_ref.py against the reference's own Lm3dNeRF + render_dynamic_face on identical weights and
    identical random draws (both consume torch's global generator in the same order)."""
    from oracle import legacy_nerf_ref as LN
    refshim.install()
    with refshim.cpu_mode():
        from modules.nerfs.commons.volume_rendering import render_dynamic_face
        from modules.nerfs.lm3d_nerf.lm3d_nerf import Lm3dNeRF
        torch.manual_seed(3)
        model = Lm3dNeRF({"cond_dim": 64, "hidden_size": 256, "use_window_cond": True, "cond_win_size": 1, "smo_win_size": 5, "with_att": True}).eval()
        w = {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("model_")}
        mine = LN.make_weights(0)
        assert set(w) == set(mine) and all(w[k].shape == mine[k].shape for k in w)       # same parameter names and shapes
        H = W = 20
        focal, cx, cy = 60.0, 10.0, 10.0
        c2w = torch.tensor([[1, 0, 0, 0.02], [0, 1, 0, -0.01], [0, 0, 1, 0.6]], dtype=torch.float32)
        g = torch.Generator().manual_seed(8)
        bg, cond = torch.rand(H, W, 3, generator=g), torch.randn(64, generator=g)
        rays_o, rays_d = LN.get_rays(H, W, focal, c2w, cx, cy)
        with torch.no_grad():
            torch.manual_seed(5)
            ref = render_dynamic_face(H, W, focal, cx, cy, chunk=2048, rays_o=rays_o, rays_d=rays_d, bc_rgb=bg, cond=cond, near=0.3, far=0.9,
                                      network_fn=model, N_samples=64, N_importance=128)[0]
            torch.manual_seed(5)
            out = LN.render(w, H, W, focal, cx, cy, c2w, bg, cond)
    assert out.shape == (H * W, 3)
    assert (out - ref.reshape(-1, 3)).abs().max().item() < 1e-6


@pytest.mark.parametrize("cfg", ["egs/datasets/videos/May/lm3d_radnerf_torso.yaml", "egs/datasets/videos/May/lm3d_radnerf.yaml",
                                 "egs/datasets/videos/Obama2/lm3d_radnerf_torso.yaml", "egs/datasets/videos/Obama/radnerf.yaml"])
def test_yaml_chain_loader_matches_reference_set_hparams(cfg):
    """geneface_amd.hparams.load_config against the reference's own set_hparams on its experiment files (two identities, landmark-
    and audio-driven), with and without command-line style overrides."""
    import os
    from geneface_amd import hparams as HP
    refshim.install()
    from utils.commons.hparams import set_hparams as ref_set
    cwd = os.getcwd()
    os.chdir(refshim.REFERENCE_ROOT)                      # the reference resolves base_config against its working directory
    try:
        for over in ("", "max_steps=24,dt_gamma=0.01,with_att=False,camera_offset=[0 0 1]"):
            ref = ref_set(config=cfg, hparams_str=over, print_hparams=False, global_hparams=False)
            ours = HP.load_config(cfg, over, root=refshim.REFERENCE_ROOT)
            ours.update({"infer": False, "debug": False, "validate": False, "exp_name": ""})     # run-mode flags set_hparams adds from argv
            assert ours == ref, {k: (ours.get(k), ref.get(k)) for k in set(ours) | set(ref) if ours.get(k) != ref.get(k)}
    finally:
        os.chdir(cwd)
    may = HP.load_config("egs/datasets/videos/May/lm3d_radnerf_torso.yaml", root=refshim.REFERENCE_ROOT)
    for k, v in HP.may_hparams(True).items():
        if k in may:
            assert may[k] == v, (k, may[k], v)


@pytest.mark.parametrize("name", ["hash", "hash_smoothstep", "smoothstep", "head_aware", "audio"])
def test_variant_hparams_match_the_reference_yaml_files(name):
    """geneface_amd.hparams.VARIANTS (the other RAD-NeRF experiment files the reference ships: what tests/test_gpu_sweep.py sweeps and
    bench.py's `variants` block times) against the reference's own set_hparams on the yaml files each entry names."""
    import os
    from geneface_amd import hparams as HP
    refshim.install()
    from utils.commons.hparams import set_hparams as ref_set
    overrides, files = HP.VARIANTS[name]
    cwd = os.getcwd()
    os.chdir(refshim.REFERENCE_ROOT)
    try:
        resolved = [ref_set(config=f, hparams_str="", print_hparams=False, global_hparams=False) for f in files]
    finally:
        os.chdir(cwd)
    head = resolved[0]
    for k, v in overrides.items():
        if k == "video_id":
            continue
        src = next((r for r in resolved if k in r and r[k] == v), None)
        assert src is not None, (name, k, v, [r.get(k) for r in resolved])
    ours = HP.variant_hparams(name, torso=len(files) > 1 or name == "head_aware")
    ref = resolved[-1] if name in ("head_aware", "audio") else head
    for k, v in ours.items():
        if k in ref and k not in ("video_id", "head_model_dir", "task_cls", "torso_train_mode"):
            assert ref[k] == v or (name not in ("head_aware",) and k in ("individual_embedding_num",) and any(r.get(k) == v for r in resolved)), (name, k, ref[k], v)
