"""Generate the committed golden vectors (run in the BUILD container only: needs /root/reference).

    python tests/golden/make_golden.py

What produces the numbers: the reference's own, unmodified Python layers
(modules/radnerfs/{renderer,radnerf,radnerf_torso,cond_encoder}.py and the op wrappers
raymarching.py / grid.py / sphere_harmonics.py / freq.py) imported from /root/reference, executing
on CPU over the C oracle's kernels registered at the `_raymarching_face / _gridencoder / _shencoder /
_freqencoder` import seam (oracle/refshim.py).  Inputs are the seeded synthetic fixture
(geneface_amd/synthetic.py); only outputs (and an input checksum) are stored.
"""
import hashlib
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from geneface_amd import hparams as H_  # noqa: E402
from geneface_amd import synthetic as S  # noqa: E402
from oracle import refshim  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sd_checksum(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.numpy().tobytes())
    return h.hexdigest()


def frame_inputs(seq, idx, ref_utils):
    H, W = seq["H"], seq["W"]
    pose = torch.from_numpy(seq["poses"][idx:idx + 1])
    rays = ref_utils.get_rays(pose, seq["intrinsics"], H, W, -1)
    return dict(rays_o=rays["rays_o"], rays_d=rays["rays_d"], bg_coords=ref_utils.get_bg_coords(H, W, "cpu"),
                cond=torch.from_numpy(seq["cond_wins"][idx]), pose6=ref_utils.convert_poses(pose),
                bg=torch.from_numpy(seq["bg_img"]).view(1, -1, 3))


def golden_frames():
    for torso in (False, True):
        hp = H_.may_hparams(torso)
        sd = S.make_state_dict(hp, torso)
        model, rhp = refshim.build_reference_model(torso)
        for k in hp:  # our restated hparams must equal the reference's yaml chain on every key we list
            if k in rhp:
                assert rhp[k] == hp[k], (k, rhp[k], hp[k])
        model.load_state_dict(sd, strict=True)
        import modules.radnerfs.utils as ref_utils
        for (Himg, idx) in ((64, 1), (96, 3)):
            seq = S.make_sequence(4, Himg, Himg, hp)
            fi = frame_inputs(seq, idx, ref_utils)
            with refshim.cpu_mode(), torch.no_grad():
                out = model.render(fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], index=0, staged=False,
                                   bg_color=fi["bg"], perturb=False, force_all_rays=True, **rhp)
                cond_feat = model.cal_cond_feat(fi["cond"])
            name = f"frame_{'torso' if torso else 'head'}_{Himg}.npz"
            payload = {k: v.detach().numpy() for k, v in out.items()}
            payload["cond_feat"] = cond_feat.numpy()
            payload["rays_d_checksum"] = np.array(float(fi["rays_d"].double().abs().sum()))
            payload["pose6"] = fi["pose6"].numpy()
            payload["state_dict_sha256"] = np.array(sd_checksum(sd))
            np.savez_compressed(os.path.join(OUT, name), **payload)
            print(name, {k: v.shape for k, v in payload.items() if hasattr(v, "shape")})


def golden_ops():
    """Wrapper-level op vectors: reference Python wrappers (layout, padding, remap) over the C kernels.
    Inputs come from tests/golden_inputs.py seeds; only outputs are stored."""
    refshim.install()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_inputs as GI
    import modules.radnerfs.raymarching as rm
    from modules.radnerfs.encoders.encoding import get_encoder
    out = {}
    with refshim.cpu_mode(), torch.no_grad():
        for D, enc_name, interp in GI.GRID_CASES:
            tag, x, table, off = GI.grid_case(D, enc_name, interp)
            enc, _ = get_encoder(enc_name, input_dim=D, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                                 desired_resolution=2048, interpolation=interp)
            assert np.array_equal(enc.offsets.numpy(), off)
            enc.embeddings.data.copy_(torch.from_numpy(table))
            out[tag + "_y"] = enc(torch.from_numpy(x), bound=1).numpy()
        sh, _ = get_encoder("spherical_harmonics")
        out["sh_y"] = sh(torch.from_numpy(GI.sh_dirs())).numpy()
        for dim, deg in ((6, 4), (2, 10)):
            fe, _ = get_encoder("frequency", input_dim=dim, multires=deg)
            out[f"freq_{dim}_{deg}_y"] = fe(torch.from_numpy(GI.freq_case(dim, deg))).numpy()
        # near/far + one march/composite round on a small ray bundle
        hp = H_.may_hparams(False)
        sd = S.make_state_dict(hp, False)
        seq = S.make_sequence(2, 32, 32, hp)
        import modules.radnerfs.utils as ref_utils
        fi = frame_inputs(seq, 0, ref_utils)
        ro, rd = fi["rays_o"].view(-1, 3).contiguous(), fi["rays_d"].view(-1, 3).contiguous()
        nears, fars = rm.near_far_from_aabb(ro, rd, sd["aabb_infer"], hp["min_near"])
        N = ro.shape[0]
        alive = torch.arange(N, dtype=torch.int32)
        rays_t = nears.clone()
        xyzs, dirs, deltas = rm.march_rays(N, 3, alive, rays_t, ro, rd, 1.0, sd["density_bitfield"], 1, 128, nears, fars, 128, False,
                                           hp["dt_gamma"], hp["max_steps"])
        sig, rgb = (torch.from_numpy(a) for a in GI.composite_inputs(xyzs.shape[0]))
        ws, dep, img = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
        rm.composite_rays(N, 3, alive, rays_t, sig, rgb, deltas, ws, dep, img, 1e-4)
        out.update(rays_o=ro.numpy(), rays_d=rd.numpy(), march_nears=nears.numpy(), march_fars=fars.numpy(), march_xyzs=xyzs.numpy(),
                   march_deltas=deltas.numpy(), comp_alive=alive.numpy(), comp_rays_t=rays_t.numpy(), comp_ws=ws.numpy(),
                   comp_depth=dep.numpy(), comp_image=img.numpy())
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("ops.npz", len(out), "arrays")


def golden_variants(size=48, idx=2):
    """One head+torso frame of every OTHER RAD-NeRF configuration the reference ships (geneface_amd.hparams.VARIANTS: hashed grids, smoothstep,
    head-aware torso with both outcomes of its per-frame coin, the audio-driven config on the second identity), rendered by the reference's own
    Python built with those hparams (round 5: the oracle's restatement of these branches -- the strided-conv AudioNet, the head-colour encoder,
    the hash / smoothstep flags on their way to the grid encoder -- had only been compared with the product, never with the reference)."""
    import random
    refshim.install()
    import modules.radnerfs.utils as ref_utils
    for name in ("hash", "hash_smoothstep", "smoothstep", "head_aware", "audio"):
        hp = H_.variant_hparams(name, True)
        seed = 1000 if name == "audio" else 0
        sd = S.make_state_dict(hp, True, seed=seed)
        model, rhp = refshim.build_reference_model(True, overrides=H_.VARIANTS[name][0])
        for k in hp:
            if k in rhp and k not in ("video_id", "head_model_dir"):
                assert rhp[k] == hp[k], (name, k, rhp[k], hp[k])
        model.load_state_dict(sd, strict=True)
        seq = S.make_sequence(4, size, size, hp, seed=seed)
        fi = frame_inputs(seq, idx, ref_utils)
        for coin in ((0.25, 0.75) if name == "head_aware" else (0.75,)):
            real = random.random
            random.random = lambda: coin          # radnerf_torso.py:175: `random.random() < 0.5` -> the torso sees the head (0.25) or zeros (0.75)
            try:
                with refshim.cpu_mode(), torch.no_grad():
                    out = model.render(fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], index=0, staged=False,
                                       bg_color=fi["bg"], perturb=False, force_all_rays=True, **rhp)
                    cond_feat = model.cal_cond_feat(fi["cond"])
            finally:
                random.random = real
            tag = name + ("_coin_heads" if coin < 0.5 and name == "head_aware" else ("_coin_tails" if name == "head_aware" else ""))
            payload = {k: v.detach().numpy() for k, v in out.items()}
            payload["cond_feat"] = cond_feat.numpy()
            payload["pose6"] = fi["pose6"].numpy()
            payload["rays_d_checksum"] = np.array(float(fi["rays_d"].double().abs().sum()))
            payload["state_dict_sha256"] = np.array(sd_checksum(sd))
            np.savez_compressed(os.path.join(OUT, f"frame_variant_{tag}_{size}.npz"), **payload)
            print(f"frame_variant_{tag}_{size}.npz", {k: v.shape for k, v in payload.items() if hasattr(v, "shape") and v.ndim})


if __name__ == "__main__":
    assert refshim.available(), "needs the reference tree"
    if "--variants-only" not in sys.argv:
        golden_frames()
        golden_ops()
    golden_variants()
