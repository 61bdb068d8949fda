"""Deterministic synthetic stand-ins for the assets the reference does not ship.

No checkpoint, dataset or landmark file exists offline (checkpoints/.gitkeep only), so tests, the
golden vectors and bench.py all draw from this generator:

* `make_state_dict`   -- a state_dict with the reference's key names / shapes
                         (modules/radnerfs/radnerf.py:12-59, radnerf_torso.py:18-49, renderer.py:78-99)
                         filled from seeded numpy streams; layer gains are chosen so that density,
                         colour, ambient coordinates and torso alpha span their useful ranges.
* `make_density_bitfield` -- Morton-ordered, LSB-first packed occupancy of an analytic head
                         (ellipsoid + neck), the layout `kernel_packbits` produces
                         (raymarching/src/raymarching.cu:268-289).
* `make_poses`        -- orbit-camera c2w matrices in ngp axes (inference/nerfs/radnerf_gui.py:21-43
                         defaults: radius 3.35, fovy 21.24 deg).
* `make_landmarks`    -- an AR(1) `[T, 204]` normalised 3-D landmark sequence.
Everything is numpy-seeded, so the GPU box regenerates bit-identical inputs.
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

from .encoders.gridencoder import grid_offsets

# Gains applied on top of the U(-1/sqrt(fan_in), 1/sqrt(fan_in)) init (see _linear); calibrated once so that
# log-density has mean ~3.3 / std ~1.2 inside the head (sigma ~ 1..1000: rays saturate after ~5-16 samples),
# ambient coordinates have std ~0.25, colours and torso alpha are not constant.
GAINS = {
    "ambient_net.net.0.weight": 3.0, "ambient_net.net.1.weight": 3.0, "ambient_net.net.2.weight": 3.0,
    "sigma_net.net.0.weight": 2.0, "sigma_net.net.1.weight": 2.0, "sigma_net.net.2.weight": 2.0,
    "color_net.net.0.weight": 2.5, "color_net.net.1.weight": 5.0,
    "torso_deform_net.net.0.weight": 2.0, "torso_deform_net.net.1.weight": 2.0, "torso_deform_net.net.2.weight": 0.5,
    "torso_canonicial_net.net.0.weight": 2.0, "torso_canonicial_net.net.1.weight": 2.0,
    "torso_canonicial_net.net.2.weight": 4.0,
}
#: the MLPs are bias-free (cond_encoder.py:102), so the density row of sigma_net's last layer is rebuilt as
#: ABS*|w| + LIN*w : the |w| part gives log-density a positive mean, the w part its spread.  Same for torso alpha.
SIGMA_ROW_ABS, SIGMA_ROW_LIN = 2.5, 10.0
TORSO_ALPHA_ROW_ABS, TORSO_ALPHA_ROW_LIN = 1.0, 2.0


def _rng(name: str, seed: int):
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def _uniform(name, seed, shape, bound):
    return _rng(name, seed).uniform(-bound, bound, size=shape).astype(np.float32)


def _linear(name, seed, out_dim, in_dim, k=1):
    bound = 1.0 / math.sqrt(in_dim * k)
    shape = (out_dim, in_dim) if k == 1 else (out_dim, in_dim, k)
    return _uniform(name, seed, shape, bound) * np.float32(GAINS.get(name, 1.0))


def cond_in_dim(hp):
    return {"esperanto": 44, "deepspeech": 29, "idexp_lm3d_normalized": 204}[hp["cond_type"]]


def make_state_dict(hp: dict, torso: bool = True, seed: int = 0, sigma_row_scale: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """`sigma_row_scale` < 1 thins the density (0.02: sigma ~ 1 everywhere, no ray ever saturates -- bench.py's stress fixture)."""
    sd = OrderedDict()
    G = hp["grid_size"]
    cascade = 1 + math.ceil(math.log2(hp["bound"]))
    bound = float(hp["bound"])
    sd["individual_embeddings"] = (_rng("individual_embeddings", seed).standard_normal(
        (hp["individual_embedding_num"], hp["individual_embedding_dim"])) * 0.1).astype(np.float32)
    if torso:
        sd["torso_individual_codes"] = (_rng("torso_individual_codes", seed).standard_normal(
            (hp["individual_embedding_num"], hp["torso_individual_embedding_dim"])) * 0.1).astype(np.float32)
    aabb = np.array([-bound, -bound / 2, -bound, bound, bound / 2, bound], dtype=np.float32)
    sd["aabb_train"] = aabb.copy()
    sd["aabb_infer"] = aabb.copy()
    sd["density_grid"] = np.zeros((cascade, G ** 3), dtype=np.float32)
    sd["density_bitfield"] = make_density_bitfield(G, cascade, bound, seed)
    sd["step_counter"] = np.zeros((16, 2), dtype=np.int32)
    if torso:
        sd["density_grid_torso"] = make_density_grid_torso(G, seed)

    cin, cout = cond_in_dim(hp), hp["cond_out_dim"]
    chans = [cin, 32, 32, 64, 64]
    for i in range(4):  # AudioNet conv stack, cond_encoder.py:25-38
        n = f"cond_prenet.encoder_conv.{2 * i}"
        sd[n + ".weight"] = _linear(n + ".weight", seed, chans[i + 1], chans[i], 3)
        sd[n + ".bias"] = _uniform(n + ".bias", seed, (chans[i + 1],), 1.0 / math.sqrt(chans[i] * 3))
    for i, (o, ii) in zip((0, 2), ((64, 64), (cout, 64))):  # cond_encoder.py:39-43
        n = f"cond_prenet.encoder_fc1.{i}"
        sd[n + ".weight"] = _linear(n + ".weight", seed, o, ii)
        sd[n + ".bias"] = _uniform(n + ".bias", seed, (o,), 1.0 / math.sqrt(ii))
    if hp["with_att"]:
        att = [cout, 16, 8, 4, 2, 1]
        for i in range(5):  # cond_encoder.py:60-72
            n = f"cond_att_net.attentionConvNet.{2 * i}"
            sd[n + ".weight"] = _linear(n + ".weight", seed, att[i + 1], att[i], 3)
            sd[n + ".bias"] = _uniform(n + ".bias", seed, (att[i + 1],), 1.0 / math.sqrt(att[i] * 3))
        s = hp["smo_win_size"]
        sd["cond_att_net.attentionNet.0.weight"] = _linear("cond_att_net.attentionNet.0.weight", seed, s, s)
        sd["cond_att_net.attentionNet.0.bias"] = _uniform("cond_att_net.attentionNet.0.bias", seed, (s,), 1 / math.sqrt(s))

    def grid(name, dim, desired):
        # U(-a_l, a_l) per level with a_l = 2 * (16/res_l)^2: a band-limited signal (coarse levels carry the
        # amplitude, fine levels only detail) like a trained table.  White noise at the 2048 level would turn 1e-7
        # of fp32 rounding in a coordinate into 1e-2 of feature change and make every parity tolerance meaningless.
        off = grid_offsets(dim, 16, 16, hp["log2_hashmap_size"], desired)
        pls = np.exp2(np.log2(desired / 16) / 15)
        table = _uniform(name + ".embeddings", seed, (int(off[-1]), 2), 2.0)
        for l in range(16):
            table[off[l]:off[l + 1]] *= np.float32((16.0 / np.ceil(16 * pls ** l)) ** 2)
        sd[name + ".embeddings"] = table
        sd[name + ".offsets"] = off

    def mlp(name, din, dout, dh, nl):
        for l in range(nl):
            n = f"{name}.net.{l}.weight"
            sd[n] = _linear(n, seed, dout if l == nl - 1 else dh, din if l == 0 else dh)

    grid("position_embedder", 3, hp["desired_resolution"] * hp["bound"])
    mlp("ambient_net", 32 + cout, hp["ambient_out_dim"], hp["hidden_dim_ambient"], hp["num_layers_ambient"])
    grid("ambient_embedder", hp["ambient_out_dim"], hp["desired_resolution"])
    mlp("sigma_net", 32 + 32, 1 + hp["geo_feat_dim"], hp["hidden_dim_sigma"], hp["num_layers_sigma"])
    last = f"sigma_net.net.{hp['num_layers_sigma'] - 1}.weight"
    sd[last][0] = (np.float32(SIGMA_ROW_ABS) * np.abs(sd[last][0]) + np.float32(SIGMA_ROW_LIN) * sd[last][0]) * np.float32(sigma_row_scale)
    mlp("color_net", 16 + hp["geo_feat_dim"] + hp["individual_embedding_dim"], 3, hp["hidden_dim_color"], hp["num_layers_color"])
    if torso:
        grid("torso_embedder", 2, 2048)
        din = 42 + 54 + hp["torso_individual_embedding_dim"]
        if hp.get("torso_head_aware", False):   # radnerf_torso.py:36-46
            for i, (o, ii) in zip((0, 2, 4), ((16, 4), (32, 16), (16, 32))):
                n = f"head_color_weights_encoder.{i}"
                sd[n + ".weight"] = _linear(n + ".weight", seed, o, ii) * np.float32(2.0)
                sd[n + ".bias"] = _uniform(n + ".bias", seed, (o,), 1.0 / math.sqrt(ii))
            din += 16
        mlp("torso_deform_net", din, 2, 64, 3)
        mlp("torso_canonicial_net", 32 + din, 4, 32, 3)
        w = sd["torso_canonicial_net.net.2.weight"]
        w[0] = np.float32(TORSO_ALPHA_ROW_ABS) * np.abs(w[0]) + np.float32(TORSO_ALPHA_ROW_LIN) * w[0]
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items())


def _morton3d(x, y, z):
    def expand(v):
        v = v.astype(np.uint64)
        v = (v * 0x00010001) & 0xFF0000FF
        v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3
        v = (v * 0x00000005) & 0x49249249
        return v
    return (expand(x) | (expand(y) << 1) | (expand(z) << 2)).astype(np.int64)


def head_occupancy(G: int, bound: float, seed: int = 0) -> np.ndarray:
    """bool [G,G,G] indexed [nx,ny,nz]; world axes are ngp axes: +x is image-up, the camera looks down -y
    and +z is image-right for the base pose of `make_poses`."""
    c = (np.arange(G, dtype=np.float64) + 0.5) / G * 2 * bound - bound
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    r = _rng("head_shape", seed)
    cx, cz = 0.12 + 0.02 * r.uniform(-1, 1), 0.02 * r.uniform(-1, 1)
    ax, ay, az = 0.40 + 0.03 * r.uniform(-1, 1), 0.30, 0.33 + 0.03 * r.uniform(-1, 1)
    head = ((X - cx) / ax) ** 2 + (Y / ay) ** 2 + ((Z - cz) / az) ** 2 <= 1.0
    neck = (X > -0.62) & (X < cx - 0.2) & ((Y + 0.03) ** 2 + (Z - cz) ** 2 <= 0.16 ** 2)
    if seed >= 1000:   # second identity (BASELINE.json configs[4]): narrower, taller head with a hair cap, thicker neck, off-centre
        head = ((X - cx - 0.05) / (ax * 1.1)) ** 2 + ((Y + 0.02) / (ay * 0.9)) ** 2 + ((Z - cz + 0.04) / (az * 0.85)) ** 2 <= 1.0
        cap = ((X - cx - 0.28) / 0.22) ** 2 + (Y / 0.26) ** 2 + ((Z - cz + 0.04) / 0.30) ** 2 <= 1.0
        neck = (X > -0.70) & (X < cx - 0.15) & ((Y + 0.02) ** 2 + (Z - cz + 0.03) ** 2 <= 0.20 ** 2)
        return head | cap | neck
    return head | neck


def make_density_bitfield(G=128, cascade=1, bound=1.0, seed=0) -> np.ndarray:
    idx = np.arange(G)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    flat = np.zeros(cascade * G ** 3, dtype=np.uint8)
    m = _morton3d(X.ravel(), Y.ravel(), Z.ravel())
    for c in range(cascade):  # cascade c covers [-min(2^c, bound), min(2^c, bound)]^3 (raymarching.cu:883-892)
        flat[c * G ** 3 + m] = head_occupancy(G, min(2.0 ** c, float(bound)), seed).ravel()
    return np.packbits(flat.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)


def make_density_grid_torso(G=128, seed=0) -> np.ndarray:
    """[G*G] soft shoulder mask sampled by F.grid_sample(bg_coords) (radnerf_torso.py:167): bg_coords are
    (row, col) in [-1,1]; grid_sample's x indexes the last axis, so axis 1 <- image row, axis 0 <- image col."""
    u = np.linspace(-1, 1, G)
    col, row = np.meshgrid(u, u, indexing="ij")  # [axis0 = col, axis1 = row]
    half_w = 0.25 + 0.55 * np.clip((row - 0.2) / 0.8, 0, 1)
    inside = (row > 0.2) & (np.abs(col) < half_w)
    soft = np.clip((half_w - np.abs(col)) / 0.1, 0, 1) * np.clip((row - 0.2) / 0.1, 0, 1)
    return np.where(inside, 0.05 + 0.95 * soft, 0.0).astype(np.float32).reshape(-1)


def intrinsics(H: int, W: int, fovy_deg: float = 21.24) -> np.ndarray:
    """[fx, fy, cx, cy]; OrbitCamera.intrinsics, radnerf_gui.py:38-41."""
    focal = H / (2 * math.tan(math.radians(fovy_deg) / 2))
    return np.array([focal, focal, W // 2, H // 2], dtype=np.float64)


def _rot(axis, a):
    c, s = math.cos(a), math.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]),
            "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def make_poses(T: int, seed: int = 7, radius: float = 3.35) -> np.ndarray:
    """[T,4,4] float32 cam2world in ngp axes: camera on +y looking at the origin, image-up = +x, gently
    wandering (yaw/pitch <= 8 deg, translation jitter <= 0.03) as a sum of three sinusoids."""
    r = _rng("poses", seed)
    ph, fr = r.uniform(0, 2 * math.pi, (3, 5)), r.uniform(0.01, 0.08, (3, 5))
    base = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], dtype=np.float64)
    poses = np.zeros((T, 4, 4), dtype=np.float32)
    for t in range(T):
        w = [float(np.sin(fr[:, k] * t + ph[:, k]).sum() / 3) for k in range(5)]
        R = _rot("x", math.radians(8) * w[0]) @ _rot("z", math.radians(8) * w[1]) @ base
        cam = R @ np.array([0, 0, -radius]) + 0.03 * np.array([w[2], w[3], w[4]])
        poses[t, :3, :3] = R
        poses[t, :3, 3] = cam
        poses[t, 3, 3] = 1
    return poses


def make_landmarks(T: int, seed: int = 11, rho: float = 0.9) -> np.ndarray:
    """[T,204] float32, AR(1) with unit stationary variance (already 'normalised': mean 0, std 1)."""
    r = _rng("landmarks", seed)
    x = np.zeros((T, 204), dtype=np.float64)
    x[0] = r.standard_normal(204)
    for t in range(1, T):
        x[t] = rho * x[t - 1] + math.sqrt(1 - rho * rho) * r.standard_normal(204)
    return x.astype(np.float32)


def make_audio_features(T: int, C: int, win: int = 16, seed: int = 13, rho: float = 0.8) -> np.ndarray:
    """[T, win, C] float32: per video frame a `win`-step window of a C-channel audio-feature stream (esperanto 44 / deepspeech 29 channels;
    the layout of `esperanto_win` / `deepspeech_win`, tasks/radnerfs/dataset_utils.py:91-94): an AR(1) stream at two feature steps per video
    frame, windows centred on the frame and zero padded at the sequence ends as the extractors do."""
    r = _rng("audio_features", seed)
    n = 2 * T + win
    x = np.zeros((n, C), dtype=np.float64)
    x[0] = r.standard_normal(C)
    for t in range(1, n):
        x[t] = rho * x[t - 1] + math.sqrt(1 - rho * rho) * r.standard_normal(C)
    x[:win // 2] = 0
    x[-(win // 2):] = 0
    return np.stack([x[2 * t:2 * t + win] for t in range(T)]).astype(np.float32)


def make_bg_img(H: int, W: int) -> np.ndarray:
    """[H,W,3] float32 in [0,1]: smooth gradient plus a checker, so blending errors are visible."""
    yy, xx = np.meshgrid(np.arange(H) / max(H - 1, 1), np.arange(W) / max(W - 1, 1), indexing="ij")
    chk = (((np.arange(H)[:, None] * 8 // H) + (np.arange(W)[None, :] * 8 // W)) % 2).astype(np.float64)
    img = np.stack([0.2 + 0.6 * xx, 0.25 + 0.5 * yy, 0.3 + 0.4 * (1 - xx) * yy], -1) + 0.08 * chk[..., None]
    return np.clip(img, 0, 1).astype(np.float32)


def make_sequence(T: int, H: int = 512, W: int = 512, hp: dict = None, seed: int = 0, radius: float = None) -> dict:
    """Everything `run_model` needs for T frames except rays (host numpy; rays are generated per frame):
    cond_wins [T,5,1,204] (landmark-driven; [T,8,16,44] for the audio-driven configs), poses [T,4,4] (smoothed, ngp axes), intrinsics [4],
    bg_img [H*W,3]."""
    from .lm3d import cond_windows, get_win_conds, normalize_and_smooth
    from .utils import smooth_camera_path
    hp = hp or {}
    if hp.get("cond_type", "idexp_lm3d_normalized") != "idexp_lm3d_normalized":
        # audio-driven RAD-NeRF (egs/egs_bases/radnerf/radnerf.yaml:4-7): per frame a [cond_win, C] feature window; cond_wins = the smo_win
        # frames around it, zero padded at the ends (get_audio_features att_mode 2, modules/radnerfs/utils.py:85-101 via dataset_utils.py:158)
        feats = make_audio_features(T, cond_in_dim(hp), hp.get("cond_win_size", 16), seed=13 + seed)
        wins = np.stack([get_win_conds(feats, i, hp.get("smo_win_size", 8), "zero") for i in range(T)])
    else:
        lm = make_landmarks(T, seed=11 + seed)
        lm_norm = normalize_and_smooth(lm, 0.0, 1.0, hp.get("infer_lm3d_clamp_std", 2.5))
        wins = cond_windows(lm_norm, hp.get("cond_win_size", 1), hp.get("smo_win_size", 5))
    poses = make_poses(T, seed=7 + seed) if radius is None else make_poses(T, seed=7 + seed, radius=radius)   # radius < 3.35: the head fills the frame
    if hp.get("infer_smooth_camera_path", True):
        poses = smooth_camera_path(poses.copy(), hp.get("infer_smooth_camera_path_kernel_size", 7)).astype(np.float32)
    return {
        "cond_wins": wins,
        "poses": poses, "intrinsics": intrinsics(H, W), "bg_img": make_bg_img(H, W).reshape(-1, 3), "H": H, "W": W,
    }


def make_dataset_dict(T: int = 9, H: int = 64, W: int = 64, seed: int = 3):
    """A `trainval_dataset.npy`-shaped dict (data_gen/nerf/binarizer.py:175-199: train_samples / val_samples with AD-NeRF convention 4x4 `c2w`,
    H, W, focal, cx, cy, uint8 bg_img, landmark statistics) for the synthetic orbit camera, and the ngp-axes poses it encodes -- what
    geneface_amd.lm3d_radnerf_infer.RADNeRFPoseSource reads in place of the real file."""
    rng = np.random.default_rng(seed)
    ngp = make_poses(T)
    scale = 4.0
    c2w = []
    for p in ngp:                                         # invert nerf_matrix_to_ngp: rows (y,z,x) <- (x,y,z), t / scale
        m = np.eye(4, dtype=np.float32)
        m[0, :3], m[1, :3], m[2, :3] = [p[2, 0], -p[2, 1], -p[2, 2]], [p[0, 0], -p[0, 1], -p[0, 2]], [p[1, 0], -p[1, 1], -p[1, 2]]
        m[0, 3], m[1, 3], m[2, 3] = p[2, 3] / scale, p[0, 3] / scale, p[1, 3] / scale
        c2w.append(m)
    K = intrinsics(H, W)
    samples = [{"c2w": m, "idx": i} for i, m in enumerate(c2w)]
    return {"train_samples": samples[:T - 2], "val_samples": samples[T - 2:], "H": H, "W": W, "focal": float(K[0]), "cx": float(K[2]),
            "cy": float(K[3]), "bg_img": (make_bg_img(H, W).reshape(H, W, 3) * 255).astype(np.uint8),
            "idexp_lm3d_mean": rng.normal(size=(1, 68, 3)).astype(np.float32) * 0.1,
            "idexp_lm3d_std": (1 + 0.1 * rng.random(size=(1, 68, 3))).astype(np.float32)}, ngp
