"""The hparams the RAD-NeRF render path reads, with the May `lm3d_radnerf(_torso)` values.

The reference resolves these through a yaml chain
(egs/datasets/videos/May/lm3d_radnerf_torso.yaml -> egs/egs_bases/radnerf/lm3d_radnerf.yaml
 -> egs/egs_bases/radnerf/base.yaml:49-122) into a global mutable dict
(utils/commons/hparams.py:25-132) and forwards the *whole* dict into render(**hparams)
(tasks/radnerfs/radnerf.py:169).  We keep plain-dict semantics; only the keys the hot path
reads are listed.  tests/test_hparams.py checks these values against the reference yaml chain
when the reference tree is present.
"""
import copy

_MAY_BASE = {
    # NeRF / marcher (base.yaml:52-80)
    "near": 0.3, "far": 0.9, "n_rays": 65536, "cuda_ray": True,
    "max_steps": 16, "num_steps": 16, "upsample_steps": 0, "update_extra_interval": 16, "max_ray_batch": 4096,
    "min_near": 0.05, "bound": 1, "camera_scale": 4.0, "camera_offset": [0, 0, 0],
    "grid_size": 128, "desired_resolution": 2048, "log2_hashmap_size": 16,
    "dt_gamma": 0.00390625, "density_thresh": 10, "density_thresh_torso": 0.01, "torso_shrink": 0.8,
    "smooth_lips": False,
    # network (base.yaml:85-102)
    "grid_type": "tiledgrid", "grid_interpolation_type": "linear", "with_att": True, "use_window_cond": True,
    "torso_head_aware": False,
    "num_layers_sigma": 3, "hidden_dim_sigma": 128, "geo_feat_dim": 128,
    "num_layers_color": 2, "hidden_dim_color": 128, "cond_out_dim": 64,
    "num_layers_ambient": 3, "hidden_dim_ambient": 128, "ambient_out_dim": 2,
    "individual_embedding_num": 13000, "individual_embedding_dim": 4, "torso_individual_embedding_dim": 8,
    # lm3d_radnerf.yaml:4-7
    "cond_type": "idexp_lm3d_normalized", "cond_win_size": 1, "smo_win_size": 5,
    "task_cls": "tasks.radnerfs.radnerf.RADNeRFTask",
    # infer (base.yaml:104-116)
    "infer_scale_factor": 1.0, "infer_lm3d_clamp_std": 2.5, "infer_smooth_camera_path": True,
    "infer_smooth_camera_path_kernel_size": 7, "infer_bg_img_fname": "",
    # gui (base.yaml:118-123)
    "gui_w": 512, "gui_h": 512, "gui_radius": 3.35, "gui_fovy": 21.24, "gui_max_spp": 1,
    "amp": True, "seed": 9999, "video_id": "May",
}

_MAY_TORSO = {
    "task_cls": "tasks.radnerfs.radnerf_torso.RADNeRFTorsoTask",
    "head_model_dir": "checkpoints/May/lm3d_radnerf",
    "torso_train_mode": 1,
}


def may_hparams(torso: bool = True) -> dict:
    hp = copy.deepcopy(_MAY_BASE)
    if torso:
        hp.update(copy.deepcopy(_MAY_TORSO))
    return hp


#: The other RAD-NeRF experiment files the reference ships (each is the May / Obama yaml chain with these keys changed); name -> (overrides,
#: the yaml files that state them).  tests/test_vs_reference.py resolves every file through the reference's own set_hparams and compares.
#: The hash / smoothstep head files have no torso twin of their own: their torso stage is lm3d_radnerf_torso.yaml with `head_model_dir`
#: pointed at them.  grid_type / grid_interpolation_type select the HEAD grids (position, ambient: radnerf.py:40-48); the torso grid is a
#: linear tiled grid whatever they say (radnerf_torso.py:30).
VARIANTS = {
    "default": ({}, ["egs/datasets/videos/May/lm3d_radnerf.yaml", "egs/datasets/videos/May/lm3d_radnerf_torso.yaml"]),
    "hash": ({"grid_type": "hashgrid", "individual_embedding_num": 10000}, ["egs/datasets/videos/May/lm3d_radnerf_hash.yaml"]),
    "hash_smoothstep": ({"grid_type": "hashgrid", "grid_interpolation_type": "smoothstep", "individual_embedding_num": 10000},
                        ["egs/datasets/videos/May/lm3d_radnerf_hash_smoothstep.yaml"]),
    "smoothstep": ({"grid_interpolation_type": "smoothstep", "individual_embedding_num": 10000},
                   ["egs/datasets/videos/May/lm3d_radnerf_smoothstep.yaml"]),
    "head_aware": ({"torso_head_aware": True}, ["egs/datasets/videos/May/lm3d_radnerf_torso_head_aware.yaml"]),
    "audio": ({"cond_type": "esperanto", "cond_win_size": 16, "smo_win_size": 8, "individual_embedding_num": 10000, "video_id": "Obama"},
              ["egs/datasets/videos/Obama/radnerf.yaml", "egs/datasets/videos/Obama2/radnerf_torso.yaml"]),
}


def variant_hparams(name: str, torso: bool = True) -> dict:
    """The hot-path hparams of one of the reference's shipped RAD-NeRF experiment files (VARIANTS)."""
    hp = may_hparams(torso)
    hp.update(copy.deepcopy(VARIANTS[name][0]))
    if name == "audio" and torso:
        hp["head_model_dir"] = "checkpoints/Obama2/radnerf"
    return hp


#: the process-global dict, mirroring `utils.commons.hparams.hparams`
hparams = {}


def set_hparams(new: dict) -> dict:
    hparams.clear()
    hparams.update(new)
    return hparams


# ----------------------------------------------------------------------------------------------- yaml chain
def _merge(old: dict, new: dict):
    """Nested dicts merge key by key, everything else is replaced (utils/commons/hparams.py:17-22)."""
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(old.get(k), dict):
            _merge(old[k], v)
        else:
            old[k] = v


def load_config(config_fn: str, hparams_str: str = "", root: str = None) -> dict:
    """Resolve one of the reference's experiment files (egs/datasets/videos/<id>/lm3d_radnerf(_torso).yaml) the way
    `set_hparams(config=..., hparams_str=...)` does (utils/commons/hparams.py:51-107): depth-first `base_config` inheritance with
    every file visited once, later files overriding earlier ones, then the command-line style overrides "a=1,b.c=2,d=[1 1 1]"
    (typed after the value they replace).  `base_config` entries are relative to the working directory of the reference
    (`root`, default: the current directory) or, when they start with '.', to the file that names them."""
    import os

    import yaml
    root = root or os.getcwd()
    seen = set()

    def resolve(path):
        return path if os.path.isabs(path) else os.path.join(root, path)

    def load(fn):
        full = resolve(fn)
        if not os.path.exists(full):
            return {}
        with open(full) as f:
            cfg = yaml.safe_load(f) or {}
        seen.add(fn)
        if "base_config" not in cfg:
            return cfg
        bases = cfg["base_config"] if isinstance(cfg["base_config"], list) else [cfg["base_config"]]
        out = {}
        for b in bases:
            if b.startswith("."):
                b = os.path.normpath(os.path.join(os.path.dirname(fn), b))
            if b not in seen:
                _merge(out, load(b))
        _merge(out, cfg)
        return out

    hp = load(config_fn)
    if not hp:
        raise FileNotFoundError(f"config {resolve(config_fn)!r} not found or empty")
    if hparams_str:
        import ast
        for item in hparams_str.split(","):
            k, v = item.split("=")
            v = v.strip("'\" ")
            node = hp
            *parents, leaf = k.split(".")
            for p in parents:
                node = node[p]
            old = node[leaf]
            if v in ("True", "False") or isinstance(old, (bool, list, dict)):
                node[leaf] = ast.literal_eval(v.replace(" ", ",") if isinstance(old, list) else v)
            else:
                node[leaf] = type(old)(v)
    return hp
