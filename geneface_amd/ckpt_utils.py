"""Checkpoint discovery and loading with the reference's semantics (utils/commons/ckpt_utils.py:7-66).

The reference's Trainer writes `<work_dir>/model_ckpt_steps_<N>.ckpt` with LEGACY (non-zip) serialization
(`torch.save(..., _use_new_zipfile_serialization=False)`, utils/commons/trainer.py:454-458) and this layout (:460-473):

    {'epoch': int, 'global_step': int, 'checkpoint_callback_best': float | numpy scalar,
     'optimizer_states': [optimizer.state_dict(), ...],
     'state_dict': {'model': OrderedDict, <other task children with parameters, e.g. 'criterion_lpips'>: OrderedDict}}

`load_ckpt` keeps the reference's signature and behaviour: `ckpt_base_dir` is a file or a directory (newest step wins, or `steps`),
the state dict is either nested per child (`state_dict['model']`, dotted `model_name` selects a sub-module by prefix) or flat with
`'<model_name>.'` prefixes, `strict=False` additionally drops shape-mismatched keys, a missing checkpoint asserts when `force`.
"""
import glob
import os
import pickle
import re

import torch


def get_all_ckpts(work_dir, steps=None):
    """Checkpoint paths of `work_dir`, newest step first (ckpt_utils.py:17-24)."""
    pattern = f"{work_dir}/model_ckpt_steps_*.ckpt" if steps is None else f"{work_dir}/model_ckpt_steps_{steps}.ckpt"
    return sorted(glob.glob(pattern), key=lambda x: -int(re.findall(r".*steps_(\d+)\.ckpt", x)[0]))


def _torch_load(path):
    """torch.load(map_location='cpu') of a checkpoint this project's users trained themselves.  torch >= 2.6 defaults to the restricted
    unpickler; the reference's files carry numpy scalars (`checkpoint_callback_best`) and optimizer state next to the tensors, so the
    restricted load is tried with numpy's scalar reconstruction allow-listed, and a file that still needs more falls back to the plain
    unpickler the reference itself uses (ckpt_utils.py:13,30) -- same trust model as the reference: load only checkpoints you wrote."""
    try:
        import numpy as np
        allow = [np.dtype, np.float64, np.float32, np.int64]
        for mod in ("numpy._core.multiarray", "numpy.core.multiarray"):
            try:
                allow.append(__import__(mod, fromlist=["scalar"]).scalar)
                break
            except (ImportError, AttributeError):
                continue
        allow += [type(np.dtype(t)) for t in ("float64", "float32", "int64")]
        with torch.serialization.safe_globals(allow):
            return torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError) as e:
        # The full unpickler can execute code: say that it is about to run, and why the restricted load refused the file (the message names
        # the global it blocked).  GENEFACE_AMD_SAFE_LOAD_ONLY=1 turns the fallback off for deployments that load third-party files.
        if os.environ.get("GENEFACE_AMD_SAFE_LOAD_ONLY", "") not in ("", "0"):
            raise RuntimeError(f"{path}: the restricted unpickler refused this checkpoint and GENEFACE_AMD_SAFE_LOAD_ONLY is set: {e}") from e
        import warnings
        warnings.warn(f"{path}: restricted checkpoint load failed ({str(e).splitlines()[0][:200]}); falling back to the full unpickler, as the "
                      f"reference does (utils/commons/ckpt_utils.py:13) -- load only checkpoints you trust", RuntimeWarning, stacklevel=2)
        return torch.load(path, map_location="cpu", weights_only=False)


def get_last_checkpoint(work_dir, steps=None):
    """(checkpoint dict, path) of the newest checkpoint, or (None, None) (ckpt_utils.py:7-14)."""
    paths = get_all_ckpts(work_dir, steps)
    if not paths:
        return None, None
    return _torch_load(paths[0]), paths[0]


def select_state_dict(checkpoint, model_name="model"):
    """The state dict of `model_name` inside a checkpoint: nested per task child, or flat with '<model_name>.' prefixes
    (ckpt_utils.py:35-47)."""
    state_dict = checkpoint["state_dict"]
    if len([k for k in state_dict.keys() if "." in k]) > 0:
        return {k[len(model_name) + 1:]: v for k, v in state_dict.items() if k.startswith(f"{model_name}.")}
    if "." not in model_name:
        return state_dict[model_name]
    base, rest = model_name.split(".")[0], model_name[len(model_name.split(".")[0]) + 1:]
    return {k[len(rest) + 1:]: v for k, v in state_dict[base].items() if k.startswith(f"{rest}.")}


def load_ckpt(cur_model, ckpt_base_dir, model_name="model", force=True, strict=True, steps=None):
    """utils/commons/ckpt_utils.py:27-66, same arguments and behaviour."""
    if os.path.isfile(ckpt_base_dir):
        base_dir, ckpt_path = os.path.dirname(ckpt_base_dir), ckpt_base_dir
        checkpoint = _torch_load(ckpt_base_dir)
    else:
        base_dir = ckpt_base_dir
        checkpoint, ckpt_path = get_last_checkpoint(ckpt_base_dir, steps)
    if checkpoint is None:
        e_msg = f"| ckpt not found in {base_dir}."
        assert not force, e_msg
        print(e_msg)
        return None
    state_dict = dict(select_state_dict(checkpoint, model_name))
    if not strict:
        cur = cur_model.state_dict()
        for key in [k for k, v in state_dict.items() if k in cur and cur[k].shape != v.shape]:
            print("| Unmatched keys: ", key, cur[key].shape, state_dict[key].shape)
            del state_dict[key]
    cur_model.load_state_dict(state_dict, strict=strict)
    print(f"| load '{model_name}' from '{ckpt_path}'.")
    return checkpoint
