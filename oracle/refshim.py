"""Run the reference's *own* Python layers on CPU on top of the C oracle (TEST INFRASTRUCTURE ONLY).

Only usable where /root/reference exists (this build container, never the GPU box).  It
  * registers the oracle's ops under the four extension names the reference wrappers import first
    (raymarching/raymarching.py:9-12, gridencoder/grid.py:9-12, shencoder/sphere_harmonics.py:9-12,
    freqencoder/freq.py:9-12), so the unmodified NeRFRenderer.render / RADNeRF.forward /
    RADNeRFTorso.render execute,
  * stubs the third-party imports the reference pulls in at module scope but never touches on the
    render path (cv2, lpips, trimesh, mcubes, tensorboardX, imageio, ...),
  * makes Tensor.cuda() the identity (the wrappers call it unconditionally on CPU tensors).
Used by tests/golden/make_golden.py to produce the committed golden vectors.
"""
import contextlib
import importlib
import os
import sys
import types

import torch

from . import kernels as _k

REFERENCE_ROOT = os.environ.get("GENEFACE_REFERENCE_ROOT", "/root/reference")

_STUB_NAMES = ["cv2", "lpips", "trimesh", "mcubes", "tensorboardX", "imageio", "dearpygui",
               "dearpygui.dearpygui", "face_alignment", "librosa", "python_speech_features",
               "skimage", "skimage.transform", "pytorch3d", "torch.utils.tensorboard"]


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modules", "radnerfs"))


def _module_from_class(name, cls):
    m = types.ModuleType(name)
    for k, v in vars(cls).items():
        if isinstance(v, staticmethod):
            setattr(m, k, v.__func__)
    m.__doc__ = f"C-oracle stand-in for the reference CUDA extension `{name}`"
    return m


class _Anything(types.ModuleType):
    """A module whose every attribute is a harmless callable/class placeholder."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)

        class _Placeholder:
            def __init__(self, *a, **k):
                pass

            def __call__(self, *a, **k):
                raise RuntimeError(f"stubbed dependency {self.__class__.__qualname__} was actually used")

        _Placeholder.__qualname__ = f"{self.__name__}.{item}"
        return _Placeholder


def install(root: str = None, backend: str = "oracle"):
    """Idempotently make `import modules.radnerfs...` work from /root/reference on CPU.
    root: another place to import the reference's Python from (the staged archive oracle/_refpy/geneface_refpy.zip on the GPU box).
    backend "oracle": the four extension names resolve to the C oracle (CPU).  backend "compat": they are left to
    geneface_amd.compat.install(), i.e. the reference's Python then drives the PRODUCT's kernels on the GPU -- the one configuration in
    which product code and this test infrastructure meet, and only inside tests/."""
    root = root or REFERENCE_ROOT
    if root == REFERENCE_ROOT and not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if root not in sys.path:
        sys.path.insert(0, root)
    if backend == "oracle":
        sys.modules.setdefault("_raymarching_face", _module_from_class("_raymarching_face", _k.raymarching_face))
        sys.modules.setdefault("_gridencoder", _module_from_class("_gridencoder", _k.gridencoder))
        sys.modules.setdefault("_shencoder", _module_from_class("_shencoder", _k.shencoder))
        sys.modules.setdefault("_freqencoder", _module_from_class("_freqencoder", _k.freqencoder))
    for name in _STUB_NAMES:
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = _Anything(name)
    try:
        importlib.import_module("numba")
    except Exception:  # decorators must stay transparent: data_gen/nerf/binarizer.py pulls numba-jitted helpers in
        nb = types.ModuleType("numba")

        def _decorator(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        nb.jit = nb.njit = nb.vectorize = _decorator
        sys.modules["numba"] = nb


@contextlib.contextmanager
def cpu_mode():
    """Tensor.cuda()/Module.cuda() become no-ops while the reference code runs."""
    orig_t, orig_m = torch.Tensor.cuda, torch.nn.Module.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = orig_t, orig_m


def reference_hparams(torso: bool):
    """The May lm3d_radnerf(+torso) config resolved through the reference's own yaml chain
    (utils/commons/hparams.py:51-72 semantics: depth-first base_config, later overrides)."""
    import yaml

    def load(path):
        with open(os.path.join(REFERENCE_ROOT, path)) as f:
            cfg = yaml.safe_load(f) or {}
        out = {}
        bases = cfg.pop("base_config", [])
        if isinstance(bases, str):
            bases = [bases]
        for b in bases:
            if b.startswith("."):
                b = os.path.normpath(os.path.join(os.path.dirname(path), b))
            out.update(load(b))
        out.update(cfg)
        return out

    name = "lm3d_radnerf_torso.yaml" if torso else "lm3d_radnerf.yaml"
    return load(os.path.join("egs/datasets/videos/May", name))


def build_reference_model(torso: bool, overrides=None):
    """Instantiate the reference's RADNeRF / RADNeRFTorso with the May hparams (CPU)."""
    install()
    hp = reference_hparams(torso)
    hp.update(overrides or {})
    from utils.commons.hparams import hparams as global_hp  # the reference's global dict
    global_hp.clear()
    global_hp.update(hp)
    if torso:
        from modules.radnerfs.radnerf_torso import RADNeRFTorso as cls
    else:
        from modules.radnerfs.radnerf import RADNeRF as cls
    model = cls(hp)
    model.eval()
    return model, hp
