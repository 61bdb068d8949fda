"""Loader for oracle/_ref (TEST INFRASTRUCTURE ONLY): the reference's own four extension modules, built for gfx950 from the
sources under /root/reference by oracle/refbuild/build_ref.py.  The prebuilt modules travel to the GPU box; nothing here reads
/root/reference at run time.  They need a GPU (their launchers dereference device pointers), take CUDA tensors and launch on
the null stream -- exactly as the reference does.

    mods = ref_kernels.load("fast")          # (raymarching_face, gridencoder, shencoder, freqencoder), hipcc's default fp contraction
    mods = ref_kernels.load("off")           # the same sources with -ffp-contract=off
    with radnerf_ref.kernel_backend(mods): radnerf_ref.render(sd_on_gpu, ...)      # the reference pipeline on the MI355X
"""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
NAMES = ("ref_raymarching_face", "ref_gridencoder", "ref_shencoder", "ref_freqencoder")
_loaded = {}


def _path(name, contract):
    return os.path.join(REF_DIR, name + ("" if contract == "fast" else "_nofma") + ".so")


def available(contract: str = "fast") -> bool:
    return all(os.path.exists(_path(n, contract)) for n in NAMES)


def load(contract: str = "fast"):
    if contract not in _loaded:
        if not available(contract):
            raise FileNotFoundError(f"oracle/_ref is not built ({_path(NAMES[0], contract)}); run oracle/refbuild/build_ref.py where "
                                    "/root/reference exists")
        import torch  # noqa: F401  (the modules link against libtorch)
        mods = []
        for n in NAMES:
            modname = os.path.basename(_path(n, contract))[:-3]
            spec = importlib.util.spec_from_file_location(modname, _path(n, contract))
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            mods.append(m)
        _loaded[contract] = tuple(mods)
    return _loaded[contract]
