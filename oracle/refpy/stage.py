"""Stage the reference's render-path PYTHON for the GPU box (TEST INFRASTRUCTURE ONLY; same status as oracle/_ref).

    python -m oracle.refpy.stage            (build container only: reads /root/reference)

/root/reference does not exist on the GPU box, so the reference's own, unmodified Python layers -- NeRFRenderer.render,
RADNeRF.forward, RADNeRFTorso.render, the cond encoder and the four autograd wrappers (raymarching.py, grid.py, sphere_harmonics.py,
freq.py) -- could only ever run here, on CPU, over the C oracle.  To run them ON the MI355X over geneface_amd.compat (the extension seam
end to end: INTEGRATION.md seam 1, tests/test_gpu_refpy.py) they have to travel.  This script packs them, byte for byte as they lie under
/root/reference, into ONE archive

    oracle/_refpy/geneface_refpy.zip        git-ignored (never in history), not gpurun-ignored (travels like the built .so files)

which Python imports from directly (zipimport): no reference source file is ever unpacked into, or committed to, this repository.
Contents: modules/radnerfs/**/*.py (without the setup.py / JIT backend.py build scripts), utils/commons/{hparams,os_utils}.py (the global
`hparams` dict the modules import), empty __init__.py for the reference's implicit namespace packages, and refpy_hparams.json = the May
lm3d_radnerf / lm3d_radnerf_torso configurations resolved through the reference's own yaml chain (oracle/refshim.reference_hparams).
"""
import hashlib
import json
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
OUT_DIR = os.path.join(ROOT, "oracle", "_refpy")
ARCHIVE = os.path.join(OUT_DIR, "geneface_refpy.zip")
EXTRA = ["utils/commons/hparams.py", "utils/commons/os_utils.py"]
SKIP = {"setup.py", "backend.py"}


def available() -> bool:
    from oracle import refshim
    return refshim.available()


def staged() -> bool:
    return os.path.exists(ARCHIVE)


def _files(ref_root):
    out = []
    # modules/nerfs (round 6): the legacy pure-PyTorch renderer -- Lm3dNeRF + render_dynamic_face, baseline B2 of BASELINE.md section 3 --
    # so that bench.py's `cpu_baseline.legacy_nerf` times the reference's OWN module on the GPU box's host cores (kind "reference") instead of
    # the restatement oracle/legacy_nerf_ref.py (kind "port")
    for sub in ("radnerfs", "nerfs"):
        for d, _, names in os.walk(os.path.join(ref_root, "modules", sub)):
            for n in sorted(names):
                if n.endswith(".py") and n not in SKIP:
                    out.append(os.path.relpath(os.path.join(d, n), ref_root))
    return sorted(out) + EXTRA


def build(force: bool = False) -> str:
    from oracle import refshim
    ref_root = refshim.REFERENCE_ROOT
    files = _files(ref_root)
    newest = max(os.path.getmtime(os.path.join(ref_root, f)) for f in files)
    if not force and staged() and os.path.getmtime(ARCHIVE) >= max(newest, os.path.getmtime(__file__)):
        return ARCHIVE
    os.makedirs(OUT_DIR, exist_ok=True)
    manifest, dirs = {}, set()
    with zipfile.ZipFile(ARCHIVE, "w", zipfile.ZIP_DEFLATED) as z:
        for rel in files:
            data = open(os.path.join(ref_root, rel), "rb").read()
            z.writestr(rel, data)
            manifest[rel] = hashlib.sha256(data).hexdigest()
            d = os.path.dirname(rel)
            while d:
                dirs.add(d)
                d = os.path.dirname(d)
        for d in sorted(dirs):   # the reference relies on implicit namespace packages; make them explicit inside the archive
            init = f"{d}/__init__.py"
            if init not in manifest:
                z.writestr(init, "")
        z.writestr("refpy_hparams.json", json.dumps({"head": refshim.reference_hparams(False), "torso": refshim.reference_hparams(True)}, default=str))
        z.writestr("MANIFEST.json", json.dumps({"source": "yerfor/GeneFace, files unmodified", "sha256": manifest}, indent=1))
    return ARCHIVE


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    print(build(force="--force" in sys.argv))
