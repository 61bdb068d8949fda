"""hipcc build of libgeneface_hip.so (gfx950 only, in-tree so it travels with the repo snapshot).

    python -m geneface_amd.csrc.build [--force]

Every translation unit is compiled to an object (cached by mtime of the source and the headers),
then linked into geneface_amd/csrc/libgeneface_hip.so.  No torch headers are involved: the library
is a plain C-ABI (include/geneface_hip.h) consumed through ctypes.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libgeneface_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"--offload-arch={ARCH}",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


# EVERY translation unit is built WITHOUT the SLP vectoriser, i.e. without packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 /
# v_pk_add_f32).  In k_head_phase<true> the low half of one v_pk_fma_f32 of the 2-D grid lookup -- the one fed by a broadcast-form
# v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[0,1] -- intermittently lost its product in lanes 32..63 whenever two workgroups shared a CU
# (NOTES.md 4.7, tools/fast_diag.py).  The mechanism is not established, so since round 3 the instruction class is banned from the whole
# library, not just from the kernel it was caught in (encoders.hip's k_encode8<2> held two instances of exactly that pair):
# tests/test_build_invariants.py disassembles every code object and asserts the count is zero.  Scalar FMAs compute the same values (the
# packed form was only ever two independent FMAs); measured cost: head fp32 719 vs 717 fps, fast tier 1890 vs 1845 fps (round 2, A/B on
# one box), stand-alone ops and the training step within noise (round 3, profiles/round3/).
COMMON.append("-fno-slp-vectorize")
PER_FILE = {}


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")) + glob.glob(os.path.join(HERE, "*.cpp")))


def headers():
    return sorted(glob.glob(os.path.join(HERE, "*.hpp")) + glob.glob(os.path.join(HERE, "*.inc")) + glob.glob(os.path.join(HERE, "..", "..", "include", "*.h")))


def source_digest() -> str:
    """sha256 (16 hex digits) over every kernel source, header and table of the library plus the compile flags: what a measurement taken on
    one tree stamps itself with (tools/pmc_summary.py) and what bench.py compares a committed PMC summary against (`traffic_stale`).
    Content, not mtimes: a snapshot on the GPU box or a fresh checkout has new mtimes and the same digest."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(sources() + headers()):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(COMMON).encode())
    return h.hexdigest()[:16]


def _compile(src, force, extra=(), obj_dir=OBJ_DIR, drop=()):
    obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
    newest = max(os.path.getmtime(p) for p in [src, __file__, *headers()])
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [HIPCC, *[f for f in COMMON if f not in drop], *PER_FILE.get(os.path.basename(src), ()), *extra, "-x", "hip", "-c", src, "-o", obj, "-I", HERE,
           "-I", os.path.join(HERE, "..", "..", "include")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force: bool = False, verbose: bool = False, trace: bool = False, variant: str = None, defines=(), drop_flags=()) -> str:
    """trace=True builds the instrumented variant (-DGF_TRACE: per-round s_memtime timeline of the head kernel, read by
    tools/trace_head.py) into libgeneface_hip_trace.so; the product library never contains the instrumentation.
    variant="x", defines=("-DFOO",) builds an experiment library libgeneface_hip_x.so (A/B runs select it with GF_HIP_LIB);
    drop_flags=("-fno-slp-vectorize",) removes flags of COMMON for that variant (tools/pk_regress.py: the packed-FP32 probe build)."""
    if drop_flags and not variant:
        raise ValueError("drop_flags is for experiment variants only: the product library is always built with COMMON")
    tag = "trace" if trace else variant
    obj_dir = OBJ_DIR + (f"_{tag}" if tag else "")
    out = OUT.replace(".so", f"_{tag}.so") if tag else OUT
    extra = (("-DGF_TRACE",) if trace else ()) + tuple(defines)
    os.makedirs(obj_dir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, force, extra, obj_dir, tuple(drop_flags)), sources()))
    objs = [o for o, _ in results]
    if force or any(ch for _, ch in results) or not os.path.exists(out):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", out, "-lz", "-lpthread"]   # zlib: the PNG frame writer (png_writer.cpp)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("linked", out)
    return out


if __name__ == "__main__":
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    _defs = tuple(a for a in sys.argv[1:] if a.startswith(("-D", "-f")))   # e.g. --variant slp -fslp-vectorize for an A/B library
    print(build(force="--force" in sys.argv, verbose=True, trace="--trace" in sys.argv, variant=_variant, defines=_defs))
