"""Build recipe for oracle/_ref (TEST INFRASTRUCTURE ONLY): the reference's own four extensions, compiled for gfx950.

The reference's per-op arithmetic lives in four CUDA translation units plus their pybind glue
(modules/radnerfs/raymarching/src/{raymarching.cu,bindings.cpp}, encoders/{grid,sh,freq}encoder/src/*).  They use nothing
beyond `<<<>>>` launches, ATen tensors and a handful of device intrinsics, all of which hipcc reads as they are, so this
recipe compiles them *where they lie under /root/reference* (nothing is copied or rewritten) with three include shims
(oracle/refbuild/shim: cuda.h / cuda_runtime.h / cuda_fp16.h -> the HIP runtime headers, ATen/cuda/CUDAContext.h ->
ATen/hip/HIPContext.h) and links them against this image's torch.  Outputs go to oracle/_ref/ only (git-ignored, not
gpurun-ignored, so the prebuilt modules travel to the GPU box; /root/reference itself does not exist there):

    oracle/_ref/ref_raymarching_face*.so   ref_gridencoder*.so   ref_shencoder*.so   ref_freqencoder*.so

Each is a Python extension module exporting exactly the reference's pybind functions; tests/test_gpu_vs_ref_kernels.py runs
them on the MI355X next to the C oracle and next to the product library.  The shims and this hipcc build are used for the
checker only -- the product (geneface_amd/csrc) contains no translated or shimmed code.

`python oracle/refbuild/build_ref.py [--force] [--contract off|fast]`: `fast` (hipcc's default, like nvcc's --fmad=true) lets
the compiler fuse a*b+c; `off` rounds every product.  Both are built (suffix _nofma for `off`) because which products a
compiler fuses is not part of the reference's source; the tests report parity against both.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_ref")
REFERENCE_ROOT = os.environ.get("GENEFACE_REFERENCE_ROOT", "/root/reference")
RAD = os.path.join(REFERENCE_ROOT, "modules", "radnerfs")

EXTENSIONS = {   # module name -> source directory under modules/radnerfs
    "ref_raymarching_face": ("raymarching/src", ["raymarching.cu", "bindings.cpp"]),
    "ref_gridencoder": ("encoders/gridencoder/src", ["gridencoder.cu", "bindings.cpp"]),
    "ref_shencoder": ("encoders/shencoder/src", ["shencoder.cu", "bindings.cpp"]),
    "ref_freqencoder": ("encoders/freqencoder/src", ["freqencoder.cu", "bindings.cpp"]),
}


def available() -> bool:
    return os.path.isdir(RAD)


def module_path(name: str, contract: str = "fast") -> str:
    return os.path.join(OUT_DIR, name + ("" if contract == "fast" else "_nofma") + ".so")


def _build_one(name, contract, force, verbose, torch_dir, cxx11_abi):
    sub, files = EXTENSIONS[name]
    out = module_path(name, contract)
    srcs = [os.path.join(RAD, sub, f) for f in files]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs + [__file__]):
        return out
    inc = ["-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(torch_dir, "include"),
           "-I" + os.path.join(torch_dir, "include", "torch", "csrc", "api", "include"), "-I" + sysconfig.get_paths()["include"]]
    libs = ["-L" + os.path.join(torch_dir, "lib"), "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
            "-Wl,-rpath," + os.path.join(torch_dir, "lib")]
    modname = os.path.basename(out)[:-3]
    objs = []
    for s in srcs:
        obj = os.path.join(OUT_DIR, f"{modname}.{os.path.basename(s)}.o")
        cmd = ["hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", f"-ffp-contract={contract}",
               "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-DTORCH_EXTENSION_NAME={modname}", "-DTORCH_API_INCLUDE_EXTENSION_H",
               f"-D_GLIBCXX_USE_CXX11_ABI={cxx11_abi}", *inc, "-c", s, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    subprocess.run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, *libs, "-o", out], check=True)
    for o in objs:
        os.remove(o)
    return out


def build(force: bool = False, contracts=("fast", "off"), verbose: bool = False) -> list:
    """Builds every extension that is out of date (the eight translation-unit pairs in parallel); returns the module paths.
    Raises if /root/reference is absent."""
    if not available():
        raise FileNotFoundError(f"{RAD} not found: oracle/_ref can only be built where the reference checkout exists")
    from concurrent.futures import ThreadPoolExecutor
    import torch
    tdir, abi = os.path.dirname(torch.__file__), int(torch._C._GLIBCXX_USE_CXX11_ABI)
    os.makedirs(OUT_DIR, exist_ok=True)
    jobs = [(n, c) for c in contracts for n in EXTENSIONS]
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        return list(ex.map(lambda j: _build_one(j[0], j[1], force, verbose, tdir, abi), jobs))


if __name__ == "__main__":
    contract = sys.argv[sys.argv.index("--contract") + 1] if "--contract" in sys.argv else None
    for p in build(force="--force" in sys.argv, contracts=(contract,) if contract else ("fast", "off"), verbose="-v" in sys.argv):
        print(p)
