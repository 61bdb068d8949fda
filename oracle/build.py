"""Build recipe for the parity oracle (TEST INFRASTRUCTURE ONLY).

Compiles oracle/radnerf_kernels.c with gcc into oracle/_build/liboracle_radnerf.so.
The output directory is git-ignored but travels to the GPU box with the snapshot.
The real reference kernels are built separately by oracle/refbuild/build_ref.py into oracle/_ref/
(hipcc reads the CUDA sources as they are); they need a GPU to run, this file's output does not.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "radnerf_kernels.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle_radnerf.so")

# -ffp-contract=off: every a*b+c rounds twice unless written as fmaf() (file header explains why)
CFLAGS = ["-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
          "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-Wextra"]


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    newest = max(os.path.getmtime(p) for p in (SRC, os.path.join(HERE, "sh_high_monomials.inc")))
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    cmd = ["gcc", *CFLAGS, SRC, "-o", OUT, "-lm"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
