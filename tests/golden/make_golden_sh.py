"""Golden vectors for SH degrees 1..8 of the `_shencoder` seam (run in the BUILD container only: needs /root/reference).

    python tests/golden/make_golden_sh.py

What produces the numbers: the reference's OWN source expressions.  kernel_sh (modules/radnerfs/encoders/shencoder/src/shencoder.cu:28-356)
is a list of assignments `outputs[k] = <polynomial in x, y, z>;` and `dx[k] / dy[k] / dz[k] = ...;`.  This script reads that file where it
lies, turns each right-hand side into a numpy fp32 expression (same operand order, one rounding per operation, no contraction) and
evaluates it on seeded directions.  Only the inputs and the resulting numbers are stored (tests/golden/sh_deg8.npz); no source text.
The CUDA file cannot run here (no GPU); on the MI355X the same kernel runs as oracle/_ref/ref_shencoder.so beside oracle and product
(tests/test_gpu_vs_ref_kernels.py).
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
SRC = "/root/reference/modules/radnerfs/encoders/shencoder/src/shencoder.cu"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sh_deg8.npz")


def sh_inputs(n_unit=192, n_free=64, seed=20):
    """Unit directions (what the renderer feeds) and free points of R^3 (the polynomials and their derivatives are defined there too)."""
    r = np.random.default_rng(seed)
    u = r.standard_normal((n_unit, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    axes = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1], [0.6, 0.0, 0.8], [0.0, -0.6, 0.8]], dtype=np.float64)
    u[:len(axes)] = axes
    f = r.uniform(-1.2, 1.2, (n_free, 3))
    return np.concatenate([u, f]).astype(np.float32)


def reference_expressions():
    pat = re.compile(r"^\s*(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*(.*?)\s*;")
    table = {"outputs": {}, "dx": {}, "dy": {}, "dz": {}}
    for line in open(SRC):
        m = pat.match(line)
        if m:
            table[m.group(1)][int(m.group(2))] = m.group(3)
    for k, v in table.items():
        assert sorted(v) == list(range(64)), (k, len(v))
    return table


def evaluate(expr, env):
    py = re.sub(r"(\d+\.\d+(?:[eE][-+]?\d+)?)f", r"np.float32(\1)", expr)
    val = eval(py, {"np": np}, env)      # noqa: S307  (the reference's own arithmetic expressions)
    return np.broadcast_to(np.asarray(val, dtype=np.float32), env["x"].shape).copy()


def main():
    X = sh_inputs()
    x, y, z = (np.ascontiguousarray(X[:, i]) for i in range(3))
    env = {"x": x, "y": y, "z": z, "xy": x * y, "xz": x * z, "yz": y * z, "x2": x * x, "y2": y * y, "z2": z * z}
    env["xyz"] = env["xy"] * z
    env["x4"], env["y4"], env["z4"] = env["x2"] * env["x2"], env["y2"] * env["y2"], env["z2"] * env["z2"]
    env["x6"], env["y6"], env["z6"] = env["x4"] * env["x2"], env["y4"] * env["y2"], env["z4"] * env["z2"]
    t = reference_expressions()
    vals = np.stack([evaluate(t["outputs"][k], env) for k in range(64)], axis=1)
    grads = np.stack([np.stack([evaluate(t[d][k], env) for k in range(64)], axis=1) for d in ("dx", "dy", "dz")], axis=1)
    assert vals.dtype == np.float32 and grads.dtype == np.float32 and grads.shape == (len(X), 3, 64)
    np.savez_compressed(OUT, inputs=X, values=vals, dy_dx=grads)
    print("sh_deg8.npz", X.shape, vals.shape, grads.shape, "max|Y|", float(np.abs(vals).max()), "max|dY|", float(np.abs(grads).max()))


if __name__ == "__main__":
    assert os.path.exists(SRC), "needs the reference tree"
    sys.exit(main())
