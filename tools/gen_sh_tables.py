#!/usr/bin/env python
"""Generate the tables of the real spherical-harmonics basis for bands 4..7 (SH degrees 5..8 of the `_shencoder` seam).

The reference's kernel_sh (encoders/shencoder/src/shencoder.cu:69-121 values, :150-356 derivatives) spells the 48 polynomials of
these bands and their 144 partial derivatives out term by term.  Nothing is taken from there: the basis is derived here from its
definition, in exact rational arithmetic,

    Y_l^m(x, y, z) = N_l^m * Q_l^|m|(z) * { A_m(x, y)  m > 0 ;  1  m = 0 ;  B_|m|(x, y)  m < 0 }
    Q_l^m = d^m P_l / dz^m   (P_l: Legendre polynomial),   A_m + i B_m = (x + i y)^m,
    N_l^0 = sqrt((2l+1) / 4pi),   N_l^m = (-1)^m sqrt(2 (2l+1) (l-|m|)! / (4pi (l+|m|)!))

in the reference's ordering (index l*l + l + m) and sign convention (Condon-Shortley phase), as polynomials on R^3 with x^2 + y^2
eliminated in favour of z -- the representative the reference differentiates.  Because dQ_l^m/dz = Q_l^(m+1) and
dA_m/dx = m A_(m-1), dA_m/dy = -m B_(m-1), dB_m/dx = m B_(m-1), dB_m/dy = m A_(m-1), values and all three partial derivatives come
out of ONE family of z-polynomials and one (A, B) recurrence:

  * geneface_amd/csrc/sh_high_tables.inc  -- per (l, m): Horner coefficients of N Q_l^m and N Q_l^(m+1), fp32 literals
                                             (the product kernel: sh_core.hpp::sh_high)
  * oracle/sh_high_monomials.inc          -- per basis function: the expanded monomial list c * x^a y^b z^c in double
                                             (the oracle evaluates those in double: a different algorithm from the product's)

Checked below before anything is written: orthonormality of the generated basis on the sphere (Lebedev-free: a product Gauss grid),
the addition theorem sum_m Y_l^m(u)^2 = (2l+1)/4pi, and the known closed forms of bands 0..3 (sh_core.hpp::sh4).

    python tools/gen_sh_tables.py          # rewrites both .inc files (idempotent)
    python tools/gen_sh_tables.py --check  # verifies the committed files are what this script generates
"""
import math
import os
import sys
from fractions import Fraction as Fr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LMAX = 7


def legendre(l):
    """coefficients (ascending powers of z) of P_l as Fractions (Bonnet recursion)."""
    p0, p1 = [Fr(1)], [Fr(0), Fr(1)]
    if l == 0:
        return p0
    for n in range(1, l):
        a = [Fr(0)] + [Fr(2 * n + 1, n + 1) * c for c in p1]
        b = [Fr(n, n + 1) * c for c in p0] + [Fr(0), Fr(0)]
        p0, p1 = p1, [x - y for x, y in zip(a, b)]
    return p1


def deriv(p):
    return [k * c for k, c in enumerate(p)][1:] or [Fr(0)]


def q_poly(l, m):
    p = legendre(l)
    for _ in range(m):
        p = deriv(p)
    return p


def norm(l, m):
    m = abs(m)
    v = (2 * l + 1) * math.factorial(l - m) / (4 * math.pi * math.factorial(l + m))
    return math.sqrt(v) if m == 0 else (-1) ** m * math.sqrt(2 * v)


def ab_poly(m):
    """A_m, B_m as {(ex, ey): int}: real and imaginary part of (x + i y)^m."""
    A, B = {(0, 0): 1}, {}
    for _ in range(m):
        nA, nB = {}, {}
        for (a, b), c in A.items():
            nA[(a + 1, b)] = nA.get((a + 1, b), 0) + c      # x * A
            nB[(a, b + 1)] = nB.get((a, b + 1), 0) + c      # y * A
        for (a, b), c in B.items():
            nA[(a, b + 1)] = nA.get((a, b + 1), 0) - c      # -y * B
            nB[(a + 1, b)] = nB.get((a + 1, b), 0) + c      # x * B
        A, B = nA, nB
    return A, B


def monomials(l, m):
    """[(coef double, ex, ey, ez)] of Y_l^m expanded."""
    q = q_poly(l, abs(m))
    A, B = ab_poly(abs(m))
    xy = A if m >= 0 else B
    n = norm(l, m)
    out = []
    for (a, b), c in sorted(xy.items()):
        for k, qc in enumerate(q):
            if qc != 0 and c != 0:
                out.append((n * float(qc * c), a, b, k))
    return out


def eval_mono(mons, x, y, z):
    return sum(c * x ** a * y ** b * z ** k for c, a, b, k in mons)


def self_check():
    # (1) bands 0..3 against the closed forms every SH table lists (and sh_core.hpp::sh4 implements)
    x, y, z = 0.36, -0.48, 0.8
    want = {(0, 0): 0.28209479177387814, (1, -1): -0.48860251190291987 * y, (1, 0): 0.48860251190291987 * z, (1, 1): -0.48860251190291987 * x,
            (2, -2): 1.0925484305920792 * x * y, (2, 0): 0.94617469575755997 * z * z - 0.31539156525251999,
            (2, 2): 0.54627421529603959 * (x * x - y * y), (3, -3): 0.59004358992664352 * y * (-3 * x * x + y * y),
            (3, -1): 0.45704579946446572 * y * (1 - 5 * z * z), (3, 0): 0.3731763325901154 * z * (5 * z * z - 3),
            (3, 2): 1.4453057213202769 * z * (x * x - y * y), (3, 3): 0.59004358992664352 * x * (-x * x + 3 * y * y)}
    for (l, m), v in want.items():
        got = eval_mono(monomials(l, m), x, y, z)
        assert abs(got - v) < 1e-14, (l, m, got, v)
    # (2) addition theorem at a few unit vectors
    for u in ((0.36, -0.48, 0.8), (0.6, 0.0, -0.8), (2 / 7, 3 / 7, 6 / 7)):
        for l in range(LMAX + 1):
            s = sum(eval_mono(monomials(l, m), *u) ** 2 for m in range(-l, l + 1))
            assert abs(s - (2 * l + 1) / (4 * math.pi)) < 1e-12, (l, s)
    # (3) orthonormality on the sphere: Gauss-Legendre in z x uniform in phi (exact for these degrees)
    import numpy as np
    zs, wz = np.polynomial.legendre.leggauss(12)
    phis = np.arange(32) * (2 * math.pi / 32)
    pts = [(math.sqrt(1 - zz * zz) * math.cos(p), math.sqrt(1 - zz * zz) * math.sin(p), zz, w * 2 * math.pi / 32) for zz, w in zip(zs, wz) for p in phis]
    idx = [(l, m) for l in range(4, LMAX + 1) for m in range(-l, l + 1)]
    vals = np.array([[eval_mono(monomials(l, m), px, py, pz) for (px, py, pz, _) in pts] for (l, m) in idx])
    w = np.array([p[3] for p in pts])
    G = (vals * w) @ vals.T
    assert np.abs(G - np.eye(len(idx))).max() < 1e-10, np.abs(G - np.eye(len(idx))).max()


def f32_lit(v):
    import numpy as np
    return "0.0f" if v == 0 else f"{float(np.float32(v))!r}f".replace("e-0", "e-").replace("e+0", "e+")


def product_inc():
    lines = ["// GENERATED by tools/gen_sh_tables.py -- do not edit.  Real SH bands 4..7: per (l, m >= 0) the Horner coefficients (ascending powers of z,",
             "// zero padded to 8) of N_l^m Q_l^m(z) and of its z-derivative N_l^m Q_l^(m+1)(z); N carries the Condon-Shortley sign and the sqrt(2).",
             "// Row index: kShHighRow[l - 4] + m."]
    rows_q, rows_d, starts, n = [], [], [], 0
    for l in range(4, LMAX + 1):
        starts.append(n)
        for m in range(0, l + 1):
            nrm = norm(l, m)
            q = [nrm * float(c) for c in q_poly(l, m)]
            d = [nrm * float(c) for c in (q_poly(l, m + 1) if m + 1 <= l else [Fr(0)])]
            rows_q.append((l, m, (q + [0.0] * 8)[:8]))
            rows_d.append((l, m, (d + [0.0] * 8)[:8]))
            n += 1
    lines.append(f"static constexpr int kShHighRows = {n};")
    lines.append("static constexpr int kShHighRow[4] = {" + ", ".join(str(s) for s in starts) + "};")
    for name, rows in (("kShHighQ", rows_q), ("kShHighDQ", rows_d)):
        lines.append(f"static constexpr float {name}[kShHighRows][8] = {{")
        for l, m, r in rows:
            lines.append("    {" + ", ".join(f32_lit(v) for v in r) + f"}},   // l = {l}, m = {m}")
        lines.append("};")
    return "\n".join(lines) + "\n"


def oracle_inc():
    lines = ["/* GENERATED by tools/gen_sh_tables.py -- do not edit.  Real SH bands 4..7 (basis functions 16..63 in the reference's order, index",
             " * l*l + l + m): each as its expanded monomial list  coef * x^ex * y^ey * z^ez  (double coefficients).  Test infrastructure. */",
             "typedef struct { double c; unsigned char ex, ey, ez; } orc_sh_mono_t;"]
    starts, mons = [], []
    for l in range(4, LMAX + 1):
        for m in range(-l, l + 1):
            starts.append(len(mons))
            mons += monomials(l, m)
    starts.append(len(mons))
    lines.append(f"static const int orc_sh_high_start[{len(starts)}] = {{" + ", ".join(str(s) for s in starts) + "};")
    lines.append(f"static const orc_sh_mono_t orc_sh_high_mono[{len(mons)}] = {{")
    for c, a, b, k in mons:
        lines.append(f"    {{{c!r}, {a}, {b}, {k}}},")
    lines.append("};")
    return "\n".join(lines) + "\n"


def main():
    self_check()
    outs = {os.path.join(ROOT, "geneface_amd", "csrc", "sh_high_tables.inc"): product_inc(),
            os.path.join(ROOT, "oracle", "sh_high_monomials.inc"): oracle_inc()}
    if "--check" in sys.argv:
        for path, text in outs.items():
            assert open(path).read() == text, f"{path} is not what tools/gen_sh_tables.py generates"
        print("sh tables: committed files match the generator")
        return
    for path, text in outs.items():
        with open(path, "w") as f:
            f.write(text)
        print("wrote", os.path.relpath(path, ROOT), len(text), "bytes")


if __name__ == "__main__":
    main()
