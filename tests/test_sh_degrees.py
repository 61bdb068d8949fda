"""SH degrees 1..8 of the `_shencoder` seam (sphere_harmonics.py:70; kernel shencoder.cu:28-356).

CPU part: the oracle against tests/golden/sh_deg8.npz -- the reference's own source expressions evaluated in fp32 by
tests/golden/make_golden_sh.py -- plus properties that do not depend on either (finite differences, the addition theorem), and the
generated tables being what tools/gen_sh_tables.py generates.  GPU part (-m gpu): the product through the C ABI against golden and oracle,
forward / dy_dx / backward, through the module API as well; the running reference kernel is in tests/test_gpu_vs_ref_kernels.py.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import kernels as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "sh_deg8.npz"))
REL = 2e-6            # of the output scale: three different association orders of one polynomial


def _oracle(x, degree):
    n = degree * degree
    out, dy = torch.empty(len(x), n), torch.empty(len(x), 3, n)
    K.shencoder.sh_encode_forward(x, out, len(x), 3, degree, dy)
    return out, dy


@pytest.mark.parametrize("degree", range(1, 9))
def test_oracle_vs_reference_source_expressions(degree):
    x = torch.from_numpy(GOLD["inputs"])
    n = degree * degree
    out, dy = _oracle(x, degree)
    gv, gd = GOLD["values"][:, :n], GOLD["dy_dx"][:, :, :n]
    assert np.abs(out.numpy() - gv).max() <= REL * max(1.0, np.abs(gv).max())
    assert np.abs(dy.numpy() - gd).max() <= REL * max(1.0, np.abs(gd).max())


def test_oracle_derivatives_are_derivatives():
    """Central differences in double of the oracle's own values (bands 4..7 are evaluated in double inside the oracle, so a step of 1e-3 in
    fp32 inputs leaves ~1e-4 relative truncation error: a structural check -- a wrong sign or a swapped row is off by O(1))."""
    r = np.random.default_rng(3)
    x = torch.from_numpy(r.uniform(-0.9, 0.9, (64, 3)).astype(np.float32))
    _, dy = _oracle(x, 8)
    h = 2.0 ** -9
    for d in range(3):
        e = torch.zeros(3)
        e[d] = h
        fd = (_oracle(x + e, 8)[0].double() - _oracle(x - e, 8)[0].double()) / (2 * h)
        err = (fd - dy[:, d].double()).abs().max().item()
        assert err < 2e-3 * max(1.0, dy[:, d].abs().max().item()), (d, err)


def test_addition_theorem_every_band():
    r = np.random.default_rng(4)
    u = r.standard_normal((256, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    out, _ = _oracle(torch.from_numpy(u.astype(np.float32)), 8)
    for l in range(8):
        s = (out[:, l * l:(l + 1) * (l + 1)].double() ** 2).sum(dim=1)
        assert (s - (2 * l + 1) / (4 * math.pi)).abs().max().item() < 2e-5, l


def test_generated_tables_are_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_sh_tables.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_oracle_rejects_degree_9():
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError):
        K.shencoder.sh_encode_forward(x, torch.empty(4, 81), 4, 3, 9, None)


# ------------------------------------------------------------------------------------------------ GPU
DEV = "cuda:0"


@pytest.mark.gpu
@pytest.mark.parametrize("degree", range(1, 9))
def test_product_vs_golden_and_oracle(degree):
    from geneface_amd.compat import _shencoder
    n = degree * degree
    x = torch.from_numpy(GOLD["inputs"])
    B = len(x)
    out, dy = torch.full((B, n), float("nan"), device=DEV), torch.full((B, 3 * n), float("nan"), device=DEV)
    _shencoder.sh_encode_forward(x.to(DEV), out, B, 3, degree, dy)
    gv, gd = GOLD["values"][:, :n], GOLD["dy_dx"][:, :, :n]
    assert np.abs(out.cpu().numpy() - gv).max() <= REL * max(1.0, np.abs(gv).max())
    assert np.abs(dy.cpu().numpy().reshape(B, 3, n) - gd).max() <= REL * max(1.0, np.abs(gd).max())
    # without dy_dx the values are the same bits
    out2 = torch.empty(B, n, device=DEV)
    _shencoder.sh_encode_forward(x.to(DEV), out2, B, 3, degree, None)
    assert torch.equal(out, out2)
    # a larger batch against the oracle, and the backward contraction
    g = torch.Generator().manual_seed(degree)
    xb = torch.cat([torch.nn.functional.normalize(torch.randn(40000, 3, generator=g), dim=-1), torch.rand(25537, 3, generator=g) * 2 - 1])
    Bb = len(xb)
    ob, db = torch.empty(Bb, n, device=DEV), torch.empty(Bb, 3 * n, device=DEV)
    _shencoder.sh_encode_forward(xb.to(DEV), ob, Bb, 3, degree, db)
    ro, rd = _oracle(xb, degree)
    assert (ob.cpu() - ro).abs().max().item() <= REL * max(1.0, ro.abs().max().item())
    assert (db.cpu().view(Bb, 3, n) - rd).abs().max().item() <= REL * max(1.0, rd.abs().max().item())
    grad = torch.randn(Bb, n, generator=g)
    gi = torch.zeros(Bb, 3, device=DEV)
    _shencoder.sh_encode_backward(grad.to(DEV), xb.to(DEV), Bb, 3, degree, db, gi)
    want = torch.einsum("bk,bdk->bd", grad.double(), rd.double())
    assert (gi.cpu().double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.gpu
def test_module_api_degree_8_with_autograd():
    """get_encoder('spherical_harmonics', degree=8) (encoding.py:22): forward shape and the gradient with respect to the directions."""
    from geneface_amd.encoders import get_encoder
    enc, n = get_encoder("spherical_harmonics", degree=8)
    assert n == 64
    x = torch.from_numpy(GOLD["inputs"][:64]).to(DEV).requires_grad_(True)
    y = enc(x)
    assert y.shape == (64, 64)
    w = torch.randn(64, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    (y * w).sum().backward()
    want = torch.einsum("bk,bdk->bd", w.cpu().double(), torch.from_numpy(GOLD["dy_dx"][:64]).double())
    assert (x.grad.cpu().double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.gpu
def test_degree_out_of_range_raises_like_the_reference():
    from geneface_amd.compat import _shencoder
    x = torch.rand(8, 3, device=DEV)
    for degree in (0, 9):
        with pytest.raises(RuntimeError, match="degree"):
            _shencoder.sh_encode_forward(x, torch.empty(8, max(degree * degree, 1), device=DEV), 8, 3, degree, None)
