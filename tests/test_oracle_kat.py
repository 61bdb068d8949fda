"""Analytic known-answer tests that pin the C oracle (the reference ships no tests or vectors)."""
import math

import numpy as np
import torch

from oracle import kernels as K
from oracle import radnerf_ref as R


def test_morton_kat_and_roundtrip(oracle_lib):
    c = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 5, 7], [127, 127, 127]], dtype=torch.int32)
    idx = torch.empty(5, dtype=torch.int32)
    K.raymarching_face.morton3D(c, 5, idx)
    assert idx[:3].tolist() == [1, 2, 4]
    assert idx[4].item() == 128 ** 3 - 1
    back = torch.empty(5, 3, dtype=torch.int32)
    K.raymarching_face.morton3D_invert(idx, 5, back)
    assert torch.equal(back, c)
    g = torch.Generator().manual_seed(0)
    c = torch.randint(0, 1024, (4096, 3), generator=g, dtype=torch.int32)
    idx = torch.empty(4096, dtype=torch.int32)
    back = torch.empty(4096, 3, dtype=torch.int32)
    K.raymarching_face.morton3D(c, 4096, idx)
    K.raymarching_face.morton3D_invert(idx, 4096, back)
    assert torch.equal(back, c)


def test_packbits_lsb_first(oracle_lib):
    grid = torch.zeros(1, 16)
    grid[0, 0] = 1.0
    grid[0, 9] = 1.0
    grid[0, 15] = 0.5  # == thresh: strict '>' leaves it clear
    bits = torch.empty(2, dtype=torch.uint8)
    K.raymarching_face.packbits(grid, 2, 0.5, bits)
    assert bits.tolist() == [1, 2]


def test_dilation_six_neighbours(oracle_lib):
    H = 8
    g = torch.zeros(1, H ** 3)
    c = torch.tensor([[3, 4, 5]], dtype=torch.int32)
    idx = torch.empty(1, dtype=torch.int32)
    K.raymarching_face.morton3D(c, 1, idx)
    g[0, idx[0]] = 2.0
    out = torch.empty_like(g)
    K.raymarching_face.morton3D_dilation(g, 1, H, out)
    assert int((out > 0).sum()) == 7
    nb = torch.tensor([[2, 4, 5], [4, 4, 5], [3, 3, 5], [3, 5, 5], [3, 4, 4], [3, 4, 6]], dtype=torch.int32)
    ni = torch.empty(6, dtype=torch.int32)
    K.raymarching_face.morton3D(nb, 6, ni)
    assert torch.all(out[0, ni.long()] == 2.0)


def test_slab_test_axis_rays(oracle_lib):
    aabb = torch.tensor([-1, -0.5, -1, 1, 0.5, 1], dtype=torch.float32)
    o = torch.tensor([[0, 3.0, 0], [0, 3.0, 0], [5.0, 3.0, 0], [0, 0.2, 0]], dtype=torch.float32)
    d = torch.tensor([[1e-9, -1, 1e-9], [1e-9, 1, 1e-9], [1e-9, -1, 1e-9], [1e-9, -1, 1e-9]], dtype=torch.float32)
    nears, fars = R.near_far_from_aabb(o, d, aabb, 0.05)
    assert abs(nears[0] - 2.5) < 1e-6 and abs(fars[0] - 3.5) < 1e-6
    fmax = float(np.finfo(np.float32).max)
    assert nears[2] == fmax and fars[2] == fmax           # misses in x
    assert nears[3] == np.float32(0.05) and abs(fars[3] - 0.7) < 1e-6  # origin inside: near clamps to min_near
    # ray 1 points away: the slab interval is behind the origin, near is clamped, far stays negative
    assert nears[1] == np.float32(0.05) and fars[1] < 0


def test_sh_axis_values(oracle_lib):
    d = torch.tensor([[0, 1.0, 0], [0, 0, 1.0], [1.0, 0, 0]])
    y = R.sh_encode(d)
    assert torch.allclose(y[:, 0], torch.full((3,), 0.28209479177387814))
    assert abs(y[0, 1] + 0.48860251190291987) < 1e-7   # Y_1^-1 at +y
    assert abs(y[1, 2] - 0.48860251190291987) < 1e-7   # Y_1^0 at +z
    assert abs(y[2, 3] + 0.48860251190291987) < 1e-7   # Y_1^1 at +x
    assert abs(y[1, 6] - (0.94617469575755997 - 0.31539156525251999)) < 1e-7
    # orthonormality on a fine quadrature: int Y_i Y_j = delta_ij
    n = 200
    th = (torch.arange(n) + 0.5) / n * math.pi
    ph = (torch.arange(2 * n) + 0.5) / (2 * n) * 2 * math.pi
    T, P = torch.meshgrid(th, ph, indexing="ij")
    dirs = torch.stack([torch.sin(T) * torch.cos(P), torch.sin(T) * torch.sin(P), torch.cos(T)], -1).reshape(-1, 3).float()
    w = (torch.sin(T) * (math.pi / n) * (2 * math.pi / (2 * n))).reshape(-1, 1).double()
    Y = R.sh_encode(dirs).double()
    gram = Y.T @ (Y * w)
    assert torch.allclose(gram, torch.eye(16, dtype=torch.float64), atol=2e-4)


def test_freq_layout(oracle_lib):
    x = torch.tensor([[0.0, math.pi / 2]])
    y = R.freq_encode(x, 2)  # [x0,x1, sin x0, sin x1, cos x0, cos x1, sin 2x0, sin 2x1, cos 2x0, cos 2x1]
    expect = torch.tensor([[0, math.pi / 2, 0, 1, 1, 0, 0, 0, 1, -1]], dtype=torch.float32)
    assert y.shape == (1, 10)
    assert torch.allclose(y, expect, atol=2e-7)


def _encode(x01, table, offsets, gridtype, interp=0, D=None):
    pls = np.exp2(np.log2(2048 / 16) / 15)
    return R.grid_encode(x01, table, offsets, pls, 16, gridtype, False, interp)


def test_grid_levels_match_python_formula(oracle_lib):
    pls = np.exp2(np.log2(2048 / 16) / 15)
    scale, res = K.grid_level_meta(16, float(np.log2(pls)), 16)
    expect = [int(np.ceil(16 * pls ** i)) for i in range(16)]  # grid.py:122
    assert res.tolist() == expect == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    # robust to a few ulps of exp2f: ceil(scale) must not sit on an integer boundary
    for s in scale.tolist():
        assert math.ceil(s * (1 - 4e-7)) == math.ceil(s * (1 + 4e-7)) or abs(s - round(s)) < 1e-3


def test_grid_lattice_node_returns_table_row(oracle_lib):
    from geneface_amd.encoders.gridencoder import grid_offsets
    for D in (2, 3):
        off = torch.from_numpy(grid_offsets(D, 16, 16, 16, 2048))
        g = torch.Generator().manual_seed(D)
        table = torch.rand(int(off[-1]), 2, generator=g)
        # level 0: scale = 15, pos = x*15 + 0.5 -> node (i + 0) when x = (i - 0.5)/15 ... choose x so that pos is integral
        node = [3, 7, 11][:D]
        x = torch.tensor([[(n - 0.5) / 15 for n in node]], dtype=torch.float32)
        y = _encode(x, table, off, 1)
        stride, row = 1, 0
        for n in node:
            row += n * stride
            stride *= 17
        assert torch.allclose(y[0, :2], table[row], atol=1e-5)
        # out-of-range input -> exact zeros on every level
        y = _encode(torch.tensor([[1.5] + [0.5] * (D - 1)]), table, off, 1)
        assert torch.count_nonzero(y) == 0


def test_tiled_drops_high_dims_and_hash_uses_primes(oracle_lib):
    from geneface_amd.encoders.gridencoder import grid_offsets
    off = torch.from_numpy(grid_offsets(3, 16, 16, 16, 2048))
    rows = int(off[-1])
    table = torch.arange(rows, dtype=torch.float32).unsqueeze(1).repeat(1, 2)
    table[:, 1] = 0
    # last level (res 2048, stride 2049): the running stride exceeds 2^16 after TWO dimensions -> tiled ignores z
    scale15 = 2047.0
    xs, ys = 100.0, 7.0
    x = torch.tensor([[(xs - 0.5) / scale15, (ys - 0.5) / scale15, (300 - 0.5) / scale15],
                      [(xs - 0.5) / scale15, (ys - 0.5) / scale15, (17 - 0.5) / scale15]], dtype=torch.float32)
    y = _encode(x, table, off, 1)
    assert abs(float(y[0, 30]) - float(y[1, 30])) < 1e-2  # same x,y -> same rows regardless of z
    assert abs(float(y[0, 30]) - (float(off[15]) + (xs + ys * 2049) % 65536)) < 40.0
    # hash: node (a,b,c) -> ((a*1) ^ (b*2654435761) ^ (c*805459861)) % 65536 at the last level
    yh = _encode(x[:1], table, off, 0)
    a, b, c = 100, 7, 300
    want = ((a * 1) ^ ((b * 2654435761) & 0xFFFFFFFF) ^ ((c * 805459861) & 0xFFFFFFFF)) % 65536
    # x was chosen on a lattice node (weights ~ (1,0,0..)), so the value is ~ the hashed row id
    assert abs(float(yh[0, 30]) - (float(off[15]) + want)) < 40.0


def test_march_respects_occupancy_and_budget(oracle_lib):
    from helpers import frame_inputs, model_fixture, sequence
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(2, 32, 32), 0)
    ro, rd = fi["rays_o"].view(-1, 3), fi["rays_d"].view(-1, 3)
    nears, fars = R.near_far_from_aabb(ro, rd, sd["aabb_infer"], hp["min_near"])
    N = ro.shape[0]
    alive = torch.arange(N, dtype=torch.int32)
    xyzs, dirs, deltas = R.march_rays(N, 4, alive, nears.clone(), ro, rd, 1.0, sd["density_bitfield"], 1, 128, nears, fars, 128,
                                      hp["dt_gamma"], hp["max_steps"])
    valid = deltas[:N * 4, 0] > 0
    assert 0 < int(valid.sum()) < N * 4
    # every emitted sample lies in an occupied cell of the analytic head
    from geneface_amd.synthetic import head_occupancy
    occ = head_occupancy(128, 1.0)
    p = xyzs[:N * 4][valid]
    cell = torch.clamp((0.5 * (p + 1) * 128).floor().long(), 0, 127)
    assert occ[cell[:, 0], cell[:, 1], cell[:, 2]].all()
    # dt is the constant 2*sqrt(3)/128 step for bound=1, max_steps=16
    assert torch.allclose(deltas[:N * 4][valid][:, 0], torch.full((int(valid.sum()),), 2 * 1.7320508075688772 / 128))
    # samples of one ray are contiguous from the front and t increases
    d4 = deltas[:N * 4].view(N, 4, 2)
    v4 = d4[..., 0] > 0
    assert torch.all(v4[:, 1:] <= v4[:, :-1])
    assert torch.all((d4[:, 1:, 1] > d4[:, :-1, 1]) | ~v4[:, 1:])


def test_sph_from_ray_known_answers():
    """raymarching.cu:161-198: far intersection with the sphere, y up.  From the origin along +y: theta = 0 -> -1; along +x: theta = pi/2 -> 0,
    phi = 0; along +z: phi = pi/2 -> 0.5; along -x: phi = pi -> 1.  A shifted origin moves the hit point, not the formula."""
    import math
    from oracle import kernels as K
    o = torch.zeros(5, 3)
    d = torch.tensor([[0, 1, 0], [1, 0, 0], [0, 0, 1], [-1, 1e-12, 0], [0.6, 0.0, 0.8]], dtype=torch.float32)
    c = torch.empty(5, 2)
    K.raymarching_face.sph_from_ray(o, d, 2.0, 5, c)
    assert abs(c[0, 0] + 1) < 1e-6
    assert abs(c[1, 0]) < 1e-6 and abs(c[1, 1]) < 1e-6
    assert abs(c[2, 0]) < 1e-6 and abs(c[2, 1] - 0.5) < 1e-6
    assert abs(c[3, 1] - 1) < 1e-6
    assert abs(c[4, 1] - math.atan2(0.8, 0.6) / math.pi) < 1e-6
    o2 = torch.tensor([[0.5, 0.0, 0.0]])
    c2 = torch.empty(1, 2)
    K.raymarching_face.sph_from_ray(o2, torch.tensor([[0.0, 0.0, 1.0]]), 1.0, 1, c2)      # hits (0.5, 0, sqrt(.75))
    assert abs(c2[0, 1] - math.atan2(math.sqrt(0.75), 0.5) / math.pi) < 1e-6 and abs(c2[0, 0]) < 1e-6


def test_grad_total_variation_known_answers():
    """gridencoder.cu:505-596 on a 1-level dense 2-D grid (resolution 4 -> 5 x 5 nodes, C = 1): a table linear in x has left/right
    differences that cancel in the interior and a one-sided difference at the border; the result is normalised by the RMS of the
    differences, so only signs and the weight / (2 D) factor remain."""
    from oracle import kernels as K
    L, D, C, H = 1, 2, 1, 4                         # scale = 2^0 * 4 - 1 = 3, resolution = 4
    offsets = torch.tensor([0, 32], dtype=torch.int32)             # 25 nodes rounded up to 32
    table = torch.zeros(32, 1)
    for y in range(5):
        for x in range(5):
            table[x + y * 5, 0] = float(x)
    grad = torch.zeros(32, 1)
    pts = torch.tensor([[0.5, 0.5], [0.0, 0.5], [1.0, 0.5]])       # nodes (2,2), (0,2), (3,2): floor(x * 3 + 0.5)
    K.gridencoder.grad_total_variation(pts, table, grad, offsets, 4.0, 3, D, C, L, 0.0, H, 1, False)
    g = grad.view(-1)
    assert abs(g[2 + 2 * 5]) < 1e-6                                 # interior: (2 - 3) + (2 - 1) = 0
    assert abs(g[0 + 2 * 5] - (-1.0)) < 1e-5                        # left border: only the right neighbour, (0 - 1) / |.| * 4 / 4
    assert abs(g[3 + 2 * 5]) < 1e-6                                 # node 3 of 0..4 still has both neighbours
    assert abs(float(g.abs().sum()) - 1.0) < 1e-5
