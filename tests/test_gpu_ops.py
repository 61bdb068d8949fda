"""-m gpu: every stand-alone HIP operator against the CPU oracle, called through the C ABI
(geneface_amd.compat modules -> ctypes -> libgeneface_hip.so)."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as GI
from helpers import frame_inputs, model_fixture, sequence
from oracle import kernels as K
from oracle import radnerf_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    return np.load(os.path.join(GOLD, "ops.npz"))


def test_extension_is_loaded_and_sees_the_gpu(hip_lib):
    assert hip_lib.gf_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libgeneface_hip.so" in maps


@pytest.mark.parametrize("D,enc,interp", GI.GRID_CASES)
def test_grid_encoder_vs_golden_and_oracle(ops, D, enc, interp):
    from geneface_amd.encoders import get_encoder
    tag, x, table, off = GI.grid_case(D, enc, interp)
    m, out_dim = get_encoder(enc, input_dim=D, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                             desired_resolution=2048, interpolation=interp)
    assert out_dim == 32 and np.array_equal(m.offsets.numpy(), off)
    m.embeddings.data.copy_(torch.from_numpy(table))
    m = m.to(DEV)
    y = m(torch.from_numpy(x).to(DEV), bound=1).detach().cpu().numpy()
    gold = ops[tag + "_y"]
    assert y.shape == gold.shape
    assert np.abs(y - gold).max() < 2e-6, np.abs(y - gold).max()   # fp32, fma vs mul+add ordering only
    assert np.array_equal(y == 0, gold == 0)                         # out-of-range rows are exact zeros


def test_grid_encoder_large_random_vs_oracle():
    """1 M random points through the seam layout [L,B,C] (3-D tiled, the head's position grid)."""
    from geneface_amd.compat import _gridencoder
    hp, sd = model_fixture(False)
    B = 1 << 20
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, generator=g)
    emb, off = sd["position_embedder.embeddings"], sd["position_embedder.offsets"]
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    out_ref = torch.empty(16, B, 2)
    K.gridencoder.grid_encode_forward(x, emb, off, out_ref, B, 3, 2, 16, S, 16, None, 1, False, 0)
    out = torch.empty(16, B, 2, device=DEV)
    _gridencoder.grid_encode_forward(x.to(DEV), emb.to(DEV), off.to(DEV), out, B, 3, 2, 16, S, 16, None, 1, False, 0)
    err = (out.cpu() - out_ref).abs().max().item()
    assert err < 1e-5, err


def test_grid_dy_dx_vs_oracle():
    from geneface_amd.compat import _gridencoder
    tag, x, table, off = GI.grid_case(3, "hashgrid", "smoothstep")
    B = x.shape[0]
    x01 = torch.from_numpy((x + 1) / 2)
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    o_ref, d_ref = torch.empty(16, B, 2), torch.empty(B, 16 * 3 * 2)
    K.gridencoder.grid_encode_forward(x01, torch.from_numpy(table), torch.from_numpy(off), o_ref, B, 3, 2, 16, S, 16, d_ref, 0, False, 1)
    o, d = torch.empty(16, B, 2, device=DEV), torch.empty(B, 96, device=DEV)
    _gridencoder.grid_encode_forward(x01.to(DEV), torch.from_numpy(table).to(DEV), torch.from_numpy(off).to(DEV), o, B, 3, 2, 16, S, 16, d,
                                     0, False, 1)
    assert (o.cpu() - o_ref).abs().max() < 2e-6
    scale = d_ref.abs().max().item()
    assert (d.cpu() - d_ref).abs().max() < 1e-5 * max(scale, 1.0)


def test_sh_and_freq_vs_golden(ops):
    from geneface_amd.encoders import get_encoder
    sh, n = get_encoder("spherical_harmonics")
    y = sh(torch.from_numpy(GI.sh_dirs()).to(DEV)).cpu().numpy()
    assert n == 16 and np.abs(y - ops["sh_y"]).max() < 1e-6
    for dim, deg in ((6, 4), (2, 10)):
        fe, c = get_encoder("frequency", input_dim=dim, multires=deg)
        y = fe(torch.from_numpy(GI.freq_case(dim, deg)).to(DEV)).cpu().numpy()
        assert c == dim + 2 * dim * deg
        assert np.abs(y - ops[f"freq_{dim}_{deg}_y"]).max() < 2e-5  # |arg| up to 2^9: fp32 range reduction


def test_unsupported_arguments_raise_like_the_reference():
    from geneface_amd.compat import _gridencoder, _shencoder
    x = torch.rand(8, 3, device=DEV)
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        _gridencoder.grid_encode_forward(x, torch.rand(64, 3, device=DEV), torch.tensor([0, 64], dtype=torch.int32, device=DEV),
                                         torch.empty(1, 8, 3, device=DEV), 8, 3, 3, 1, 0.5, 16, None, 1, False, 0)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        _shencoder.sh_encode_forward(torch.rand(8, 3), torch.empty(8, 16, device=DEV), 8, 3, 4, None)
    with pytest.raises(RuntimeError, match="contiguous"):
        _shencoder.sh_encode_forward(torch.rand(3, 8, device=DEV).t(), torch.empty(8, 16, device=DEV), 8, 3, 4, None)


def test_near_far_march_composite_bit_exact(ops):
    """Integer/discrete outputs of the marcher (which samples exist, their positions and t) must be identical."""
    from geneface_amd import raymarching as rm
    hp, sd = model_fixture(False)
    ro, rd = torch.from_numpy(ops["rays_o"]).to(DEV), torch.from_numpy(ops["rays_d"]).to(DEV)
    nears, fars = rm.near_far_from_aabb(ro, rd, sd["aabb_infer"].to(DEV), hp["min_near"])
    assert np.array_equal(nears.cpu().numpy(), ops["march_nears"]) and np.array_equal(fars.cpu().numpy(), ops["march_fars"])
    N = ro.shape[0]
    alive = torch.arange(N, dtype=torch.int32, device=DEV)
    rays_t = nears.clone()
    bits = sd["density_bitfield"].to(DEV)
    xyzs, dirs, deltas = rm.march_rays(N, 3, alive, rays_t, ro, rd, 1.0, bits, 1, 128, nears, fars, 128, False, hp["dt_gamma"], hp["max_steps"])
    assert np.array_equal(xyzs.cpu().numpy(), ops["march_xyzs"])
    assert np.array_equal(deltas.cpu().numpy(), ops["march_deltas"])
    sig, rgb = (torch.from_numpy(a).to(DEV) for a in GI.composite_inputs(xyzs.shape[0]))
    ws, dep, img = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV)
    rm.composite_rays(N, 3, alive, rays_t, sig, rgb, deltas, ws, dep, img, 1e-4)
    assert np.array_equal(alive.cpu().numpy(), ops["comp_alive"])            # who terminated: exact
    assert np.array_equal(rays_t.cpu().numpy(), ops["comp_rays_t"])          # copied t values: exact
    assert np.abs(ws.cpu().numpy() - ops["comp_ws"]).max() < 2e-6            # __expf vs expf
    assert np.abs(dep.cpu().numpy() - ops["comp_depth"]).max() < 1e-5
    assert np.abs(img.cpu().numpy() - ops["comp_image"]).max() < 2e-6


@pytest.mark.parametrize("size,n_step", [(64, 1), (128, 2), (512, 8)])
def test_march_full_image_bit_exact(size, n_step):
    from geneface_amd import raymarching as rm
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, size, size), 2)
    ro, rd = fi["rays_o"].view(-1, 3), fi["rays_d"].view(-1, 3)
    nears, fars = R.near_far_from_aabb(ro, rd, sd["aabb_infer"], hp["min_near"])
    N = ro.shape[0]
    alive = torch.arange(N, dtype=torch.int32)
    xr, dr, der = R.march_rays(N, n_step, alive, nears.clone(), ro, rd, 1.0, sd["density_bitfield"], 1, 128, nears, fars, 128,
                               hp["dt_gamma"], hp["max_steps"])
    n_g, f_g = rm.near_far_from_aabb(ro.to(DEV), rd.to(DEV), sd["aabb_infer"].to(DEV), hp["min_near"])
    assert torch.equal(n_g.cpu(), nears) and torch.equal(f_g.cpu(), fars)
    xg, dg, deg = rm.march_rays(N, n_step, alive.to(DEV), n_g.clone(), ro.to(DEV), rd.to(DEV), 1.0, sd["density_bitfield"].to(DEV), 1, 128,
                                n_g, f_g, 128, False, hp["dt_gamma"], hp["max_steps"])
    assert torch.equal(xg.cpu(), xr) and torch.equal(deg.cpu(), der) and torch.equal(dg.cpu(), dr)
    assert int((der[:, 0] > 0).sum()) > 0


def test_grid_maintenance_ops_bit_exact():
    from geneface_amd import raymarching as rm
    g = torch.Generator().manual_seed(9)
    coords = torch.randint(0, 128, (5000, 3), generator=g, dtype=torch.int32)
    idx_ref = torch.empty(5000, dtype=torch.int32)
    K.raymarching_face.morton3D(coords, 5000, idx_ref)
    idx = rm.morton3D(coords.to(DEV))
    assert torch.equal(idx.cpu(), idx_ref)
    assert torch.equal(rm.morton3D_invert(idx).cpu(), coords)
    grid = torch.rand(1, 64 ** 3, generator=g) * 20 - 1
    bits_ref = torch.empty(64 ** 3 // 8, dtype=torch.uint8)
    K.raymarching_face.packbits(grid, 64 ** 3 // 8, 10.0, bits_ref)
    assert torch.equal(rm.packbits(grid.to(DEV), 10.0).cpu(), bits_ref)
    dil_ref = torch.empty_like(grid)
    K.raymarching_face.morton3D_dilation(grid, 1, 64, dil_ref)
    assert torch.equal(rm.morton3D_dilation(grid.to(DEV)).cpu(), dil_ref)


def test_empty_inputs_are_noops():
    from geneface_amd import raymarching as rm
    from geneface_amd.encoders import get_encoder
    z = torch.zeros(0, 3, device=DEV)
    n, f = rm.near_far_from_aabb(z, z, torch.tensor([-1, -.5, -1, 1, .5, 1.], device=DEV), 0.05)
    assert n.numel() == 0 and f.numel() == 0
    sh, _ = get_encoder("spherical_harmonics")
    assert sh(z).shape == (0, 16)


@pytest.mark.parametrize("D,desired", [(3, 2048), (2, 2048)])
@pytest.mark.parametrize("gridtype,interp", [(1, 0), (1, 1), (0, 0)])
def test_fused_lookup_equals_generic_operator(D, desired, gridtype, interp):
    """grid_core.hpp::encode8 (what the fused head kernel evaluates: per-level strides / mask / hash flag instead of the
    generic stride walk + integer modulo) against the generic operator on 2 M random points PLUS points constructed to sit
    in the last row of a wrapped level, where the x-neighbour wraps to row 0 (index & mask == mask)."""
    from geneface_amd.encoders.gridencoder import grid_offsets, per_level_scale_for
    from geneface_amd.lib import check, current_stream, lib, ptr
    L, Hres, log2 = 16, 16, 16
    off = grid_offsets(D, L, Hres, log2, desired)
    pls = per_level_scale_for(desired, Hres, L)
    S = float(np.log2(pls))
    g = torch.Generator().manual_seed(D * 10 + gridtype)
    table = (torch.rand(int(off[-1]), 2, generator=g) * 2 - 1).to(DEV)
    offsets = torch.from_numpy(off).to(DEV)
    pts = [torch.rand(2_000_000, D, generator=g)]
    # cells whose tiled index lands on the last row of a wrapped level (mask = 65535)
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(l) * np.float32(S)) * Hres - 1)
        res1 = int(np.ceil(scale)) + 1 + 1
        size = int(off[l + 1] - off[l])
        if res1 ** D <= size:
            continue
        s1 = res1 if 1 <= size else 0
        ys = torch.arange(0, res1 - 1)
        xs = (65535 - (ys * s1) % 65536) % 65536
        ok = xs < res1 - 1
        cells = torch.stack([xs[ok], ys[ok]] + ([torch.zeros_like(xs[ok])] if D == 3 else []), dim=1).float()
        if D == 3 and res1 * res1 <= size:   # z participates in the index on this level: keep z = 0 (contributes nothing)
            pass
        pts.append(((cells + 0.25) - 0.5) / float(scale))   # pos = x * scale + 0.5 lands inside cell `cells`
    x = torch.cat(pts).clamp(0, 1).contiguous().to(DEV)
    B = x.shape[0]
    ref = torch.empty(B, 32, device=DEV)
    out = torch.empty(B, 32, device=DEV)
    check(lib().gf_grid_encode_forward_blc(ptr(x), ptr(table), ptr(offsets, torch.int32), ptr(ref), B, D, 2, L, S, Hres, None, gridtype, 0, interp,
                                           current_stream(x.device)))
    check(lib().gf_grid_encode_fused_lookup(ptr(x), ptr(table), ptr(offsets, torch.int32), ptr(out), B, D, S, Hres, gridtype, interp,
                                            current_stream(x.device)))
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() <= 1e-6
    assert B > 2_000_000   # the wrap cases were generated


def test_grid_encoder_seam_accepts_half_tables_like_the_reference_under_autocast():
    """grid.py:41-44 hands the backend half tables / outputs / dy_dx / gradients under autocast (the May config's amp: true).  The compat
    module converts at the seam and computes in fp32; checked against the fp32 call, and against the reference's own half kernels when
    oracle/_ref is present."""
    from geneface_amd.compat import _gridencoder
    hp, sd = model_fixture(False)
    B = 20000
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, 3, generator=g).to(DEV)
    emb, off = sd["position_embedder.embeddings"].to(DEV), sd["position_embedder.offsets"].to(DEV)
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    o32, d32 = torch.empty(16, B, 2, device=DEV), torch.empty(B, 96, device=DEV)
    _gridencoder.grid_encode_forward(x, emb, off, o32, B, 3, 2, 16, S, 16, d32, 1, False, 0)
    eh = emb.half()
    o16, d16 = torch.empty(16, B, 2, device=DEV, dtype=torch.float16), torch.empty(B, 96, device=DEV, dtype=torch.float16)
    _gridencoder.grid_encode_forward(x, eh, off, o16, B, 3, 2, 16, S, 16, d16, 1, False, 0)
    assert (o16.float() - o32).abs().max() < 2e-3 and torch.isfinite(d16).all()
    grad = torch.randn(16, B, 2, generator=g).to(DEV)
    ge32, gi32 = torch.zeros_like(emb), torch.zeros(B, 3, device=DEV)
    _gridencoder.grid_encode_backward(grad, x, emb, off, ge32, B, 3, 2, 16, S, 16, d32, gi32, 1, False, 0)
    ge16, gi16 = torch.zeros_like(eh), torch.zeros(B, 3, device=DEV, dtype=torch.float16)
    _gridencoder.grid_encode_backward(grad.half(), x, eh, off, ge16, B, 3, 2, 16, S, 16, d16, gi16, 1, False, 0)
    assert (ge16.float() - ge32).abs().max() < 2e-2 * max(1.0, float(ge32.abs().max()))
    assert (gi16.float() - gi32).abs().max() < 2e-2 * max(1.0, float(gi32.abs().max()))
    from oracle import ref_kernels
    if ref_kernels.available("fast"):
        GE = ref_kernels.load("fast")[1]
        r16 = torch.empty(16, B, 2, device=DEV, dtype=torch.float16)
        GE.grid_encode_forward(x, eh, off, r16, B, 3, 2, 16, S, 16, None, 1, False, 0)
        torch.cuda.synchronize()
        assert (o16.float() - r16.float()).abs().max() < 4e-3          # the reference rounds every corner product to half


@pytest.mark.parametrize("D", [2, 3, 4])
@pytest.mark.parametrize("C", [2, 4, 8])
@pytest.mark.parametrize("gridtype", [0, 1])
def test_half_table_forward_is_the_fp32_forward_on_the_widened_table(D, C, gridtype):
    """gf_grid_encode_forward_f16 (the branch the reference reaches through AT_DISPATCH_FLOATING_TYPES_AND_HALF, gridencoder.cu:375-398, when
    its wrapper casts the table to half under autocast, grid.py:41-44) widens each row on load and computes in fp32: outputs and dy_dx must
    equal gf_grid_encode_forward on `table.half().float()` BIT FOR BIT, for every channel count the half branch serves and both index rules."""
    from geneface_amd.encoders.gridencoder import grid_offsets, per_level_scale_for
    from geneface_amd.lib import check, current_stream, lib, ptr
    L, Hres, log2, desired = 8, 16, 14, 256
    off = grid_offsets(D, L, Hres, log2, desired)          # row offsets (independent of the channel count)
    S = float(np.log2(per_level_scale_for(desired, Hres, L)))
    g = torch.Generator().manual_seed(100 * D + 10 * C + gridtype)
    table = (torch.rand(int(off[-1]), C, generator=g) * 2 - 1).to(DEV)
    th = table.half()
    offsets = torch.from_numpy(np.asarray(off, dtype=np.int32)).to(DEV)
    B = 50_000
    x = torch.rand(B, D, generator=g).to(DEV)
    x[:5] = 1.5       # out of range: zeros
    o32, d32 = torch.empty(L, B, C, device=DEV), torch.empty(B, L * D * C, device=DEV)
    o16, d16 = torch.empty(L, B, C, device=DEV), torch.empty(B, L * D * C, device=DEV)
    check(lib().gf_grid_encode_forward(ptr(x), ptr(th.float().contiguous()), ptr(offsets, torch.int32), ptr(o32), B, D, C, L, S, Hres, ptr(d32), gridtype, 0, 0,
                                       current_stream(x.device)))
    check(lib().gf_grid_encode_forward_f16(ptr(x), ptr(th, torch.float16), ptr(offsets, torch.int32), ptr(o16), B, D, C, L, S, Hres, ptr(d16), gridtype, 0, 0,
                                           current_stream(x.device)))
    torch.cuda.synchronize()
    assert torch.equal(o16, o32) and torch.equal(d16, d32)
    assert float(o16[:, :5].abs().max()) == 0.0 and float(o16.abs().max()) > 0.1
    # and it is the half table that was read, not something wider: against the fp32 table the outputs differ by the table's rounding
    o_full = torch.empty(L, B, C, device=DEV)
    check(lib().gf_grid_encode_forward(ptr(x), ptr(table), ptr(offsets, torch.int32), ptr(o_full), B, D, C, L, S, Hres, None, gridtype, 0, 0, current_stream(x.device)))
    err = float((o16 - o_full).abs().max())
    assert 0 < err < 2e-3


@pytest.mark.parametrize("D", [2, 3])
def test_fused_lookup_far_out_of_range_points_read_nothing_and_give_zeros(D):
    """The specialised lookup of the fused kernels on points far outside [0,1] (and NaN / inf): zeros like the generic operator and the
    reference (gridencoder.cu:117-131), and -- the point of the test -- no table address is formed from such coordinates (on a dense level
    nothing wraps the index: x = 1e6 used to index ~1e18 rows past the table).  The module API accepts any position (RADNeRF.forward)."""
    from geneface_amd.encoders.gridencoder import grid_offsets, per_level_scale_for
    from geneface_amd.lib import check, current_stream, lib, ptr
    L, Hres, log2, desired = 16, 16, 16, 2048
    off = grid_offsets(D, L, Hres, log2, desired)
    S = float(np.log2(per_level_scale_for(desired, Hres, L)))
    g = torch.Generator().manual_seed(77 + D)
    table = (torch.rand(int(off[-1]), 2, generator=g) * 2 - 1).to(DEV)
    offsets = torch.from_numpy(off).to(DEV)
    x = torch.rand(4096, D, generator=g)
    bad = torch.tensor([1e6, -1e6, 3.4e38, -3.4e38, float("inf"), float("-inf"), float("nan"), 1.0000001, -1e-7, 65536.5])
    for k, v in enumerate(bad):
        x[k * 3:(k * 3) + 3] = torch.rand(3, D, generator=g)
        x[k * 3, 0] = v
        x[k * 3 + 1, D - 1] = v
        x[k * 3 + 2, :] = v
    x = x.to(DEV)
    for gridtype in (1, 0):
        ref, out = torch.empty(4096, 32, device=DEV), torch.full((4096, 32), 7.0, device=DEV)
        check(lib().gf_grid_encode_forward_blc(ptr(x), ptr(table), ptr(offsets, torch.int32), ptr(ref), 4096, D, 2, L, S, Hres, None, gridtype, 0, 0, current_stream(x.device)))
        check(lib().gf_grid_encode_fused_lookup(ptr(x), ptr(table), ptr(offsets, torch.int32), ptr(out), 4096, D, S, Hres, gridtype, 0, current_stream(x.device)))
        torch.cuda.synchronize()
        rows = torch.arange(30, device=DEV)
        assert float(out[rows].abs().max()) == 0.0                       # every planted row is out of range (or NaN): zeros
        # the generic operator agrees on EVERY row, NaN rows included (it used to test `x < 0 || x > 1`, let a NaN through and
        # form an index from (uint32_t)floorf(NaN)); with dy_dx requested those rows are zeros as well
        assert float(ref[rows].abs().max()) == 0.0
        assert torch.equal(out, ref) or (out - ref).abs().max().item() <= 1e-6
        ref2, dy = torch.empty(4096, 32, device=DEV), torch.full((4096, D * 32), 7.0, device=DEV)
        check(lib().gf_grid_encode_forward_blc(ptr(x), ptr(table), ptr(offsets, torch.int32), ptr(ref2), 4096, D, 2, L, S, Hres, ptr(dy), gridtype, 0, 0, current_stream(x.device)))
        torch.cuda.synchronize()
        assert torch.equal(ref2, ref) and float(dy[rows].abs().max()) == 0.0 and bool(torch.isfinite(dy).all())
        assert float(out[30:].abs().max()) > 0.1


def test_field_forward_accepts_any_position():
    """RADNeRF.forward on positions far outside the bound: the one-launch field must not fault, and returns what the oracle returns (zero
    grid features -> the field of the biases)."""
    from test_gpu_render import build
    hp, sd, model = build(False, "fused")
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(2000, 3, generator=g) * 2 - 1) * 0.4
    x[:8] = torch.tensor([[1e6, 0, 0], [0, -1e6, 0], [0, 0, 3e38], [50, 50, 50], [-50, 0.1, 0.2], [1.0001, 0, 0], [0, -1.0001, 0], [7, -7, 7]])
    d = torch.nn.functional.normalize(torch.randn(2000, 3, generator=g), dim=-1)
    cond = torch.randn(5, 1, 204, generator=g)
    cf = R.cal_cond_feat(sd, hp, cond)
    s_ref, c_ref, a_ref = R.head_field(sd, hp, x, d, cf, sd["individual_embeddings"][0])
    with torch.no_grad():
        s, c, a = model(x.to(DEV), d.to(DEV), cf.to(DEV), model.individual_embeddings[0])
    torch.cuda.synchronize()
    assert torch.isfinite(s).all() and torch.isfinite(c).all()
    assert ((s.cpu() - s_ref).abs() / s_ref.abs().clamp(min=1e-3)).max() < 2e-3 and (c.cpu() - c_ref).abs().max() < 1e-4 and (a.cpu() - a_ref).abs().max() < 1e-5
