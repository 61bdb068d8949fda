"""Seeded inputs of the op-level golden vectors (shared by tests/golden/make_golden.py and the tests,
so only OUTPUTS need to be committed)."""
import zlib

import numpy as np

from geneface_amd.encoders.gridencoder import grid_offsets

GRID_CASES = [(D, enc, interp) for D in (2, 3) for enc in ("tiledgrid", "hashgrid") for interp in ("linear", "smoothstep")]


def _rng(tag):
    return np.random.default_rng([zlib.crc32(tag.encode()), 1234])


def grid_case(D, enc, interp):
    """-> x [257,D] in [-1.1,1.1] (some out of range), table [rows,2] float32, offsets int32[17]"""
    tag = f"grid_D{D}_{enc}_{interp}"
    off = grid_offsets(D, 16, 16, 16, 2048)
    r = _rng(tag)
    table = r.uniform(-0.5, 0.5, (int(off[-1]), 2)).astype(np.float32)
    x = (r.uniform(0, 1, (257, D)) * 2.2 - 1.1).astype(np.float32)
    x[0] = 1.0
    x[1] = -1.0
    x[2] = 0.0
    return tag, x, table, off


def sh_dirs():
    d = _rng("sh").standard_normal((300, 3))
    return (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)


def freq_case(dim, deg):
    return (_rng(f"freq_{dim}_{deg}").uniform(-1, 1, (64, dim))).astype(np.float32)


def composite_inputs(M):
    r = _rng("composite")
    return (r.uniform(0, 60, M)).astype(np.float32), r.uniform(0, 1, (M, 3)).astype(np.float32)
