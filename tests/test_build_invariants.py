"""CPU: properties of the BUILT device code that the design relies on, read from the code objects (tools/kernel_resources.py).

* no packed-FP32 VALU instruction anywhere in the library (NOTES.md 4.7: a v_pk_mul_f32 -> v_pk_fma_f32 pair lost a term when two
  workgroups shared a CU; the instruction class is banned, not just the one kernel it was caught in);
* no kernel uses scratch (a launch that touches scratch at all costs ~45 us more, NOTES.md 4.7) and none spills VGPRs;
* the field kernels run on the matrix pipe, the stand-alone ops do not pretend to.
"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import kernel_resources as KR  # noqa: E402

pytestmark = pytest.mark.skipif(not KR.available(), reason="ROCm LLVM binutils not present")


@pytest.fixture(scope="module")
def census(hip_lib):
    return KR.report()


def test_every_translation_unit_is_present(census):
    assert set(census) == {"cond_encode.hip", "cond_train.hip", "encoders.hip", "field_wgrad.hip", "frame_head.hip", "frame_torso.hip", "grid_update.hip", "raymarch.hip", "torso_blend_train.hip", "torso_wgrad.hip"}
    assert sum(len(d["kernels"]) for d in census.values()) >= 90


def test_no_packed_fp32_instructions(census):
    for tu, d in census.items():
        c = d["instructions"]
        assert c["v_pk_fma_f32"] == 0 and c["v_pk_mul_f32"] == 0 and c["v_pk_add_f32"] == 0, (tu, d["packed_fp32_by_function"])


def test_no_scratch_no_vgpr_spills(census):
    for tu, d in census.items():
        assert d["instructions"]["scratch_"] == 0, tu
        for k in d["kernels"]:
            assert k.get("private_segment_fixed_size", 0) == 0, (tu, k["name"])
            assert k.get("vgpr_spill_count", 0) == 0, (tu, k["name"])
            assert k.get("vgpr_count", 0) <= 512, (tu, k["name"])   # unified VGPR + AGPR file of a gfx950 SIMD lane


def test_matrix_pipe_is_used_where_the_design_says(census):
    assert census["frame_head.hip"]["instructions"]["v_mfma"] > 10_000      # head field, training forward / backward, grid density
    assert census["frame_torso.hip"]["instructions"]["v_mfma"] > 100
    for tu in ("raymarch.hip", "encoders.hip", "grid_update.hip", "cond_encode.hip", "cond_train.hip", "torso_blend_train.hip", "torso_wgrad.hip"):   # HBM / issue-bound byte and index work: no GEMM reshaping
        assert census[tu]["instructions"]["v_mfma"] == 0, tu
