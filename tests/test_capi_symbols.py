"""The C-ABI library builds for gfx950, loads on a CPU-only host and exports every symbol the public header declares."""
import ctypes
import os
import subprocess

from geneface_amd import lib as L


def test_header_declares_the_expected_surface():
    names = [n for _, n, _ in L.header_prototypes()]
    for must in ("gf_near_far_from_aabb", "gf_march_rays", "gf_composite_rays", "gf_packbits", "gf_morton3D", "gf_morton3D_invert",
                 "gf_morton3D_dilation", "gf_grid_encode_forward", "gf_sh_encode_forward", "gf_freq_encode_forward",
                 "gf_last_error"):
        assert must in names
    assert len(names) == len(set(names))


def test_library_exports_every_declared_symbol(hip_lib):
    raw = ctypes.CDLL(L.LIB_PATH)
    for _, name, _ in L.header_prototypes():
        assert hasattr(raw, name), f"{name} declared in include/geneface_hip.h but not exported"
    assert hip_lib.gf_version().decode().startswith("geneface_hip")


def test_library_contains_gfx950_code_object(hip_lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", L.LIB_PATH], capture_output=True, text=True).stdout \
        if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") else ".hip_fatbin"
    assert ".hip_fatbin" in out
    blob = open(L.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_errors_are_reported_not_swallowed(hip_lib):
    # argument validation happens before any device work, so this is safe without a GPU
    rc = hip_lib.gf_freq_encode_forward(None, 4, 2, 3, 999, None, None)
    assert rc != 0 and b"C must equal" in hip_lib.gf_last_error()
    rc = hip_lib.gf_sh_encode_forward(None, None, 4, 2, 4, None, None)
    assert rc != 0 and b"input dim == 3" in hip_lib.gf_last_error()
    rc = hip_lib.gf_grid_encode_forward(None, None, None, None, 4, 7, 2, 16, 0.5, 16, None, 1, 0, 0, None)
    assert rc != 0


def test_host_level_meta_matches_oracle(hip_lib):
    import numpy as np
    from oracle import kernels as K
    pls = np.exp2(np.log2(2048 / 16) / 15)
    S = float(np.log2(pls))
    sc = (ctypes.c_float * 16)()
    rs = (ctypes.c_uint32 * 16)()
    assert hip_lib.gf_grid_level_meta(16, S, 16, ctypes.addressof(sc), ctypes.addressof(rs)) == 0
    scale, res = K.grid_level_meta(16, S, 16)
    assert list(rs) == res.tolist()
    assert list(sc) == scale.tolist()


def test_product_never_imports_the_oracle():
    """The parity oracle is test infrastructure: nothing under geneface_amd/ may reference it."""
    root = os.path.dirname(L.__file__)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f


def test_reference_kernel_build_exports_the_reference_api():
    """oracle/_ref (the reference's CUDA sources built for gfx950, GPU-side checker): when present, each module loads on CPU and
    exports exactly the functions the reference's pybind glue defines; the compat modules of the product carry the same names."""
    import pytest
    from oracle import ref_kernels
    if not ref_kernels.available("fast"):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    import geneface_amd.compat._freqencoder as cf
    import geneface_amd.compat._gridencoder as cg
    import geneface_amd.compat._raymarching_face as cr
    import geneface_amd.compat._shencoder as cs
    expect = {0: ["packbits", "near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "morton3D_dilation", "march_rays_train",
                  "march_rays_train_backward", "composite_rays_train_forward", "composite_rays_train_backward", "march_rays", "composite_rays"],
              1: ["grid_encode_forward", "grid_encode_backward", "grad_total_variation"],
              2: ["sh_encode_forward", "sh_encode_backward"], 3: ["freq_encode_forward", "freq_encode_backward"]}
    for contract in ("fast", "off"):
        if not ref_kernels.available(contract):
            continue
        mods = ref_kernels.load(contract)
        for i, compat in enumerate((cr, cg, cs, cf)):
            for name in expect[i]:
                assert callable(getattr(mods[i], name)), (contract, i, name)
                assert callable(getattr(compat, name)), (i, name)


def test_null_descriptors_are_refused_by_every_struct_entry_point(hip_lib):
    """The struct-taking entry points of the frame loop and of the training tier check their descriptor before they touch the device: a NULL
    (what a binding in another host language hands over when its marshalling went wrong) gives GF_ERR_INVALID (1) and a message that names the
    call, not a crash -- so this runs without a GPU.  Zero-sized work is success without a launch."""
    import ctypes as C
    L = hip_lib
    calls = {
        "gf_render_head": lambda: L.gf_render_head(None, None),
        "gf_render_torso": lambda: L.gf_render_torso(None, None),
        "gf_cond_encode": lambda: L.gf_cond_encode(None, None),
        "gf_cond_encode_batch": lambda: L.gf_cond_encode_batch(None, C.c_uint32(3), None),
        "gf_field_forward": lambda: L.gf_field_forward(None, None, None, C.c_uint32(5), None, None, None, None, None),
        "gf_field_forward_train": lambda: L.gf_field_forward_train(None, None, None, C.c_uint32(5), None, None, None, None, None, None),
        "gf_field_forward_train16": lambda: L.gf_field_forward_train16(None, None, None, C.c_uint32(5), None, None, None, None, None, None),
        "gf_field_backward": lambda: L.gf_field_backward(None, None, C.c_uint32(5), None, None),
        "gf_field_wgrad16": lambda: L.gf_field_wgrad16(C.c_uint32(5), None, None),
        "gf_field_wgrad32": lambda: L.gf_field_wgrad32(C.c_uint32(5), None, None),
        "gf_cond_train_forward": lambda: L.gf_cond_train_forward(None, None),
        "gf_cond_train_backward": lambda: L.gf_cond_train_backward(None, None),
        "gf_torso_train_forward": lambda: L.gf_torso_train_forward(None, None),
        "gf_torso_train_backward": lambda: L.gf_torso_train_backward(None, None),
        "gf_torso_wgrad": lambda: L.gf_torso_wgrad(None, None),
        "gf_torso_blend_train_forward": lambda: L.gf_torso_blend_train_forward(None, None),
        "gf_torso_blend_train_backward": lambda: L.gf_torso_blend_train_backward(None, None),
    }
    for name, call in calls.items():
        rc = call()
        msg = hip_lib.gf_last_error()
        assert rc == 1, (name, rc, msg)
        assert msg and (b"null" in msg.lower()), (name, msg)
    # nothing to do is not an error (and needs no pointers)
    assert L.gf_field_forward(None, None, None, C.c_uint32(0), None, None, None, None, None) == 0
    assert L.gf_field_backward(None, None, C.c_uint32(0), None, None) == 0
