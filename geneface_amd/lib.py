"""ctypes binding of libgeneface_hip.so (C ABI: include/geneface_hip.h).

There is NO fallback: if the shared library cannot be loaded, or a call reports an error, a
RuntimeError is raised.  Nothing here (or anywhere under geneface_amd/) imports the CPU oracle.
"""
import ctypes as C
import os
import re
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GF_HIP_LIB selects another build of the same library (tools/trace_head.py points it at the instrumented one)
LIB_PATH = os.environ.get("GF_HIP_LIB") or os.path.join(_HERE, "csrc", "libgeneface_hip.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "geneface_hip.h")

_lock = threading.Lock()
_lib = None

_CTYPE = {
    "void*": C.c_void_p, "const void*": C.c_void_p,
    "const float*": C.c_void_p, "float*": C.c_void_p,
    "const int32_t*": C.c_void_p, "int32_t*": C.c_void_p,
    "const uint8_t*": C.c_void_p, "uint8_t*": C.c_void_p,
    "const uint32_t*": C.c_void_p, "uint32_t*": C.c_void_p,
    "uint32_t": C.c_uint32, "int": C.c_int, "float": C.c_float, "uint64_t": C.c_uint64,
}


def header_prototypes(path: str = HEADER_PATH):
    """[(return_type, name, [(ctype_string, arg_name), ...])] parsed from the public header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = []
    for m in re.finditer(r"\b(int|uint32_t|uint64_t|const char\*|void\*)\s+(gf_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                typ = mm.group(1).replace(" *", "*").strip()
                parsed.append((typ, mm.group(2)))
        protos.append((ret, name, parsed))
    return protos


def _bind(lib):
    for ret, name, args in header_prototypes():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = {"int": C.c_int, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64, "void*": C.c_void_p}.get(ret, C.c_char_p)
        argtypes = []
        for typ, _ in args:
            if typ in _CTYPE:
                argtypes.append(_CTYPE[typ])
            elif typ.endswith("*"):
                argtypes.append(C.c_void_p)  # pointer to a parameter struct
            else:
                raise RuntimeError(f"geneface_hip.h: unknown C type '{typ}' in {name}")
        fn.argtypes = argtypes
    return lib


def lib():
    """The loaded library (built on first use when the toolchain is present and the .so is absent)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    from .csrc import build as _b
                    _b.build()
                try:
                    _lib = _bind(C.CDLL(LIB_PATH))
                except OSError as e:  # pragma: no cover
                    raise RuntimeError(f"geneface_amd: cannot load the HIP extension {LIB_PATH}: {e}. "
                                       "Run `python -c 'import __graft_entry__ as g; g.build()'`.") from e
    return _lib


def check(rc: int):
    if rc != 0:
        raise RuntimeError(lib().gf_last_error().decode())


def ptr(t, dtype=None, allow_none=False):
    """Device pointer of a contiguous CUDA(HIP) tensor, with the checks the reference launchers make
    (CHECK_CUDA / CHECK_CONTIGUOUS / dtype, gridencoder.cu:448-464)."""
    if t is None:
        if allow_none:
            return None
        raise RuntimeError("expected a tensor, got None")
    if not t.is_cuda:
        raise RuntimeError("tensor must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be a contiguous tensor")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"tensor must have dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


def current_stream(device=None):
    return torch.cuda.current_stream(device).cuda_stream
