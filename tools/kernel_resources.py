#!/usr/bin/env python
"""Per-kernel resources and instruction census of the built library, from the device code objects embedded in geneface_amd/csrc/_obj/*.o.

    python tools/kernel_resources.py [--json]

For every translation unit: unbundle the gfx950 code object (.hip_fatbin -> clang-offload-bundler), read the kernel descriptors' notes
(VGPRs, SGPR spills, scratch = private_segment_fixed_size, static LDS) and count the packed-FP32 VALU instructions (v_pk_fma_f32 /
v_pk_mul_f32 / v_pk_add_f32: NOTES.md 4.7 -- the library is built without them) and the MFMAs in the disassembly.  No GPU needed.
tests/test_build_invariants.py asserts on this.
"""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf", "llvm-objdump"))


def code_object(obj_path, tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, os.path.basename(obj_path) + ".co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj_path, os.path.join(tmp, "discard.o")], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={fat}", f"--output={co}"],
                   check=True, capture_output=True)
    return co


def kernels_of(co):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "args" or (k == "agpr_count" and cur.get("name")):   # a new kernel record starts with .agpr_count / .args
            if cur.get("name"):
                out.append(cur)
                cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count"):
            cur[k] = v if k == "name" else int(v)
    if cur.get("name"):
        out.append(cur)
    # de-duplicate (the walk above may split one record in two); keep complete ones
    seen = {}
    for k in out:
        seen.setdefault(k["name"], {}).update(k)
    return list(seen.values())


def census(co):
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    counts = {"v_pk_fma_f32": 0, "v_pk_mul_f32": 0, "v_pk_add_f32": 0, "v_mfma": 0, "scratch_": 0}
    per_fn, fn = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            fn = m.group(1)
            continue
        for key in counts:
            if key in line:
                counts[key] += 1
                if key.startswith("v_pk"):
                    per_fn.setdefault(fn, {}).setdefault(key, 0)
                    per_fn[fn][key] += 1
    return counts, per_fn


def report(obj_dir=None):
    obj_dir = obj_dir or os.path.join(ROOT, "geneface_amd", "csrc", "_obj")
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(obj_dir, "*.hip.o"))):
            co = code_object(obj, tmp)
            counts, per_fn = census(co)
            res[os.path.basename(obj)[:-2]] = {"kernels": kernels_of(co), "instructions": counts, "packed_fp32_by_function": per_fn}
    return res


if __name__ == "__main__":
    r = report()
    if "--json" in sys.argv:
        print(json.dumps(r, indent=1))
    else:
        for tu, d in r.items():
            print(f"== {tu}: {d['instructions']}")
            for k in d["kernels"]:
                print(f"   {k.get('name', '?')[:90]:90s} vgpr {k.get('vgpr_count', -1):3d} sgpr_spill {k.get('sgpr_spill_count', 0):3d} "
                      f"scratch {k.get('private_segment_fixed_size', -1)} lds {k.get('group_segment_fixed_size', -1)}")
