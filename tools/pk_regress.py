#!/usr/bin/env python
"""Does a build WITH packed-FP32 VALU instructions still miscompute?  (NOTES.md 4.7)

Round 2 caught `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[0,1]` -> `v_pk_fma_f32` losing a product in lanes 32..63 of the fast-tier head
kernel whenever two workgroups shared a CU: 49 of 80 renders of ONE 512x512 frame differed from the first.  The instruction class has been
banned library-wide since (`-fno-slp-vectorize`, tests/test_build_invariants.py) with the mechanism unexplained.  This tool builds the
library with the vectoriser back ON (libgeneface_hip_slp.so, an experiment variant: never shipped), takes a census of the packed-FP32
instructions the compiler now emits -- including whether the broadcast-form pair of round 2 still appears -- and renders each tier's frames
over and over, alone and with four frames in flight, counting renders whose bytes differ from the first render of the same frame.

    python tools/pk_regress.py --build-only            # CPU: build the variant, print the census
    GF_HIP_LIB=.../libgeneface_hip_slp.so python tools/pk_regress.py --frames 4000     # GPU: the render loop on that library
    python tools/pk_regress.py --frames 4000            # GPU: the same loop on the product library (control)
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]


def census(obj_dir):
    import kernel_resources as KR
    out = {"compiler": subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[:2]}
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(os.listdir(obj_dir)):
            if not o.endswith(".hip.o"):
                continue
            co = KR.code_object(os.path.join(obj_dir, o), tmp)
            counts, per_fn = KR.census(co)
            dis = subprocess.run([f"{KR.LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout.splitlines()
            fn, bcast, bcast_then_fma = None, {}, {}
            for i, line in enumerate(dis):
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    fn = m.group(1)
                    continue
                if "v_pk_mul_f32" in line and "op_sel:[0,1]" in line and "op_sel_hi:[0,1]" in line:       # the scalar-broadcast form of round 2
                    bcast[fn] = bcast.get(fn, 0) + 1
                    dst = line.split("v_pk_mul_f32")[1].split(",")[0].strip()
                    if any("v_pk_fma_f32" in nxt and dst in nxt for nxt in dis[i + 1:i + 6]):
                        bcast_then_fma[fn] = bcast_then_fma.get(fn, 0) + 1
            scr = {k["name"]: k.get("private_segment_fixed_size", 0) for k in KR.kernels_of(co) if k.get("private_segment_fixed_size")}
            out[o] = {"packed_fp32": {k: v for k, v in counts.items() if k.startswith("v_pk")}, "broadcast_form_v_pk_mul": sum(bcast.values()),
                      "broadcast_form_feeding_v_pk_fma_within_5": sum(bcast_then_fma.values()), "kernels_with_scratch_bytes": scr}
    return out


def render_loop(frames, size):
    import torch
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.infer import FramePipeline
    from geneface_amd.lib import LIB_PATH
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = HP.may_hparams(True)
    seq = S.make_sequence(8, size, size, hp)
    sd = S.make_state_dict(hp, True)
    res = {"library": os.path.basename(LIB_PATH), "size": size}
    for precision in ("fast", "split", "fp32"):
        m = RADNeRFTorso(hp)
        m.load_state_dict(sd, strict=True)
        m = m.to("cuda:0").eval()
        m.render_impl, m.render_precision = "fused", precision
        n = frames if precision != "fp32" else max(frames // 3, 8)
        rec = {}
        for in_flight in (1, 4):
            pipe = FramePipeline(m, hp, seq, "cuda:0", impl="fused", in_flight=in_flight)
            first, deviating, bytes_off, lanes = {}, 0, 0, {}
            with torch.no_grad():
                for k, (i, frame) in enumerate(_stream(pipe, n)):
                    t = torch.from_numpy(frame)
                    if i not in first:
                        first[i] = t.clone()
                    elif not torch.equal(first[i], t):
                        deviating += 1
                        d = (first[i] != t)
                        bytes_off += int(d.sum())
                        if len(lanes) < 6:
                            px = d.any(dim=-1).nonzero()
                            lanes[f"render {k} frame {i}"] = [[int(a), int(b)] for a, b in px[:6]]
            rec[f"in_flight_{in_flight}"] = {"renders": n, "deviating_renders": deviating, "bytes_differing": bytes_off, "examples": lanes}
        res[precision] = rec
    return res


def _stream(pipe, n):
    """(frame index, host uint8 frame) for n renders cycling over the pipeline's 8 frames, the pipeline kept full."""
    idx = [j % 8 for j in range(n)]
    for a in range(0, n, 8):
        for i, frame in pipe.stream(idx[a:a + 8]):
            yield i, frame


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--frames", type=int, default=4000)
    ap.add_argument("--size", type=int, default=512)
    args = ap.parse_args()
    if args.build_only:
        from geneface_amd.csrc import build as B
        B.build(variant="slp", drop_flags=("-fno-slp-vectorize",))
        print(json.dumps(census(os.path.join(ROOT, "geneface_amd", "csrc", "_obj_slp")), indent=1))
        return
    print(json.dumps(render_loop(args.frames, args.size)))


if __name__ == "__main__":
    main()
