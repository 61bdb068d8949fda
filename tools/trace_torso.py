#!/usr/bin/env python
"""Where a tile of k_torso_field spends its time (instrumented build, like tools/trace_head.py for the head kernel).

    python -m geneface_amd.csrc.build --trace
    python tools/trace_torso.py [--size 512] [--frame 10]

Lane 0 of every wave of the first 64 field workgroups stamps s_memtime at the segment boundaries of its first tile (frame_torso.hip,
GF_TSTAMP); this script renders a frame with that build, alone on the GPU, and prints per segment the mean / p50 / p90 shader cycles over the
traced waves.  Measurement tooling only."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GF_HIP_LIB", os.path.join(ROOT, "geneface_amd", "csrc", "libgeneface_hip_trace.so"))

NAMES = ["read list length", "weights -> LDS (44 KB) + barrier", "list entry + pixel coordinates (2 dependent loads)", "24 frequency encodings",
         "deform L1 (48 MFMA)", "deform L2 (64 MFMA)", "deform L3 rows (VALU)", "2-D grid lookup (8 levels x 4 corners per lane)",
         "canonical L1 (40 MFMA)", "canonical L2 (16 MFMA)", "canonical L3 rows + sigmoid inputs (VALU)", "sigmoids + stores"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frame", type=int, default=10)
    args = ap.parse_args()
    import numpy as np
    import torch
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.infer import FramePipeline
    from geneface_amd.lib import lib
    from geneface_amd.radnerf_torso import RADNeRFTorso
    dev = torch.device("cuda", 0)
    hp = HP.may_hparams(True)
    seq = S.make_sequence(args.frame + 1, args.size, args.size, hp)
    m = RADNeRFTorso(hp)
    m.load_state_dict(S.make_state_dict(hp, True), strict=True)
    m = m.to(dev).eval()
    m.render_impl = "fused"
    pipe = FramePipeline(m, hp, seq, dev, impl="fused", in_flight=1)
    L = lib()
    L.gf_torso_trace_set.argtypes = [C.c_void_p]
    L.gf_torso_trace_set.restype = None
    buf = torch.zeros(64 * 4 * 16, dtype=torch.int64, device=dev)
    with torch.no_grad():
        pipe.render_frame(args.frame)          # warm: LDS limit, mask list, caches
        pipe.wait()
        L.gf_torso_trace_set(buf.data_ptr())
        pipe.render_frame(args.frame)
        pipe.wait()
        L.gf_torso_trace_set(None)
    t = buf.cpu().numpy().reshape(64, 4, 16)
    ok = (t[:, :, 12] > 0) & (t[:, :, 0] > 0)
    print(f"traced waves with a full tile: {int(ok.sum())} of 256")
    d = np.diff(t[:, :, :13].astype(np.int64), axis=2)[ok]          # [waves, 12]
    tot = (t[:, :, 12] - t[:, :, 0])[ok]
    print(f"first tile of a wave, start to last store: mean {tot.mean():.0f} cycles, p50 {np.median(tot):.0f}, p90 {np.percentile(tot, 90):.0f}")
    for k, name in enumerate(NAMES):
        c = d[:, k]
        print(f"  {name:58s} mean {c.mean():8.0f}  p50 {np.median(c):8.0f}  p90 {np.percentile(c, 90):8.0f}  {100 * c.mean() / tot.mean():5.1f}%")
    starts = t[:, :, 0][ok]
    print(f"workgroup starts spread over {(starts.max() - starts.min())} cycles; last traced store {(t[:, :, 12][ok].max() - starts.min())} cycles after the first start")


if __name__ == "__main__":
    main()
