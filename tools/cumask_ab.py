#!/usr/bin/env python
"""VERDICT r5 next #7, measured: throughput-mode frame latency with R compute units RESERVED for the small kernels of the frames in flight.

    GF_HIP_LIB=geneface_amd/csrc/libgeneface_hip_cumask.so GF_CUMASK_RESERVE=<R> python tools/cumask_ab.py [--precision fp32|split]

The A/B library (python -m geneface_amd.csrc.build --variant cumask -DGF_CUMASK) launches the two persistent head kernels of a frame on a
stream created with hipExtStreamCreateWithCUMask that excludes R CUs; everything else of the frame stays on the pipeline's own stream.
R = 0 (or the product library) is the baseline.  Printed: one JSON line -- frames/s of the pipelined loop (the headline's configuration),
device time per frame p50 / p99 (HIP events on the frame's stream: from the stream reaching the frame to its D2H copy finishing), and an md5
over the first frames (the picture must not change)."""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.infer import FramePipeline
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = HP.may_hparams(True)
    n, warm = args.frames, 10
    seq = S.make_sequence(warm + n, 512, 512, hp)
    m = RADNeRFTorso(hp)
    m.load_state_dict(S.make_state_dict(hp, True), strict=True)
    m = m.to("cuda:0").eval()
    m.render_impl, m.render_precision = "fused", args.precision
    pipe = FramePipeline(m, hp, seq, "cuda:0", impl="fused", frames=(0, warm + n))
    pct = lambda v, q: sorted(v)[min(len(v) - 1, int(q * len(v)))]
    with torch.no_grad():
        md5 = hashlib.md5()
        for i in range(warm):
            u8 = pipe.render_frame(i)
            pipe.wait()
            md5.update(u8.numpy().tobytes())
        for i in range(warm):                 # the pipelined rotation, warm
            pipe.render_frame(i)
        pipe.wait()
        pipe.frame_timing = []
        t0 = time.perf_counter()
        for rep in range(args.reps):
            pipe.prepare(warm, warm + n)
            for i in range(warm, warm + n):
                pipe.render_frame(i)
        pipe.wait()
        wall = time.perf_counter() - t0
        torch.cuda.synchronize()
        dev = [a.elapsed_time(b) for _, a, b in pipe.frame_timing]
    print(json.dumps({"library": os.path.basename(os.environ.get("GF_HIP_LIB", "libgeneface_hip.so")), "reserved_cus": int(os.environ.get("GF_CUMASK_RESERVE", "0")),
                      "high_bits": int(os.environ.get("GF_CUMASK_HIGH", "0")), "precision": args.precision, "frames_in_flight": pipe.in_flight,
                      "fps": len(dev) / wall, "device_ms_per_frame_p50": pct(dev, 0.5), "device_ms_per_frame_p99": pct(dev, 0.99),
                      "device_ms_per_frame_max": max(dev), "frames": len(dev), "md5_first_frames": md5.hexdigest()}))


if __name__ == "__main__":
    main()
