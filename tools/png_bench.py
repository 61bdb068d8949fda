#!/usr/bin/env python
"""Encoder-side measurement for the PNG leg (base_nerf_infer.py:97-101): real rendered 512x512 head+torso frames through the native writer
at several zlib (level, strategy) settings and worker counts -- frames/s of the encoders alone and file size.  -> profiles/round3/png_bench.json"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from geneface_amd import hparams as HP, synthetic as S
from geneface_amd.infer import FramePipeline
from geneface_amd.png import FrameWriter, decode_rgb8
from geneface_amd.radnerf_torso import RADNeRFTorso

hp = HP.may_hparams(True)
m = RADNeRFTorso(hp); m.load_state_dict(S.make_state_dict(hp, True), strict=True); m = m.to("cuda:0").eval()
seq = S.make_sequence(32, 512, 512, hp)
pipe = FramePipeline(m, hp, seq, "cuda:0", impl="fused")
frames = [f.copy() for _, f in pipe.stream(range(32))]
res = []
for level, strategy, name in ((1, 0, "level 1, default strategy (round 2)"), (1, 3, "level 1, Z_RLE"), (1, 2, "Z_HUFFMAN_ONLY"), (1, 1, "level 1, Z_FILTERED"), (3, 0, "level 3")):
    for workers in (8, 16, 32, 64):
        d = tempfile.mkdtemp(prefix="gf_pngb_", dir="/dev/shm")
        w = FrameWriter(d, workers=workers, level=level, strategy=strategy)
        t0 = time.perf_counter()
        for k in range(256):
            w.submit(k, frames[k % 32])
        w.close()
        dt = time.perf_counter() - t0
        st = w.stage_seconds()
        assert np.array_equal(decode_rgb8(open(os.path.join(d, "00005.png"), "rb").read()), frames[5])
        shutil.rmtree(d)
        res.append({"setting": name, "level": level, "strategy": strategy, "workers": workers, "fps": 256 / dt, "MB_per_frame": st["bytes"] / 256 / 1e6,
                    "deflate_ms_per_frame": st["deflate_sum_over_workers"] / 256 * 1e3, "write_ms_per_frame": st["write_sum_over_workers"] / 256 * 1e3})
        print(res[-1])
out = sys.argv[1] if len(sys.argv) > 1 else None
if out:
    json.dump({"host_cores": os.cpu_count(), "frames": "32 rendered 512x512 head+torso frames of the bench fixture, 256 submissions", "results": res}, open(out, "w"), indent=1)
