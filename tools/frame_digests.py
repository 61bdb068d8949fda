#!/usr/bin/env python
"""Digests of rendered frames, for byte-identity A/Bs between two builds of the library:

    GF_HIP_LIB=.../libgeneface_hip_<variant>.so python tools/frame_digests.py > a.txt ;  python tools/frame_digests.py > b.txt ;  diff a.txt b.txt

25 frames of the bench sequence at 512x512 head+torso through the frame loop (uint8) and the module API (fp32 rgb + depth), fp32 and split
tiers, plus the thin-density fixture (long phase 1)."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.infer import FramePipeline
    from geneface_amd.radnerf_torso import RADNeRFTorso
    import random
    seq = S.make_sequence(25, 512, 512, HP.may_hparams(True))
    for name, kw in (("default", {}), ("thin", dict(sigma_row_scale=0.02)), ("head_aware", {})):
        hp = HP.variant_hparams("head_aware", True) if name == "head_aware" else HP.may_hparams(True)     # head_aware: k_torso_field<true>, both coin outcomes
        sd = S.make_state_dict(hp, True, **kw)
        for precision in ("fp32", "split"):
            m = RADNeRFTorso(hp)
            m.load_state_dict(sd, strict=True)
            m = m.to("cuda:0").eval()
            m.render_impl, m.render_precision = "fused", precision
            pipe = FramePipeline(m, hp, seq, "cuda:0", impl="fused")
            for i in range(0, 25, 2 if name == "default" else 6):
                with torch.no_grad():
                    random.seed(100 + i)
                    u8 = pipe.render_frame(i)
                    pipe.wait()
                    d8 = hashlib.md5(u8.numpy().tobytes()).hexdigest()
                    random.seed(100 + i)
                    out = pipe.run_model(pipe.sample(i))
                    df = hashlib.md5(out["rgb_map"].cpu().numpy().tobytes() + out["depth_map"].cpu().numpy().tobytes()).hexdigest()
                print(name, precision, i, d8, df)


if __name__ == "__main__":
    main()
