#!/usr/bin/env python
"""Per-round timeline of the head kernel (k_head_phase) from the instrumented build.

    python -m geneface_amd.csrc.build --trace          # builds libgeneface_hip_trace.so (-DGF_TRACE)
    python tools/trace_head.py [--size 512] [--frame 10]

Thread 0 of the first 16 workgroups stamps s_memtime at every segment boundary of every round (frame_head.hip, GF_STAMP);
this script renders one frame with that build and prints, per segment, the mean / p50 / p90 shader cycles over the traced
rounds, so the share of a round spent in refill, march, scan, every weight-chunk barrier and every MFMA segment is
visible.  Measurement tooling only: the product library carries no instrumentation.
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GF_HIP_LIB", os.path.join(ROOT, "geneface_amd", "csrc", "libgeneface_hip_trace.so"))

NAMES = {
    (0, 1): "refill (queue atomic + ray loads)", (1, 2): "census barrier", (2, 3): "march", (3, 4): "barrier after march",
    (4, 5): "scan + dense map", (5, 6): "barrier (dense map)", (6, 7): "encode 3-D grid -> H",
    (7, 8): "BARRIER", (8, 9): "mfma amb L1 + sig L1a (K=32+32)",
    (9, 10): "BARRIER ", (10, 11): "store H", (11, 12): "BARRIER  ",
    (12, 13): "mfma amb L2 (K=128)", (13, 14): "amb L3 partial sums from the accumulators + barrier", (14, 15): "publish partials", (15, 16): "BARRIER    ",
    (16, 17): "amb L3 collect + tanh + encode 2-D grid -> H", (17, 18): "BARRIER     ",
    (18, 19): "mfma sig L1b (K=32)", (19, 20): "BARRIER      ", (20, 21): "store H  ", (21, 22): "BARRIER       ",
    (22, 23): "mfma sig L2 (K=128)", (23, 24): "BARRIER        ", (24, 25): "store H + density-row partials", (25, 26): "BARRIER         ",
    (26, 27): "sigma collect + exp + mfma sig L3 (K=128)", (27, 28): "BARRIER          ", (28, 29): "store H    ", (29, 30): "BARRIER           ",
    (30, 31): "SH + mfma col L1 (K=16+128)", (31, 32): "col L2 partial sums from the accumulators + barrier", (32, 33): "publish partials ", (33, 34): "BARRIER             ",
    (34, 35): "col L2 collect + sigmoid", (35, 36): "barrier end of field", (36, 37): "composite + retire",
}
SLOT_END, SLOT_MV, SLOT_NPOOL, SLOT_N = 37, 38, 39, 40


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frame", type=int, default=10)
    ap.add_argument("--json", default=None)
    ap.add_argument("--rounds", type=int, default=0, help="print every traced round of the first N workgroups (start, duration, gap to the previous round, samples, pool)")
    ap.add_argument("--spans", default=None, help="save the per-workgroup [start tick, end tick, rounds | samples << 32, XCC, start / end realtime] array (.npy)")
    ap.add_argument("--fast", action="store_true", help="trace the f16 fast-tier kernel (render_precision='fast')")
    ap.add_argument("--split", action="store_true", help="trace the split-tier kernel (render_precision='split'; density L1 is one K=64 layer there: slots 8-9 = "
                                                         "ambient L1 alone, 18-19 = density L1)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.fused import GfFrame, _fill_pose_frame, frame_stats
    from geneface_amd.infer import FramePipeline
    from geneface_amd.lib import check, current_stream, lib
    from geneface_amd.radnerf_torso import RADNeRFTorso

    dev = torch.device("cuda", 0)
    hp = HP.may_hparams(True)
    seq = S.make_sequence(args.frame + 1, args.size, args.size, hp)
    model = RADNeRFTorso(hp)
    model.load_state_dict(S.make_state_dict(hp, True), strict=True)
    model = model.to(dev).eval()
    if args.fast:
        model.render_precision = "fast"
    if args.split:
        model.render_precision = "split"
    pipe = FramePipeline(model, hp, seq, dev, impl="fused")
    L = lib()
    L.gf_trace_dims.restype = C.c_uint32
    nwg, nrounds, nslots = (L.gf_trace_dims(C.c_uint32(i)) for i in range(3))
    buf = torch.zeros(2 * nwg * nrounds * nslots, dtype=torch.int32, device=dev)
    L.gf_trace_set.argtypes = [C.c_void_p]
    for warm in range(3):
        pipe.render_frame(args.frame)
    torch.cuda.synchronize()
    L.gf_trace_set(C.c_void_p(buf.data_ptr()))
    buf.zero_()
    spans = torch.zeros(2 * 512 * 8, dtype=torch.int64, device=dev)
    L.gf_trace_set_spans.argtypes = [C.c_void_p]
    L.gf_trace_set_spans(C.c_void_p(spans.data_ptr()))
    with torch.no_grad():
        f = GfFrame()
        st, bufs, keep = _fill_pose_frame(pipe, args.frame, f, None)
        ms = (C.c_float * 4)()
        nph = C.c_uint32(0)
        check(L.gf_render_head_timed(C.byref(f), current_stream(dev), ms, C.byref(nph)))
    torch.cuda.synchronize()
    N = args.size * args.size
    fs = frame_stats(st.workspace(N)[1], N, hp["max_steps"])
    t = buf.cpu().numpy().astype(np.int64).reshape(2, nwg, nrounds, nslots) & 0xFFFFFFFF
    print(f"frame {args.frame} {args.size}x{args.size}: phase ms = {ms[0]:.3f} {ms[1]:.3f}; stats = {json.dumps(fs)}")
    report = {"phase_ms": [ms[0], ms[1]], "stats": fs, "phases": []}
    sp = spans.cpu().numpy().reshape(2, 512, 8)
    if args.spans:
        np.save(args.spans, sp)
    sp = sp.copy()
    wg_samples = sp[:, :, 2] >> 32
    sp[:, :, 2] &= 0xFFFFFFFF
    for ph in range(2):
        live = sp[ph][sp[ph][:, 1] != 0]
        if not len(live):
            continue
        life = live[:, 1] - live[:, 0]
        q = np.percentile(life, [10, 50, 90])
        print(f"phase {ph}: {len(live)} workgroups; lifetime min/p10/p50/p90/max = {life.min()}/{q[0]:.0f}/{q[1]:.0f}/{q[2]:.0f}/{life.max()} ticks "
              f"(mean {life.mean():.0f}); max lifetime over {ms[ph]:.3f} ms -> shader clock >= {life.max() / (ms[ph] * 1e-3) / 1e9:.3f} GHz; "
              f"rounds min/mean/max {live[:, 2].min()}/{live[:, 2].mean():.1f}/{live[:, 2].max()}; perfectly balanced the phase would take "
              f"{100 * life.mean() / life.max():.1f}% of its time")
        # the constant 100 MHz counter (s_memrealtime) next to the shader-clock counter: the true clock of every workgroup's lifetime, and
        # when the workgroups really start and stop inside the launch
        rt = (live[:, 5] - live[:, 4]).astype(np.float64)
        clk = life[rt > 0] / rt[rt > 0] * 0.1
        t0, t1 = live[:, 4].min(), live[:, 5].max()
        print(f"         true shader clock over a workgroup's life (ticks / 100 MHz realtime): min {clk.min():.3f} p50 {np.median(clk):.3f} max {clk.max():.3f} GHz; "
              f"first start -> last end {(t1 - t0) / 100:.1f} us of the {ms[ph] * 1e3:.1f} us between the HIP events; "
              f"starts spread over {(live[:, 4].max() - t0) / 100:.1f} us, ends over {(t1 - live[:, 5].min()) / 100:.1f} us; "
              f"workgroups leave the kernel {(live[:, 7].min() - t0) / 100:.1f} .. {(live[:, 7].max() - t0) / 100:.1f} us after the first start")
        report.setdefault("spans", []).append({"phase": ph, "clock_ghz_p50": float(np.median(clk)), "first_start_to_last_end_us": float((t1 - t0) / 100),
                                               "event_us": float(ms[ph] * 1e3), "start_spread_us": float((live[:, 4].max() - t0) / 100),
                                               "end_spread_us": float((t1 - live[:, 5].min()) / 100)})
    for w in range(args.rounds):
        tp0 = t[0][w]
        idx = np.nonzero(tp0[:, SLOT_END] != 0)[0]
        if not len(idx):
            continue
        base0 = int(tp0[idx[0], 0])
        print(f"workgroup {w}: round (start kticks, duration kticks, gap kticks, samples, pool, n, march kticks)")
        prev_end = base0
        for r in idx:
            st0, en = int(tp0[r, 0]), int(tp0[r, SLOT_END])
            print(f"   {r:2d} {((st0 - base0) & 0xFFFFFFFF) / 1e3:8.1f} {((en - st0) & 0xFFFFFFFF) / 1e3:7.1f} {((st0 - prev_end) & 0xFFFFFFFF) / 1e3:7.1f} "
                  f"{int(tp0[r, SLOT_MV]):4d} {int(tp0[r, SLOT_NPOOL]):4d} {int(tp0[r, SLOT_N]):2d} {((int(tp0[r, 3]) - int(tp0[r, 2])) & 0xFFFFFFFF) / 1e3:7.1f}")
            prev_end = en
            if ((en - st0) & 0xFFFFFFFF) > 400000:   # a slow round: every stage
                print("        stages (kticks): " + " ".join(f"{a}-{b}:{((int(tp0[r, b]) - int(tp0[r, a])) & 0xFFFFFFFF) / 1e3:.1f}" for (a, b) in NAMES if tp0[r, a] and tp0[r, b]))
    for ph in range(2):
        tp = t[ph]
        used = tp[:, :, SLOT_END] != 0
        if not used.any():
            continue
        full = used & (tp[:, :, SLOT_MV] >= 97)   # rounds with >= 3 full tiles of samples: the steady state
        sel = full if full.any() else used
        rounds = tp[sel]
        total = (rounds[:, SLOT_END] - rounds[:, 0]) & 0xFFFFFFFF
        print(f"\n== phase {ph}: {int(used.sum())} traced rounds over {nwg} workgroups, {int(sel.sum())} selected "
              f"(Mv>=97: {bool(full.any())}); round = {total.mean():.0f} cycles mean, p50 {np.median(total):.0f}, p90 {np.percentile(total, 90):.0f}; "
              f"Mv mean {rounds[:, SLOT_MV].mean():.1f}, n_pool mean {rounds[:, SLOT_NPOOL].mean():.1f}, n mean {rounds[:, SLOT_N].mean():.2f}")
        rows = []
        for (a, b), name in NAMES.items():
            d = (rounds[:, b] - rounds[:, a]) & 0xFFFFFFFF
            ok = (rounds[:, a] != 0) & (rounds[:, b] != 0) & (d < (1 << 30))
            if not ok.any():
                continue
            d = d[ok]
            rows.append((name, float(d.mean()), float(np.median(d)), float(np.percentile(d, 90)), float(d.mean() / total.mean())))
        for name, mean, p50, p90, share in rows:
            print(f"  {name:58s} mean {mean:9.0f}  p50 {p50:9.0f}  p90 {p90:9.0f}  {100 * share:5.1f}%")
        bar = sum(r[1] for r in rows if r[0].startswith("BARRIER"))
        mf = sum(r[1] for r in rows if "mfma" in r[0])
        print(f"  -> layer barriers {100 * bar / total.mean():.1f}%  mfma segments {100 * mf / total.mean():.1f}% of the round")
        # rounds per workgroup and the span of each workgroup's activity
        per_wg = used.sum(axis=1)
        print(f"  rounds per traced workgroup: {per_wg.tolist()}")
        # lifetime of each traced workgroup in shader ticks vs the launch's wall time -> effective shader clock
        spans = []
        for w in range(nwg):
            idx = np.nonzero(used[w])[0]
            if len(idx):
                spans.append(int((tp[w, idx[-1], SLOT_END] - tp[w, idx[0], 0]) & 0xFFFFFFFF))
        if spans:
            print(f"  workgroup lifetimes (ticks): min {min(spans)} max {max(spans)} -> effective clock >= {max(spans) / (ms[ph] * 1e-3) / 1e9:.3f} GHz "
                  f"(phase {ms[ph]:.3f} ms; traced rounds are capped at {nrounds})")
        report["phases"].append({"phase": ph, "round_cycles_mean": float(total.mean()), "segments": rows})
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
