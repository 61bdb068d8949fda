"""Diagnostics for run-to-run deviations of the fused head kernel (needs a GPU; uses libgeneface_hip_diag.so, built with
`python -m geneface_amd.csrc.build --variant diag -DGF_DIAG`).

    python tools/fast_diag.py [--size 160] [--frames 3000] [--in-flight 3] [--precision fast] [--plain] [--steps poison,group,stress]

Steps
  poison : every LDS word the workgroup has not written itself reads as NaN (at workgroup start and, for the activation buffer, at
           every round).  A frame that changes under poisoning proves an uninitialised LDS read.
  group  : force other ray groupings (grid size, pool capacity) on a frame rendered alone.  A frame that changes proves that a
           sample's value depends on which other samples share its round.
  stress : frames in flight on several streams against the same frames rendered alone; for a deviating frame the per-sample records
           (marcher, grid features, ambient, density, colour, compositor inputs; keyed by ray and cumulative sample index) of both
           renders are compared quantity by quantity.
--plain runs the stress step with the product library (no records) to see whether this box shows deviations at all.
"""
import argparse
import ctypes as C
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=160)
ap.add_argument("--frames", type=int, default=3000)
ap.add_argument("--in-flight", type=int, default=3)
ap.add_argument("--precision", default="fast")
ap.add_argument("--plain", action="store_true")
ap.add_argument("--lib", default=None)
ap.add_argument("--steps", default="poison,group,stress")
ap.add_argument("--max-dumps", type=int, default=4)
ap.add_argument("--place-table", default=None, help="hex low dword: move the ambient table to a device address whose low 32 bits are >= this value")
args = ap.parse_args()
if args.lib:
    os.environ["GF_HIP_LIB"] = args.lib
elif not args.plain:
    os.environ["GF_HIP_LIB"] = os.path.join(REPO, "geneface_amd", "csrc", "libgeneface_hip_diag.so")
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

from helpers import model_fixture, sequence  # noqa: E402
from geneface_amd import fused  # noqa: E402
from geneface_amd.infer import FramePipeline  # noqa: E402
from geneface_amd.lib import lib  # noqa: E402
from geneface_amd.radnerf_torso import RADNeRFTorso  # noqa: E402

DEV = "cuda:0"
STRIDE = 32
FIELDS = {0: "sigma", 1: "r", 2: "g", 3: "b", 4: "dt", 5: "t", 12: "amb0", 13: "amb1", 14: "h0", 15: "x", 16: "c3d", 17: "c2d",
          20: "x2_0", 21: "x2_1", 22: "c2d32", 24: "e0", 25: "e1"}


def hw(q):
    v = int(q[23])
    return f"se {(v >> 13) & 7} cu {(v >> 8) & 15} simd {(v >> 4) & 3} wave {v & 15}"


def build(precision):
    hp, sd = model_fixture(True)
    m = RADNeRFTorso(hp)
    m.load_state_dict(sd, strict=True)
    m.render_impl = "fused"
    m.render_precision = precision
    return hp, m.to(DEV).eval()


class Diag:
    def __init__(self, model, N, slots):
        self.L = lib()
        self.on = hasattr(self.L, "gf_diag_register") if not args.plain else False
        try:
            self.L.gf_diag_words.restype = C.c_uint32
            self.L.gf_diag_last_tag.restype = C.c_uint32
            self.W = int(self.L.gf_diag_words())
            self.on = True
        except AttributeError:
            self.on = False
            return
        st = fused.get_state(model)
        self.bufs = []
        for s in range(slots):
            ws = st.workspace(N, s)[0]
            buf = torch.zeros(N * STRIDE * self.W, dtype=torch.float32, device=DEV)
            self.L.gf_diag_register(C.c_uint32(s), C.c_void_p(ws.data_ptr()), C.c_void_p(buf.data_ptr()), C.c_uint32(STRIDE))
            self.bufs.append(buf)
        self.N = N

    def config(self, poison=0, poison_round=0, grid=0, pool_cap=0):
        if self.on:
            self.L.gf_diag_config(C.c_uint32(poison), C.c_uint32(poison_round), C.c_uint32(grid), C.c_uint32(pool_cap))

    def tag(self, slot):
        return int(self.L.gf_diag_last_tag(C.c_uint32(slot))) if self.on else 0

    def records(self, slot):
        torch.cuda.synchronize()
        return self.bufs[slot].cpu().numpy().reshape(self.N, STRIDE, self.W)


def render_solo(pipe, diag, i, want_records=False):
    slot = pipe._slot
    f = pipe.render_frame(i)
    pipe.wait()
    ws_slot = slot % max(2, pipe.in_flight)
    out = f.clone().numpy()
    if want_records and diag.on:
        # phase 1 launches carry the tag of the same launch_head call as phase 0
        return out, (diag.records(ws_slot).copy(), diag.tag(ws_slot))
    return out, None


def compare_records(a, tag_a, b, tag_b, log, limit=12):
    ua, ub = a.view(np.uint32), b.view(np.uint32)
    va, vb = ua[..., 6] == tag_a, ub[..., 6] == tag_b
    log(f"    samples recorded: alone {int(va.sum())}, in flight {int(vb.sum())}, only one side {int((va != vb).sum())}")
    both = va & vb
    first = None
    for w, name in FIELDS.items():
        d = (ua[..., w] != ub[..., w]) & both
        n = int(d.sum())
        if n:
            fa, fb = a[..., w][d], b[..., w][d]
            log(f"    {name:6s}: {n} samples differ, max |d| = {float(np.abs(fa.astype(np.float64) - fb.astype(np.float64)).max()):.3e}")
    anyd = np.zeros_like(both)
    for w in FIELDS:
        anyd |= (ua[..., w] != ub[..., w]) & both
    rays, ks = np.nonzero(anyd)
    log(f"    rays with a differing sample: {sorted(set(rays.tolist()))[:40]}")
    for ray, k in list(zip(rays.tolist(), ks.tolist()))[:limit]:
        ra, rb = a[ray, k], b[ray, k]
        qa, qb = ua[ray, k], ub[ray, k]
        diffs = [FIELDS[w] for w in FIELDS if qa[w] != qb[w]]
        log(f"    ray {ray} sample {k}: differs in {diffs}")
        for side, r, q in (("alone    ", ra, qa), ("in flight", rb, qb)):
            log(f"      {side} wg {q[7]:3d} round {q[8]:3d} dense {q[9]:3d}/Mv {q[10]:3d} n_pool {q[11] >> 8:3d} n {q[11] & 255} phase {q[18]} raw {q[19]:3d} | "
                f"{hw(q)} | amb ({r[12]:.8f},{r[13]:.8f}) e ({r[24]:.9g},{r[25]:.9g}) [{q[24]:08x},{q[25]:08x}] x2 ({r[20]:.9g},{r[21]:.9g}) [{q[20]:08x},{q[21]:08x}] "
                f"c2d32 {r[22]:.8f} c2d {r[17]:.6f} h0 {r[14]:.6f} sigma {r[0]:.6f} xcc {q[58] & 15 if len(q) > 58 else -1}")
        if a.shape[-1] >= 58:
            lv = [(i // 2, i % 2, float(ra[26 + i]), float(rb[26 + i])) for i in range(32) if qa[26 + i] != qb[26 + i]]
            log("      levels: " + ", ".join(f"L{l}.{c} {x:.8f}->{y:.8f} ({y - x:+.2e})" for l, c, x, y in lv))
    return rays, ks


def main():
    out_dir = os.path.join(REPO, "gpurun_out", "diag")
    os.makedirs(out_dir, exist_ok=True)
    name = f"report_{args.precision}_{'plain' if args.plain else 'diag'}_{args.size}_if{args.in_flight}.txt"
    fh = open(os.path.join(out_dir, name), "w")

    def log(msg):
        print(msg, flush=True)
        fh.write(msg + "\n")
        fh.flush()

    log(f"# fast_diag: precision={args.precision} size={args.size} frames={args.frames} in_flight={args.in_flight} lib={os.environ.get('GF_HIP_LIB', 'product')}")
    log(f"# device: {torch.cuda.get_device_name(0)}")
    hp, model = build(args.precision)
    seq = sequence(4, args.size, args.size)
    N = args.size * args.size
    if args.place_table:
        # a stale (not yet returned) gather destination still holds the 64-bit address it was loaded from: make its low dword a large float
        want_lo = int(args.place_table, 16)
        emb = model.ambient_embedder.embeddings
        big = torch.empty(6 << 30, dtype=torch.uint8, device=DEV)
        base = big.data_ptr()
        offb = (want_lo - (base & 0xFFFFFFFF)) % (1 << 32)
        offb = (offb + 255) & ~255
        view = big[offb:offb + emb.numel() * 4].view(torch.float32).view_as(emb)
        view.copy_(emb.detach())
        emb.data = view
        fused.invalidate(model)
        log(f"ambient table moved to {emb.data_ptr():#x}")
        model._keep_big = big
    solo = FramePipeline(model, hp, seq, DEV, impl="fused", overlap=False)
    diag = Diag(model, N, max(2, args.in_flight))
    log(f"# per-sample records: {'on' if diag.on else 'off'}")
    steps = args.steps.split(",")

    diag.config()
    want, want_rec = [], []
    for i in range(4):
        o, r = render_solo(solo, diag, i, True)
        want.append(o)
        want_rec.append(r)
    # reproducibility alone
    nrep, bad = 20, 0
    for rep in range(nrep):
        for i in range(4):
            o, _ = render_solo(solo, diag, i)
            bad += int(not np.array_equal(o, want[i]))
    log(f"alone, {nrep * 4} renders: {bad} differ from the first render")

    if "solo" in steps and diag.on:
        n_bad, dumped, per_field = 0, 0, {}
        for rep in range(args.frames):
            i = rep % 4
            o, r = render_solo(solo, diag, i, True)
            ua, ub = want_rec[i][0].view(np.uint32), r[0].view(np.uint32)
            both = (ua[..., 6] == want_rec[i][1]) & (ub[..., 6] == r[1])
            cnt = {FIELDS[w]: int(((ua[..., w] != ub[..., w]) & both).sum()) for w in FIELDS}
            if any(cnt.values()):
                n_bad += 1
                for k_, v_ in cnt.items():
                    per_field[k_] = per_field.get(k_, 0) + v_
                if dumped < args.max_dumps:
                    dumped += 1
                    log(f"  solo render {rep} (frame {i}): uint8 bytes differing {int((o != want[i]).sum())}")
                    compare_records(want_rec[i][0], want_rec[i][1], r[0], r[1], log, limit=8)
        log(f"solo: {args.frames} renders, {n_bad} with differing records; differing words per field {per_field}")

    if "l15" in steps and diag.on:
        # Bit-exact emulation of the last level of the 2-D lookup for the samples whose L15 channel-0 feature differs between two
        # renders: which of the two is the correct one, and what exactly does the other one contain?
        ae = model.ambient_embedder
        tab = ae.embeddings.detach().cpu().numpy().astype(np.float32)
        off = ae.offsets.detach().cpu().numpy()
        base_ptr = int(ae.embeddings.data_ptr())
        log(f"ambient table at {base_ptr:#x}")
        f32 = np.float32
        l = 15
        scale = f32(np.exp2(f32(l) * f32(np.log2(ae.per_level_scale))) * f32(ae.base_resolution) - f32(1.0))
        res = int(np.ceil(scale)) + 1
        size = int(off[l + 1] - off[l])
        stride, sdim = 1, [0, 0]
        for d in range(2):
            if stride <= size:
                sdim[d] = stride
                stride *= res + 1
        mask = 0xFFFFFFFF if stride <= size else size - 1

        def fma(a, b, c):
            return f32(np.float64(a) * np.float64(b) + np.float64(c))

        def corners(x0, x1):
            p = [fma(f32(x0), scale, f32(0.5)), fma(f32(x1), scale, f32(0.5))]
            fl = [np.floor(q) for q in p]
            fr = [f32(q - f) for q, f in zip(p, fl)]
            g = [int(f) for f in fl]
            out = []
            for c in range(4):
                bx, by = c & 1, (c >> 1) & 1
                w = f32(fr[0] if bx else f32(1) - fr[0])
                w = f32(w * (fr[1] if by else f32(1) - fr[1]))
                idx = ((g[0] + bx) + (g[1] + by) * sdim[1]) & mask
                row = int(off[l]) + idx
                out.append((w, tab[row, 0], tab[row, 1], base_ptr + row * 8))
            return out

        def chain(cs, ch, repl=None, skip=None):
            o = f32(0)
            for c, (w, v0, v1, addr) in enumerate(cs):
                if skip == c:
                    continue
                v = (v0, v1)[ch]
                if repl is not None and repl[0] == c:
                    v = repl[1]
                o = fma(w, v, o)
            return o

        stats = {}
        for rep in range(args.frames):
            i = rep % 4
            o, r = render_solo(solo, diag, i, True)
            A, B = want_rec[i][0], r[0]
            ua, ub = A.view(np.uint32), B.view(np.uint32)
            both = (ua[..., 6] == want_rec[i][1]) & (ub[..., 6] == r[1])
            for col in range(26, 58):
                d = (ua[..., col] != ub[..., col]) & both
                if d.any():
                    stats[f"feature word {col - 26} (level {(col - 26) // 2}, channel {(col - 26) % 2}) differs"] = stats.get(f"feature word {col - 26} (level {(col - 26) // 2}, channel {(col - 26) % 2}) differs", 0) + int(d.sum())
            d = (ua[..., 56] != ub[..., 56]) & both
            for ray, k in zip(*np.nonzero(d)):
                cs = corners(A[ray, k, 20], A[ray, k, 21])
                t0 = chain(cs, 0)
                sides = {"first": A[ray, k, 56], "later": B[ray, k, 56]}
                good = [n for n, v in sides.items() if v.view(np.uint32) == t0.view(np.uint32)]
                stats["emulation reproduces one side exactly" if len(good) == 1 else "emulation reproduces neither/both"] = stats.get("emulation reproduces one side exactly" if len(good) == 1 else "emulation reproduces neither/both", 0) + 1
                if len(good) != 1:
                    continue
                bad = sides["later" if good[0] == "first" else "first"]
                hyp = {}
                for c in range(4):
                    hyp[f"corner {c} skipped"] = chain(cs, 0, skip=c)
                    lo = np.array([cs[c][3] & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0]
                    hi = np.array([cs[c][3] >> 32], dtype=np.uint32).view(np.float32)[0]
                    hyp[f"corner {c} value = low dword of its address"] = chain(cs, 0, repl=(c, lo))
                    hyp[f"corner {c} value = high dword of its address"] = chain(cs, 0, repl=(c, hi))
                    hyp[f"corner {c} value = its channel-1 value"] = chain(cs, 0, repl=(c, cs[c][2]))
                    for c2 in range(4):
                        if c2 != c:
                            hyp[f"corner {c} value = corner {c2} value"] = chain(cs, 0, repl=(c, cs[c2][1]))
                hit = [n for n, v in hyp.items() if v.view(np.uint32) == bad.view(np.uint32)]
                key = "glitched value == " + (" | ".join(hit) if hit else "none of the hypotheses")
                stats[key] = stats.get(key, 0) + 1
        for k_, v_ in sorted(stats.items(), key=lambda kv: -kv[1]):
            log(f"  l15: {v_:6d}  {k_}")

    if "poison" in steps and diag.on:
        for pz, pr in ((0x7FC07FC0, 0), (0x7FC07FC0, 0x7FC07FC0), (0x3C003C00, 0x3C003C00)):
            diag.config(poison=pz, poison_round=pr)
            nd = []
            for i in range(4):
                o, _ = render_solo(solo, diag, i)
                nd.append(int((o != want[i]).sum()))
            log(f"poison start={pz:#x} round={pr:#x}: differing bytes per frame {nd}")
        diag.config()

    if "group" in steps and diag.on:
        for grid, cap in ((0, 16), (0, 40), (0, 100), (25, 0), (64, 0), (100, 0), (137, 0), (64, 24), (333, 0), (512, 64)):
            diag.config(grid=grid, pool_cap=cap)
            nd, dumped = [], False
            for i in range(4):
                o, r = render_solo(solo, diag, i, True)
                nbytes = int((o != want[i]).sum())
                nd.append(nbytes)
                # records may differ even when the uint8 frame does not
                if r is not None:
                    ua, ub = want_rec[i][0].view(np.uint32), r[0].view(np.uint32)
                    both = (ua[..., 6] == want_rec[i][1]) & (ub[..., 6] == r[1])
                    nrec = sum(int(((ua[..., w] != ub[..., w]) & both).sum()) for w in FIELDS)
                    nd[-1] = (nbytes, nrec, int(((ua[..., 6] == want_rec[i][1]) != (ub[..., 6] == r[1])).sum()))
                    if nrec and not dumped:
                        dumped = True
                        log(f"  grid={grid} pool_cap={cap} frame {i}: records differ")
                        compare_records(want_rec[i][0], want_rec[i][1], r[0], r[1], log, limit=6)
            log(f"grouping grid={grid or 'default'} pool_cap={cap or 'default'}: (differing bytes, differing record words, samples on one side only) per frame {nd}")
        diag.config()

    if "stress" in steps:
        pipe = FramePipeline(model, hp, seq, DEV, impl="fused", in_flight=args.in_flight)
        depth = len(pipe._pinned)
        order = [0, 1, 2, 3, 3, 1, 0, 2]
        pending, n_dev, dumps, t0 = [], 0, 0, time.time()
        dev_frames = []

        def drain(entry):
            nonlocal n_dev, dumps
            k, idx, buf, ev, ws_slot, tag = entry
            ev.synchronize()
            got = buf.numpy()
            if not np.array_equal(got, want[idx]):
                n_dev += 1
                d = np.abs(got.astype(np.int32) - want[idx].astype(np.int32)).max(-1)
                ys, xs = np.nonzero(d)
                dev_frames.append(k)
                if dumps < args.max_dumps:
                    dumps += 1
                    log(f"  frame #{k} (sequence frame {idx}, slot {ws_slot}): {len(ys)} pixels differ, max {int(d.max())} LSB, rays {[int(y * args.size + x) for y, x in zip(ys, xs)][:16]}")
                    if diag.on:
                        rec = diag.records(ws_slot)
                        compare_records(want_rec[idx][0], want_rec[idx][1], rec, tag, log)

        for k in range(args.frames):
            idx = order[k % len(order)]
            slot = pipe._slot
            buf = pipe.render_frame(idx)
            ws_slot = slot % max(2, pipe.in_flight)
            pending.append((k, idx, buf, pipe._events[slot], ws_slot, diag.tag(ws_slot)))
            if len(pending) == depth:
                drain(pending.pop(0))
        for e in pending:
            drain(e)
        dt = time.time() - t0
        log(f"stress: {args.frames} frames, {args.in_flight} in flight, {dt:.1f} s ({args.frames / dt:.0f} fps): {n_dev} deviating frames {dev_frames[:20]}")
    fh.close()


if __name__ == "__main__":
    main()
