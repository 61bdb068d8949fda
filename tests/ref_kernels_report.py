"""Test infrastructure (GPU box): the reference's own kernels (oracle/_ref, the CUDA sources built for gfx950 by
oracle/refbuild/build_ref.py) next to the C oracle and the product's ops, same inputs, every output compared.
`python tests/ref_kernels_report.py` prints the full difference report (profiles/round1/r1z_ref_kernels_report.json);
tests/test_gpu_vs_ref_kernels.py asserts on it.  `frames()` also times the reference pipeline -- the torch restatement of the
reference's Python over the reference's kernels, one launch per op and one host sync per march iteration, as the reference
runs it -- on this GPU."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from helpers import frame_inputs, model_fixture, sequence   # noqa: E402
from oracle import kernels as K                              # noqa: E402
from oracle import radnerf_ref as R                          # noqa: E402
from oracle import ref_kernels                               # noqa: E402

DEV = "cuda:0"
S3 = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))


def backends():
    from geneface_amd.compat import _freqencoder, _gridencoder, _raymarching_face, _shencoder
    out = {"oracle": ((K.raymarching_face, K.gridencoder, K.shencoder, K.freqencoder), "cpu"),
           "product": ((_raymarching_face, _gridencoder, _shencoder, _freqencoder), DEV)}
    for c in ("fast", "off"):
        if ref_kernels.available(c):
            out["ref_" + c] = (ref_kernels.load(c), DEV)
    return out


def cases():
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, 256, 256), 2)
    ro, rd = fi["rays_o"].view(-1, 3).contiguous(), fi["rays_d"].view(-1, 3).contiguous()
    N = ro.shape[0]
    g = torch.Generator().manual_seed(17)

    def near_far(m, dev):
        RM = m[0]
        n, f = torch.empty(N, device=dev), torch.empty(N, device=dev)
        RM.near_far_from_aabb(ro.to(dev), rd.to(dev), sd["aabb_infer"].to(dev), N, hp["min_near"], n, f)
        return {"nears": n, "fars": f}

    nears, fars = R.near_far_from_aabb(ro, rd, sd["aabb_infer"], hp["min_near"])

    def march(n_step):
        def run(m, dev):
            RM = m[0]
            M = N * n_step
            M += 128 - M % 128
            xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
            alive, t = torch.arange(N, dtype=torch.int32, device=dev), nears.clone().to(dev)
            RM.march_rays(N, n_step, alive, t, ro.to(dev), rd.to(dev), 1.0, hp["dt_gamma"], hp["max_steps"], 1, 128, sd["density_bitfield"].to(dev),
                          nears.to(dev), fars.to(dev), xyzs, dirs, deltas, torch.zeros(N, device=dev))
            sig = (torch.rand(M, generator=torch.Generator().manual_seed(3)) * 60).to(dev)
            rgb = torch.rand(M, 3, generator=torch.Generator().manual_seed(4)).to(dev)
            ws, dep, img = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, 3, device=dev)
            RM.composite_rays(N, n_step, 1e-4, alive, t, sig, rgb, deltas, ws, dep, img)
            return {"xyzs": xyzs, "dirs": dirs, "deltas": deltas, "i:alive": alive, "rays_t": t, "ws": ws, "depth": dep, "image": img}
        return run

    coords = torch.randint(0, 128, (5000, 3), generator=g, dtype=torch.int32)
    grid = torch.rand(1, 64 ** 3, generator=g) * 20 - 1

    def maintenance(m, dev):
        RM = m[0]
        idx = torch.empty(5000, dtype=torch.int32, device=dev)
        RM.morton3D(coords.to(dev), 5000, idx)
        back = torch.empty(5000, 3, dtype=torch.int32, device=dev)
        RM.morton3D_invert(idx, 5000, back)
        bits = torch.empty(64 ** 3 // 8, dtype=torch.uint8, device=dev)
        RM.packbits(grid.to(dev), 64 ** 3 // 8, 10.0, bits)
        dil = torch.empty(1, 64 ** 3, device=dev)
        RM.morton3D_dilation(grid.to(dev), 1, 64, dil)
        return {"i:morton": idx, "i:invert": back, "i:bits": bits, "dilation": dil}

    def grid_fwd(D, gridtype, interp, table, off, B=1 << 18, with_dx=True):
        x = torch.rand(B, D, generator=torch.Generator().manual_seed(D * 10 + gridtype))
        x[:7] = 1.25
        grad = torch.randn(16, B, 2, generator=torch.Generator().manual_seed(8))

        def run(m, dev):
            GE = m[1]
            out = torch.empty(16, B, 2, device=dev)
            dy = torch.empty(B, 16 * D * 2, device=dev) if with_dx else None
            GE.grid_encode_forward(x.to(dev), table.to(dev), off.to(dev), out, B, D, 2, 16, S3, 16, dy, gridtype, False, interp)
            g_emb, g_in = torch.zeros(table.shape, device=dev), torch.zeros(B, D, device=dev)
            GE.grid_encode_backward(grad.to(dev), x.to(dev), table.to(dev), off.to(dev), g_emb, B, D, 2, 16, S3, 16, dy, g_in, gridtype, False, interp)
            return {"out": out, "dy_dx": dy, "g_emb": g_emb, "g_in": g_in}
        return run

    hp_t, sd_t = model_fixture(True)
    pe, po = sd["position_embedder.embeddings"], sd["position_embedder.offsets"]
    ae, ao = sd["ambient_embedder.embeddings"], sd["ambient_embedder.offsets"]
    from geneface_amd.encoders import get_encoder
    hm, _ = get_encoder("hashgrid", input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048,
                        interpolation="smoothstep")
    he = (torch.rand(hm.embeddings.shape, generator=g) * 2 - 1)

    B = 1 << 18
    d = torch.randn(B, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    gsh = torch.randn(B, 16, generator=g)

    def sh_case(degree):
        """degree 4 is GeneFace's; 5..8 are the rest of the extension's range (shencoder.cu:69-121,150-356), on unit directions and on free
        points of R^3 (the polynomials are differentiated there, not on the sphere)."""
        n = degree * degree
        pts = d if degree == 4 else torch.cat([d[:B // 2], torch.rand(B // 2, 3, generator=torch.Generator().manual_seed(degree)) * 2 - 1])
        gk = gsh if degree == 4 else torch.randn(B, n, generator=torch.Generator().manual_seed(100 + degree))

        def run(m, dev):
            SH = m[2]
            out, dy = torch.empty(B, n, device=dev), torch.empty(B, 3 * n, device=dev)
            SH.sh_encode_forward(pts.to(dev), out, B, 3, degree, dy)
            gi = torch.zeros(B, 3, device=dev)
            SH.sh_encode_backward(gk.to(dev), pts.to(dev), B, 3, degree, dy, gi)
            return {"out": out, "dy_dx": dy, "g_in": gi}
        return run

    sh = sh_case(4)

    xf = torch.rand(B, 2, generator=g) * 2 - 1
    gfq = torch.randn(B, 42, generator=g)

    def freq(m, dev):
        FQ = m[3]
        out = torch.empty(B, 42, device=dev)
        FQ.freq_encode_forward(xf.to(dev), B, 2, 10, 42, out)
        gi = torch.zeros(B, 2, device=dev)
        FQ.freq_encode_backward(gfq.to(dev), out, B, 2, 10, 42, gi)
        return {"out": out, "g_in": gi}

    fi64 = frame_inputs(sequence(4, 96, 96), 1)
    ro6, rd6 = fi64["rays_o"].view(-1, 3).contiguous(), fi64["rays_d"].view(-1, 3).contiguous()
    n6, f6 = R.near_far_from_aabb(ro6, rd6, sd["aabb_infer"], hp["min_near"])
    N6 = ro6.shape[0]
    noises = torch.rand(N6, generator=g)

    def train(m, dev):
        RM = m[0]
        max_steps = 64
        M = N6 * max_steps
        xyzs, dirs, deltas = torch.zeros(M, 3, device=dev), torch.zeros(M, 3, device=dev), torch.zeros(M, 2, device=dev)
        rays, counter = torch.empty(N6, 3, dtype=torch.int32, device=dev), torch.zeros(2, dtype=torch.int32, device=dev)
        RM.march_rays_train(ro6.to(dev), rd6.to(dev), sd["density_bitfield"].to(dev), 1.0, hp["dt_gamma"], max_steps, N6, 1, 128, M, n6.to(dev), f6.to(dev),
                            xyzs, dirs, deltas, rays, counter, noises.to(dev))
        # canonical (ray-major) order: the reference hands out offsets with atomics
        r = rays.cpu().numpy().astype(np.int64)
        order = np.argsort(r[:, 0], kind="stable")
        r = r[order]
        cnt = r[:, 2]
        tot = int(cnt.sum())
        rep = np.repeat(np.arange(N6), cnt)
        excl = np.cumsum(cnt) - cnt
        src = torch.from_numpy(r[rep, 1] + (np.arange(tot) - excl[rep]))
        xs, ds, des = xyzs.cpu()[src], dirs.cpu()[src], deltas.cpu()[src]
        # composite on canonicalised data (so every backend sees the same layout): rays table rebuilt in ray order
        rays_c = torch.from_numpy(np.stack([r[:, 0], excl, cnt], 1).astype(np.int32))
        sig = torch.rand(tot, generator=torch.Generator().manual_seed(5)) * 40
        rgb = torch.rand(tot, 3, generator=torch.Generator().manual_seed(6))
        amb = torch.rand(tot, generator=torch.Generator().manual_seed(7))
        ws, am, dep, img = (torch.empty(N6, device=dev), torch.empty(N6, device=dev), torch.empty(N6, device=dev), torch.empty(N6, 3, device=dev))
        RM.composite_rays_train_forward(sig.to(dev), rgb.to(dev), amb.to(dev), des.to(dev).contiguous(), rays_c.to(dev), tot, N6, 1e-4, ws, am, dep, img)
        gg = torch.Generator().manual_seed(9)
        gws, gam, gim = torch.rand(N6, generator=gg), torch.rand(N6, generator=gg), torch.rand(N6, 3, generator=gg)
        gs, gc, ga = torch.zeros(tot, device=dev), torch.zeros(tot, 3, device=dev), torch.zeros(tot, device=dev)
        RM.composite_rays_train_backward(gws.to(dev), gam.to(dev), gim.to(dev), sig.to(dev), rgb.to(dev), amb.to(dev), des.to(dev).contiguous(), rays_c.to(dev),
                                         ws, am, img, tot, N6, 1e-4, gs, gc, ga)
        gx, gd = torch.randn(tot, 3, generator=gg), torch.randn(tot, 3, generator=gg)
        go, gdd = torch.zeros(N6, 3, device=dev), torch.zeros(N6, 3, device=dev)
        RM.march_rays_train_backward(gx.to(dev), gd.to(dev), rays_c.to(dev), des.to(dev).contiguous(), N6, tot, go, gdd)
        return {"i:counts": torch.from_numpy(cnt), "i:counter": counter, "xyzs": xs, "dirs": ds, "deltas": des, "ws": ws, "amb": am, "depth": dep, "image": img,
                "g_sig": gs, "g_rgb": gc, "g_amb": ga, "g_ro": go, "g_rd": gdd}

    def sph_from_ray(m, dev):
        RM = m[0]
        c = torch.empty(N, 2, device=dev)
        RM.sph_from_ray(ro.to(dev), rd.to(dev), 3.0, N, c)
        return {"coords": c}

    def grad_tv(D, gridtype, table, off, B=1 << 16):
        x = torch.rand(B, D, generator=torch.Generator().manual_seed(31 + D))
        x[:5] = -0.25

        def run(m, dev):
            GE = m[1]
            gr = torch.zeros(table.shape, device=dev)
            GE.grad_total_variation(x.to(dev), table.to(dev), gr, off.to(dev), 1e-2, B, D, 2, 16, S3, 16, gridtype, False)
            return {"g_tv": gr}
        return run

    return {"sph_from_ray": sph_from_ray, "grad_tv3_tiled": grad_tv(3, 1, pe, po), "grad_tv2_tiled": grad_tv(2, 1, ae, ao),
            "grad_tv3_hash": grad_tv(3, 0, he, hm.offsets),
            "near_far": near_far, "march1": march(1), "march2": march(2), "march8": march(8), "maintenance": maintenance,
            "grid3_tiled_lin": grid_fwd(3, 1, 0, pe, po), "grid2_tiled_lin": grid_fwd(2, 1, 0, ae, ao),
            "grid3_hash_smooth": grid_fwd(3, 0, 1, he, hm.offsets), "sh": sh, "sh5": sh_case(5), "sh6": sh_case(6), "sh7": sh_case(7), "sh8": sh_case(8),
            "freq": freq, "train": train}


def diff(a, b, name):
    if a is None or b is None:
        return None
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.shape != b.shape:
        return {"shape": [list(a.shape), list(b.shape)]}
    if name.startswith("i:") or not a.dtype.is_floating_point:
        return {"mismatch": int((a != b).sum()), "n": a.numel()}
    dd = (a.double() - b.double()).abs()
    return {"max": float(dd.max()) if dd.numel() else 0.0, "n_diff": int((a != b).sum()), "n": a.numel(), "scale": float(a.abs().max()) if a.numel() else 0.0}


def frames(report, timing_iters=20, sizes=(256, 512)):
    """Whole frames: the torch restatement over the reference's kernels on the GPU vs (a) the same over the C oracle on CPU,
    (b) the product's fused path; and the reference pipeline's frame rate on this GPU."""
    hp, sd = model_fixture(True)
    sd_g = {k: v.to(DEV) for k, v in sd.items()}
    for size in sizes:
        seq = sequence(4, size, size)
        fi = frame_inputs(seq, 2)
        fg = {k: v.to(DEV) for k, v in fi.items()}
        for c in ("fast", "off"):
            if not ref_kernels.available(c):
                continue
            with R.kernel_backend(ref_kernels.load(c)):
                for _ in range(3):
                    out = R.render(sd_g, hp, fg["rays_o"], fg["rays_d"], fg["cond"], fg["bg_coords"], fg["pose6"], fg["bg"], True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = timing_iters
                for _ in range(n):
                    R.render(sd_g, hp, fg["rays_o"], fg["rays_d"], fg["cond"], fg["bg_coords"], fg["pose6"], fg["bg"], True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
            key = f"frame{size}_{c}"
            report[key] = {"ref_pipeline_ms": dt * 1e3, "ref_pipeline_fps": 1.0 / dt}
            if size == 256:
                o = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True)
                report[key]["oracle_vs_ref"] = {k: diff(o[k], out[k], k) for k in ("rgb_map", "depth_map", "weights_sum")}
            model = model_for(hp, sd)
            prod = {}
            for impl in ("fused", "ops"):
                model.render_impl = impl
                with torch.no_grad():
                    po = model.render(fg["rays_o"], fg["rays_d"], fg["cond"], fg["bg_coords"], fg["pose6"], index=0, staged=False, bg_color=fg["bg"],
                                      perturb=False, force_all_rays=True, **hp)
                torch.cuda.synchronize()
                prod[impl] = {k: diff(po[k].reshape(out[k].shape), out[k], k) for k in ("rgb_map", "depth_map")}
            report[key]["product_vs_ref"] = prod


def model_for(hp, sd):
    from geneface_amd.radnerf_torso import RADNeRFTorso
    m = RADNeRFTorso(hp)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def op_report():
    report = {}
    be = backends()
    for cname, fn in cases().items():
        outs = {}
        for bname, (mods, dev) in be.items():
            outs[bname] = fn(mods, dev)
            if dev != "cpu":
                torch.cuda.synchronize()
        for ref in [b for b in be if b.startswith("ref_")]:
            for other in ("oracle", "product"):
                report[f"{cname}:{other}_vs_{ref}"] = {k: diff(outs[other][k], outs[ref][k], k) for k in outs[ref]}
        if "ref_fast" in outs and "ref_off" in outs:
            report[f"{cname}:ref_off_vs_ref_fast"] = {k: diff(outs["ref_off"][k], outs["ref_fast"][k], k) for k in outs["ref_fast"]}
    return report


def main():
    report = op_report()
    frames(report)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
