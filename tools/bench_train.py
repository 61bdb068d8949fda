#!/usr/bin/env python
"""Training-step rate of the RAD-NeRF head on one MI355X (secondary measurement).

A step = what RADNeRFTask._training_step does around the model (tasks/radnerfs/radnerf.py:185-216): every 16 steps
`update_extra_state`, then `render` in training mode on n_rays = 65536 rays of one frame (base.yaml:55), MSE + regularisers,
backward, Adam.  The only number the reference publishes for this path is "~6 h for the head on an RTX 3090 Ti"
(docs/train_models/train_models.md:91) for max_updates = 250 000 (base.yaml:64), i.e. ~11.6 steps/s.
Synthetic fixture (no dataset offline): the rate, not the loss, is what is measured."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--n-rays", type=int, default=65536)
    ap.add_argument("--foreach-adam", action="store_true", help="torch.optim.Adam's default multi-tensor path (what the reference's trainer builds) "
                    "instead of fused=True")
    ap.add_argument("--amp", action="store_true", help="fp16 autocast + GradScaler, as the May config trains (lm3d_radnerf.yaml:5 amp: true)")
    ap.add_argument("--amp-f32-field", action="store_true", help="--amp with the exact-fp32 field node under autocast (RADNeRF.amp_field = 'f32': round 5's "
                    "behaviour) instead of the f16 tier")
    ap.add_argument("--amp-f32-backward", action="store_true", help="--amp with the fp32 dX chain (RADNeRF.amp_backward = 'f32': stage 1 of round 6)")
    ap.add_argument("--cond-ops", action="store_true", help="the condition encoder through the torch modules (RADNeRF.cond_impl = 'ops': the tree before "
                    "round 6's gf_cond_train_forward / _backward)")
    ap.add_argument("--gemm-wgrad", action="store_true", help="fp32 step with the weight gradients as batched library products (RADNeRF.wgrad_impl = 'gemm': "
                    "the tree before round 6's gf_field_wgrad32)")
    ap.add_argument("--amp-gemm-wgrad", action="store_true", help="--amp with the weight gradients as batched library products (RADNeRF.amp_wgrad = 'gemm': "
                    "stage 2 of round 6) instead of the fused kernel (gf_field_wgrad16)")
    ap.add_argument("--torso", action="store_true", help="the TORSO task's step (tasks/radnerfs/radnerf_torso.py:30-122): head frozen and rendered under no_grad, "
                    "only torso parameters in the optimizer (networks at lr, the 2-D grid at 10 lr), mse on rgb_map + the alpha entropy term, "
                    "RADNeRFTorso.update_extra_state (the 128x128 torso occupancy) every 16 steps")
    ap.add_argument("--torso-compact", action="store_true", help="--torso: the reference's boolean-mask compaction of the masked pixels (one host sync per "
                    "step: RADNeRFTorso.torso_train_dense = False) instead of the dense evaluation of round 6")
    ap.add_argument("--torso-gemm-wgrad", action="store_true", help="--torso: the torso weight gradients as batched library products and torch glue "
                    "(RADNeRFTorso.torso_wgrad_impl = 'gemm') instead of gf_torso_wgrad")
    ap.add_argument("--torso-blend-ops", action="store_true", help="--torso: the tail of the training branch (mask, blends, clamp) as torch expressions "
                    "(RADNeRFTorso.torso_blend_impl = 'ops') instead of the fused node")
    ap.add_argument("--op-graph", action="store_true", help="--torso: pin the torso field to the torch op graph (RADNeRFTorso.field_impl = 'ops' for the torso "
                    "field only: the tree before round 6's fused node), for same-box before / after")
    args = ap.parse_args()
    if args.torso:
        return main_torso(args)
    import torch
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd import utils
    from geneface_amd.radnerf import RADNeRF

    dev = torch.device("cuda", 0)
    hp = HP.may_hparams(False)
    sd = S.make_state_dict(hp, False)
    model = RADNeRF(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    if args.amp_f32_field:
        model.amp_field = "f32"
    if args.amp_f32_backward:
        model.amp_backward = "f32"
    if args.amp_gemm_wgrad:
        model.amp_wgrad = "gemm"
    if args.gemm_wgrad:
        model.wgrad_impl = "gemm"
    if args.cond_ops:
        model.cond_impl = "ops"
    seq = S.make_sequence(8, 512, 512, hp)
    poses = torch.from_numpy(seq["poses"]).to(dev)
    cond = torch.from_numpy(seq["cond_wins"]).to(dev)
    model.conds = cond[:, cond.shape[1] // 2]                      # [T, cond_win, C] for update_extra_state's random window
    bg = torch.from_numpy(seq["bg_img"]).to(dev).view(1, -1, 3)
    bgc = utils.get_bg_coords(512, 512, dev)
    target = torch.rand(1, 512 * 512, 3, device=dev)
    # fused=True: one multi-tensor kernel, and under a GradScaler the skip-on-overflow decision stays on the device -- the default (foreach)
    # optimizer makes GradScaler.step() read found_inf on the host every step, which serialises host and GPU
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.99), eps=1e-15, fused=not args.foreach_adam)
    torch.manual_seed(0)
    scaler = torch.amp.GradScaler("cuda", enabled=args.amp)

    def step(i):
        if i % hp["update_extra_interval"] == 0:
            model.update_extra_state()
        f = i % len(poses)
        rays = utils.get_rays(poses[f:f + 1], seq["intrinsics"], 512, 512, args.n_rays)   # random pixels, as the reference's dataset draws them
        sel = rays["inds"][0]
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            out = model.render(rays["rays_o"], rays["rays_d"], cond[f], bgc[:, sel], None, index=f, bg_color=bg[:, sel],
                               perturb=True, force_all_rays=False, **hp)
            loss = ((out["rgb_map"] - target[:, sel]) ** 2).mean() + 1e-3 * out["ambient"].mean()
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        return out

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = []
    for i in range(args.warmup, args.warmup + args.steps):
        h0 = time.perf_counter()
        out = step(i)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rate = args.steps / dt
    # how long the host needs to ENQUEUE a step (no sync inside one, except the occupancy refresh's read of mean_density every 16th): when this
    # is below ms_per_step the device is the bound, when it equals it the step is host-bound
    quiet = sorted(h for k, h in enumerate(host) if (args.warmup + k) % hp["update_extra_interval"] != 0)
    print(f"host enqueue per step: median {1e3 * quiet[len(quiet) // 2]:.2f} ms, min {1e3 * quiet[0]:.2f}, max {1e3 * quiet[-1]:.2f} "
          f"(steps without an occupancy refresh); step {1e3 * dt / args.steps:.2f} ms", file=sys.stderr)
    print(json.dumps({"metric": f"RAD-NeRF head training steps/s (n_rays {args.n_rays}, {('fp16 autocast, field on the ' + ('exact-fp32 node' if args.amp_f32_field else 'f16 tier')) if args.amp else 'fp32'}, {'foreach' if args.foreach_adam else 'fused'} Adam, grid update every 16 steps)", "value": rate,
                      "ms_per_step": 1e3 / rate, "hours_for_250k_steps": 250000 / rate / 3600, "points_last_step": int(model.step_counter[(model.local_step - 1) % 16, 0]),
                      "amp": {"field": getattr(model, "amp_field", "f16"), "backward": getattr(model, "amp_backward", "f16"),
                              "weight_gradients": getattr(model, "amp_wgrad", "fused")} if args.amp else None,
                      "weight_gradients": getattr(model, "amp_wgrad", "fused") if args.amp else getattr(model, "wgrad_impl", "fused"),
                      "cond_encoder": getattr(model, "cond_impl", "auto"),
                      "reference_published": "~6 h for 250 000 steps on an RTX 3090 Ti (~11.6 steps/s), docs/train_models/train_models.md:91",
                      "data": "synthetic"}))


def main_torso(args):
    import torch
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd import utils
    from geneface_amd.radnerf_torso import RADNeRFTorso

    dev = torch.device("cuda", 0)
    hp = HP.may_hparams(True)
    model = RADNeRFTorso(hp)
    model.load_state_dict(S.make_state_dict(hp, True), strict=True)
    model = model.to(dev).train()
    if args.torso_compact:
        model.torso_train_dense = False
    if args.torso_gemm_wgrad:
        model.torso_wgrad_impl = "gemm"
    if args.torso_blend_ops:
        model.torso_blend_impl = "ops"
    if args.op_graph:      # the torso field and the frozen head's condition encoder as round 5 ran them (the head's own fused field stays)
        model._fused_torso_train_ok = lambda *a, **k: False
        model._cond_feat_no_grad = model.cal_cond_feat
    seq = S.make_sequence(8, 512, 512, hp)
    poses = torch.from_numpy(seq["poses"]).to(dev)
    cond = torch.from_numpy(seq["cond_wins"]).to(dev)
    model.poses = poses                                             # update_extra_state draws its pose from them (radnerf_torso.py:44-46, :203-207)
    pose6 = utils.convert_poses(poses)
    bg = torch.from_numpy(seq["bg_img"]).to(dev).view(1, -1, 3)
    bgc = utils.get_bg_coords(512, 512, dev)
    target = torch.rand(1, 512 * 512, 3, device=dev)
    emb = [p for k, p in model.named_parameters() if "torso_embedder" in k]
    net = [p for k, p in model.named_parameters() if "torso_embedder" not in k and "torso" in k]
    for k, p in model.named_parameters():
        if "torso" not in k:
            p.requires_grad_(False)
    opt = torch.optim.Adam(net, lr=5e-4, betas=(0.9, 0.99), eps=1e-15, fused=not args.foreach_adam)
    opt.add_param_group({"params": emb, "lr": 5e-3, "betas": (0.9, 0.99), "eps": 1e-15})
    torch.manual_seed(0)
    scaler = torch.amp.GradScaler("cuda", enabled=args.amp)
    masked = [0]

    def step(i):
        if i % hp["update_extra_interval"] == 0:
            model.update_extra_state()
        f = i % len(poses)
        rays = utils.get_rays(poses[f:f + 1], seq["intrinsics"], 512, 512, args.n_rays)
        sel = rays["inds"][0]
        with torch.autocast("cuda", dtype=torch.float16, enabled=args.amp):
            out = model.render(rays["rays_o"], rays["rays_d"], cond[f], bgc[:, sel], pose6[f:f + 1], index=0, bg_color=bg[:, sel],
                               perturb=True, force_all_rays=False, **hp)
            alphas = out["torso_alpha_map"].clamp(1e-5, 1 - 1e-5)
            loss = ((out["rgb_map"] - target[:, sel]) ** 2).mean() \
                + 1e-3 * torch.mean(-alphas * torch.log2(alphas) - (1 - alphas) * torch.log2(1 - alphas))
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        return out

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rate = args.steps / dt
    print(json.dumps({"metric": f"RAD-NeRF TORSO training steps/s (head frozen; n_rays {args.n_rays}, {'fp16 autocast' if args.amp else 'fp32'}, "
                                f"{'foreach' if args.foreach_adam else 'fused'} Adam, torso occupancy update every 16 steps)", "value": rate,
                      "ms_per_step": 1e3 / rate, "hours_for_250k_steps": 250000 / rate / 3600,
                      "masked_pixels_last_step": int((out["torso_alpha_map"] > 0).sum()), "torso_train_dense": bool(model.torso_train_dense), "head_points_last_step": int(model.step_counter[(model.local_step - 1) % 16, 0]),
                      "reference_published": "~4 h for the torso on an RTX 3090 Ti, docs/train_models/train_models.md:93", "data": "synthetic"}))


if __name__ == "__main__":
    main()
