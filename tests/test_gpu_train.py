"""-m gpu: training-tier ray-marching ops through the C ABI against the oracle, and the autograd wrappers
of geneface_amd/raymarching.py (raymarching.py:185-342 of the reference)."""
import numpy as np
import pytest
import torch

from helpers import frame_inputs, model_fixture, sequence
from oracle import kernels as K
from oracle import radnerf_ref as R
from test_oracle_train import march_train

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RM = K.raymarching_face


def _scene(size=64, idx=1):
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, size, size), idx)
    ro, rd = fi["rays_o"].view(-1, 3).contiguous(), fi["rays_d"].view(-1, 3).contiguous()
    nears, fars = R.near_far_from_aabb(ro, rd, sd["aabb_infer"], hp["min_near"])
    return hp, sd, ro, rd, nears, fars


@pytest.mark.parametrize("size,max_steps", [(64, 16), (37, 64), (128, 16)])
def test_march_rays_train_bit_exact_vs_oracle(size, max_steps):
    from geneface_amd import raymarching as GR
    hp, sd, ro, rd, nears, fars = _scene(size)
    ref = march_train(hp, sd, ro, rd, nears, fars, max_steps=max_steps)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    xyzs, dirs, deltas, rays = GR.march_rays_train(ro.to(DEV), rd.to(DEV), float(hp["bound"]), sd["density_bitfield"].to(DEV), 1, hp["grid_size"],
                                                   nears.to(DEV), fars.to(DEV), counter, -1, False, 128, True, hp["dt_gamma"], max_steps)
    tot = ref[4][0].item()
    assert counter.cpu().tolist() == ref[4].tolist()
    assert torch.equal(rays.cpu(), ref[3])                      # (ray, offset, count): integer decisions identical, ray order
    assert xyzs.shape[0] == tot + (128 - tot % 128)             # align=128 padding rule of the wrapper
    assert torch.equal(xyzs[:tot].cpu(), ref[0][:tot]) and torch.equal(dirs[:tot].cpu(), ref[1][:tot]) and torch.equal(deltas[:tot].cpu(), ref[2][:tot])
    assert not xyzs[tot:].any()


def test_march_rays_train_overflow_and_perturb():
    """A buffer three times too small, with jitter.  The reference hands out point offsets with atomics (any order); the oracle restates the
    serial order, the product the ray order started at the 256-ray block floor(noises[0] * nblocks) -- so the rays that lose their samples
    move with the call's jitter instead of always being the tail of the batch.  Rolling the oracle's inputs by that start reproduces the
    product's buffers exactly."""
    from geneface_amd.compat import _raymarching_face as B
    hp, sd, ro, rd, nears, fars = _scene(48)
    N = ro.shape[0]
    g = torch.Generator().manual_seed(3)
    noises = torch.rand(N, generator=g)
    noises[0] = 0.6                                   # the entry that also picks the starting block
    nblocks = (N + 255) // 256
    r0 = min(int(float(noises[0]) * nblocks), nblocks - 1) * 256
    assert r0 > 0
    roll = lambda t: torch.roll(t, -r0, 0).contiguous()
    full = march_train(hp, sd, roll(ro), roll(rd), roll(nears), roll(fars), noises=roll(noises))
    M = full[4][0].item() // 3
    ref = march_train(hp, sd, roll(ro), roll(rd), roll(nears), roll(fars), M=M, noises=roll(noises))
    d = lambda t: t.to(DEV)
    xyzs, dirs, deltas = torch.zeros(M, 3, device=DEV), torch.zeros(M, 3, device=DEV), torch.zeros(M, 2, device=DEV)
    rays, counter = torch.empty(N, 3, dtype=torch.int32, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV)
    B.march_rays_train(d(ro), d(rd), d(sd["density_bitfield"]), float(hp["bound"]), hp["dt_gamma"], hp["max_steps"], N, 1, hp["grid_size"], M,
                       d(nears), d(fars), xyzs, dirs, deltas, rays, counter, d(noises))
    assert counter.cpu().tolist() == ref[4].tolist()
    got = rays.cpu()
    assert torch.equal(got[:, 0], torch.arange(N, dtype=torch.int32))                  # row n describes ray n
    assert torch.equal(roll(got[:, 1:]), ref[3][:, 1:])                                # offsets / counts of the rolled order
    assert torch.equal(xyzs.cpu(), ref[0]) and torch.equal(deltas.cpu(), ref[2])
    dropped = (got[:, 2] > 0) & (got[:, 1] + got[:, 2] > M)
    assert dropped.any() and not dropped[r0:r0 + 256].any()                            # the victims are the tail of the ROTATED order
    # a second call accumulates into the counter like the reference's atomics
    B.march_rays_train(d(ro), d(rd), d(sd["density_bitfield"]), float(hp["bound"]), hp["dt_gamma"], hp["max_steps"], N, 1, hp["grid_size"], M,
                       d(nears), d(fars), xyzs, dirs, deltas, rays, counter, d(noises))
    assert counter.cpu().tolist() == [2 * ref[4][0].item(), 2 * N]


def test_composite_rays_train_and_backward_vs_oracle_and_autograd():
    from geneface_amd import raymarching as GR
    hp, sd, ro, rd, nears, fars = _scene(64)
    xyzs, dirs, deltas, rays, counter = march_train(hp, sd, ro, rd, nears, fars)
    N, M = ro.shape[0], counter[0].item()
    g = torch.Generator().manual_seed(4)
    sigmas, rgbs, ambient = torch.rand(M, generator=g) * 40, torch.rand(M, 3, generator=g), torch.rand(M, generator=g)
    gws, gamb, gimg = torch.rand(N, generator=g), torch.rand(N, generator=g), torch.rand(N, 3, generator=g)
    de = deltas[:M].contiguous()
    for T_thresh in (1e-4, 0.2):
        ws, amb, dep, img = torch.empty(N), torch.empty(N), torch.empty(N), torch.empty(N, 3)
        RM.composite_rays_train_forward(sigmas, rgbs, ambient, de, rays, M, N, T_thresh, ws, amb, dep, img)
        gs, gc, ga = torch.zeros(M), torch.zeros(M, 3), torch.zeros(M)
        RM.composite_rays_train_backward(gws, gamb, gimg, sigmas, rgbs, ambient, de, rays, ws, amb, img, M, N, T_thresh, gs, gc, ga)
        s_g, c_g, a_g = (t.to(DEV).requires_grad_(True) for t in (sigmas, rgbs, ambient))
        ws_g, amb_g, dep_g, img_g = GR.composite_rays_train(s_g, c_g, a_g, de.to(DEV), rays.to(DEV), T_thresh)
        assert (ws_g.cpu() - ws).abs().max() < 1e-5 and (img_g.cpu() - img).abs().max() < 1e-5
        assert (dep_g.cpu() - dep).abs().max() < 1e-4 and (amb_g.cpu() - amb).abs().max() < 1e-4
        ((ws_g * gws.to(DEV)).sum() + (amb_g * gamb.to(DEV)).sum() + (img_g * gimg.to(DEV)).sum() + 0.0 * dep_g.sum()).backward()
        assert (c_g.grad.cpu() - gc).abs().max() < 1e-5 and (a_g.grad.cpu() - ga).abs().max() < 1e-6
        assert (s_g.grad.cpu() - gs).abs().max() < 1e-4 * max(1.0, float(gs.abs().max()))


def test_march_rays_train_backward_vs_oracle():
    from geneface_amd import raymarching as GR
    hp, sd, ro, rd, nears, fars = _scene(48)
    N = ro.shape[0]
    ro_g, rd_g = ro.to(DEV).requires_grad_(True), rd.to(DEV).requires_grad_(True)
    xyzs, dirs, deltas, rays = GR.march_rays_train(ro_g, rd_g, float(hp["bound"]), sd["density_bitfield"].to(DEV), 1, hp["grid_size"], nears.to(DEV),
                                                   fars.to(DEV), None, -1, False, -1, True, hp["dt_gamma"], hp["max_steps"])
    M = xyzs.shape[0]
    g = torch.Generator().manual_seed(5)
    gx, gd = torch.randn(M, 3, generator=g), torch.randn(M, 3, generator=g)
    ((xyzs * gx.to(DEV)).sum() + (dirs * gd.to(DEV)).sum()).backward()
    go, gdd = torch.zeros(N, 3), torch.zeros(N, 3)
    RM.march_rays_train_backward(gx, gd, rays.cpu(), deltas.detach().cpu().contiguous(), N, M, go, gdd)
    assert (ro_g.grad.cpu() - go).abs().max() < 1e-4 and (rd_g.grad.cpu() - gdd).abs().max() < 1e-3


# ----------------------------------------------------------------------------------------------- encoders, backward
@pytest.mark.parametrize("D,gridtype,interp", [(3, "tiled", "linear"), (2, "tiled", "linear"), (3, "hash", "linear"), (2, "hash", "smoothstep")])
def test_grid_encoder_autograd_vs_oracle(D, gridtype, interp):
    """GridEncoder.forward / backward through torch autograd (grid.py:24-90): table gradient (f32-atomic scatter) and input gradient
    against the oracle's kernels on the same inputs."""
    from geneface_amd.encoders.gridencoder import GridEncoder
    enc = GridEncoder(input_dim=D, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=14, desired_resolution=1024,
                      gridtype=gridtype, interpolation=interp).to(DEV)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        enc.embeddings.copy_((torch.rand(enc.embeddings.shape, generator=g) * 2 - 1).to(DEV))
    B = 50_000
    x = (torch.rand(B, D, generator=g) * 2 - 1)
    x[:5] = 1.5                                            # outside [-bound, bound]: zero features, zero gradients
    grad = torch.randn(B, 32, generator=g)
    x_g = x.to(DEV).requires_grad_(True)
    out = enc(x_g, bound=1)
    (out * grad.to(DEV)).sum().backward()
    # oracle on the same numbers
    off, emb = enc.offsets.cpu(), enc.embeddings.detach().cpu()
    S = float(torch.log2(torch.tensor(enc.per_level_scale, dtype=torch.float64)))
    x01 = ((x + 1) / 2).contiguous()
    o_ref, dy = torch.empty(16, B, 2), torch.empty(B, 16 * D * 2)
    K.gridencoder.grid_encode_forward(x01, emb, off, o_ref, B, D, 2, 16, S, 16, dy, enc.gridtype_id, False, enc.interp_id)
    assert (out.detach().cpu() - o_ref.permute(1, 0, 2).reshape(B, 32)).abs().max() < 1e-5
    g_emb, g_in = torch.zeros_like(emb), torch.zeros(B, D)
    K.gridencoder.grid_encode_backward(grad.view(B, 16, 2).permute(1, 0, 2).contiguous(), x01, emb, off, g_emb, B, D, 2, 16, S, 16, dy, g_in,
                                       enc.gridtype_id, False, enc.interp_id)
    ge = enc.embeddings.grad.cpu()
    assert (ge - g_emb).abs().max() < 2e-4 * max(1.0, float(g_emb.abs().max()))     # accumulation order differs (atomics)
    gi = x_g.grad.cpu() * 2                                   # d/dx of (x + bound) / (2 bound) is 1/2
    assert (gi - g_in).abs().max() < 1e-3 * max(1.0, float(g_in.abs().max()))
    assert not x_g.grad[:5].any()
    # inference-style call: no graph, no dy_dx
    with torch.no_grad():
        assert torch.equal(enc(x.to(DEV), bound=1), out.detach())


def test_sh_and_freq_encoder_autograd_vs_oracle():
    from geneface_amd.encoders.freqencoder import FreqEncoder
    from geneface_amd.encoders.shencoder import SHEncoder
    g = torch.Generator().manual_seed(12)
    B = 20_000
    d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    grad = torch.randn(B, 16, generator=g)
    d_g = d.to(DEV).requires_grad_(True)
    out = SHEncoder(3, 4)(d_g)
    (out * grad.to(DEV)).sum().backward()
    o_ref, dy, gi = torch.empty(B, 16), torch.empty(B, 3, 16), torch.zeros(B, 3)
    K.shencoder.sh_encode_forward(d, o_ref, B, 3, 4, dy)
    K.shencoder.sh_encode_backward(grad, d, B, 3, 4, dy, gi)
    assert (out.detach().cpu() - o_ref).abs().max() < 1e-6 and (d_g.grad.cpu() - gi).abs().max() < 1e-5
    x = torch.randn(B, 6, generator=g)
    fe = FreqEncoder(6, 4)
    gr = torch.randn(B, fe.output_dim, generator=g)
    x_g = x.to(DEV).requires_grad_(True)
    fo = fe(x_g)
    (fo * gr.to(DEV)).sum().backward()
    f_ref, gin = torch.empty(B, fe.output_dim), torch.zeros(B, 6)
    K.freqencoder.freq_encode_forward(x, B, 6, 4, fe.output_dim, f_ref)
    K.freqencoder.freq_encode_backward(gr, f_ref, B, 6, 4, fe.output_dim, gin)
    assert (fo.detach().cpu() - f_ref).abs().max() < 1e-5 and (x_g.grad.cpu() - gin).abs().max() < 2e-4


# ----------------------------------------------------------------------------------------------- training branch of render()
@pytest.mark.parametrize("torso", [False, True])
@pytest.mark.parametrize("field_impl", ["auto", "ops"])
def test_render_training_branch_gradients_vs_oracle(torso, field_impl):
    """model.train(); model.render(...) takes the reference's training branch (renderer.py:296-313 / radnerf_torso.py:93-198): one
    loss, one backward, every parameter gradient against the oracle's differentiable restatement on the CPU."""
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.radnerf_torso import RADNeRFTorso
    from test_oracle_train import _loss
    hp, sd = model_fixture(torso)
    fi = frame_inputs(sequence(4, 40, 40), 2)
    g = torch.Generator().manual_seed(8)
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=g)
    # oracle
    sd_g = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.startswith(("aabb", "density")) else v) for k, v in sd.items()}
    ref = R.render_train(sd_g, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso)
    _loss(ref, target).backward()
    # GPU
    model = (RADNeRFTorso if torso else RADNeRF)(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    model.field_impl = field_impl      # "auto": the field is one autograd node (fused forward, hand-written backward); "ops": the torch graph
    to = lambda t: t.to(DEV)
    out = model.render(to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]), index=0, staged=False,
                       bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
    assert (out["rgb_map"].detach().cpu() - ref["rgb_map"].detach()).abs().max() < 5e-4
    assert (out["weights_sum"].detach().cpu() - ref["weights_sum"].detach()).abs().max() < 5e-4
    _loss(out, to(target)).backward()
    assert int(model.step_counter[0, 0]) == ref["n_points"] and model.local_step == 1
    checked = 0
    for name, p in model.named_parameters():
        gr = sd_g[name].grad
        if gr is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name      # head frozen under no_grad in the torso model
            continue
        assert p.grad is not None, name
        # fp32 on both sides, different summation orders, exp() of the density head in between; and a sample whose ambient coordinate
        # sits within rounding of a cell face of the 2048-level 2-D grid takes its (piecewise constant) dy_dx from the neighbouring cell
        # on one side only -- so: tight in the L2 sense, loose on the single worst entry
        diff = (p.grad.cpu() - gr).double()
        l2 = float(diff.norm() / gr.double().norm().clamp(min=1e-20))
        worst = float(diff.abs().max()) / max(float(gr.abs().max()), 1e-12)
        assert l2 < 1e-2 and worst < 0.1, (name, l2, worst)
        checked += 1
    assert checked >= (6 if torso else 20)
    # eval() switches back to the inference branch (fused kernels)
    model.eval()
    with torch.no_grad():
        ev = model.render(to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]), index=0, staged=False,
                          bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
    assert "weights_sum" not in ev and ev["rgb_map"].shape == out["rgb_map"].shape


def test_fused_training_field_vs_op_graph():
    """RADNeRF.forward under autograd, one node (train_field.head_field) against the op-by-op torch graph on the same 20 000 points: the
    three outputs and the gradient of every tensor the field touches (two grid tables, eight weights, the condition encoder through
    cond_feat, the identity code), for an arbitrary downstream loss."""
    from geneface_amd.radnerf import RADNeRF
    hp, sd = model_fixture(False)
    g = torch.Generator().manual_seed(6)
    M = 20000
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * torch.tensor([0.5, 0.3, 0.5])).to(DEV)
    d = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1).to(DEV)
    cond = torch.randn(5, 1, 204, generator=g).to(DEV)
    ws, wc, wa = torch.rand(M, generator=g).to(DEV), torch.rand(M, 3, generator=g).to(DEV), torch.randn(M, 2, generator=g).to(DEV)
    res = {}
    for impl in ("ops", "auto"):
        m = RADNeRF(hp)
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).train()
        m.field_impl = impl
        cf = m.cal_cond_feat(cond)
        s_, c_, a_ = m(x, d, cf, m.individual_embeddings[3])
        loss = (torch.log1p(s_) * ws).sum() * 1e-3 + (c_ * wc).sum() * 1e-2 + (a_ * wa).sum() * 1e-2 + a_.abs().sum() * 1e-3
        loss.backward()
        res[impl] = (s_.detach(), c_.detach(), a_.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    so, co, ao, go = res["ops"]
    sf, cf_, af, gf = res["auto"]
    assert ((sf - so).abs() / so.abs().clamp(min=1e-3)).max() < 2e-3 and (cf_ - co).abs().max() < 1e-4 and (af - ao).abs().max() < 1e-5
    assert set(gf) == set(go) and len(gf) >= 24
    for n in go:
        diff = (gf[n] - go[n]).double()
        l2 = float(diff.norm() / go[n].double().norm().clamp(min=1e-20))
        worst = float(diff.abs().max()) / max(float(go[n].abs().max()), 1e-12)
        # the single-element attention bias sits on a cancelling sum (the same constant added to all five window logits in front of a
        # softmax): its gradient is rounding residue of the forward's summation order -- 3e-3 with the row sums of rounds 2-3, 6.5e-3 with
        # round 4's per-wave partial sums -- so it gets the "worst element" bar; every real tensor keeps 5e-3
        if n == "cond_att_net.attentionConvNet.8.bias":
            # mathematically zero (a constant in front of a softmax): what both sides hold is rounding residue, so the statement is "tiny
            # against the gradients that matter", not a relative error between two residues
            scale = max(float(go[k].abs().max()) for k in go if k.startswith("cond_att_net.attentionConvNet.8"))
            assert float(gf[n].abs().max()) <= 1e-3 * max(scale, 1e-12) or l2 < 2e-2, (n, float(gf[n].abs().max()), scale, l2)
            continue
        assert l2 < 5e-3 and worst < 2e-2, (n, l2, worst)


def test_training_branch_under_fp16_autocast():
    """The May config trains with amp: true (lm3d_radnerf.yaml:5; utils/commons/trainer.py wraps the step in autocast).  The ops keep
    the reference's custom_fwd(cast_inputs=float32) semantics: the nn.Linear layers run in fp16, every HIP op in fp32, one GradScaler
    step leaves finite gradients and the image stays within fp16 noise of the fp32 run."""
    from geneface_amd.radnerf import RADNeRF
    from test_oracle_train import _loss
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, 40, 40), 2)
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    model = RADNeRF(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    ref = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    with torch.autocast("cuda", dtype=torch.float16):
        out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
        loss = _loss(out, target)
    assert out["rgb_map"].dtype == torch.float32                                  # compositing is forced to fp32
    assert (out["rgb_map"].detach() - ref["rgb_map"].detach()).abs().max() < 3e-2
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    n = 0
    for name, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
            n += 1
    assert n >= 20 and scaler.get_scale() >= 1024.0                               # no inf/nan step was skipped


def test_grid_backward_keeps_tiny_gradients():
    """the table scatter accumulates in fixed point scaled by the level's LARGEST gradient.  Entries that only ever receive tiny
    gradients (samples behind T ~ 1e-4, rarely hit hash rows) must still get them with fp32 relative accuracy -- the reference trains the
    tables with Adam eps = 1e-15 (tasks/radnerfs/radnerf.py:63) so that exactly those entries take full-size steps.  Three bands of one
    dense 2-D grid receive gradients of relative size 1, 1e-7 and 1e-14; per-entry error against the oracle's scatter, relative to the
    entry's own sum of |contributions|."""
    from geneface_amd.encoders.gridencoder import GridEncoder
    D, B = 2, 120_000
    enc = GridEncoder(input_dim=D, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=512, gridtype="tiled").to(DEV)
    g = torch.Generator().manual_seed(5)
    x01 = torch.rand(B, D, generator=g)
    band = torch.randint(0, 3, (B,), generator=g)
    x01[:, 0] = (x01[:, 0] * 0.27 + band * 0.36).clamp(0, 1)             # bands [0, .27], [.36, .63], [.72, .99]: no cell of any level is shared
    scale = torch.tensor([1.0, 1e-7, 1e-14])[band]
    grad = torch.randn(B, 32, generator=g) * scale[:, None]
    x = (x01 * 2 - 1).to(DEV)
    out = enc(x, bound=1)
    (out * grad.to(DEV)).sum().backward()
    got = enc.embeddings.grad.cpu()
    off, emb = enc.offsets.cpu(), enc.embeddings.detach().cpu()
    S = float(torch.log2(torch.tensor(enc.per_level_scale, dtype=torch.float64)))
    x01c = ((x.cpu() + 1) / 2).contiguous()
    dy, gi = torch.empty(B, 16 * D * 2), torch.zeros(B, D)
    ref, mass = torch.zeros_like(emb), torch.zeros_like(emb)
    glbc = grad.view(B, 16, 2).permute(1, 0, 2).contiguous()
    o_ref = torch.empty(16, B, 2)
    K.gridencoder.grid_encode_forward(x01c, emb, off, o_ref, B, D, 2, 16, S, 16, dy, enc.gridtype_id, False, enc.interp_id)
    K.gridencoder.grid_encode_backward(glbc, x01c, emb, off, ref, B, D, 2, 16, S, 16, dy, gi, enc.gridtype_id, False, enc.interp_id)
    K.gridencoder.grid_encode_backward(glbc.abs(), x01c, emb, off, mass, B, D, 2, 16, S, 16, dy, gi, enc.gridtype_id, False, enc.interp_id)
    hit = mass > 0
    rel = ((got - ref).abs()[hit] / mass[hit])
    assert float(rel.max()) < 1e-4, float(rel.max())      # measured 2.4e-5: the ORACLE's fp32 running sum over the ~400 contributions of a coarse-level entry
    tiny = hit & (mass < 1e-10)                                            # entries fed by the 1e-14 band alone
    assert int(tiny.sum()) > 10_000 and int((got[tiny] != 0).sum()) > 0.99 * int((ref[tiny] != 0).sum())
    assert not got[~hit].any()


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_non_finite_gradients_reach_the_tables(bad):
    """A GradScaler decides from the parameter gradients whether a step overflowed.  The table scatter accumulates in fixed point scaled by
    the level's largest gradient, so an inf / NaN in the incoming gradient must not be rounded away: the table gradient of that level turns
    non-finite -- through the stand-alone op (its own max pass) and through the fused field (maxima written by the backward kernel)."""
    from geneface_amd.encoders.gridencoder import GridEncoder
    from geneface_amd.radnerf import RADNeRF
    g = torch.Generator().manual_seed(2)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048).to(DEV)
    x = torch.rand(5000, 3, generator=g).to(DEV)
    out = enc(x, bound=1)
    w = torch.ones_like(out)
    w[123, 7] = bad
    (out * w).sum().backward()
    assert not torch.isfinite(enc.embeddings.grad).all()
    hp, sd = model_fixture(False)
    m = RADNeRF(hp)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    M = 4096
    xs = ((torch.rand(M, 3, generator=g) * 2 - 1) * 0.4).to(DEV)
    d = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1).to(DEV)
    cf = m.cal_cond_feat(torch.randn(5, 1, 204, generator=g).to(DEV))
    s_, c_, a_ = m(xs, d, cf, m.individual_embeddings[0])
    ws = torch.ones(M, device=DEV)
    ws[77] = bad
    ((s_ * ws).sum() + c_.sum() + a_.sum()).backward()
    assert not torch.isfinite(m.position_embedder.embeddings.grad).all()


# ----------------------------------------------------------------------------------------------- fused torso training field (round 6)
@pytest.mark.parametrize("M", [1, 37, 4096 + 77])
def test_fused_torso_field_vs_op_graph(M):
    """forward_torso as ONE autograd node (train_torso.py: gf_torso_train_forward / _backward) against the same module's torch op graph on the
    same pixels: outputs, and the gradients of every torso parameter (grid table, six Linear weights, identity code) for a loss that
    touches alpha, colour AND the deformation (the reference returns `deform`; a regulariser on it must reach the deform net too).
    fp32 on both sides; what differs is summation order (MFMA k-order vs rocBLAS) and the fused sine of the encodings (<= 9.3e-8 abs)."""
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp, sd = model_fixture(True)
    model = RADNeRFTorso(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(M, 2, device=DEV, generator=g) * 2.4 - 1.2          # some pixels beyond the picture: the clamp's zero-gradient branch
    pose = torch.tensor([[0.1, -0.2, 0.05, 0.3, -0.1, 3.2]], device=DEV)
    wa, wc, wd = torch.rand(M, 1, device=DEV, generator=g), torch.rand(M, 3, device=DEV, generator=g), torch.rand(M, 2, device=DEV, generator=g)
    names = [n for n, _ in model.named_parameters() if "torso" in n]
    res = {}
    for impl in ("auto", "ops"):
        model.field_impl = impl
        model.zero_grad(set_to_none=True)
        code = model.torso_individual_codes[0]
        assert model._fused_torso_train_ok(x, code, None) == (impl == "auto")
        a, c, dx = model.forward_torso(x, pose, code)
        ((a * wa).sum() + (c * wc).sum() + 0.3 * (dx * wd).sum()).backward()
        res[impl] = ([t.detach().clone() for t in (a, c, dx)], {n: p.grad.detach().clone() for n, p in model.named_parameters() if n in names and p.grad is not None})
    for u, v, tol in zip(res["auto"][0], res["ops"][0], (2e-6, 2e-6, 2e-6)):
        assert u.shape == v.shape and float((u - v).abs().max()) < tol, float((u - v).abs().max())
    assert set(res["auto"][1]) == set(res["ops"][1]) and len(res["ops"][1]) >= 8, sorted(res["ops"][1])
    for n, gr in res["ops"][1].items():
        d = (res["auto"][1][n] - gr).double()
        l2 = float(d.norm() / gr.double().norm().clamp(min=1e-20))
        assert l2 < 2e-4, (n, l2)         # measured 1e-6 .. 3e-5


# ----------------------------------------------------------------------------------------------- AMP training tier (round 6)
@pytest.mark.parametrize("backward", ["f16", "f32"])
def test_amp_field_node_vs_fp32_node(backward):
    """Under torch.autocast(float16) the fused field runs on the f16 tier (train_field._HeadFieldAMP: f16 MFMA operands re-gathered from the
    fp32 master weights, fp32 accumulation, binary16 saves, half operands in the weight-gradient products) -- the arithmetic of the
    reference's AMP step (base.yaml:49 amp: true; trainer.py:307-382).  Against the exact-fp32 node on the same points: outputs and every
    gradient within half-precision bars (operands rounded to 11 bits: ~1e-3 relative per product, sqrt(K)-averaged over K = 32..144)."""
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.train_field import _HeadField, _HeadFieldAMP, head_field
    hp, sd = model_fixture(False)
    model = RADNeRF(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    model.amp_backward = backward      # the dX chain on the f16 matrix pipe (k_field_backward16), or the fp32 chain with binary16 outputs
    g = torch.Generator(device=DEV).manual_seed(3)
    M = 5000 + 77
    xyz = torch.rand(M, 3, device=DEV, generator=g) * 1.6 - 0.8
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=DEV, generator=g), dim=-1)
    cond = torch.randn(64, device=DEV, generator=g) * 0.3
    code = model.individual_embeddings[0]
    ws, wc, wa = torch.rand(M, device=DEV, generator=g), torch.rand(M, 3, device=DEV, generator=g), torch.rand(M, 2, device=DEV, generator=g)
    res = {}
    for tier in ("f32", "f16"):
        model.amp_field = tier
        model.zero_grad(set_to_none=True)
        cf = cond.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb, amb = head_field(model, xyz, dirs, cf, code)
            assert sigma.dtype == rgb.dtype == amb.dtype == torch.float32
        # log-density weighs the (exp-amplified) sigma so that no single sample dominates the comparison
        ((torch.log1p(sigma) * ws).sum() + (rgb * wc).sum() + (amb * wa).sum()).backward()
        res[tier] = ([t.detach().clone() for t in (sigma, rgb, amb)], {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None},
                     cf.grad.detach().float().clone())
    model.amp_field = "f16"
    (s32, c32, a32), (s16, c16, a16) = res["f32"][0], res["f16"][0]
    assert float((c16 - c32).abs().max()) < 3e-2 and float((a16 - a32).abs().max()) < 3e-2
    assert float(((s16 - s32).abs() / (s32.abs() + 1e-3)).median()) < 2e-2
    assert set(res["f16"][1]) == set(res["f32"][1]) and len(res["f32"][1]) >= 11
    for n, gr in res["f32"][1].items():
        d = (res["f16"][1][n] - gr).double()
        l2 = float(d.norm() / gr.double().norm().clamp(min=1e-20))
        # the two tables sit behind the longest chains (six layers of f16-rounded operands and the ReLU masks of the f16 forward: a
        # pre-activation within rounding of zero flips its mask bit, and with it that sample's path): measured 6.8 % on the 3-D table,
        # <= 3 % on every weight matrix
        assert l2 < (1e-1 if "embedder" in n else 5e-2), (n, l2)
    dcf = float((res["f16"][2] - res["f32"][2]).double().norm() / res["f32"][2].double().norm())
    assert dcf < 5e-2, dcf


def test_amp_training_steps_follow_the_fp32_steps():
    """Eight optimizer steps of the training branch under fp16 autocast + GradScaler on the f16 tier against the same eight steps in fp32:
    the loss curves agree within half-precision noise and no step is skipped (finite gradients at the initial scale)."""
    from geneface_amd.radnerf import RADNeRF
    from test_oracle_train import _loss
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, 40, 40), 2)
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    curves = {}
    for amp in (False, True):
        model = RADNeRF(hp)
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).train()
        opt = torch.optim.Adam(model.parameters(), lr=2e-4)
        scaler = torch.amp.GradScaler("cuda", init_scale=256.0, enabled=amp)
        losses = []
        for _ in range(8):
            with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
                out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
                loss = _loss(out, target)
            opt.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(float(loss))
        curves[amp] = losses
        if amp:
            assert scaler.get_scale() >= 256.0          # no inf / nan step
    a, b = curves[True], curves[False]
    assert all(abs(x / y - 1.0) < 5e-2 for x, y in zip(a, b)), (a, b)
    assert b[-1] < b[0] and a[-1] < a[0]


@pytest.mark.parametrize("tier", ["f16", "f32"])
@pytest.mark.parametrize("M", [1, 37, 4173, 300_001])
def test_fused_weight_gradient_products_vs_float64(M, tier):
    """gf_field_wgrad16 / gf_field_wgrad32 (csrc/field_wgrad.hip): the eight tall products G^T X of the backward in one launch + a fixed-order
    reduction, on binary16 rows (AMP tier, f16 MFMA) or fp32 rows (exact tier, f32 MFMA).
    Against the same operands multiplied in float64: fp32 accumulation only (1e-5 of each tensor's norm); the blocks the kernel
    does not own (identity-code / condition columns) untouched; the same bits on a second call; row counts that are not multiples of the
    32-row stage, below one stage, and large enough that every workgroup of every group has rows."""
    import ctypes as C
    from geneface_amd.lib import check, current_stream, lib
    from geneface_amd.train_field import GfFieldWgrad
    g = torch.Generator(device=DEV).manual_seed(11 + M)
    f = lambda *s: torch.randn(*s, device=DEV, generator=g) * 0.5
    h = (lambda *s: f(*s).half()) if tier == "f16" else f
    fn, ws_bytes = (lib().gf_field_wgrad16, lib().gf_field_wgrad16_ws_bytes) if tier == "f16" else (lib().gf_field_wgrad32, lib().gf_field_wgrad32_ws_bytes)
    t = {"f3": h(M, 32), "ha1": h(M, 128), "ha2": h(M, 128), "f2": h(M, 32), "hs1": h(M, 128), "hs2": h(M, 128), "geo": h(M, 128), "hc1": h(M, 128),
         "sh": h(M, 16), "g_hc1": h(M, 128), "g_geo": h(M, 128), "g_hs2": h(M, 128), "g_hs1": h(M, 128), "g_ha2": h(M, 128), "g_ha1": h(M, 128),
         "g_zc": f(M, 3), "g_h0": f(M), "g_za": f(M, 2)}
    shapes = {"gw_color1": (3, 128), "gw_color0": (128, 148), "gw_sigma2": (129, 128), "gw_sigma1": (128, 128), "gw_sigma0": (128, 64),
              "gw_ambient2": (2, 128), "gw_ambient1": (128, 128), "gw_ambient0": (128, 96)}
    ws = torch.empty(ws_bytes() // 4, device=DEV)
    runs = []
    for fill in (7.0, -3.0):
        out = {n: torch.full(s, fill, device=DEV) for n, s in shapes.items()}
        ws.fill_(float("nan"))
        wg = GfFieldWgrad(**{n: v.data_ptr() for n, v in {**t, **out}.items()}, ld_color0=148, ld_ambient0=96, workspace=ws.data_ptr())
        check(fn(M, C.byref(wg), current_stream(torch.device(DEV))))
        torch.cuda.synchronize()
        assert bool((out["gw_color0"][:, 144:] == fill).all()) and bool((out["gw_ambient0"][:, 32:] == fill).all())
        runs.append(out)
    for n in shapes:
        assert torch.equal(runs[0][n][:, :144] if n == "gw_color0" else runs[0][n][:, :32] if n == "gw_ambient0" else runs[0][n],
                           runs[1][n][:, :144] if n == "gw_color0" else runs[1][n][:, :32] if n == "gw_ambient0" else runs[1][n]), n
    # the skinny gradients enter the f16 kernel as binary16
    d = {n: v.double() if (v.dtype == torch.float16 or tier == "f32") else v.half().double() for n, v in t.items()}
    tn = lambda a, b: a.t() @ b
    want = {"gw_color1": tn(d["g_zc"], d["hc1"]), "gw_color0": torch.cat([tn(d["g_hc1"], d["sh"]), tn(d["g_hc1"], d["geo"])], 1),
            "gw_sigma2": torch.cat([tn(d["g_h0"][:, None], d["hs2"]), tn(d["g_geo"], d["hs2"])], 0), "gw_sigma1": tn(d["g_hs2"], d["hs1"]),
            "gw_sigma0": torch.cat([tn(d["g_hs1"], d["f3"]), tn(d["g_hs1"], d["f2"])], 1), "gw_ambient2": tn(d["g_za"], d["ha2"]),
            "gw_ambient1": tn(d["g_ha2"], d["ha1"]), "gw_ambient0": tn(d["g_ha1"], d["f3"])}
    for n, w in want.items():
        got = runs[0][n][:, :w.shape[1]].double()
        err = float((got - w).norm() / w.norm().clamp(min=1e-30))
        assert err < 1e-5, (n, err)
        assert float((got - w).abs().max()) < 1e-4 * max(1.0, float(w.abs().max())), n


def test_amp_node_fused_weight_gradients_vs_library_products():
    """The AMP node with the fused weight-gradient kernel (model.amp_wgrad = "fused", the default) against the same node over the batched
    library products of stage 2 ("gemm"): same operands, so every parameter gradient agrees to the rounding of the library path's binary16
    partial products (it rounds each 4096-row slab's product to half before the fp32 sum; the fused kernel never leaves fp32)."""
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.train_field import head_field
    hp, sd = model_fixture(False)
    model = RADNeRF(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    g = torch.Generator(device=DEV).manual_seed(5)
    M = 20000 + 13
    xyz = torch.rand(M, 3, device=DEV, generator=g) * 1.6 - 0.8
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=DEV, generator=g), dim=-1)
    cond = torch.randn(64, device=DEV, generator=g) * 0.3
    code = model.individual_embeddings[0]
    ws, wc, wa = torch.rand(M, device=DEV, generator=g), torch.rand(M, 3, device=DEV, generator=g), torch.rand(M, 2, device=DEV, generator=g)
    res = {}
    for impl in ("gemm", "fused"):
        model.amp_wgrad = impl
        model.zero_grad(set_to_none=True)
        cf = cond.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, rgb, amb = head_field(model, xyz, dirs, cf, code)
        ((torch.log1p(sigma) * ws).sum() + (rgb * wc).sum() + (amb * wa).sum()).backward()
        res[impl] = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
        res[impl]["cond"] = cf.grad.detach().float().clone()
    del model.amp_wgrad
    assert set(res["gemm"]) == set(res["fused"])
    for n, gr in res["gemm"].items():
        l2 = float((res["fused"][n] - gr).double().norm() / gr.double().norm().clamp(min=1e-20))
        assert l2 < (1e-6 if ("embedder" in n or n in ("cond", "individual_embeddings")) else 2e-3), (n, l2)


def test_fp32_node_fused_weight_gradients_vs_library_products():
    """The exact node with the fused weight-gradient kernel (gf_field_wgrad32, model.wgrad_impl = "fused", the default) against the same node
    over the batched library products ("gemm"): the same fp32 operands, two summation orders."""
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.train_field import head_field
    hp, sd = model_fixture(False)
    model = RADNeRF(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    g = torch.Generator(device=DEV).manual_seed(6)
    M = 70000 + 5
    xyz = torch.rand(M, 3, device=DEV, generator=g) * 1.6 - 0.8
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=DEV, generator=g), dim=-1)
    cond = torch.randn(64, device=DEV, generator=g) * 0.3
    code = model.individual_embeddings[0]
    ws, wc, wa = torch.rand(M, device=DEV, generator=g), torch.rand(M, 3, device=DEV, generator=g), torch.rand(M, 2, device=DEV, generator=g)
    res = {}
    for impl in ("gemm", "fused", "fused"):
        model.wgrad_impl = impl
        model.zero_grad(set_to_none=True)
        cf = cond.clone().requires_grad_(True)
        sigma, rgb, amb = head_field(model, xyz, dirs, cf, code)
        ((torch.log1p(sigma) * ws).sum() + (rgb * wc).sum() + (amb * wa).sum()).backward()
        cur = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        cur["cond"] = cf.grad.detach().clone()
        if impl in res:                     # the fused kernel reproduces itself bit for bit
            for n in ("ambient_net.net.0.weight", "ambient_net.net.1.weight", "ambient_net.net.2.weight", "sigma_net.net.0.weight", "sigma_net.net.1.weight",
                      "sigma_net.net.2.weight", "color_net.net.0.weight", "color_net.net.1.weight"):
                own = 32 if n == "ambient_net.net.0.weight" else 144 if n == "color_net.net.0.weight" else cur[n].shape[1]   # beyond: outer products
                assert torch.equal(cur[n][:, :own], res[impl][n][:, :own]), n                                             # with atomic column sums
        res[impl] = cur
    del model.wgrad_impl
    assert set(res["gemm"]) == set(res["fused"])
    for n, gr in res["gemm"].items():
        l2 = float((res["fused"][n] - gr).double().norm() / gr.double().norm().clamp(min=1e-20))
        assert l2 < 2e-5, (n, l2)


@pytest.mark.parametrize("D,gridtype,interp,B", [(3, 1, 0, 300_001), (2, 1, 0, 300_001), (3, 0, 0, 70_001), (2, 0, 1, 70_001), (3, 1, 1, 5_000)])
def test_binned_table_scatter_is_the_unbinned_one(D, gridtype, interp, B):
    """gf_grid_encode_backward_binned (k_grid_bin in front of the scatter: each row partition reads its own list of points) against
    gf_grid_encode_backward_scaled on the same inputs: the same contributions in integer partials over other groupings of the points: the two
    table gradients agree to fp32 rounding of the flush, and both agree with a float64 scatter.  Tiled and hashed tables of the May size, points out of range, zero-gradient points (terminated
    samples), a level with an all-zero gradient."""
    import ctypes as C
    from geneface_amd.encoders.gridencoder import grid_offsets
    from geneface_amd.lib import check, current_stream, lib
    L_ = lib()
    g = torch.Generator(device=DEV).manual_seed(100 * D + B % 97)
    off_h = grid_offsets(D, 16, 16, 16, 2048)
    off = torch.from_numpy(off_h).to(DEV)
    x = torch.rand(B, D, device=DEV, generator=g)
    x[::501] = 1.25                                         # out of range
    x[7::733] = float("nan")
    grad = torch.randn(16, B, 2, device=DEV, generator=g)
    grad[:, 3::5] = 0.0                                     # samples behind a ray's end carry exact zeros at every level
    grad[5] = 0.0                                           # a level nobody contributes to
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    lmax = grad.abs().amax(dim=(1, 2)).contiguous().view(torch.int32)
    st = current_stream(torch.device(DEV))
    ws = torch.empty(L_.gf_grid_backward_ws_bytes(B, 16), dtype=torch.uint8, device=DEV)
    tabs = []
    for binned in (False, True, True):
        t = torch.zeros(int(off_h[-1]), 2, device=DEV)
        ws.fill_(0xAB)
        if binned:
            check(L_.gf_grid_encode_backward_binned(grad.data_ptr(), x.data_ptr(), off.data_ptr(), t.data_ptr(), B, D, 2, 16, S, 16, gridtype, 0, interp,
                                                    lmax.data_ptr(), ws.data_ptr(), st))
        else:
            check(L_.gf_grid_encode_backward_scaled(grad.data_ptr(), x.data_ptr(), off.data_ptr(), t.data_ptr(), B, D, 2, 16, S, 16, gridtype, 0, interp,
                                                    lmax.data_ptr(), st))
        torch.cuda.synchronize()
        tabs.append(t)
    plain, b1, b2 = tabs
    scale = float(plain.abs().max())
    assert scale > 0 and torch.isfinite(plain).all()
    for other in (b1, b2):
        # not bit for bit: which points share a slice's integer partial differs (list order), and every partial reaches the table as a float
        # atomic -- last-ulp differences, as between two runs of either form
        assert float((other - plain).abs().max()) <= 2e-6 * scale
    assert not plain[int(off_h[5]):int(off_h[6])].any() and not b1[int(off_h[5]):int(off_h[6])].any()
    # float64 scatter of the same contributions (the oracle's kernel, double accumulators)
    g64 = torch.zeros(int(off_h[-1]), 2, dtype=torch.float32)
    xc = x.cpu().clone()
    xc[torch.isnan(xc)] = 1.25                              # the product treats NaN as out of range; the restated reference kernel must not see it
    K.gridencoder.grid_encode_backward(grad.cpu().contiguous(), xc.contiguous(), torch.zeros_like(g64), torch.from_numpy(off_h), g64, B, D, 2, 16, S, 16,
                                       None, None, gridtype, False, interp)
    assert float((b1.cpu() - g64).abs().max()) < 2e-4 * max(1.0, float(g64.abs().max()))


@pytest.mark.parametrize("amp", [False, True])
def test_point_lists_beyond_the_kernels_row_addressing_go_through_in_slabs(amp, monkeypatch):
    """The training kernels address their [M,128] rows with 32-bit byte offsets (M < 2^23 fp32, 2^24 binary16); head_field sends a longer list
    through the node in slabs.  With the limit lowered to 7 000: the same outputs and the same gradients as the single call on 20 000 points."""
    import geneface_amd.train_field as TF
    from geneface_amd.radnerf import RADNeRF
    hp, sd = model_fixture(False)
    model = RADNeRF(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    g = torch.Generator(device=DEV).manual_seed(9)
    M = 20000
    xyz = torch.rand(M, 3, device=DEV, generator=g) * 1.6 - 0.8
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device=DEV, generator=g), dim=-1)
    cond = torch.randn(64, device=DEV, generator=g) * 0.3
    code = model.individual_embeddings[0]
    ws, wc, wa = torch.rand(M, device=DEV, generator=g), torch.rand(M, 3, device=DEV, generator=g), torch.rand(M, 2, device=DEV, generator=g)
    res = []
    for limit in (None, 7000):
        if limit:
            monkeypatch.setattr(TF, "_MAX_POINTS_F32", limit)
            monkeypatch.setattr(TF, "_MAX_POINTS_AMP", limit)
        model.zero_grad(set_to_none=True)
        cf = cond.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            sigma, rgb, amb = TF.head_field(model, xyz, dirs, cf, code)
        ((torch.log1p(sigma) * ws).sum() + (rgb * wc).sum() + (amb * wa).sum()).backward()
        res.append(([t.detach().clone() for t in (sigma, rgb, amb)], {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None},
                    cf.grad.detach().float().clone()))
    (o1, g1, c1), (o2, g2, c2) = res
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)                             # a point's outputs do not depend on its neighbours in the list
    assert set(g1) == set(g2) and len(g1) >= 11
    for n in g1:
        l2 = float((g2[n] - g1[n]).double().norm() / g1[n].double().norm().clamp(min=1e-20))
        assert l2 < 1e-5, (n, l2)                            # sums over the points in another grouping
    assert float((c2 - c1).double().norm() / c1.double().norm()) < 1e-5


@pytest.mark.parametrize("config", ["lm3d", "audio"])
def test_condition_encoder_node_vs_torch_modules(config):
    """train_cond.cond_feat_train (gf_cond_train_forward / _backward: AudioNet + AudioAttNet under autograd as two launches) against the
    torch modules on the same weights: the features, and the gradient of every one of the encoder's 24 parameter tensors for an arbitrary
    downstream gradient.  The landmark config (window [5, 1, 204]: the k = 3 convolutions see one tap) and the audio config (window
    [8, 16, 44]: strided convolutions that shrink the window to 1)."""
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd import train_cond
    from geneface_amd.radnerf import RADNeRF
    hp = HP.may_hparams(False) if config == "lm3d" else HP.variant_hparams("audio", False)
    model = RADNeRF(hp)
    model.load_state_dict(S.make_state_dict(hp, False), strict=True)
    model = model.to(DEV).train()
    g = torch.Generator(device=DEV).manual_seed(3)
    S_, T_, C_ = int(hp["smo_win_size"]), int(hp["cond_win_size"]), int(model.cond_prenet.encoder_conv[0].in_channels)
    cond = torch.randn(S_, T_, C_, device=DEV, generator=g) * 0.7
    gout = torch.randn(int(model.cond_prenet.dim_aud), device=DEV, generator=g)
    assert train_cond.supported(model, cond)
    res = {}
    for impl in ("ops", "auto"):
        model.cond_impl = impl
        model.zero_grad(set_to_none=True)
        feat = model.cal_cond_feat(cond)
        assert feat.shape == (int(model.cond_prenet.dim_aud),)
        (feat * gout).sum().backward()
        res[impl] = (feat.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    del model.cond_impl
    (f0, g0), (f1, g1) = res["ops"], res["auto"]
    assert float((f1 - f0).abs().max()) < 1e-5 * max(1.0, float(f0.abs().max()))
    assert set(g0) == set(g1) and len(g1) == 24 and all(n.startswith(("cond_prenet", "cond_att_net")) for n in g1)
    for n in g0:
        err = float((g1[n] - g0[n]).double().norm() / g0[n].double().norm().clamp(min=1e-20))
        assert err < 2e-5, (n, err)
    # the inference-side kernel computes the same features (same operation order: bit for bit)
    from geneface_amd import fused
    with torch.no_grad():
        r = fused.cond_encode_batch(model, fused.get_state(model), cond[None].contiguous())
    assert r is not None and torch.equal(r[0][0], f1)


def test_condition_encoder_node_in_the_training_step():
    """One training step of the head with the encoder as the fused node against the same step over the torch modules: the image and every
    parameter gradient (the encoder's reach it through d cond_feat of the field node)."""
    from geneface_amd.radnerf import RADNeRF
    from test_oracle_train import _loss
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, 40, 40), 2)
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    res = {}
    for impl in ("ops", "auto"):
        model = RADNeRF(hp)
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).train()
        model.cond_impl = impl
        out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
        _loss(out, target).backward()
        res[impl] = (out["rgb_map"].detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    (i0, g0), (i1, g1) = res["ops"], res["auto"]
    assert float((i1 - i0).abs().max()) < 1e-5
    assert set(g0) == set(g1)
    for n in g0:
        err = float((g1[n] - g0[n]).double().norm() / g0[n].double().norm().clamp(min=1e-20))
        # the two encoders agree to the last few ulp of cond_feat (different summation order in the FC layers); behind it sit the ambient
        # coordinate's 2-D hash lookup and exp(): 6e-4 on the 3-D table, 1e-5 .. 1e-4 elsewhere (measured).  The exact comparison of the node
        # is test_condition_encoder_node_vs_torch_modules; this one checks that the step as a whole is the same step.
        assert err < 5e-3, (n, err)


def test_torso_training_branch_dense_vs_compacted():
    """RADNeRFTorso's training branch with the fused torso field on EVERY sampled pixel and the mask applied afterwards (torso_train_dense, round
    6: no compaction, no host sync per step) against the reference's boolean-mask gather / scatter (radnerf_torso.py:174-184) on the same rays:
    the same picture, the same alpha map, and the same gradient for every torso parameter (the sums run over other groupings of the pixels)."""
    from geneface_amd.radnerf_torso import RADNeRFTorso
    from test_oracle_train import _loss
    hp, sd = model_fixture(True)
    fi = frame_inputs(sequence(4, 48, 48), 1)
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    res = {}
    for dense in (False, True, "fused_blend"):
        model = RADNeRFTorso(hp)
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).train()
        model.torso_train_dense = bool(dense)
        model.torso_blend_impl = "fused" if dense == "fused_blend" else "ops"
        for k, p in model.named_parameters():
            p.requires_grad_("torso" in k)
        out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
        alphas = out["torso_alpha_map"].clamp(1e-5, 1 - 1e-5)
        (_loss(out, target) + 1e-3 * torch.mean(-alphas * torch.log2(alphas) - (1 - alphas) * torch.log2(1 - alphas))).backward()
        res[dense] = (out["rgb_map"].detach().clone(), out["torso_alpha_map"].detach().clone(),
                      {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, out.get("deform"))
    # the tail (mask, both blends, clamp) as one node against the torch expressions, both on the dense field: the same operations in the same
    # order -- the picture bit for bit, the gradients to fp32 rounding of another association
    (i1, a1, g1, d1), (i2, a2, g2, d2) = res[True], res["fused_blend"]
    assert torch.equal(i2, i1) and torch.equal(a2, a1) and torch.equal(d2, d1)
    for n in g1:
        err = float((g2[n] - g1[n]).double().norm() / g1[n].double().norm().clamp(min=1e-20))
        assert err < 2e-6, (n, err)
    (i0, a0, g0, d0), (i1, a1, g1, d1) = res[False], res[True]
    masked = a0.reshape(-1) > 0
    assert 0 < int(masked.sum()) < masked.numel()                  # both kinds of pixel are in the frame
    assert float((i1 - i0).abs().max()) < 1e-6 and float((a1 - a0).abs().max()) < 1e-6
    assert d1.shape[0] == masked.numel() and d0.shape[0] == int(masked.sum()) and not d1[~masked].any()
    assert float((d1[masked] - d0).abs().max()) < 1e-6
    assert set(g0) == set(g1) and len(g0) >= 8 and all("torso" in n for n in g0)
    for n in g0:
        err = float((g1[n] - g0[n]).double().norm() / g0[n].double().norm().clamp(min=1e-20))
        assert err < 2e-5, (n, err)


def test_torso_training_step_under_fp16_autocast():
    """The torso task under the Trainer's autocast + GradScaler (base.yaml:49 amp: true applies to both tasks): the torso nodes cast their
    inputs to fp32 (custom_fwd(cast_inputs=float32): the torso field, its weight gradients and the blend run in fp32 whatever the autocast
    state), so a step under autocast gives the fp32 step's picture and gradients, finite at the scaler's scale, and no step is skipped."""
    from geneface_amd.radnerf_torso import RADNeRFTorso
    from test_oracle_train import _loss
    hp, sd = model_fixture(True)
    fi = frame_inputs(sequence(4, 48, 48), 1)
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    res = {}
    for amp in (False, True):
        model = RADNeRFTorso(hp)
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).train()
        for k, p in model.named_parameters():
            p.requires_grad_("torso" in k)
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-4)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, enabled=amp)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
            loss = _loss(out, target)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
        assert all(torch.isfinite(g).all() for g in grads.values())
        scaler.step(opt)
        scaler.update()
        if amp:
            assert scaler.get_scale() >= 1024.0
        res[amp] = (out["rgb_map"].detach().float().clone(), grads)
    (i0, g0), (i1, g1) = res[False], res[True]
    # the frozen head's image is rendered under no_grad on the fp32 field either way; what autocast touches is torch glue around the nodes
    assert float((i1 - i0).abs().max()) < 2e-3
    assert set(g0) == set(g1) and len(g0) >= 8
    for n in g0:
        err = float((g1[n] - g0[n]).double().norm() / g0[n].double().norm().clamp(min=1e-20))
        assert err < 2e-2, (n, err)


@pytest.mark.parametrize("amp", [False, True])
@pytest.mark.parametrize("force_all_rays", [True, False])
def test_training_step_on_an_empty_occupancy_grid(force_all_rays, amp):
    """Edge case of the training branch: no ray meets an occupied cell (a bitfield of zeros -- e.g. the first steps after `reset_extra_state`,
    or a crop that looks past the head).  The point list is empty (or all padding), the picture is the background, every gradient is finite and
    zero where nothing was sampled, and the optimizer step goes through (no skipped step under the scaler)."""
    from geneface_amd.radnerf import RADNeRF
    from test_oracle_train import _loss
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, 40, 40), 2)
    model = RADNeRF(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    model.density_bitfield.zero_()
    model.density_grid.zero_()
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, enabled=amp)
    for step in range(3):         # the second and third call run on the step counter's running mean of zero samples
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=step > 0, force_all_rays=force_all_rays, **hp)
            loss = _loss(out, target)
        assert torch.equal(out["rgb_map"].float().reshape(-1, 3), to(fi["bg"]).reshape(-1, 3).clamp(0, 1))
        assert float(out["weights_sum"].abs().max()) == 0.0
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        for n, p in model.named_parameters():
            if p.grad is not None:
                assert torch.isfinite(p.grad).all(), n
                assert float(p.grad.abs().max()) == 0.0, n
    assert scaler.get_scale() >= 1024.0 or not amp


@pytest.mark.parametrize("dense", [True, False])
def test_torso_training_step_with_an_empty_torso_mask(dense):
    """The torso task when no sampled pixel lies on the torso (a zero torso occupancy: `mask.any()` is false, radnerf_torso.py:174): the picture
    is head over background, every torso gradient is finite and zero, on the dense branch and on the compacting one."""
    from geneface_amd.radnerf_torso import RADNeRFTorso
    from test_oracle_train import _loss
    hp, sd = model_fixture(True)
    fi = frame_inputs(sequence(4, 40, 40), 1)
    model = RADNeRFTorso(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    model.torso_train_dense = dense
    model.density_grid_torso.zero_()
    model.mean_density_torso = 0
    for k, p in model.named_parameters():
        p.requires_grad_("torso" in k)
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
    assert float(out["torso_alpha_map"].abs().max()) == 0.0
    loss = _loss(out, target)
    if loss.requires_grad:
        loss.backward()
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all() and float(p.grad.abs().max()) == 0.0, n
    # the picture is the head over the plain background: the torso contributes nothing anywhere
    from geneface_amd.radnerf import RADNeRF
    head = RADNeRF(model_fixture(False)[0])
    head.load_state_dict({k: v for k, v in sd.items() if k in head.state_dict()}, strict=True)
    head = head.to(DEV).train()
    with torch.no_grad():
        want = head.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
    assert (want["rgb_map"] - out["rgb_map"].detach()).abs().max() < 2e-6


def test_grad_scaler_recovers_from_an_overflowing_scale_on_the_f16_tier():
    """The GradScaler contract end to end on the AMP tier (utils/commons/trainer.py:307-382): with a scale far too large the f16 chain's gradients
    leave the binary16 range, the parameter gradients the scaler inspects are non-finite, the step is SKIPPED (weights untouched) and the scale
    halves; after enough halvings the gradients are finite, the step is taken, and no weight ever holds a non-finite value."""
    from geneface_amd.radnerf import RADNeRF
    from test_oracle_train import _loss
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, 40, 40), 2)
    model = RADNeRF(hp)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).train()
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40, backoff_factor=1.0 / 64, growth_interval=1000)
    snapshot = lambda: torch.cat([p.detach().reshape(-1).float() for p in model.parameters()])
    skipped = taken = 0
    for step in range(12):
        before, scale = snapshot(), scaler.get_scale()
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
            loss = _loss(out, target)
        assert model._last_field_node == "amp_f16"
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        after = snapshot()
        assert torch.isfinite(after).all()
        if scaler.get_scale() < scale:                    # the scaler saw a non-finite gradient: nothing may have moved
            assert torch.equal(before, after)
            skipped += 1
        else:
            assert not torch.equal(before, after)
            taken += 1
    assert skipped >= 1 and taken >= 1, (skipped, taken, scaler.get_scale())


def test_training_step_under_bfloat16_autocast_runs_the_exact_node():
    """torch.autocast(bfloat16) is not what the reference trains with (torch.cuda.amp.autocast: float16), but nothing forbids it: there is no
    bfloat16 tier, so the field takes the exact fp32 node (as `amp_field = "f32"` does under float16), every HIP op computes in fp32, and the
    gradients are those of the fp32 step up to what torch's own bfloat16 glue (the loss arithmetic) does to them."""
    from geneface_amd.radnerf import RADNeRF
    from test_oracle_train import _loss
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, 40, 40), 2)
    to = lambda t: t.to(DEV)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=torch.Generator().manual_seed(8)).to(DEV)
    res = {}
    for bf in (False, True):
        model = RADNeRF(hp)
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).train()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
            out = model.render(*args, index=0, staged=False, bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
        assert model._last_field_node == "f32" and out["rgb_map"].dtype == torch.float32
        _loss(out, target).backward()                                       # the loss itself in fp32 on both sides: what is compared is the product
        res[bf] = (out["rgb_map"].detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    assert torch.equal(res[False][0], res[True][0])
    assert set(res[False][1]) == set(res[True][1]) and len(res[True][1]) >= 20
    for n, g in res[False][1].items():
        err = float((res[True][1][n] - g).double().norm() / g.double().norm().clamp(min=1e-30))
        assert err < 5e-4, (n, err)      # two runs of the same fp32 step differ by this much on the attention net (the few fp32 atomic sums left; cf. test_gpu_ddp.py)
