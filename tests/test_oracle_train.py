"""CPU: the oracle's training-tier ray-marching kernels against independent statements of the same maths:
torch autograd of a differentiable restatement for the compositor's backward, the inference marcher for the training marcher."""
import numpy as np
import torch

from helpers import frame_inputs, model_fixture, sequence
from oracle import kernels as K
from oracle import radnerf_ref as R

RM = K.raymarching_face


def _scene(size=32, idx=1):
    hp, sd = model_fixture(False)
    fi = frame_inputs(sequence(4, size, size), idx)
    ro, rd = fi["rays_o"].view(-1, 3).contiguous(), fi["rays_d"].view(-1, 3).contiguous()
    nears, fars = R.near_far_from_aabb(ro, rd, sd["aabb_infer"], hp["min_near"])
    return hp, sd, ro, rd, nears, fars


def march_train(hp, sd, ro, rd, nears, fars, M=None, noises=None, max_steps=None):
    N = ro.shape[0]
    max_steps = max_steps or hp["max_steps"]
    M = N * max_steps if M is None else M
    xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
    rays, counter = torch.empty(N, 3, dtype=torch.int32), torch.zeros(2, dtype=torch.int32)
    noises = torch.zeros(N) if noises is None else noises
    RM.march_rays_train(ro, rd, sd["density_bitfield"], float(hp["bound"]), hp["dt_gamma"], max_steps, N, 1, hp["grid_size"], M, nears, fars,
                        xyzs, dirs, deltas, rays, counter, noises)
    return xyzs, dirs, deltas, rays, counter


def test_train_marcher_matches_inference_marcher():
    hp, sd, ro, rd, nears, fars = _scene()
    N, ms = ro.shape[0], hp["max_steps"]
    xyzs, dirs, deltas, rays, counter = march_train(hp, sd, ro, rd, nears, fars)
    assert counter[1].item() == N and counter[0].item() == rays[:, 2].sum().item() > 0
    assert torch.equal(rays[:, 0], torch.arange(N, dtype=torch.int32))
    assert torch.equal(rays[:, 1].long(), torch.cumsum(rays[:, 2].long(), 0) - rays[:, 2].long())     # offsets = exclusive prefix sum
    # the inference marcher asked for max_steps samples from `near` yields the same samples ray by ray
    x2, d2, de2 = R.march_rays(N, ms, torch.arange(N, dtype=torch.int32), nears.clone(), ro, rd, float(hp["bound"]), sd["density_bitfield"], 1,
                               hp["grid_size"], nears, fars, -1, hp["dt_gamma"], ms)
    cnt2 = (de2[:, 0].view(N, ms) > 0).sum(1)
    assert torch.equal(cnt2.int(), rays[:, 2])
    for n in torch.nonzero(rays[:, 2] > 0).flatten()[:200].tolist():
        o, c = rays[n, 1].item(), rays[n, 2].item()
        assert torch.equal(xyzs[o:o + c], x2.view(N, ms, 3)[n, :c]) and torch.equal(deltas[o:o + c], de2.view(N, ms, 2)[n, :c])
        assert torch.equal(dirs[o:o + c], rd[n].expand(c, 3))
    # points beyond the total stay zero (the caller pre-zeroes), every point inside is a sample
    tot = counter[0].item()
    assert (deltas[:tot, 0] > 0).all() and not deltas[tot:].any()


def test_train_marcher_overflow_and_noise():
    hp, sd, ro, rd, nears, fars = _scene()
    full = march_train(hp, sd, ro, rd, nears, fars)
    tot = full[4][0].item()
    M = tot // 2
    xyzs, dirs, deltas, rays, counter = march_train(hp, sd, ro, rd, nears, fars, M=M)
    assert counter[0].item() == tot and torch.equal(rays, full[3])          # counting is unaffected by the buffer size
    fits = (rays[:, 1] + rays[:, 2]).long() <= M
    last = int(torch.nonzero(fits & (rays[:, 2] > 0)).flatten()[-1])
    end = (rays[last, 1] + rays[last, 2]).item()
    assert torch.equal(xyzs[:end], full[0][:end]) and not xyzs[end:].any()   # rays that do not fit write nothing (raymarching.cu:459)
    # perturbation shifts the start by noise * dt: sample counts may change, structure stays valid
    g = torch.Generator().manual_seed(0)
    pert = march_train(hp, sd, ro, rd, nears, fars, noises=torch.rand(ro.shape[0], generator=g))
    assert pert[4][0].item() > 0 and not torch.equal(pert[2][:100], full[2][:100])


def _torch_composite(sigmas, rgbs, ambient, deltas, rays, N):
    """differentiable restatement (no early termination): w_i = alpha_i * prod_{j<i} (1 - alpha_j)"""
    ws, amb, dep, img = [], [], [], []
    for n in range(N):
        o, c = rays[n, 1].item(), rays[n, 2].item()
        s, col, dt, t = sigmas[o:o + c], rgbs[o:o + c], deltas[o:o + c, 0], deltas[o:o + c, 1]
        alpha = 1 - torch.exp(-s * dt)
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=alpha.dtype), 1 - alpha[:-1]]), 0) if c else alpha
        w = alpha * T
        ws.append(w.sum()); amb.append(ambient[o:o + c].sum()); dep.append((w * t).sum()); img.append((w[:, None] * col).sum(0))
    return torch.stack(ws), torch.stack(amb), torch.stack(dep), torch.stack(img)


def test_composite_train_forward_backward_vs_autograd():
    hp, sd, ro, rd, nears, fars = _scene(24)
    xyzs, dirs, deltas, rays, counter = march_train(hp, sd, ro, rd, nears, fars)
    N, M = ro.shape[0], counter[0].item()
    g = torch.Generator().manual_seed(1)
    sigmas = (torch.rand(M, generator=g) * 30).double().requires_grad_(True)
    rgbs = torch.rand(M, 3, generator=g).double().requires_grad_(True)
    ambient = torch.rand(M, generator=g).double().requires_grad_(True)
    ws_t, amb_t, dep_t, img_t = _torch_composite(sigmas, rgbs, ambient, deltas[:M].double(), rays, N)
    gws, gamb, gimg = torch.rand(N, generator=g), torch.rand(N, generator=g), torch.rand(N, 3, generator=g)
    loss = (ws_t * gws.double()).sum() + (amb_t * gamb.double()).sum() + (img_t * gimg.double()).sum()
    loss.backward()
    s32, c32, a32 = sigmas.detach().float(), rgbs.detach().float(), ambient.detach().float()
    ws, amb, dep, img = torch.empty(N), torch.empty(N), torch.empty(N), torch.empty(N, 3)
    RM.composite_rays_train_forward(s32, c32, a32, deltas[:M].contiguous(), rays, M, N, 0.0, ws, amb, dep, img)
    assert (ws - ws_t.float()).abs().max() < 1e-5 and (img - img_t.float()).abs().max() < 1e-5
    assert (dep - dep_t.float()).abs().max() < 1e-4 and (amb - amb_t.float()).abs().max() < 1e-4
    gs, gc, ga = torch.zeros(M), torch.zeros(M, 3), torch.zeros(M)
    RM.composite_rays_train_backward(gws, gamb, gimg, s32, c32, a32, deltas[:M].contiguous(), rays, ws, amb, img, M, N, 0.0, gs, gc, ga)
    assert (gc - rgbs.grad.float()).abs().max() < 1e-5
    assert (ga - ambient.grad.float()).abs().max() < 1e-6
    assert (gs - sigmas.grad.float()).abs().max() < 2e-5 * max(1.0, float(sigmas.grad.abs().max()))
    # early termination: once T < T_thresh the later samples receive neither weight nor gradient
    ws2, amb2, dep2, img2 = torch.empty(N), torch.empty(N), torch.empty(N), torch.empty(N, 3)
    RM.composite_rays_train_forward(s32, c32, a32, deltas[:M].contiguous(), rays, M, N, 0.5, ws2, amb2, dep2, img2)
    assert (ws2 <= ws + 1e-6).all() and (ws2 < ws - 1e-3).any()
    gs2, gc2, ga2 = torch.zeros(M), torch.zeros(M, 3), torch.zeros(M)
    RM.composite_rays_train_backward(gws, gamb, gimg, s32, c32, a32, deltas[:M].contiguous(), rays, ws2, amb2, img2, M, N, 0.5, gs2, gc2, ga2)
    assert (gc2.abs().sum(1) == 0).sum() > (gc.abs().sum(1) == 0).sum()


def test_march_train_backward_is_the_chain_rule_of_o_plus_t_d():
    """xyz = o + t d, dir = d  =>  d/do = sum g_xyz, d/dd = sum (g_xyz * t + g_dir); the reference evaluates t as deltas[:, 1]."""
    hp, sd, ro, rd, nears, fars = _scene(16)
    xyzs, dirs, deltas, rays, counter = march_train(hp, sd, ro, rd, nears, fars)
    N, M = ro.shape[0], counter[0].item()
    g = torch.Generator().manual_seed(2)
    gx, gd = torch.randn(M, 3, generator=g), torch.randn(M, 3, generator=g)
    go, gdd = torch.zeros(N, 3), torch.zeros(N, 3)
    RM.march_rays_train_backward(gx, gd, rays, deltas[:M].contiguous(), N, M, go, gdd)
    for n in torch.nonzero(rays[:, 2] > 0).flatten()[:50].tolist():
        o, c = rays[n, 1].item(), rays[n, 2].item()
        np.testing.assert_allclose(go[n].numpy(), gx[o:o + c].sum(0).numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gdd[n].numpy(), (gx[o:o + c] * deltas[o:o + c, 1:2] + gd[o:o + c]).sum(0).numpy(), rtol=1e-5, atol=1e-5)
    assert not go[rays[:, 2] == 0].any()


# ----------------------------------------------------------------------------------------------- encoders, backward
def _grid_setup(D, gridtype, interp, B=4000, seed=0):
    from geneface_amd.encoders.gridencoder import grid_offsets, per_level_scale_for
    off = grid_offsets(D, 16, 16, 12, 512)                     # small tables: every row is hit
    S = float(np.log2(per_level_scale_for(512, 16, 16)))
    g = torch.Generator().manual_seed(seed)
    emb = torch.rand(int(off[-1]), 2, generator=g) * 2 - 1
    x = torch.rand(B, D, generator=g)
    x[:7] = torch.tensor([1.5] + [0.5] * (D - 1))              # out-of-range points contribute nothing
    return torch.from_numpy(off), S, emb, x, g


def _grid_fwd(x, emb, off, S, D, gridtype, interp, dy_dx=None):
    B = x.shape[0]
    out = torch.empty(16, B, 2)
    K.gridencoder.grid_encode_forward(x, emb, off, out, B, D, 2, 16, S, 16, dy_dx, gridtype, False, interp)
    return out


import pytest


@pytest.mark.parametrize("D,gridtype,interp", [(3, 1, 0), (2, 1, 0), (3, 0, 0), (2, 0, 1), (3, 1, 1)])
def test_grid_backward_is_the_adjoint_of_the_forward(D, gridtype, interp):
    """The lookup is linear in the table: <grad, F(E + dE) - F(E)> == <grad_E, dE>; and grad_x = J^T grad with J = dy_dx, which for
    linear interpolation is also the finite difference inside a cell."""
    off, S, emb, x, g = _grid_setup(D, gridtype, interp)
    B = x.shape[0]
    grad = torch.randn(16, B, 2, generator=g)
    dy_dx = torch.empty(B, 16 * D * 2)
    out = _grid_fwd(x, emb, off, S, D, gridtype, interp, dy_dx)
    g_emb, g_in = torch.zeros_like(emb), torch.zeros(B, D)
    K.gridencoder.grid_encode_backward(grad, x, emb, off, g_emb, B, D, 2, 16, S, 16, dy_dx, g_in, gridtype, False, interp)
    d_emb = torch.randn(emb.shape, generator=g)
    lhs = ((_grid_fwd(x, emb + d_emb, off, S, D, gridtype, interp).double() - out.double()) * grad.double()).sum()
    rhs = (g_emb.double() * d_emb.double()).sum()
    assert abs(lhs - rhs) < 2e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    assert not g_in[:7].any()
    want = torch.einsum("lbc,bldc->bd", grad.double(), dy_dx.view(B, 16, D, 2).double())
    assert (g_in.double() - want).abs().max() < 1e-3 * max(1.0, float(want.abs().max()))
    if interp == 0:   # piecewise linear: a central difference that stays inside the finest cell is exact up to rounding
        eps = 2e-5
        sel = slice(7, 400)
        for d in range(D):
            xp, xm = x.clone(), x.clone()
            xp[:, d] += eps; xm[:, d] -= eps
            fd = (_grid_fwd(xp.clamp(0, 1), emb, off, S, D, gridtype, interp).double() - _grid_fwd(xm.clamp(0, 1), emb, off, S, D, gridtype, interp).double()) / (2 * eps)
            jac = dy_dx.view(B, 16, D, 2)[:, :, d].permute(1, 0, 2).double()
            # points whose +-eps neighbourhood crosses a cell face at some level are not differentiable there: compare the median
            err = (fd[:, sel] - jac[:, sel]).abs().flatten()
            assert err.median() < 5e-2 * max(1.0, float(jac.abs().median()))


def test_sh_and_freq_backward_vs_autograd():
    g = torch.Generator().manual_seed(6)
    B = 300
    x = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    out, dy = torch.empty(B, 16), torch.empty(B, 3, 16)
    K.shencoder.sh_encode_forward(x, out, B, 3, 4, dy)
    xd = x.double().requires_grad_(True)
    X, Y, Z = xd[:, 0], xd[:, 1], xd[:, 2]
    sh = torch.stack([torch.full_like(X, 0.28209479177387814), -0.48860251190291987 * Y, 0.48860251190291987 * Z, -0.48860251190291987 * X,
                      1.0925484305920792 * X * Y, -1.0925484305920792 * Y * Z, 0.94617469575755997 * Z * Z - 0.31539156525251999,
                      -1.0925484305920792 * X * Z, 0.54627421529603959 * (X * X - Y * Y), 0.59004358992664352 * Y * (-3 * X * X + Y * Y),
                      2.8906114426405538 * X * Y * Z, 0.45704579946446572 * Y * (1 - 5 * Z * Z), 0.3731763325901154 * Z * (5 * Z * Z - 3),
                      0.45704579946446572 * X * (1 - 5 * Z * Z), 1.4453057213202769 * Z * (X * X - Y * Y),
                      0.59004358992664352 * X * (-X * X + 3 * Y * Y)], dim=1)
    assert (sh.float() - out).abs().max() < 1e-6
    grad = torch.randn(B, 16, generator=g)
    (sh * grad.double()).sum().backward()
    gi = torch.zeros(B, 3)
    K.shencoder.sh_encode_backward(grad, x, B, 3, 4, dy, gi)
    assert (gi - xd.grad.float()).abs().max() < 1e-5
    # frequency encoding: [x, sin(2^f x), cos(2^f x)]
    D, deg = 6, 4
    Cc = D + 2 * D * deg
    xin = torch.randn(B, D, generator=g)
    fo = torch.empty(B, Cc)
    K.freqencoder.freq_encode_forward(xin, B, D, deg, Cc, fo)
    xq = xin.double().requires_grad_(True)
    parts = [xq] + [f(xq * 2.0 ** k) for k in range(deg) for f in (torch.sin, torch.cos)]
    enc = torch.cat(parts, dim=1)
    assert (enc.float() - fo).abs().max() < 1e-5
    gr = torch.randn(B, Cc, generator=g)
    (enc * gr.double()).sum().backward()
    gin = torch.zeros(B, D)
    K.freqencoder.freq_encode_backward(gr, fo, B, D, deg, Cc, gin)
    assert (gin - xq.grad.float()).abs().max() < 2e-4


# ----------------------------------------------------------------------------------------------- training branch
def _train_setup(torso, size=20, idx=1):
    hp, sd = model_fixture(torso)
    fi = frame_inputs(sequence(4, size, size), idx)
    return hp, sd, fi


def _loss(out, target):
    return ((out["rgb_map"] - target) ** 2).mean() + 1e-3 * out["weights_sum"].mean() + 1e-4 * out["ambient"].mean()


@pytest.mark.parametrize("torso", [False, True])
def test_render_train_gradients_match_finite_differences(torso):
    """The oracle's training branch (march_rays_train -> field -> composite_rays_train through custom autograd over the C kernels):
    directional derivatives of the loss along random directions of a few weight tensors against central differences."""
    hp, sd, fi = _train_setup(torso)
    names = (["torso_canonicial_net.net.2.weight", "torso_deform_net.net.0.weight", "torso_embedder.embeddings"] if torso else
             ["color_net.net.1.weight", "sigma_net.net.1.weight", "ambient_net.net.2.weight", "position_embedder.embeddings"])
    g = torch.Generator().manual_seed(7)
    target = torch.rand(1, fi["rays_o"].shape[1], 3, generator=g)

    def run(sd_):
        return R.render_train(sd_, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso)

    sd_g = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    out = run(sd_g)
    assert out["n_points"] > 0
    loss = _loss(out, target)
    loss.backward()
    for k in names:
        grad = sd_g[k].grad
        assert grad is not None and torch.isfinite(grad).all() and grad.abs().max() > 0, k
        direction = torch.randn(sd[k].shape, generator=g)
        direction /= direction.norm()
        eps = 2e-3 * max(1.0, float(sd[k].abs().mean())) if "embeddings" not in k else 1e-2
        lp = _loss(run({**sd, k: sd[k] + eps * direction}), target).item()
        lm = _loss(run({**sd, k: sd[k] - eps * direction}), target).item()
        fd, an = (lp - lm) / (2 * eps), float((grad * direction).sum())
        assert abs(fd - an) <= 0.08 * max(abs(fd), abs(an)) + 2e-6, (k, fd, an)
