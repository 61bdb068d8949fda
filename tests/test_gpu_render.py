"""-m gpu: whole frames through the reference-compatible module API against the oracle and the golden frames."""
import os

import numpy as np
import pytest
import torch

from helpers import frame_inputs, model_fixture, oracle_u8, pipe_inputs, psnr, sequence
from oracle import radnerf_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"

# fp32 everywhere; what differs from the oracle is summation order inside the MLPs (MFMA / rocBLAS vs
# MKL) and __expf, amplified by exp(h) of the density head.  Strict tolerance of BASELINE.md section 4,
# widened where the amplification is visible, plus a PSNR floor far above the 40 dB "fast" bar.
RGB_ATOL = 1e-4      # BASELINE.md section 4, strict tier: max|d rgb| <= 1e-4
PSNR_MIN = 65.0


def build(torso, impl):
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp, sd = model_fixture(torso)
    m = (RADNeRFTorso if torso else RADNeRF)(hp)
    m.load_state_dict(sd, strict=True)
    m.render_impl = impl
    return hp, sd, m.to(DEV).eval()


def render_gpu(model, hp, fi):
    to = lambda t: t.to(DEV)
    return model.render(to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]), index=0, staged=False,
                        bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)


def check(out, ref, torso):
    rgb, rgb_ref = out["rgb_map"].cpu(), ref["rgb_map"] if torch.is_tensor(ref["rgb_map"]) else torch.from_numpy(ref["rgb_map"])
    assert rgb.shape == rgb_ref.shape
    err = (rgb - rgb_ref).abs().max().item()
    assert err < RGB_ATOL, err
    assert psnr(rgb, rgb_ref) > PSNR_MIN
    u8 = (rgb * 255).to(torch.uint8).int() - (rgb_ref * 255).to(torch.uint8).int()
    assert (u8.abs() <= 1).float().mean().item() > 0.999
    dref = ref["depth_map"] if torch.is_tensor(ref["depth_map"]) else torch.from_numpy(ref["depth_map"])
    assert (out["depth_map"].cpu() - dref).abs().max().item() < 2e-4      # measured 4e-5
    if torso:
        for k, tol in (("torso_alpha_map", 2e-5), ("torso_rgb_map", 2e-5)):
            r = ref[k] if torch.is_tensor(ref[k]) else torch.from_numpy(ref[k])
            assert (out[k].cpu() - r).abs().max().item() < tol, k


def check_u8(frame, ref8, rerender=None):
    """Bar for uint8 frames whose rays were generated inside the kernel: >= 99.9 % of the bytes identical, nothing off by more than 1 LSB,
    PSNR >= 55 dB.  In-kernel rays can differ from torch's get_rays in the last ulp (the reference builds them with a matmul whose summation
    order is the BLAS library's); a ray that grazes an occupied cell of the density grid within that ulp gains or loses a sample, and its
    pixel changes by a visible amount (0-1 such pixels per frame).  No pixel is excused on that suspicion: when any is off by more than
    1 LSB, `rerender()` must return the reference frame rendered FROM THE KERNEL'S OWN INPUT BITS (helpers.pipe_inputs: rays from gf_pinhole_rays,
    the device function k_frame_init runs, and the device's background coordinates / euler pose) and the whole frame must then agree within
    1 LSB.  (Until round 4 up to 4 pixels per frame were excused as "grazing rays"; at 128 x 128 the three excused pixels were in fact torso-mask
    pixels -- `occ > 0` flips where the CPU's and the GPU's get_bg_coords differ in the last ulp -- i.e. a harness that fed the two sides
    different bits, not a property of the kernel.)"""
    frame, ref8 = torch.as_tensor(frame).reshape(-1, 3), torch.as_tensor(ref8).reshape(-1, 3)
    diff = (frame.int() - ref8.int()).abs()
    off = (diff > 1).any(dim=1)
    n_off = int(off.sum())
    if n_off:
        assert rerender is not None, f"{n_off} pixels off by more than 1 LSB and no arbitration on the kernel's own rays"
        assert n_off <= 8, n_off           # grazing rays are rare; more than a handful is something else
        ref8 = torch.as_tensor(rerender()).reshape(-1, 3)
        diff = (frame.int() - ref8.int()).abs()
        assert int(diff.max()) <= 1, (n_off, int((diff > 1).any(dim=1).sum()), "still off after re-rendering the reference on the kernel's rays")
    assert (diff == 0).float().mean().item() > 0.999
    assert psnr(frame.float() / 255, ref8.float() / 255) > 55
    return n_off


@pytest.mark.parametrize("impl", ["ops", "fused"])
@pytest.mark.parametrize("torso", [False, True])
@pytest.mark.parametrize("size,idx", [(64, 1), (96, 3)])
def test_frame_vs_reference_golden(impl, torso, size, idx):
    hp, sd, model = build(torso, impl)
    fi = frame_inputs(sequence(4, size, size), idx)
    out = render_gpu(model, hp, fi)
    gold = np.load(os.path.join(GOLD, f"frame_{'torso' if torso else 'head'}_{size}.npz"))
    check(out, gold, torso)


@pytest.mark.parametrize("impl", ["ops", "fused"])
def test_frame_256_vs_oracle(impl):
    hp, sd, model = build(True, impl)
    fi = frame_inputs(sequence(4, 256, 256), 0)
    trace = []
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True, trace=trace)
    out = render_gpu(model, hp, fi)
    check(out, ref, True)
    if impl == "fused":
        # the fused path has no iterations: it must arrive at the reference's budget and sample totals by replaying the
        # n_step schedule from its terminal-index histogram (frame_head.hip header)
        from geneface_amd.fused import frame_stats
        fs = frame_stats(model.last_ctrl, 256 * 256, hp["max_steps"])
        assert [n for _, n in fs["schedule"]] == [t["n_step"] for t in trace]
        assert fs["budget"] == fs["budget_device"] == sum(t["n_step"] for t in trace)
        for (a, _), t in zip(fs["schedule"], trace):
            assert abs(a - t["n_alive"]) <= max(3, 1e-3 * t["n_alive"]), (a, t["n_alive"])
        # field evaluations actually performed: at least every sample the reference composites, at most every sample the
        # reference marches (its iterations evaluate a dying ray's whole n_step chunk; the pool rounds waste fewer)
        total, hi, lo = sum(fs["samples"]), sum(t["n_valid"] for t in trace), sum(t["n_composited"] for t in trace)
        assert lo - max(16, 1e-3 * lo) <= total <= hi + max(16, 1e-3 * hi), (lo, total, hi)
        # what the compositor consumed is the reference's composited count (up to the rays whose transmittance sits within rounding of
        # T_thresh).  The samples evaluated beyond it are the rest of a dying ray's slots in its last round: never more than the reference's
        # own iterations evaluate in vain (here, at 256x256, the pools are a third full and a ray gets 2-3 slots per round: ~10 %; at the
        # headline 512x512 every pool is full, one slot per ray and round: < 5 %, asserted in test_frame_loop_512_head_torso_vs_oracle)
        comp = sum(fs["composited"])
        assert abs(comp - lo) <= max(16, 1e-3 * lo), (comp, lo)
        assert total - comp <= hi - lo, (total, comp, hi, lo)
        assert fs["n_hit"] == trace[1]["n_alive"] or abs(fs["n_hit"] - trace[0]["n_valid"]) == 0
    elif hasattr(model, "last_schedule") and model.last_schedule:
        # the n_step schedule is a discrete function of the alive counts: it must be identical; the counts themselves may
        # differ by the few rays whose transmittance sits within rounding of T_thresh when an iteration ends
        assert [s for _, s in model.last_schedule] == [t["n_step"] for t in trace]
        for (a, _), t in zip(model.last_schedule, trace):
            assert abs(a - t["n_alive"]) <= max(3, 1e-3 * t["n_alive"]), (a, t["n_alive"])


def test_field_query_vs_oracle():
    """RADNeRF.forward / density / cal_cond_feat / forward_torso as stand-alone calls (the GUI and the
    density-grid update call them directly)."""
    hp, sd, model = build(True, "ops")
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(5000, 3, generator=g) * 2 - 1) * torch.tensor([0.5, 0.3, 0.5])
    d = torch.nn.functional.normalize(torch.randn(5000, 3, generator=g), dim=-1)
    cond = torch.randn(5, 1, 204, generator=g)
    cf_ref = R.cal_cond_feat(sd, hp, cond)
    s_ref, c_ref, a_ref = R.head_field(sd, hp, x, d, cf_ref, sd["individual_embeddings"][0])
    with torch.no_grad():
        cf = model.cal_cond_feat(cond.to(DEV))
        assert (cf.cpu() - cf_ref).abs().max() < 1e-5
        s, c, a = model(x.to(DEV), d.to(DEV), cf_ref.to(DEV), model.individual_embeddings[0])
        dd = model.density(x.to(DEV), cf_ref.to(DEV))
    assert (a.cpu() - a_ref).abs().max() < 1e-5
    assert ((s.cpu() - s_ref).abs() / s_ref.abs().clamp(min=1e-3)).max() < 2e-3   # exp() amplifies 1e-6-level logit noise
    assert (c.cpu() - c_ref).abs().max() < 1e-4
    assert torch.equal(dd["sigma"], s) and dd["geo_feat"].shape == (5000, 128)
    xy = torch.rand(3000, 2, generator=g) * 2 - 1
    p6 = torch.tensor([[0.1, -0.05, 0.02, 0.0, 3.3, 0.0]])
    ta_ref, tc_ref, tdx_ref = R.torso_field(sd, hp, xy, p6, sd["torso_individual_codes"][0])
    with torch.no_grad():
        ta, tc, tdx = model.forward_torso(xy.to(DEV), p6.to(DEV), model.torso_individual_codes[0])
    assert (ta.cpu() - ta_ref).abs().max() < 1e-4 and (tc.cpu() - tc_ref).abs().max() < 1e-4 and (tdx.cpu() - tdx_ref).abs().max() < 1e-4


def test_field_forward_one_launch_vs_oracle():
    """model(x, d, cond_feat, code) outside autograd (viewer, frozen head of torso training, the op-by-op render loop) is ONE launch of the
    renderer's own field core over the point list (gf_field_forward), with the caller's condition vector and identity code; under
    autograd the same call builds the torch graph."""
    hp, sd, model = build(True, "fused")
    model.render_impl = "auto"
    g = torch.Generator().manual_seed(5)
    for M in (5000, 1, 129):
        x = (torch.rand(M, 3, generator=g) * 2 - 1) * torch.tensor([0.5, 0.3, 0.5])
        d = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
        cf = R.cal_cond_feat(sd, hp, torch.randn(5, 1, 204, generator=g))
        code = sd["individual_embeddings"][7]
        s_ref, c_ref, a_ref = R.head_field(sd, hp, x, d, cf, code)
        with torch.no_grad():
            s, c, a = model(x.to(DEV), d.to(DEV), cf.to(DEV), model.individual_embeddings[7])
        assert not s.requires_grad and s.shape == (M,) and c.shape == (M, 3) and a.shape == (M, 2)
        assert (a.cpu() - a_ref).abs().max() < 1e-5
        assert ((s.cpu() - s_ref).abs() / s_ref.abs().clamp(min=1e-3)).max() < 2e-3   # exp() amplifies 1e-6-level logit noise
        assert (c.cpu() - c_ref).abs().max() < 1e-4
    with torch.no_grad():
        s0, c0, a0 = model(torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV), cf.to(DEV), model.individual_embeddings[0])
    assert s0.numel() == 0 and c0.shape == (0, 3)
    s2, c2, a2 = model(x.to(DEV), d.to(DEV), cf.to(DEV), model.individual_embeddings[7])      # grad mode: the autograd route
    assert s2.requires_grad and (s2.detach() - s).abs().max() / s.abs().max() < 2e-3


@pytest.mark.parametrize("torso", [False, True])
def test_frame_pipeline_pose_mode_vs_oracle(torso):
    """FramePipeline (the frame loop of base_nerf_infer.py:81-106): rays generated inside the kernel from the pose,
    uint8 frame copied to pinned host memory.  Bar: check_u8."""
    from geneface_amd.infer import FramePipeline
    hp, sd, model = build(torso, "fused")
    seq = sequence(4, 128, 128)
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused")
    for i in (0, 3):
        frame = pipe.render_frame(i)
        torch.cuda.synchronize()
        fi = frame_inputs(seq, i)
        ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso)
        ref8 = (ref["rgb_map"] * 255).view(128, 128, 3).to(torch.uint8)
        check_u8(frame, ref8, rerender=lambda: oracle_u8(sd, hp, pipe_inputs(pipe, i), torso))
        # the ops-path pipeline must give the same picture (its rays are torch's get_rays: arbitration = the ops path on the kernel's rays)
        pipe_ops = FramePipeline(model, hp, seq, DEV, impl="ops")
        frame_ops = pipe_ops.render_frame(i).clone()
        torch.cuda.synchronize()

        def ops_on_kernel_rays():
            with torch.no_grad():
                return (pipe_ops.run_model(pipe_ops.kernel_sample(i))["rgb_map"] * 255).to(torch.uint8).cpu()
        check_u8(frame, frame_ops, rerender=ops_on_kernel_rays)
        # and the module API fed the kernel's own rays IS the frame loop's frame, byte for byte
        with torch.no_grad():
            same = (pipe.run_model(pipe.kernel_sample(i))["rgb_map"] * 255).to(torch.uint8).view(128, 128, 3).cpu()
        assert torch.equal(same, frame.cpu())


def test_cond_encode_kernel_vs_oracle():
    """gf_cond_encode (AudioNet + AudioAttNet + the two bias folds in one launch) against the oracle's cal_cond_feat
    (the reference's cond_encoder.py modules in torch fp32) and torch folds of the same matrices."""
    from geneface_amd.fused import _per_frame_vectors, get_state
    hp, sd, model = build(True, "fused")
    st = get_state(model)
    assert st.cond is not None, "the May config must be served by the HIP condition encoder"
    g = torch.Generator().manual_seed(5)
    for trial in range(3):
        cond = torch.randn(5, 1, 204, generator=g)
        p6 = torch.tensor([[0.1 * trial, -0.05, 0.02, 0.01, 3.3, -0.02]])
        cf_ref = R.cal_cond_feat(sd, hp, cond)
        with torch.no_grad():
            cf, amb_bias, torso_bias = _per_frame_vectors(model, st, cond.to(DEV), p6.to(DEV))
            torch.cuda.synchronize()
            assert (cf.cpu() - cf_ref).abs().max() < 1e-5
            assert (amb_bias.cpu() - torch.mv(st.W_cond.cpu(), cf_ref)).abs().max() < 2e-5
            v = torch.cat([model.torso_pose_embedder(p6.to(DEV)).reshape(-1), model.torso_individual_codes[0]]).cpu()
            assert (torso_bias.cpu() - torch.mv(st.W_tconst.cpu(), v)).abs().max() < 2e-5
            # and the module-API encoder (torch modules on the GPU) agrees with both
            assert (model.cal_cond_feat(cond.to(DEV)).cpu() - cf_ref).abs().max() < 1e-5


@pytest.mark.parametrize("grid_type,interp", [("hashgrid", "linear"), ("hashgrid", "smoothstep"), ("tiledgrid", "smoothstep")])
def test_fused_other_grid_configs_vs_oracle(grid_type, interp):
    """The specialised lookup of the fused head kernel (grid_core.hpp::encode8: strides / mask / hash flag per level) on the
    grid variants GridEncoder offers besides the May default: xor-prime hashing on the levels whose lattice does not fit
    the table (gridencoder.cu:50-63) and smoothstep interpolation (:152-157).  Same tables, different index rule."""
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp, sd = model_fixture(True)
    hp = dict(hp, grid_type=grid_type, grid_interpolation_type=interp)
    model = RADNeRFTorso(hp)
    model.load_state_dict(sd, strict=True)
    model.render_impl = "fused"
    model = model.to(DEV).eval()
    fi = frame_inputs(sequence(4, 96, 96), 2)
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True)
    out = render_gpu(model, hp, fi)
    check(out, ref, True)
    model.render_impl = "ops"
    out_ops = render_gpu(model, hp, fi)
    check(out_ops, ref, True)


@pytest.mark.parametrize("size", [512, 1024])
def test_full_size_fused_vs_ops_and_invariants(size):
    """BASELINE.json's full size (512x512 head+torso) and four times that (2^20 rays: the largest frame the viewer asks for), where the CPU
    oracle takes 6 - 25 s per frame: the two GPU execution
    strategies (reference loop structure over stand-alone ops vs the fused two-phase kernels) must produce the same picture,
    and size-independent properties of the path must hold: the schedule replayed on the device equals the one the op-by-op
    loop actually ran; pixels whose ray misses the occupancy are exactly the blended background; weights stay in [0, 1]."""
    from geneface_amd.fused import frame_stats
    hp, sd, model = build(True, "ops")
    fi = frame_inputs(sequence(4, size, size), 1)
    out_ops = render_gpu(model, hp, fi)
    sched_ops = [s for _, s in model.last_schedule]
    model.render_impl = "fused"
    out = render_gpu(model, hp, fi)
    fs = frame_stats(model.last_ctrl, size * size, hp["max_steps"])
    assert [n for _, n in fs["schedule"]] == sched_ops
    assert fs["budget"] == fs["budget_device"] == sum(sched_ops)
    a, b = out["rgb_map"].float().cpu(), out_ops["rgb_map"].float().cpu()
    assert (a - b).abs().max().item() < RGB_ATOL and psnr(a, b) > 65
    u8 = (a * 255).to(torch.uint8).int() - (b * 255).to(torch.uint8).int()
    assert (u8.abs() <= 1).float().mean().item() > 0.999
    assert (out["depth_map"].cpu() - out_ops["depth_map"].cpu()).abs().max().item() < 2e-3
    # rays that pass outside the (one cell padded) box around the occupied cells can have no sample at all:
    # rgb == torso-over-background exactly (image = 0, weights_sum = 0)
    from geneface_amd.fused import get_state
    box = torch.tensor(get_state(model).occ_aabb)
    cell = 2.0 * hp["bound"] / hp["grid_size"]
    lo, hi = box[:3] - cell, box[3:] + cell
    o, d = fi["rays_o"].reshape(-1, 3), fi["rays_d"].reshape(-1, 3)
    t0, t1 = (lo - o) / d, (hi - o) / d
    tn, tf = torch.minimum(t0, t1).max(dim=1).values, torch.maximum(t0, t1).min(dim=1).values
    miss = tn > tf
    assert 0.3 < miss.float().mean().item() < 0.9
    bgmix = out["torso_rgb_map"].reshape(-1, 3).cpu()
    assert torch.equal(a.reshape(-1, 3)[miss], bgmix[miss].clamp(0, 1))
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0


@pytest.mark.parametrize("cond_type,cin", [("esperanto", 44), ("deepspeech", 29)])
def test_audio_driven_identity_vs_oracle(cond_type, cin):
    """The audio-driven RAD-NeRF variant the reference ships for its second identity (egs/datasets/videos/Obama/radnerf.yaml ->
    egs/egs_bases/radnerf/radnerf.yaml:4-7: 16-frame audio-feature windows, smo_win_size 8) with different weights and a
    different seed: same kernels, the condition encoder runs its strided-conv branch (cond_encoder.py:14-41, T=16 -> 1)."""
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.fused import _per_frame_vectors, get_state
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = dict(HP.may_hparams(True), cond_type=cond_type, cond_win_size=16, smo_win_size=8)
    sd = S.make_state_dict(hp, True, seed=1000)
    model = RADNeRFTorso(hp)
    model.load_state_dict(sd, strict=True)
    model.render_impl = "fused"
    model = model.to(DEV).eval()
    fi = frame_inputs(sequence(4, 64, 64), 1)
    cond = torch.randn(8, 16, cin, generator=torch.Generator().manual_seed(9))
    st = get_state(model)
    assert st.cond is not None and (st.cond.S, st.cond.T, st.cond.C) == (8, 16, cin)
    cf_ref = R.cal_cond_feat(sd, hp, cond)
    with torch.no_grad():
        cf, _, _ = _per_frame_vectors(model, st, cond.to(DEV), fi["pose6"].to(DEV))
    assert (cf.cpu() - cf_ref).abs().max() < 2e-5
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], cond, fi["bg_coords"], fi["pose6"], fi["bg"], torso=True)
    to = lambda t: t.to(DEV)
    out = model.render(to(fi["rays_o"]), to(fi["rays_d"]), to(cond), to(fi["bg_coords"]), to(fi["pose6"]), index=0, staged=False,
                       bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, **hp)
    check(out, ref, True)


def test_two_cascades_vs_oracle():
    """bound = 2 (two occupancy cascades, renderer.py:67): the marcher picks the mip level per sample from position and step
    size (raymarching.cu:42-54,868-878), the occupancy bounding box spans both cascades, the 3-D grid covers [-2, 2]."""
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = dict(HP.may_hparams(True), bound=2)
    sd = S.make_state_dict(hp, True, seed=3)
    for impl in ("ops", "fused"):
        model = RADNeRFTorso(hp)
        model.load_state_dict(sd, strict=True)
        model.render_impl = impl
        model = model.to(DEV).eval()
        assert model.cascade == 2
        fi = frame_inputs(sequence(4, 96, 96), 1)
        ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True)
        out = render_gpu(model, hp, fi)
        check(out, ref, True)


def _render_both(hp, sd, fi, torso=True, **over):
    """oracle + both GPU strategies for one frame with hparams overrides."""
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = dict(hp, **over)
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso)
    outs = {}
    for impl in ("ops", "fused"):
        m = (RADNeRFTorso if torso else RADNeRF)(hp)
        m.load_state_dict(sd, strict=True)
        m.render_impl = impl
        m = m.to(DEV).eval()
        outs[impl] = render_gpu(m, hp, fi)
    return ref, outs


def test_edge_empty_occupancy():
    """No occupied cell: every ray terminates in the set-up kernel, both field launches find empty queues, the frame is the
    (torso-blended) background exactly and depth is 0."""
    hp, sd = model_fixture(True)
    sd = dict(sd, density_bitfield=torch.zeros_like(sd["density_bitfield"]))
    fi = frame_inputs(sequence(4, 64, 64), 0)
    ref, outs = _render_both(hp, sd, fi)
    for impl, out in outs.items():
        check(out, ref, True)
        assert torch.equal(out["rgb_map"].cpu().reshape(-1, 3), out["torso_rgb_map"].cpu().reshape(-1, 3).clamp(0, 1)), impl
        assert float(out["depth_map"].abs().max()) == 0.0


def test_edge_full_occupancy_and_long_budgets():
    """Every cell occupied (rays march from the box entry, budgets are what the schedule gives them) and the budget regimes of the fused
    path: max_steps below, at and above the May value, through the old fused limit of 64 to the reference's own default of 1024
    (renderer.py:263; the top of its viewer's slider, radnerf_gui.py:466-471) -- terminal indices beyond the LDS histogram go to the control
    block directly, phase 1 replays the schedule from an LDS copy.  With dt_gamma = 1/256 and with dt_gamma = 0 (fixed step 2 sqrt(3) /
    max_steps: every ray takes hundreds of samples at max_steps = 1024)."""
    from geneface_amd.fused import frame_stats
    hp, sd = model_fixture(False)
    sd = dict(sd, density_bitfield=torch.full_like(sd["density_bitfield"], 255))
    fi = frame_inputs(sequence(4, 48, 48), 2)
    for ms, dtg in ((4, None), (16, None), (64, None), (65, None), (128, None), (256, 0.0), (1024, None), (1024, 0.0)):
        over = dict(max_steps=ms) if dtg is None else dict(max_steps=ms, dt_gamma=dtg)
        hp_o = dict(hp, **over)
        trace = []
        ref = R.render(sd, hp_o, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=False, trace=trace)
        for impl in ("ops", "fused"):
            from geneface_amd.radnerf import RADNeRF
            m = RADNeRF(hp_o)
            m.load_state_dict(sd, strict=True)
            m.render_impl = impl
            m = m.to(DEV).eval()
            assert m._pick_impl("auto", False, ms) == "fused"
            out = render_gpu(m, hp_o, fi)
            check(out, ref, False)
            if impl == "fused":      # the schedule replay arrives at the reference's iteration list and total budget, whatever the histogram's home
                fs = frame_stats(m.last_ctrl, 48 * 48, ms)
                assert [n for _, n in fs["schedule"]] == [t["n_step"] for t in trace], (ms, dtg)
                assert fs["budget"] == fs["budget_device"] == sum(t["n_step"] for t in trace)


def test_long_budget_occupancy_fixture_and_the_split_tier():
    """max_steps = 1024 on the analytic head (most rays end by leaving the occupied region after a few dozen samples, some terminate on
    transmittance): head+torso 96 x 96 against the oracle, fp32 and split tiers, and the pose-mode frame loop."""
    from geneface_amd.infer import FramePipeline
    hp, sd = model_fixture(True)
    for dtg in (1.0 / 256, 0.0):
        hp_o = dict(hp, max_steps=1024, dt_gamma=dtg)
        seq = sequence(4, 96, 96)
        fi = frame_inputs(seq, 1)
        ref = R.render(sd, hp_o, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True)
        for precision in ("fp32", "split"):
            from geneface_amd.radnerf_torso import RADNeRFTorso
            m = RADNeRFTorso(hp_o)
            m.load_state_dict(sd, strict=True)
            m.render_impl, m.render_precision = "fused", precision
            m = m.to(DEV).eval()
            check(render_gpu(m, hp_o, fi), ref, True)
            pipe = FramePipeline(m, hp_o, seq, DEV, impl="fused")
            frame = pipe.render_frame(1)
            pipe.wait()
            check_u8(frame, (ref["rgb_map"] * 255).view(96, 96, 3).to(torch.uint8), rerender=lambda: oracle_u8(sd, hp_o, pipe_inputs(pipe, 1), True))


@pytest.mark.parametrize("H,W", [(1, 1), (1, 3), (37, 50), (5, 129)])
def test_edge_ragged_ray_counts(H, W):
    """Ray counts that are not multiples of the wave / workgroup / tile sizes, down to a single ray."""
    hp, sd = model_fixture(True)
    seq = sequence(4, 64, 64)
    fi64 = frame_inputs(seq, 1)
    idx = torch.linspace(0, 64 * 64 - 1, H * W).long()        # a ragged subset of the 64x64 frame's rays
    fi = dict(fi64, rays_o=fi64["rays_o"][:, idx].contiguous(), rays_d=fi64["rays_d"][:, idx].contiguous(),
              bg_coords=fi64["bg_coords"][:, idx].contiguous(), bg=fi64["bg"][:, idx].contiguous())
    ref, outs = _render_both(hp, sd, fi)
    for out in outs.values():
        check_small(out, ref)


def check_small(out, ref):
    rgb, rgb_ref = out["rgb_map"].cpu(), ref["rgb_map"]
    assert rgb.shape == rgb_ref.shape
    assert (rgb - rgb_ref).abs().max().item() < RGB_ATOL
    assert (out["depth_map"].cpu() - ref["depth_map"]).abs().max().item() < 2e-3
    for k in ("torso_alpha_map", "torso_rgb_map"):
        assert (out[k].cpu().reshape(ref[k].shape) - ref[k]).abs().max().item() < 2e-5


def test_edge_thresholds_and_step_scale():
    """T_thresh (early termination) and dt_gamma (distance-proportional steps, raymarching.cu:868) away from the May values."""
    hp, sd = model_fixture(True)
    fi = frame_inputs(sequence(4, 64, 64), 3)
    for over in ({"dt_gamma": 0.0}, {"dt_gamma": 1.0 / 64}, {"max_steps": 24}):
        ref, outs = _render_both(hp, sd, fi, **over)
        for out in outs.values():
            check(out, ref, True)
    # T_thresh is a render() argument, not an hparam
    from geneface_amd.radnerf_torso import RADNeRFTorso
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True, T_thresh=0.05)
    to = lambda t: t.to(DEV)
    for impl in ("ops", "fused"):
        m = RADNeRFTorso(hp)
        m.load_state_dict(sd, strict=True)
        m.render_impl = impl
        m = m.to(DEV).eval()
        out = m.render(to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]), index=0, staged=False,
                       bg_color=to(fi["bg"]), perturb=False, force_all_rays=True, T_thresh=0.05, **hp)
        check(out, ref, True)


def test_viewer_path_downscale_and_accumulation_vs_oracle():
    """a free orbit-camera view rendered at half resolution and resized (tasks/radnerfs/radnerf.py:333-380), then the
    viewer's spp accumulation (radnerf_gui.py:222-229).  Oracle: the same rays through radnerf_ref.render + the same resize."""
    import torch.nn.functional as F
    from geneface_amd import gui
    hp, sd, model = build(True, "fused")
    seq = sequence(4, 128, 128)
    cam = gui.OrbitCamera(128, 128, r=3.35, fovy=21.24)
    cam.update_intrinsics(seq["intrinsics"])
    cam.orbit(150, 60)
    cond = torch.from_numpy(seq["cond_wins"][1])
    bg = torch.from_numpy(seq["bg_img"]).view(1, -1, 3)
    out = gui.test_gui_with_editable_data(model, hp, cam.pose, cam.intrinsics, 128, 128, cond, 0, bg, 1, 0.5, DEV)
    assert out["image"].shape == (128, 128, 3) and out["depth"].shape == (128, 128)
    pose = torch.from_numpy(cam.pose).unsqueeze(0)
    ro, rd = R.get_rays(pose, cam.intrinsics * 0.5, 64, 64)
    bg64 = F.interpolate(bg.view(1, 128, 128, 3).permute(0, 3, 1, 2), size=(64, 64), mode="bilinear").permute(0, 2, 3, 1).reshape(1, -1, 3)
    ref = R.render(sd, hp, ro, rd, cond, R.get_bg_coords(64, 64), R.convert_poses(pose), bg64, True)
    img = F.interpolate(ref["rgb_map"].view(1, 64, 64, 3).permute(0, 3, 1, 2), size=(128, 128), mode="bilinear").permute(0, 2, 3, 1)[0]
    assert (torch.from_numpy(out["image"]) - img).abs().max().item() < RGB_ATOL
    dep = F.interpolate(ref["depth_map"].view(1, 1, 64, 64), size=(128, 128), mode="nearest")[0, 0]
    assert (torch.from_numpy(out["depth"]) - dep).abs().max().item() < 2e-3
    v = gui.Viewer(model, dict(hp, gui_max_spp=3), 128, 128, cond_features=torch.from_numpy(seq["cond_wins"][:, 2]), bg_color=bg, device=DEV)
    v.cam = cam
    first = v.test_step().copy()
    for _ in range(4):
        buf = v.test_step()
    assert v.spp == 3 and np.abs(buf - first).max() < 1e-6            # deterministic renderer: the running mean is the frame
    cam.orbit(40, 0)
    v.need_update = True
    assert np.abs(v.test_step() - first).max() > 1e-3 and v.spp == 1


def test_fast_precision_tier_vs_oracle():
    """BASELINE.md section 4, "fast": f16 MFMA operands and activations with fp32 accumulation (the reference's autocast / .half()
    arithmetic) against the fp32 oracle: PSNR >= 40 dB on the float image, <= 1 LSB on >= 99.9 % of the uint8 pixels."""
    hp, sd, model = build(True, "fused")
    fi = frame_inputs(sequence(4, 256, 256), 2)
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True)
    strict = render_gpu(model, hp, fi)["rgb_map"].cpu()
    model.render_precision = "fast"
    out = render_gpu(model, hp, fi)
    rgb, rgb_ref = out["rgb_map"].cpu(), ref["rgb_map"]
    p = psnr(rgb, rgb_ref)
    u8 = (rgb * 255).to(torch.uint8).int() - (rgb_ref * 255).to(torch.uint8).int()
    within = (u8.abs() <= 1).float().mean().item()
    err = (rgb - rgb_ref).abs().max().item()
    print(f"fast tier: PSNR {p:.1f} dB, max|drgb| {err:.2e}, uint8 within 1 LSB {within * 100:.3f} %")
    assert p >= 40.0 and within >= 0.999, (p, within, err)
    assert (out["depth_map"].cpu() - ref["depth_map"]).abs().max().item() < 2e-2
    assert (strict - rgb_ref).abs().max().item() < RGB_ATOL            # the default stays the strict fp32 path
    assert (rgb - strict).abs().max().item() > 0                        # and the fast path really is a different kernel
    with pytest.raises(ValueError):
        model.render_precision = "bf8"
        render_gpu(model, hp, fi)


def _workspace_field(model, N, field, shape, dtype=torch.float32, slot=0):
    """A per-ray array of the frame workspace the last fused render on `slot` left behind (include/geneface_hip.h: gf_frame_field_offset)."""
    from geneface_amd import fused
    from geneface_amd.lib import lib
    ws = fused.get_state(model).workspace(N, slot)[0]
    off = int(lib().gf_frame_field_offset(N, field))
    n = int(np.prod(shape)) * 4
    return ws[off:off + n].view(dtype).view(*shape).clone()


def test_in_kernel_rays_vs_get_rays_512():
    """k_frame_init generates the pinhole rays from pose + intrinsics (utils.py:282-363 restated per lane) and runs the slab test
    (raymarching.cu:92-145).  Max-abs against the torch get_rays / the oracle's near_far_from_aabb at the full 512x512: directions
    within 2 ulp of a unit vector's component (a 3-term rotation sum, fused differently by torch's matmul), origins exact, near / far
    within a few ulp of t ~ 3."""
    from geneface_amd.infer import FramePipeline
    hp, sd, model = build(True, "fused")
    seq = sequence(4, 512, 512)
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused", overlap=False)
    N = 512 * 512
    for i in (0, 3):
        slot = pipe._slot
        pipe.render_frame(i)
        pipe.wait()
        torch.cuda.synchronize()
        fi = frame_inputs(seq, i)
        ws_slot = slot % 2
        rays_d = _workspace_field(model, N, 7, (N, 3), slot=ws_slot).cpu()
        nears = _workspace_field(model, N, 0, (N,), slot=ws_slot).cpu()
        fars = _workspace_field(model, N, 1, (N,), slot=ws_slot).cpu()
        from geneface_amd import fused
        n_hit = int(fused.get_state(model).workspace(N, ws_slot)[1][1])
        hits = _workspace_field(model, N, 9, (N,), dtype=torch.int32, slot=ws_slot).cpu()[:n_hit].long()
        assert 0.2 * N < n_hit < 0.6 * N and hits.unique().numel() == n_hit      # per-ray marcher state exists for the hit rays only
        ro, rd = fi["rays_o"].view(N, 3), fi["rays_d"].view(N, 3)
        assert torch.equal(ro[0], torch.from_numpy(seq["poses"][i][:3, 3]))          # the one origin, handed to the kernel by value
        err_d = (rays_d[hits] - rd[hits]).abs().max().item()
        assert err_d <= 2 ** -22, err_d                       # <= 2 ulp of a component of magnitude <= 1 (measured: 1.5)
        # gf_pinhole_rays is the same device function as a stand-alone launch: the rays the kernel kept for its hit rays, bit for bit, and the
        # same thing for every other pixel (what the oracle is fed to arbitrate a pose-mode pixel, helpers.pipe_inputs)
        from geneface_amd.fused import pinhole_rays
        ko, kd = pinhole_rays(torch.from_numpy(seq["poses"][i]), seq["intrinsics"], 512, 512, DEV)
        assert torch.equal(kd.cpu().view(N, 3)[hits], rays_d[hits]) and torch.equal(ko.cpu().view(N, 3), ro)
        assert (kd.cpu().view(N, 3) - rd).abs().max().item() <= 2 ** -22
        assert (rays_d[hits].norm(dim=-1) - 1).abs().max().item() < 2e-7
        n_ref, f_ref = R.near_far_from_aabb(ro, rd, sd["aabb_infer"], hp["min_near"])
        hit = f_ref < 1e30
        assert torch.equal(hit, fars < 1e30)
        assert (nears[hit] - n_ref[hit]).abs().max().item() < 2e-6 and (fars[hit] - f_ref[hit]).abs().max().item() < 2e-6
        assert torch.equal(nears[~hit], n_ref[~hit]) and torch.equal(fars[~hit], f_ref[~hit])


def test_head_only_512_vs_oracle():
    """BASELINE.json configs[1]: May lm3d_radnerf head-only at the full 512x512 against the CPU oracle (explicit rays through the module
    API -> fp32 rgb_map, and the frame loop's in-kernel rays -> uint8)."""
    from geneface_amd.infer import FramePipeline
    hp, sd, model = build(False, "fused")
    seq = sequence(4, 512, 512)
    fi = frame_inputs(seq, 2)
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=False)
    out = render_gpu(model, hp, fi)
    check(out, ref, False)
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused")
    frame = pipe.render_frame(2)
    pipe.wait()
    ref8 = (ref["rgb_map"] * 255).view(512, 512, 3).to(torch.uint8)
    d = (frame.int() - ref8.int()).abs()
    assert int(d.max()) <= 1 and (d == 0).float().mean().item() > 0.999


@pytest.mark.parametrize("impl", ["ops", "fused"])
def test_second_identity_256_vs_oracle(impl):
    """BASELINE.json configs[4] (second identity, Obama2-style: same architecture, other weights, another occupancy shape): head+torso at
    256x256 against the oracle."""
    hp = model_fixture(True)[0]
    from geneface_amd import synthetic as S
    sd = S.make_state_dict(hp, True, seed=1000)
    assert not torch.equal(sd["density_bitfield"], model_fixture(True)[1]["density_bitfield"])
    from geneface_amd.radnerf_torso import RADNeRFTorso
    m = RADNeRFTorso(hp)
    m.load_state_dict(sd, strict=True)
    m.render_impl = impl
    m = m.to(DEV).eval()
    fi = frame_inputs(sequence(4, 256, 256, seed=5), 1)
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True)
    check(render_gpu(m, hp, fi), ref, True)


@pytest.mark.parametrize("impl", ["ops", "fused"])
@pytest.mark.parametrize("branch", [False, True])
def test_torso_head_aware_vs_oracle(branch, impl, monkeypatch):
    """`torso_head_aware: true` (radnerf_torso.py:36-46, :68-74, :175-179): the head-colour encoder widens both torso MLPs; the reference
    flips a coin per frame between showing the torso the rendered head and zeros.  Both outcomes against the oracle, on the op-by-op path
    and (round 3) on the fused one: 'zeros' is a per-frame constant folded into the torso bias, 'head' a per-pixel encoder launch plus 8
    extra MFMA steps in both first layers of k_torso_finish<true>.  The frame loop flips the same coin."""
    import random
    from geneface_amd import synthetic as S
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = dict(model_fixture(True)[0], torso_head_aware=True)
    sd = S.make_state_dict(hp, True, seed=2)
    m = RADNeRFTorso(hp)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    assert m._pick_impl("auto", False, hp["max_steps"]) == "fused"
    m.render_impl = impl
    monkeypatch.setattr(random, "random", lambda: 0.25 if branch else 0.75)
    fi = frame_inputs(sequence(4, 96, 96), 2)
    out = render_gpu(m, hp, fi)
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True, head_aware_branch=branch)
    check(out, ref, True)
    other = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True, head_aware_branch=not branch)
    assert (other["torso_alpha_map"] - ref["torso_alpha_map"]).abs().max() > 1e-3     # the two branches really differ
    if impl == "fused":
        from geneface_amd.infer import FramePipeline
        seq = sequence(4, 96, 96)
        pipe = FramePipeline(m, hp, seq, DEV, impl="fused")
        frame = pipe.render_frame(2)
        pipe.wait()
        check_u8(frame, (ref["rgb_map"] * 255).view(96, 96, 3).to(torch.uint8),
                 rerender=lambda: oracle_u8(sd, hp, pipe_inputs(pipe, 2), True, head_aware_branch=branch))
        # a weight update reaches the head-aware packs as well (index-map refresh, no host round trip)
        with torch.no_grad():
            m.head_color_weights_encoder[4].weight.mul_(0.5)
            m.torso_deform_net.net[0].weight[:, 104:].mul_(2.0)
        sd2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        ref2 = R.render(sd2, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True, head_aware_branch=branch)
        check(render_gpu(m, hp, fi), ref2, True)


def test_fused_state_follows_the_weights():
    """The fused path renders from packed COPIES of the MLP weights, fold matrices and the occupancy box.  Whatever changes the model after
    a first render -- load_state_dict, an in-place optimizer-style update, a new occupancy bitfield, a dtype round trip that reallocates the
    parameters -- must show in the next render without any manual invalidation (train -> validate loops, checkpoint reloads)."""
    hp, sd, model = build(True, "fused")
    fi = frame_inputs(sequence(4, 64, 64), 1)
    first = render_gpu(model, hp, fi)["rgb_map"].clone()
    # (1) other weights through load_state_dict: same as a model built from them
    hp2, sd2 = model_fixture(True, seed=1)
    model.load_state_dict(sd2, strict=True)
    fresh = build(True, "fused")[2]
    fresh.load_state_dict(sd2, strict=True)
    got, want = render_gpu(model, hp, fi)["rgb_map"], render_gpu(fresh, hp, fi)["rgb_map"]
    assert torch.equal(got, want) and not torch.equal(got, first)
    # (2) an in-place step on one MLP weight and on the identity code
    with torch.no_grad():
        model.color_net.net[1].weight.mul_(0.5)
        fresh.color_net.net[1].weight.mul_(0.5)
        model.individual_embeddings[0].add_(0.25)
        fresh.individual_embeddings[0].add_(0.25)
    from geneface_amd import fused
    fused.invalidate(fresh)
    got2, want2 = render_gpu(model, hp, fi)["rgb_map"], render_gpu(fresh, hp, fi)["rgb_map"]
    assert torch.equal(got2, want2) and not torch.equal(got2, got)
    # (3) a thinner occupancy grid written in place (the march box is derived from the bitfield)
    with torch.no_grad():
        model.density_bitfield[model.density_bitfield.numel() // 2:] = 0
        fresh.density_bitfield[fresh.density_bitfield.numel() // 2:] = 0
    fused.invalidate(fresh)
    got3, want3 = render_gpu(model, hp, fi)["rgb_map"], render_gpu(fresh, hp, fi)["rgb_map"]
    assert torch.equal(got3, want3) and not torch.equal(got3, got2)
    # (4) parameters reallocated (half -> float round trip rounds the weights and moves every data_ptr)
    model.half().float()
    fresh.half().float()
    fused.invalidate(fresh)
    assert torch.equal(render_gpu(model, hp, fi)["rgb_map"], render_gpu(fresh, hp, fi)["rgb_map"])


def _fast_model(torso, sd=None):
    hp, sd0, model = build(torso, "fused")
    if sd is not None:
        model.load_state_dict(sd, strict=True)     # the packed copies follow by themselves (fused.get_state)
    model.render_precision = "fast"
    return hp, (sd if sd is not None else sd0), model


def _check_fast(out, ref, min_psnr=40.0):
    rgb, rgb_ref = out["rgb_map"].cpu().reshape(ref["rgb_map"].shape), ref["rgb_map"]
    assert psnr(rgb, rgb_ref) >= min_psnr, psnr(rgb, rgb_ref)
    u8 = (rgb * 255).to(torch.uint8).int() - (rgb_ref * 255).to(torch.uint8).int()
    assert (u8.abs() <= 1).float().mean().item() >= 0.999


@pytest.mark.parametrize("torso", [False, True])
@pytest.mark.parametrize("size,idx", [(64, 1), (96, 3)])
def test_fast_tier_golden_frames(torso, size, idx):
    """The fast tier on the committed golden frames (generated by the reference's own Python), head-only and head+torso."""
    hp, sd, model = _fast_model(torso)
    fi = frame_inputs(sequence(4, size, size), idx)
    gold = np.load(os.path.join(GOLD, f"frame_{'torso' if torso else 'head'}_{size}.npz"))
    out = render_gpu(model, hp, fi)
    _check_fast(out, {"rgb_map": torch.from_numpy(gold["rgb_map"])})


def test_fast_tier_edge_cases():
    """Empty and full occupancy, ragged ray counts, a long sample budget: the fast kernel shares the pools / schedule code with the
    strict one, but its LDS carve (f16 activations, 3-D feature buffer, SH table) is its own."""
    hp, sd = model_fixture(True)
    fi = frame_inputs(sequence(4, 64, 64), 0)
    empty = dict(sd, density_bitfield=torch.zeros_like(sd["density_bitfield"]))
    _, _, m = _fast_model(True, empty)
    out = render_gpu(m, hp, fi)
    ref = R.render(empty, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True)
    assert (out["rgb_map"].cpu() - ref["rgb_map"]).abs().max().item() < 2e-5 and float(out["depth_map"].abs().max()) == 0.0
    hp_h, sd_h = model_fixture(False)
    full = dict(sd_h, density_bitfield=torch.full_like(sd_h["density_bitfield"], 255))
    fi48 = frame_inputs(sequence(4, 48, 48), 2)
    for ms in (4, 16, 64):
        hpm = dict(hp_h, max_steps=ms)
        _, _, m = _fast_model(False, full)
        ref = R.render(full, hpm, fi48["rays_o"], fi48["rays_d"], fi48["cond"], fi48["bg_coords"], fi48["pose6"], fi48["bg"], False)
        _check_fast(render_gpu(m, hpm, fi48), ref, min_psnr=38.0)     # dense fog: every sample contributes, f16 noise accumulates
    idx = torch.linspace(0, 64 * 64 - 1, 37 * 5).long()
    fi_r = dict(fi, rays_o=fi["rays_o"][:, idx].contiguous(), rays_d=fi["rays_d"][:, idx].contiguous(),
                bg_coords=fi["bg_coords"][:, idx].contiguous(), bg=fi["bg"][:, idx].contiguous())
    _, _, m = _fast_model(True)
    ref = R.render(sd, hp, fi_r["rays_o"], fi_r["rays_d"], fi_r["cond"], fi_r["bg_coords"], fi_r["pose6"], fi_r["bg"], True)
    _check_fast(render_gpu(m, hp, fi_r), ref)


@pytest.mark.parametrize("precision", ["fp32", "fast", "split"])
@pytest.mark.parametrize("in_flight", [1, 2, 3, 4])
def test_frames_in_flight_do_not_interfere(in_flight, precision):
    """Several frames enqueued on separate streams share the model, the packed weights and the tables but nothing else (one workspace,
    one set of output buffers and one pinned host buffer per slot).  A pattern of four different frames rendered five times over through
    FramePipeline.stream must reproduce what each frame gives alone, bit for bit, on both precision tiers."""
    from geneface_amd.infer import FramePipeline
    hp, sd, model = build(True, "fused")
    model.render_precision = precision
    seq = sequence(4, 160, 160)
    solo = FramePipeline(model, hp, seq, DEV, impl="fused", overlap=False)
    want = []
    for i in range(4):
        f = solo.render_frame(i)
        solo.wait()
        want.append(f.clone())
    assert not torch.equal(want[0], want[1])
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused", in_flight=in_flight)
    order = [0, 1, 2, 3, 3, 1, 0, 2] * 5
    got = 0
    for (i, frame), k in zip(pipe.stream(order), order):
        assert i == k
        assert np.array_equal(frame, want[k].numpy()), (in_flight, precision, got)
        got += 1
    assert got == len(order)


@pytest.mark.parametrize("thin", [False, True])
def test_pipeline_depth_follows_the_scene(thin, monkeypatch):
    """in_flight=None on a process with eight hardware queues: the pipeline starts with three frames in flight and takes a fourth after the
    first rotation if few rays outlive phase 0 (the saturating fixture), and keeps three on a thin-density scene.  Frames are what each gives
    alone either way, bit for bit, across the switch."""
    from geneface_amd.infer import FramePipeline
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    from geneface_amd import hparams as HP
    from geneface_amd import synthetic as S
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = HP.may_hparams(True)
    model = RADNeRFTorso(hp)
    model.load_state_dict(S.make_state_dict(hp, True, sigma_row_scale=0.02 if thin else 1.0), strict=True)
    model.render_impl = "fused"
    model = model.to(DEV).eval()
    seq = sequence(4, 160, 160)
    solo = FramePipeline(model, hp, seq, DEV, impl="fused", overlap=False)
    want = []
    for i in range(4):
        f = solo.render_frame(i)
        solo.wait()
        want.append(f.clone().numpy())
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused")
    assert pipe.in_flight == 3 and pipe.max_in_flight == 4
    order = [0, 1, 2, 3, 3, 1, 0, 2] * 4
    for n, ((i, frame), k) in enumerate(zip(pipe.stream(order), order)):
        assert i == k and np.array_equal(frame, want[k]), (thin, n)
    assert pipe.in_flight == (3 if thin else 4)


@pytest.mark.parametrize("precision", ["fp32", "fast", "split"])
def test_full_size_frames_are_reproducible(precision):
    """Run-to-run reproducibility where it was once lost: at 512x512 the persistent head grid puts two workgroups on every CU (a 160x160
    frame leaves each CU with one, which is why the test above never saw it).  The fast tier's 2-D grid lookup then dropped one corner of
    its last level now and then -- a packed-FP32 instruction pair the compiler had formed (NOTES.md 4.7; tools/fast_diag.py pins it sample
    by sample) -- in about half of all frames rendered ALONE.  40 renders alone and 120 with three in flight must be identical bytes."""
    from geneface_amd.infer import FramePipeline
    hp, sd, model = build(True, "fused")
    model.render_precision = precision
    seq = sequence(4, 512, 512)
    solo = FramePipeline(model, hp, seq, DEV, impl="fused", overlap=False)
    want = []
    for i in range(4):
        f = solo.render_frame(i)
        solo.wait()
        want.append(f.clone().numpy())
    for rep in range(40):
        f = solo.render_frame(rep % 4)
        solo.wait()
        assert np.array_equal(f.numpy(), want[rep % 4]), (precision, "alone", rep)
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused", in_flight=3)
    order = [0, 1, 2, 3, 3, 1, 0, 2] * 15
    for n, ((i, frame), k) in enumerate(zip(pipe.stream(order), order)):
        assert np.array_equal(frame, want[k]), (precision, "in flight", n)


# ----------------------------------------------------------------------------------------------- round 3: the configs at their real shape
def _identity_model(seed, impl="fused"):
    from geneface_amd import synthetic as S
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = model_fixture(True)[0]
    sd = S.make_state_dict(hp, True, seed=seed)
    m = RADNeRFTorso(hp)
    m.load_state_dict(sd, strict=True)
    m.render_impl = impl
    return hp, sd, m.to(DEV).eval()


@pytest.mark.parametrize("seed", [0, 1000])
def test_frame_loop_512_head_torso_vs_oracle(seed):
    """BASELINE.json configs[2] (seed 0, the bench fixture) and configs[4] (seed 1000: second identity -- other weights, another occupancy
    shape) at the headline shape: 512x512 head+torso through BOTH seams against the CPU oracle -- the module API with explicit rays
    (fp32 rgb / depth / torso maps, strict tolerance) and the pose-mode frame loop (in-kernel rays, uint8 to pinned host memory)."""
    from geneface_amd.infer import FramePipeline
    hp, sd, model = _identity_model(seed)
    seq = sequence(4, 512, 512, seed=5 if seed else 0)
    fi = frame_inputs(seq, 1)
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True)
    check(render_gpu(model, hp, fi), ref, True)
    from geneface_amd.fused import frame_stats
    fs = frame_stats(model.last_ctrl, 512 * 512, hp["max_steps"])
    total, comp = sum(fs["samples"]), sum(fs["composited"])
    assert 0 <= total - comp <= 0.05 * total, (total, comp)     # field evaluations a terminating ray no longer needed: < 5 % at the headline shape
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused")
    frame = pipe.render_frame(1)
    pipe.wait()
    ref8 = (ref["rgb_map"] * 255).view(512, 512, 3).to(torch.uint8)
    d = (frame.int() - ref8.int()).abs()
    assert int(d.max()) <= 1 and (d == 0).float().mean().item() > 0.999
    assert psnr(frame.float() / 255, ref8.float() / 255) > 55


def test_prepared_pass_is_bit_identical():
    """FramePipeline.prepare: ONE condition-encoder launch for a whole pass (gf_cond_encode_batch, a workgroup per frame) instead of one
    single-workgroup launch per frame.  Row k of the batch is the single launch's result bit for bit, so are the frames -- alone, in
    flight, and through stream(), which prepares its block by itself."""
    from geneface_amd import fused
    from geneface_amd.infer import FramePipeline
    hp, sd, model = build(True, "fused")
    seq = sequence(12, 160, 160)
    st = fused.get_state(model)
    conds = torch.from_numpy(seq["cond_wins"]).float().to(DEV)
    pipe0 = FramePipeline(model, hp, seq, DEV, impl="fused", overlap=False)
    feat, amb, tb = fused.cond_encode_batch(model, st, conds, pipe0.pose6)
    for k in (0, 5, 11):
        f1, a1, t1 = fused._per_frame_vectors(model, st, conds[k], pipe0.pose6[k:k + 1])
        assert torch.equal(f1, feat[k]) and torch.equal(a1, amb[k]) and torch.equal(t1, tb[k])
        assert torch.equal(f1, model.cal_cond_feat(conds[k]).reshape(-1)) or (f1 - model.cal_cond_feat(conds[k]).reshape(-1)).abs().max() < 1e-5
    want = []
    for i in range(12):
        f = pipe0.render_frame(i)
        pipe0.wait()
        want.append(f.clone().numpy())
    for in_flight in (1, 3):
        pipe = FramePipeline(model, hp, seq, DEV, impl="fused", in_flight=in_flight)
        for rep in range(3):
            pipe.prepare(2, 12)                                   # frames 0, 1 stay outside the prepared block: they take the per-frame launch
            assert pipe.prepared(1) is None and pipe.prepared(2) is not None
            for i in range(12):
                fr = pipe.render_frame(i)
                pipe.wait()
                assert np.array_equal(fr.numpy(), want[i]), (in_flight, rep, i)
        pipe2 = FramePipeline(model, hp, seq, DEV, impl="fused", in_flight=in_flight)
        for (i, frame), k in zip(pipe2.stream(range(12)), range(12)):
            assert i == k and np.array_equal(frame, want[k])
        assert pipe2.prepared(0) is not None                       # stream() encoded its block in one launch


def test_ops_path_and_stand_alone_encoders_are_reproducible():
    """run-to-run identity outside the fused head kernel, at sizes that put two and more workgroups on every CU --
    the op-by-op render path at 512x512 (stand-alone marcher, encoders, field, compositor, torso) and the stand-alone encoders on a
    million points.  (The library carries no packed-FP32 instruction any more: tests/test_build_invariants.py.)"""
    from geneface_amd.encoders.freqencoder import FreqEncoder
    from geneface_amd.encoders.gridencoder import GridEncoder
    from geneface_amd.encoders.shencoder import SHEncoder
    hp, sd, model = build(True, "ops")
    fi = frame_inputs(sequence(4, 512, 512), 2)
    first = render_gpu(model, hp, fi)
    for rep in range(5):
        out = render_gpu(model, hp, fi)
        for k in ("rgb_map", "depth_map", "torso_alpha_map", "torso_rgb_map"):
            assert torch.equal(out[k], first[k]), (rep, k)
    g = torch.Generator().manual_seed(3)
    B = 1 << 20
    for D, gridtype in ((3, "hash"), (2, "hash"), (2, "tiled")):
        enc = GridEncoder(input_dim=D, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048, gridtype=gridtype).to(DEV)
        with torch.no_grad():
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, generator=g) * 2 - 1).to(DEV))
            x = (torch.rand(B, D, generator=g) * 2 - 1).to(DEV)
            ref = enc(x, bound=1)
            for rep in range(8):
                assert torch.equal(enc(x, bound=1), ref), (D, gridtype, rep)
    with torch.no_grad():
        d = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).to(DEV)
        sh, fq = SHEncoder().to(DEV), FreqEncoder(input_dim=3, degree=4).to(DEV)
        r1, r2 = sh(d), fq(d)
        for rep in range(8):
            assert torch.equal(sh(d), r1) and torch.equal(fq(d), r2), rep


# ----------------------------------------------------------------------------------------------- round 3: the split tier (precision = 2)
# fp32 VALUES carried as two-term f16 splits on the f16 matrix pipe (frame_head.hip, field_round_split): not fp32 bit patterns, but
# fp32-level accuracy -- so it is held to the STRICT bar of BASELINE.md section 4 (check(): max|d rgb| < 1e-4, PSNR > 65 dB, >= 99.9 % of
# the bytes within 1 LSB, depth < 2e-4, torso maps < 2e-5), on the same frames as the fp32 tier.
def _split_model(torso, sd=None, hp_over=None):
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp, sd0 = model_fixture(torso)
    hp = dict(hp, **(hp_over or {}))
    m = (RADNeRFTorso if torso else RADNeRF)(hp)
    m.load_state_dict(sd if sd is not None else sd0, strict=True)
    m.render_impl, m.render_precision = "fused", "split"
    return hp, (sd if sd is not None else sd0), m.to(DEV).eval()


@pytest.mark.parametrize("torso", [False, True])
@pytest.mark.parametrize("size,idx", [(64, 1), (96, 3)])
def test_split_tier_golden_frames(torso, size, idx):
    hp, sd, model = _split_model(torso)
    fi = frame_inputs(sequence(4, size, size), idx)
    check(render_gpu(model, hp, fi), np.load(os.path.join(GOLD, f"frame_{'torso' if torso else 'head'}_{size}.npz")), torso)


def test_split_tier_256_vs_oracle_and_fp32_tier():
    """Strict bar against the oracle, the reference's n_step schedule replayed exactly (the schedule is a discrete function of which rays
    terminated when: the split arithmetic must not move it), and the distance to the fp32 tier of the same library."""
    from geneface_amd.fused import frame_stats
    hp, sd, model = _split_model(True)
    fi = frame_inputs(sequence(4, 256, 256), 0)
    trace = []
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True, trace=trace)
    out = render_gpu(model, hp, fi)
    check(out, ref, True)
    fs = frame_stats(model.last_ctrl, 256 * 256, hp["max_steps"])
    assert [n for _, n in fs["schedule"]] == [t["n_step"] for t in trace]
    assert fs["budget"] == fs["budget_device"] == sum(t["n_step"] for t in trace)
    for (a, _), t in zip(fs["schedule"], trace):
        assert abs(a - t["n_alive"]) <= max(3, 1e-3 * t["n_alive"]), (a, t["n_alive"])
    _, _, m32 = build(True, "fused")
    o32 = render_gpu(m32, hp, fi)
    assert (o32["rgb_map"] - out["rgb_map"]).abs().max().item() < 5e-5


@pytest.mark.parametrize("seed", [0, 1000])
def test_split_tier_512_head_torso_vs_oracle(seed):
    """configs[2] / configs[4] at the headline shape on the split tier: module API (strict bar) and the pose-mode frame loop (uint8)."""
    from geneface_amd.infer import FramePipeline
    hp, sd, model = _identity_model(seed)
    model.render_precision = "split"
    seq = sequence(4, 512, 512, seed=5 if seed else 0)
    fi = frame_inputs(seq, 1)
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=True)
    check(render_gpu(model, hp, fi), ref, True)
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused")
    frame = pipe.render_frame(1)
    pipe.wait()
    ref8 = (ref["rgb_map"] * 255).view(512, 512, 3).to(torch.uint8)
    d = (frame.int() - ref8.int()).abs()
    assert int(d.max()) <= 1 and (d == 0).float().mean().item() > 0.999
    assert psnr(frame.float() / 255, ref8.float() / 255) > 55


def test_split_tier_edge_cases_and_grid_variants():
    """Empty / full occupancy, the three budget regimes (max_steps 4 / 16 / 64), ragged ray counts, thin density (no early termination:
    every sample of every ray contributes), the other grid index rules -- all at the strict bar."""
    hp, sd = model_fixture(True)
    fi = frame_inputs(sequence(4, 64, 64), 0)
    empty = dict(sd, density_bitfield=torch.zeros_like(sd["density_bitfield"]))
    _, _, m = _split_model(True, empty)
    out = render_gpu(m, hp, fi)
    check(out, R.render(empty, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True), True)
    assert float(out["depth_map"].abs().max()) == 0.0
    hp_h, sd_h = model_fixture(False)
    full = dict(sd_h, density_bitfield=torch.full_like(sd_h["density_bitfield"], 255))
    fi48 = frame_inputs(sequence(4, 48, 48), 2)
    for ms in (4, 16, 64):
        hpm, _, m = _split_model(False, full, dict(max_steps=ms))
        ref = R.render(full, hpm, fi48["rays_o"], fi48["rays_d"], fi48["cond"], fi48["bg_coords"], fi48["pose6"], fi48["bg"], False)
        check(render_gpu(m, hpm, fi48), ref, False)
    idx = torch.linspace(0, 64 * 64 - 1, 37 * 5).long()
    fi_r = dict(fi, rays_o=fi["rays_o"][:, idx].contiguous(), rays_d=fi["rays_d"][:, idx].contiguous(),
                bg_coords=fi["bg_coords"][:, idx].contiguous(), bg=fi["bg"][:, idx].contiguous())
    _, _, m = _split_model(True)
    check(render_gpu(m, hp, fi_r), R.render(sd, hp, fi_r["rays_o"], fi_r["rays_d"], fi_r["cond"], fi_r["bg_coords"], fi_r["pose6"], fi_r["bg"], True), True)
    from geneface_amd import synthetic as S
    thin = S.make_state_dict(hp, True, sigma_row_scale=0.02)
    _, _, m = _split_model(True, thin)
    fi96 = frame_inputs(sequence(4, 96, 96), 2)
    check(render_gpu(m, hp, fi96), R.render(thin, hp, fi96["rays_o"], fi96["rays_d"], fi96["cond"], fi96["bg_coords"], fi96["pose6"], fi96["bg"], True), True)
    for grid_type, interp in (("hashgrid", "linear"), ("hashgrid", "smoothstep"), ("tiledgrid", "smoothstep")):
        hpg, _, m = _split_model(True, None, dict(grid_type=grid_type, grid_interpolation_type=interp))
        check(render_gpu(m, hpg, fi96), R.render(sd, hpg, fi96["rays_o"], fi96["rays_d"], fi96["cond"], fi96["bg_coords"], fi96["pose6"], fi96["bg"], True), True)


def test_split_tier_survives_tiny_and_large_values():
    """The two ends of the f16 range.  Grid tables scaled down by 2^-12 (features ~1e-4 and below -- the reference's own table init is
    U(-1e-4, 1e-4), grid.py:138-140) with the first layers scaled up to compensate: the hi term of such a value is an f16 denormal or zero,
    the scaled lo' term carries it.  And first-layer weights scaled up / hidden activations in the hundreds.  Strict bar both times."""
    hp, sd = model_fixture(True)
    fi = frame_inputs(sequence(4, 96, 96), 1)
    k = 2.0 ** -12
    tiny = dict(sd)
    for name in ("position_embedder.embeddings", "ambient_embedder.embeddings"):
        tiny[name] = sd[name] * k
    tiny["ambient_net.net.0.weight"] = sd["ambient_net.net.0.weight"].clone()
    tiny["ambient_net.net.0.weight"][:, :32] /= k
    tiny["sigma_net.net.0.weight"] = sd["sigma_net.net.0.weight"] / k
    _, _, m = _split_model(True, tiny)
    check(render_gpu(m, hp, fi), R.render(tiny, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True), True)
    big = dict(sd)
    big["sigma_net.net.0.weight"] = sd["sigma_net.net.0.weight"] * 64.0        # hidden activations x64 ...
    big["sigma_net.net.1.weight"] = sd["sigma_net.net.1.weight"] / 64.0        # ... undone by the next layer: the same field, other magnitudes inside
    _, _, m = _split_model(True, big)
    check(render_gpu(m, hp, fi), R.render(big, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], True), True)
    # a weight beyond the f16 range cannot be split: refused at pack time, loudly
    bad = dict(sd)
    bad["color_net.net.0.weight"] = sd["color_net.net.0.weight"].clone()
    bad["color_net.net.0.weight"][3, 20] = 1e5
    _, _, m = _split_model(True, bad)
    with pytest.raises(RuntimeError, match="f16 range"):
        render_gpu(m, hp, fi)


@pytest.mark.parametrize("torso", [False, True])
def test_perturb_first_iteration_jitter_vs_oracle(torso):
    """perturb=True at inference (the GUI's and validation's jitter, renderer.py:338-342): the FIRST march iteration starts every ray at
    near + clamp(near * dt_gamma, dt_min, dt_max) * noise.  With the caller's draws (`perturb_noise`, an extension of the signature: the
    reference always draws inside the wrapper) both paths must reproduce the oracle at the strict bar; round 3 moved this into the fused
    path (k_frame_init), which used to refuse it."""
    hp, sd, model = build(torso, "fused")
    hp = dict(hp, dt_gamma=1.0 / 64)         # a distance-proportional step: the jitter then scales with the ray's near distance
    fi = frame_inputs(sequence(4, 96, 96), 1)
    noise = torch.rand(96 * 96, generator=torch.Generator().manual_seed(4))
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso, perturb_noise=noise)
    plain = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso)
    assert (ref["rgb_map"] - plain["rgb_map"]).abs().max() > 1e-3         # the jitter is visible
    to = lambda t: t.to(DEV)
    for impl in ("ops", "fused"):
        model.render_impl = impl
        out = model.render(to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]), index=0, staged=False,
                           bg_color=to(fi["bg"]), perturb=True, force_all_rays=True, perturb_noise=to(noise), **hp)
        check(out, ref, torso)
    # without caller-supplied draws: fresh noise per call (two renders differ from each other and from the unperturbed frame)
    kw = dict(index=0, staged=False, bg_color=to(fi["bg"]), perturb=True, force_all_rays=True)
    a = model.render(to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]), **kw, **hp)["rgb_map"]
    b = model.render(to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]), **kw, **hp)["rgb_map"]
    assert not torch.equal(a, b) and (a.cpu() - plain["rgb_map"].view_as(a.cpu())).abs().max() > 1e-3
    assert model._pick_impl("auto", True, hp["max_steps"]) == "fused"


@pytest.mark.parametrize("precision", ["fp32", "split"])
def test_full_size_properties_512(precision):
    """Size-independent properties at BASELINE.json's full frame size (the oracle needs ~6 s per 512x512 frame; these need none):
    * rays are independent: a permuted ray batch gives the permuted frame, bit for bit (the pools regroup every ray, the per-frame sample
      budget is a function of the whole batch's terminal-index histogram, which a permutation leaves alone);
    * the background enters affinely: rgb(bg = 1) - rgb(bg = 0) = 1 - weights_sum on all three channels, depth does not move;
    * a camera that looks away from the head sees the (torso-blended) background exactly, depth 0, and no field evaluation at all."""
    from geneface_amd.fused import frame_stats
    hp, sd, model = build(True, "fused")
    model.render_precision = precision
    seq = sequence(4, 512, 512)
    fi = frame_inputs(seq, 3)
    N = 512 * 512
    to = lambda t: t.to(DEV)
    base = render_gpu(model, hp, fi)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(9))
    fi_p = dict(fi, rays_o=fi["rays_o"][:, perm].contiguous(), rays_d=fi["rays_d"][:, perm].contiguous(),
                bg_coords=fi["bg_coords"][:, perm].contiguous(), bg=fi["bg"][:, perm].contiguous())
    out_p = render_gpu(model, hp, fi_p)
    pd = perm.to(DEV)
    for k in ("rgb_map", "depth_map", "torso_alpha_map", "torso_rgb_map"):
        a, b = base[k].reshape(N, -1), out_p[k].reshape(N, -1)
        assert torch.equal(b, a[pd]), (precision, k)
    # head-only model for the background property (the torso pass blends its own colour under the head)
    hp_h, sd_h, head = build(False, "fused")
    head.render_precision = precision
    kw = dict(index=0, staged=False, perturb=False, force_all_rays=True)
    args = (to(fi["rays_o"]), to(fi["rays_d"]), to(fi["cond"]), to(fi["bg_coords"]), to(fi["pose6"]))
    r0 = head.render(*args, bg_color=torch.zeros(1, N, 3, device=DEV), **kw, **hp_h)
    r1 = head.render(*args, bg_color=torch.ones(1, N, 3, device=DEV), **kw, **hp_h)
    assert torch.equal(r0["depth_map"], r1["depth_map"])
    du = (r1["rgb_map"] - r0["rgb_map"]).reshape(N, 3)                     # = 1 - weights_sum (colours <= 1, so image + 1 - weights_sum never clamps)
    assert float((du - du[:, :1]).abs().max()) < 2e-6 and float(du.min()) > -2e-6 and float(du.max()) <= 1.0 + 2e-6
    assert float(du.max()) == 1.0 and float(du.min()) < 0.01                 # rays that miss the head entirely, rays that saturate
    # looking away: ngp pose rotated by 180 degrees about the image-up axis
    away = fi["pose44"].clone()
    away[0, :3, 0] *= -1
    away[0, :3, 2] *= -1
    ro, rd = R.get_rays(away, seq["intrinsics"], 512, 512)
    out_a = model.render(to(ro), to(rd), to(fi["cond"]), to(fi["bg_coords"]), to(R.convert_poses(away)), bg_color=to(fi["bg"]), **kw, **hp)
    fs = frame_stats(model.last_ctrl, N, hp["max_steps"])
    assert fs["n_hit"] == 0 and sum(fs["samples"]) == 0
    assert float(out_a["depth_map"].abs().max()) == 0.0
    assert torch.equal(out_a["rgb_map"].reshape(N, 3), out_a["torso_rgb_map"].reshape(N, 3).clamp(0, 1))


def test_schedule_boundary_one_ray_changes_every_budget():
    """VERDICT r4 weak #7.  The reference's march schedule is a STEP FUNCTION of integer counts: n_step = clamp(N // n_alive, 1, 8)
    (renderer.py:338), so one ray more or less -- in N, or among the alive ones: a ray whose transmittance ends an iteration within rounding
    of T_thresh -- flips n_step wherever N is an exact multiple of n_alive, and with it the sample budget of EVERY surviving ray.  That is a
    property of the reference (its CUDA run with __expf and a CPU run with expf can disagree on such a ray, like they disagree on a grazing
    ray); what the product owes is to follow ITS alive counts exactly as the reference's loop would.  Constructed here: a 128 x 128 frame
    padded with rays that miss everything (they die in iteration 0 and only count in N) until N = k * n_alive at one iteration, and the same
    frame with one padding ray less -- the two schedules differ at that iteration, the budgets differ, and on BOTH sides of the boundary the
    fused path's replayed schedule, the op-by-op loop's schedule and the oracle's agree step for step, with every pixel inside 1e-4."""
    from geneface_amd.fused import frame_stats
    hp, sd, model = build(False, "fused")
    size = 128
    fi = frame_inputs(sequence(4, size, size), 2)
    N0 = size * size
    trace = []
    R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=False, trace=trace)
    # an iteration whose alive count a gives k = N0 // a + 1 in 2..7 (inside the clamp), so that N = k * a sits exactly on the boundary
    j, a = next((j, t["n_alive"]) for j, t in enumerate(trace) if j >= 1 and t["n_alive"] > 0 and 2 <= N0 // t["n_alive"] + 1 <= 7)
    k = N0 // a + 1
    schedules, budgets = {}, {}
    for name, N in (("on", k * a), ("below", k * a - 1)):
        pad = N - N0
        assert 0 <= pad
        away = -fi["rays_d"].reshape(-1, 3)[:1]                                   # looks away from the head: no sample, ever
        ro = torch.cat([fi["rays_o"].reshape(-1, 3), fi["rays_o"].reshape(-1, 3)[:1].expand(pad, 3)]).reshape(1, N, 3).contiguous()
        rd = torch.cat([fi["rays_d"].reshape(-1, 3), away.expand(pad, 3)]).reshape(1, N, 3).contiguous()
        bgc = torch.cat([fi["bg_coords"].reshape(-1, 2), torch.zeros(pad, 2)]).reshape(1, N, 2)
        bg = torch.cat([fi["bg"].reshape(-1, 3), torch.full((pad, 3), 0.25)]).reshape(1, N, 3)
        f2 = dict(fi, rays_o=ro, rays_d=rd, bg_coords=bgc, bg=bg)
        tr = []
        ref = R.render(sd, hp, ro, rd, fi["cond"], bgc, fi["pose6"], bg, torso=False, trace=tr)
        assert tr[j]["n_alive"] == a and tr[j]["n_step"] == (k if name == "on" else k - 1)
        model.render_impl = "fused"
        out = render_gpu(model, hp, f2)
        fs = frame_stats(model.last_ctrl, N, hp["max_steps"])
        model.render_impl = "ops"
        out_ops = render_gpu(model, hp, f2)
        want = [t["n_step"] for t in tr]
        assert [n for _, n in fs["schedule"]] == want == [s for _, s in model.last_schedule], (name, fs["schedule"], want, model.last_schedule)
        assert fs["budget"] == fs["budget_device"] == sum(want)
        for o in (out, out_ops):
            assert (o["rgb_map"].cpu() - ref["rgb_map"]).abs().max().item() < RGB_ATOL
            assert torch.equal(o["rgb_map"].cpu().reshape(-1, 3)[N0:], bg.reshape(-1, 3)[N0:])      # the padding rays: background, exactly
        schedules[name], budgets[name] = want, sum(want)
    assert schedules["on"][j] == k and schedules["below"][j] == k - 1 and schedules["on"][:j] == schedules["below"][:j]
    assert schedules["on"] != schedules["below"]
    print(f"schedule boundary at iteration {j}: n_alive {a}, N = {k} x {a}: {schedules['on']} (budget {budgets['on']}) vs one ray less: "
          f"{schedules['below']} (budget {budgets['below']})")


def test_torso_mask_list_is_rebuilt_when_its_inputs_change():
    """Round 5: the frame loop builds the torso mask's dense list once (fused.torso_mask_list: a property of bg_coords, the torso occupancy and
    its threshold) instead of once per frame.  The cached list and the per-frame list give the same bytes (the module API on the kernel's own
    rays builds it per call), and an in-place change of the occupancy -- what RADNeRFTorso.update_extra_state does -- or of the threshold is
    seen by the very next frame."""
    from geneface_amd.fused import get_state
    from geneface_amd.infer import FramePipeline
    hp, sd, model = build(True, "fused")
    seq = sequence(4, 256, 256)
    pipe = FramePipeline(model, hp, seq, DEV, impl="fused", in_flight=2)

    def both(i):
        u8 = pipe.render_frame(i)
        pipe.wait()
        u8 = u8.clone().reshape(-1, 3)
        api = (pipe.run_model(pipe.kernel_sample(i))["rgb_map"].reshape(-1, 3) * 255).to(torch.uint8).cpu()
        assert torch.equal(u8, api), int((u8 != api).sum())
        return u8
    a = both(1)
    entry = get_state(model)._mask_list
    n_masked = int(entry[1][2].item())
    assert 0 < n_masked < 256 * 256
    lst, dense_of = entry[1][0][:n_masked].cpu().long(), entry[1][1].cpu().long()
    assert torch.equal(dense_of[lst], torch.arange(n_masked)) and int((dense_of >= 0).sum()) == n_masked      # a permutation and its inverse
    mask_torch = model.torso_mask(pipe.bg_coords.reshape(-1, 2)).cpu()
    assert int((mask_torch != (dense_of >= 0)).sum()) <= 8                       # torch's grid_sample may round a threshold pixel the other way
    both(2)
    assert get_state(model)._mask_list is entry                                   # second frame: the same list, no new launch
    with torch.no_grad():
        model.density_grid_torso.view(128, 128)[:, 96:] = 0.0                     # in place: the lower quarter of the picture loses its torso
    b = both(1)
    assert get_state(model)._mask_list is not entry and int(get_state(model)._mask_list[1][2].item()) < n_masked
    assert not torch.equal(a, b)
    model.density_thresh_torso, model.mean_density_torso = 0.5, 1.0               # threshold = min of the two (radnerf_torso.py:170): part of the key as well
    c = both(1)
    assert not torch.equal(b, c)


def test_fused_torso_takes_any_finite_background_coordinate():
    """gf_render_torso is a public entry.  Coordinates far outside the picture (|bg_coords * torso_shrink| > 15.9: the bounded-argument
    sine of the fused encodings no longer applies) can only reach the field when the mask threshold is negative -- zero padding puts the
    occupancy sample of such a pixel at 0 -- and then the fused path must give what the op-by-op path (full-range sine) gives, not finite
    nonsense.  A wave with one such pixel takes the full-range sine for all of its pixels: same bits for the in-range ones."""
    hp, sd, fused = build(True, "fused")
    _, _, ops = build(True, "ops")
    fi = frame_inputs(sequence(2, 64, 64), 1)
    far = fi["bg_coords"].clone()
    far[:, 5::7] *= 37.0           # every seventh pixel far outside, the others where they were
    for m in (fused, ops):
        m.density_thresh_torso = -1.0          # min(density_thresh_torso, mean_density_torso): every pixel is masked
    outs = {}
    for name, m in (("fused", fused), ("ops", ops)):
        for tag, bgc in (("near", fi["bg_coords"]), ("far", far)):
            with torch.no_grad():
                outs[name, tag] = render_gpu(m, hp, dict(fi, bg_coords=bgc))
    for k, tol in (("torso_alpha_map", 2e-5), ("torso_rgb_map", 2e-5), ("rgb_map", 1e-4)):
        for tag in ("near", "far"):
            a, b = outs["fused", tag][k], outs["ops", tag][k]
            assert torch.isfinite(a).all()
            assert (a - b).abs().max().item() < tol, (k, tag, (a - b).abs().max().item())
    # the pixels that stayed where they were: the far frame's waves took the other sine for them, the bits are the near frame's
    keep = torch.ones(64 * 64, dtype=torch.bool)
    keep[5::7] = False
    assert torch.equal(outs["fused", "near"]["torso_alpha_map"].reshape(-1)[keep], outs["fused", "far"]["torso_alpha_map"].reshape(-1)[keep])


@pytest.mark.parametrize("precision,impl", [("fp32", "fused"), ("fp32", "ops"), ("split", "fused")])
@pytest.mark.parametrize("tag", ["hash", "hash_smoothstep", "smoothstep", "head_aware_coin_heads", "head_aware_coin_tails", "audio"])
def test_variant_frame_vs_reference_golden(tag, precision, impl, monkeypatch):
    """The product against frames the REFERENCE'S OWN PYTHON rendered for the other configurations it ships (tests/golden/frame_variant_*_48.npz,
    make_golden.py::golden_variants): hashed grids, smoothstep, the head-aware torso with either outcome of its coin, the audio-driven config on
    the second identity -- fused, op-by-op and split tier, the golden fixtures' bars."""
    import random
    from test_oracle_golden import variant_case
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp, sd, fi, branch, gold = variant_case(tag)
    m = RADNeRFTorso(hp)
    m.load_state_dict(sd, strict=True)
    m.render_impl, m.render_precision = impl, precision
    m = m.to(DEV).eval()
    monkeypatch.setattr(random, "random", lambda: 0.25 if branch else 0.75)
    out = render_gpu(m, hp, fi)
    check(out, gold, True)
    assert (out["deform"].cpu().numpy() - gold["deform"]).shape == gold["deform"].shape and np.abs(out["deform"].cpu().numpy() - gold["deform"]).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("torso", [False, True])
def test_everything_the_task_calls_under_its_autocast_is_the_fp32_result(torso, dtype):
    """The reference wraps every entry of this path in torch.autocast (the Trainer around _training_step, which calls
    update_extra_state: utils/commons/trainer.py:326 + tasks/radnerfs/radnerf.py:185-194; validation, test and both GUI entries:
    radnerf.py:359, 390).  The fused path computes in fp32 whatever that state (its tiers are render_precision's, not autocast's): each call
    under fp16 autocast must give the bytes it gives outside it -- torch glue between the launches (torch.mv bias folds, the state packing,
    the occupancy update's reductions) must not slip into half (or, should somebody train under bfloat16 autocast, into that)."""
    from geneface_amd import gui
    from geneface_amd.infer import FramePipeline
    seq = sequence(4, 64, 64)
    fi = frame_inputs(seq, 1)
    cam = gui.OrbitCamera(64, 64, r=3.35, fovy=21.24)
    cam.update_intrinsics(seq["intrinsics"])
    cam.orbit(30, 10)
    res = {}
    for amp in (False, True):
        hp, sd, model = build(torso, "fused")        # a fresh model: the packed state is built inside the autocast region too
        got = {}
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=amp):
            out = render_gpu(model, hp, fi)
            got["render"] = [out[k].float().cpu() for k in sorted(out) if torch.is_tensor(out[k])]
            g = gui.test_gui_with_editable_data(model, hp, cam.pose, cam.intrinsics, 64, 64, torch.from_numpy(seq["cond_wins"][2]), 0,
                                                torch.from_numpy(seq["bg_img"]).view(1, -1, 3), 1, 0.5, DEV)
            got["gui"] = [torch.from_numpy(g["image"]), torch.from_numpy(g["depth"])]
            pipe = FramePipeline(model, hp, seq, DEV)
            got["pipe"] = []
            for i in range(3):
                frame = pipe.render_frame(i)
                pipe.wait(frame)          # the pinned host frame is valid once its copy event has passed
                got["pipe"].append(frame.clone())
        # the training-time occupancy update, called from inside the Trainer's autocast region with gradients enabled
        import random
        model.train()
        random.seed(3)
        gen = torch.Generator(device=DEV).manual_seed(5)
        poses = torch.from_numpy(seq["poses"]).to(DEV)
        with torch.autocast("cuda", dtype=dtype, enabled=amp):
            if torso:
                model.poses = poses
                for _ in range(2):
                    model.update_extra_state(generator=gen)
                got["occupancy"] = [model.density_grid_torso.float().cpu().clone(), torch.tensor(float(model.mean_density_torso))]
            else:
                model.conds = torch.from_numpy(seq["cond_wins"][:, seq["cond_wins"].shape[1] // 2]).to(DEV)
                model.density_grid.zero_()
                model.mark_untrained_grid(poses, seq["intrinsics"])
                for _ in range(2):
                    model.update_extra_state(generator=gen)
                got["occupancy"] = [model.density_grid.float().cpu().clone(), model.density_bitfield.cpu().clone(),
                                    torch.tensor(float(model.mean_density))]
        res[amp] = got
    for name in res[False]:
        assert len(res[False][name]) == len(res[True][name]) > 0
        for a, b in zip(res[False][name], res[True][name]):
            assert a.dtype == b.dtype and a.shape == b.shape, name
            assert torch.equal(a, b), (name, float((a.double() - b.double()).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("torso", [False, True])
def test_a_halved_model_renders_the_fast_tier(torso):
    """The viewer's `nerf_task.half()` under `amp` (inference/nerfs/radnerf_gui.py:604-605) followed by its autocast render
    (tasks/radnerfs/radnerf.py:388-392): the bytes of `render_precision = "fast"`, within the fast tier's bars of the oracle."""
    hp, sd, fast = _fast_model(torso)
    fi = frame_inputs(sequence(4, 64, 64), 1)
    want = render_gpu(fast, hp, fi)
    _, _, model = build(torso, "fused")

    class Task(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.model = m

    task = Task(model).half()
    assert model.render_precision == "fast" and all(p.dtype == torch.float32 for p in model.parameters())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        got = render_gpu(task.model, hp, fi)
    assert torch.equal(got["rgb_map"], want["rgb_map"]) and torch.equal(got["depth_map"], want["depth_map"])
    ref = R.render(sd, hp, fi["rays_o"], fi["rays_d"], fi["cond"], fi["bg_coords"], fi["pose6"], fi["bg"], torso=torso)
    _check_fast(got, ref)
    task.float()
    exact = render_gpu(model, hp, fi)
    check(exact, ref, torso)                                              # back on the exact tier: the strict bars


@pytest.mark.gpu
def test_a_model_that_has_rendered_can_be_copied_and_pickled(tmp_path):
    """copy.deepcopy (an EMA copy, a snapshot before fine-tuning) and torch.save(model) after the first frame: the packed device state of the
    fused path (ctypes structures, workspaces) stays with the original; the copy renders the same bytes from a state of its own."""
    import copy
    hp, sd, model = build(True, "fused")
    fi = frame_inputs(sequence(4, 64, 64), 2)
    want = render_gpu(model, hp, fi)
    assert hasattr(model, "_fused_state")
    twin = copy.deepcopy(model)
    assert not hasattr(twin, "_fused_state")
    got = render_gpu(twin, hp, fi)
    assert torch.equal(got["rgb_map"], want["rgb_map"]) and twin._fused_state is not model._fused_state
    with torch.no_grad():                       # the twin's packed copies follow the twin's weights only
        twin.sigma_net.net[0].weight.mul_(1.5)
    assert not torch.equal(render_gpu(twin, hp, fi)["rgb_map"], want["rgb_map"])
    assert torch.equal(render_gpu(model, hp, fi)["rgb_map"], want["rgb_map"])
    path = tmp_path / "model.pt"
    torch.save(model, path)
    loaded = torch.load(path, weights_only=False)
    assert torch.equal(render_gpu(loaded, hp, fi)["rgb_map"], want["rgb_map"])


@pytest.mark.gpu
def test_inference_mode_gives_the_no_grad_frames():
    """torch.inference_mode() instead of the reference's torch.no_grad(): tensors made inside it carry no version counter, so everything this
    package caches by `_version` (torso mask list, prepared condition batches, the head-aware coin's mask) is rebuilt instead of trusted."""
    from geneface_amd.infer import FramePipeline
    hp, sd, model = build(True, "fused")
    seq = sequence(4, 64, 64)
    fi = frame_inputs(seq, 1)
    with torch.no_grad():
        want = render_gpu(model, hp, fi)
        pipe = FramePipeline(model, hp, seq, DEV)
        frames = []
        for i in range(3):
            f = pipe.render_frame(i)
            pipe.wait(f)
            frames.append(f.clone())
    _, _, fresh = build(True, "fused")
    with torch.inference_mode():
        got = render_gpu(fresh, hp, fi)
        again = render_gpu(fresh, hp, fi)
        pipe = FramePipeline(fresh, hp, seq, DEV)
        pipe.prepare(0, 3)
        for i in range(3):
            f = pipe.render_frame(i)
            pipe.wait(f)
            assert torch.equal(f, frames[i]), i
    for k in ("rgb_map", "depth_map", "torso_alpha_map"):
        assert torch.equal(got[k], want[k]) and torch.equal(again[k], want[k]), k
