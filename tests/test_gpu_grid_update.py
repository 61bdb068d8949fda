"""-m gpu: occupancy-grid maintenance -- NeRFRenderer.update_extra_state / mark_untrained_grid and
RADNeRFTorso.update_extra_state over the HIP field, Morton, dilation and packbits ops, against the CPU oracle with the same
condition window and the same (CPU-generated) cell jitter."""
import numpy as np
import pytest
import torch

from geneface_amd import hparams as HP
from geneface_amd import synthetic as S
from oracle import radnerf_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = 32   # a 32^3 grid keeps the CPU oracle at a second; the code paths are size independent


def _model(torso=True, G=G):
    from geneface_amd.radnerf import RADNeRF
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = dict(HP.may_hparams(torso), grid_size=G)
    sd = S.make_state_dict(HP.may_hparams(torso), torso)
    model = (RADNeRFTorso if torso else RADNeRF)(hp)
    skip = {"density_grid", "density_bitfield"} | ({"density_grid_torso"} if torso else set())
    missing = model.load_state_dict({k: v for k, v in sd.items() if k not in skip}, strict=False)
    assert set(missing.missing_keys) == skip and not missing.unexpected_keys
    return hp, sd, model.to(DEV).eval()


def test_update_extra_state_head_vs_oracle():
    hp, sd, model = _model(torso=False)   # RADNeRFTorso overrides the method: it only refreshes the torso grid (radnerf_torso.py:200-203)
    cond = torch.randn(5, 1, 204, generator=torch.Generator().manual_seed(2))
    grid0 = torch.zeros(1, G ** 3)
    grid0[0, ::7] = -1.0     # cells marked untrained stay untouched
    grid0[0, 1::7] = 3.0     # cells with history decay towards the new sample
    model.density_grid.copy_(grid0.to(DEV))
    model.update_extra_state(cond=cond.to(DEV), generator=torch.Generator().manual_seed(11))
    ref_grid, ref_mean, ref_bits = R.update_density_grid(sd, hp, grid0, cond, torch.Generator().manual_seed(11))
    got = model.density_grid.cpu()
    assert torch.equal(got < 0, ref_grid < 0)
    rel = (got - ref_grid).abs() / ref_grid.abs().clamp(min=1e-2)
    assert rel.max().item() < 5e-3, rel.max().item()              # exp() of the density head amplifies 1e-6 logit noise
    assert abs(model.mean_density - ref_mean) < 1e-3 * max(ref_mean, 1e-3)
    # the bitfield is exactly the packing of the model's own grid (LSB first, raymarching.cu:268-289) ...
    own = torch.zeros_like(ref_bits)
    R.RM.packbits(got.contiguous(), own.numel(), min(model.mean_density, hp["density_thresh"]), own)
    assert torch.equal(model.density_bitfield.cpu(), own)
    # ... and differs from the oracle's only in cells whose density sits at the threshold
    diff = np.unpackbits((model.density_bitfield.cpu() ^ ref_bits).numpy()).sum()
    assert diff <= 1e-3 * G ** 3, diff
    assert model.iter_density == 1
    # the fused path re-derives its occupancy bounding box after the update
    assert not hasattr(model, "_fused_state")


def test_update_extra_state_device_path_vs_ops_full_grid():
    """The full 128^3 grid: the three-launch device path (density head on the matrix pipe -> dilation + EMA-max + mean -> packbits)
    against the op-by-op route (grid-encode ops + library GEMMs through `density()`) on the same jitter; smaller S only changes how the
    reference's loop would consume the jitter stream, which `_cell_jitter` reproduces."""
    import time
    hp, sd, model = _model(torso=False, G=128)
    cond = torch.randn(5, 1, 204, generator=torch.Generator().manual_seed(5)).to(DEV)
    grid0 = torch.zeros(1, 128 ** 3)
    grid0[0, ::11] = -1.0
    grid0[0, 3::11] = 2.0
    noise = model._cell_jitter(64, torch.Generator().manual_seed(21))
    a = model._density_grid_fused(cond, noise)
    b = model._density_grid_ops(cond, noise)
    rel = ((a - b).abs() / b.abs().clamp(min=1e-2)).max().item()
    assert rel < 5e-3, rel
    model.density_grid.copy_(grid0.to(DEV))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.update_extra_state(cond=cond, generator=torch.Generator().manual_seed(21), S=64)
    torch.cuda.synchronize()
    print(f"update_extra_state 128^3 (incl. host-side jitter draw): {(time.perf_counter() - t0) * 1e3:.1f} ms")
    got = model.density_grid.cpu()
    assert torch.equal(got < 0, grid0 < 0)
    own = torch.zeros(128 ** 3 // 8, dtype=torch.uint8)
    R.RM.packbits(got.contiguous(), own.numel(), min(model.mean_density, hp["density_thresh"]), own)
    assert torch.equal(model.density_bitfield.cpu(), own)
    assert abs(model.mean_density - got.clamp(min=0).double().mean().item()) < 1e-6 * max(1.0, model.mean_density)
    # run-to-run identical (the mean is reduced in a fixed order)
    bits1 = model.density_bitfield.clone()
    model.density_grid.copy_(grid0.to(DEV))
    model.update_extra_state(cond=cond, generator=torch.Generator().manual_seed(21), S=64)
    assert torch.equal(model.density_bitfield, bits1)


def test_update_extra_state_torso_vs_oracle():
    hp, sd, model = _model()
    p6 = torch.tensor([[0.05, -0.03, 0.02, 0.01, 3.3, -0.02]])
    grid0 = torch.rand(G * G, generator=torch.Generator().manual_seed(4)) * 0.2
    model.density_grid_torso.copy_(grid0.to(DEV))
    model.update_extra_state(pose6=p6.to(DEV), generator=torch.Generator().manual_seed(12))
    ref_grid, ref_mean = R.update_density_grid_torso(sd, hp, grid0, p6, sd["torso_individual_codes"][0], torch.Generator().manual_seed(12))
    assert (model.density_grid_torso.cpu() - ref_grid).abs().max().item() < 2e-4
    assert abs(model.mean_density_torso - ref_mean) < 1e-4


def test_mark_untrained_grid_vs_oracle():
    hp, sd, model = _model(torso=False)
    seq = S.make_sequence(6, 64, 64, hp)
    poses = torch.from_numpy(seq["poses"]).float()
    model.density_grid.zero_()
    model.mark_untrained_grid(poses, seq["intrinsics"], S=16)
    ref = R.mark_untrained_grid(hp, torch.zeros(1, G ** 3), poses, seq["intrinsics"], S=16)
    got = model.density_grid.cpu()
    # a cell on the edge of a frustum may flip with the matmul's rounding: allow a handful
    assert (got != ref).sum().item() <= 8
    assert 0 < (got < 0).sum().item() < G ** 3
