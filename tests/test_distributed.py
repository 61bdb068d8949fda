"""World-size-2 CPU (gloo) coverage of the multi-GPU path: frame sharding (base_nerf_infer.py:150-155) and the one
collective of the render path, the flattened weight broadcast (the DDP constructor's implicit broadcast, :126,145).
The GPU bench launches the same code with backend nccl (= RCCL); nothing here needs a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from geneface_amd import hparams as HP
from geneface_amd.infer import broadcast_model_, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("T,W", [(100, 1), (100, 2), (272, 8), (3000, 8), (7, 8), (5, 2)])
def test_shard_range_partitions_all_frames(T, W):
    """Contiguous blocks of T // W, last rank takes the remainder: every frame exactly once, in order."""
    seen = []
    for r in range(W):
        a, b = shard_range(T, r, W)
        assert 0 <= a <= b <= T
        if r < W - 1:
            assert b - a == T // W
        seen.extend(range(a, b))
    assert seen == list(range(T))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp = HP.may_hparams(True)
    torch.manual_seed(1234 + rank)  # replicas start DIFFERENT: only rank 0's weights must survive
    model = RADNeRFTorso(hp)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(float(rank))
        model.density_bitfield.fill_(rank + 1)
        model.density_grid_torso.fill_(0.25 * (rank + 1))
    broadcast_model_(model, src=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    torch.save(sd, os.path.join(out_dir, f"sd_{rank}.pt"))
    # every rank renders its own contiguous block and nothing else: gather the block bounds
    T = 101
    a, b = shard_range(T, rank, world)
    bounds = [None] * world
    dist.all_gather_object(bounds, (a, b))
    if rank == 0:
        covered = [i for lo, hi in bounds for i in range(lo, hi)]
        assert covered == list(range(T))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sd0 = torch.load(tmp_path / "sd_0.pt")
    sd1 = torch.load(tmp_path / "sd_1.pt")
    assert sd0.keys() == sd1.keys()
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
    # rank 0's distinguishing marks won (uint8 bitfield and fp32 buffers ride in the same broadcast path)
    assert int(sd1["density_bitfield"][0]) == 1
    assert float(sd1["density_grid_torso"].flatten()[0]) == 0.25


def test_broadcast_is_noop_without_process_group():
    from geneface_amd.radnerf import RADNeRF
    m = RADNeRF(HP.may_hparams(False))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    assert broadcast_model_(m) is m
    for k, v in m.state_dict().items():
        assert torch.equal(before[k], v)


def test_forward_system_spawns_one_rank_per_gpu_world2_gloo(tmp_path, monkeypatch):
    """The entry point's own fan-out (base_nerf_infer.py:131-193) on two spawned gloo ranks with a CPU stand-in for the frame pipeline:
    every `%05d.png` index is written exactly once, by the rank that owns it (:150-155), from rank 0's weights (the replicas are built
    fresh in every process and receive them through the one broadcast), in frame order; the parent gets the frames back in order."""
    import numpy as np
    from helpers import StubPipeline
    from geneface_amd import synthetic as S
    from geneface_amd.lm3d_radnerf_infer import LM3d_RADNeRFInfer, RADNeRFPoseSource
    from geneface_amd.png import decode_rgb8
    from geneface_amd.radnerf_torso import RADNeRFTorso
    from test_entry_point import _ds_dict
    hp = HP.may_hparams(True)
    T, H, W = 11, 16, 16
    dd, _ = _ds_dict(T=T, H=H, W=W)
    model = RADNeRFTorso(hp)
    model.load_state_dict(S.make_state_dict(hp, True, seed=7), strict=True)
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "0,1")          # what the reference reads to decide on the fan-out (:63-69)
    inf = LM3d_RADNeRFInfer(hp, model=model, dataset=RADNeRFPoseSource(dd, hp), device="cpu")
    assert inf.num_gpus == 2 and inf.use_ddp
    inf.pipeline_cls = StubPipeline
    lm = S.make_landmarks(T).astype(np.float32)
    cond_path = str(tmp_path / "lm.npy")
    np.save(cond_path, lm[None])
    imgs = str(tmp_path / "imgs")
    frames = inf.infer_once({"cond_name": cond_path, "out_video_name": str(tmp_path / "out.npy"), "audio_source_name": "", "tmp_imgs_dir": imgs})
    assert frames.shape == (T, H, W, 3)
    assert sorted(os.listdir(imgs)) == [f"{i:05d}.png" for i in range(T)]            # every index exactly once, nothing else
    want = StubPipeline(model, hp, {"H": H, "W": W, "cond_wins": np.stack([s["cond_wins"] for s in inf.get_cond_from_input({"cond_name": cond_path})])},
                        "cpu", frames=(0, T))
    for (k, ref), i in zip(want.stream(range(T)), range(T)):
        png = decode_rgb8(open(os.path.join(imgs, f"{i:05d}.png"), "rb").read())
        np.testing.assert_array_equal(png, ref)                  # global index, rank-0 weights' checksum, this frame's condition
        np.testing.assert_array_equal(frames[i], ref)
    np.testing.assert_array_equal(np.load(str(tmp_path / "out.npy")), frames)
    # world_size=1 through the same entry point gives the same files
    inf1 = LM3d_RADNeRFInfer(hp, model=model, dataset=RADNeRFPoseSource(dd, hp), device="cpu")
    inf1.pipeline_cls, inf1.use_ddp = StubPipeline, False
    frames1 = inf1.infer_once({"cond_name": cond_path, "out_video_name": "", "audio_source_name": "", "tmp_imgs_dir": str(tmp_path / "imgs1")})
    np.testing.assert_array_equal(frames1, frames)


def _torchrun_style_rank(rank, world, port, out_dir, cond_path, T, H, W):
    """What a rank of `torchrun ... lm3d_radnerf_infer` does: the process group exists BEFORE the entry point is constructed."""
    import numpy as np
    from helpers import StubPipeline
    from geneface_amd import synthetic as S
    from geneface_amd.lm3d_radnerf_infer import LM3d_RADNeRFInfer, RADNeRFPoseSource
    from geneface_amd.radnerf_torso import RADNeRFTorso
    from test_entry_point import _ds_dict
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hp = HP.may_hparams(True)
    dd, _ = _ds_dict(T=T, H=H, W=W)
    model = RADNeRFTorso(hp)
    model.load_state_dict(S.make_state_dict(hp, True, seed=7 + rank), strict=True)     # replicas differ until the broadcast
    inf = LM3d_RADNeRFInfer(hp, model=model, dataset=RADNeRFPoseSource(dd, hp), device="cpu")
    inf.pipeline_cls = StubPipeline
    out_name = os.path.join(out_dir, "out.npy")
    frames = inf.infer_once({"cond_name": cond_path, "out_video_name": out_name, "audio_source_name": "", "tmp_imgs_dir": os.path.join(out_dir, "imgs")})
    np.save(os.path.join(out_dir, f"ret_{rank}.npy"), frames)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,world", [(11, 2), (272, 8)])
def test_forward_system_under_an_existing_process_group_gathers_on_rank0(tmp_path, T, world):
    """under torchrun rank 0 used to write ITS block to out_video_name as if it were the sequence.  Now both multi-GPU entry
    paths return the same thing on rank 0: every frame, in order, rendered from rank 0's weights.  (272, 8): the length of the reference's
    demo audio (zozo.wav) on a full node -- 34 frames per rank, none left over; the uneven case is (11, 2); VERDICT r4 next #7."""
    import numpy as np
    from geneface_amd import synthetic as S
    H, W = (16, 16) if world == 2 else (8, 8)
    cond_path = str(tmp_path / "lm.npy")
    np.save(cond_path, S.make_landmarks(T).astype(np.float32)[None])
    mp.spawn(_torchrun_style_rank, args=(world, _free_port(), str(tmp_path), cond_path, T, H, W), nprocs=world, join=True)
    full = np.load(tmp_path / "out.npy")
    assert full.shape == (T, H, W, 3)
    assert [int(full[i, 0, 0, 0]) for i in range(T)] == [i % 256 for i in range(T)]   # StubPipeline encodes the global frame index in channel 0
    assert len({int(full[i, 0, 0, 1]) for i in range(T)}) == 1                     # one weight checksum: rank 0's, on both ranks' frames
    np.testing.assert_array_equal(np.load(tmp_path / "ret_0.npy"), full)           # rank 0 returns the whole sequence
    for r in range(1, world):                                                       # the other ranks their own blocks
        a, b = shard_range(T, r, world)
        np.testing.assert_array_equal(np.load(tmp_path / f"ret_{r}.npy"), full[a:b])
    assert sorted(os.listdir(tmp_path / "imgs")) == [f"{i:05d}.png" for i in range(T)]
