"""CPU (gloo, world size 2): bench.py's N > 1 control flow -- the part of the scaling bench that no 1-GPU box can execute.

The same `run_rank` the GPU launch runs, over gloo with a stub pipeline: shard ranges, the all-reduced rank count, every rank deriving
the same repeat count from the max-reduced first pass, EXACTLY K frames per pass, the batched encoder call inside the pass, the
max-over-ranks time, and ONE JSON line from rank 0 only.  Plus the self fan-out (`python bench.py --gpus N` without a launcher)."""
import json
import os
import socket
import sys
import time

import pytest
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _StubPipe:
    """Stands in for FramePipeline: remembers what it was asked to do; a frame costs `cost` seconds of host time."""
    in_flight = 3

    def __init__(self, frames, cost):
        self.frames, self.cost = frames, cost
        self.passes, self.cur, self.prepared = [], None, []

    def prepare(self, first, stop):
        self.prepared.append((first, stop))
        self.cur = []
        self.passes.append(self.cur)

    def render_frame(self, i):
        if self.cur is not None:
            self.cur.append(i)
        time.sleep(self.cost)

    def wait(self):
        pass


def _rank(rank, world, port, out_dir, gpus_flag):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    args = bench.parse(["--gpus", str(gpus_flag), "--steps", "6", "--warmup", "2", "--min-seconds", "0.2"])
    pipes, lines = [], []

    def make_pipe(a, job, frames):
        pipes.append(_StubPipe(frames, cost=0.002 if job.rank == 0 else 0.006))     # rank 1 is 3x slower: the line must carry ITS time
        return pipes[-1]
    try:
        bench.run_rank(args, backend="gloo", make_pipe=make_pipe, emit=lines.append)
    except SystemExit as e:
        json.dump({"exit": str(e)}, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))
        return
    p = pipes[0]
    json.dump({"frames": list(p.frames), "passes": p.passes, "prepared": p.prepared, "lines": lines}, open(os.path.join(out_dir, f"rank{rank}.json"), "w"))


def test_bench_control_flow_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path), world), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world)]
    # contiguous blocks of K + W frames per rank (base_nerf_infer.py:150-155), disjoint, covering
    assert r[0]["frames"] == [0, 8] and r[1]["frames"] == [8, 16]
    # rank 0 alone prints, once
    assert len(r[0]["lines"]) == 1 and r[1]["lines"] == []
    line = r[0]["lines"][0]
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["config"]["rccl_ranks"] == 2 and line["config"]["collective_backend"] == "gloo"
    assert line["config"]["frames_total"] == 12 and "frame-shard x2" in line["config"]["parallelism"]
    # every rank ran the same number of passes (a disagreement would have dead-locked in the barrier), each EXACTLY K frames of its own shard,
    # local indices [W, W + K), the batched encoder call inside the pass
    assert len(r[0]["passes"]) == len(r[1]["passes"]) == line["repeats"] >= 2
    for rk in r:
        assert all(p == list(range(2, 8)) for p in rk["passes"])
        assert all(tuple(x) == (2, 8) for x in rk["prepared"])
    # the time is the MAX over ranks: at least the slow rank's 6 x 6 ms, and value = all ranks' frames / that time
    assert line["ms_per_step"] >= 6.0
    assert abs(line["value"] - 2 * 6 / (line["ms_per_step"] * 6 / 1e3)) < 1e-6 * line["value"]
    assert line["higher_is_better"] is True and line["vs_baseline"] is None
    # every rank's own clock and shard are in the line: the slow rank is visible as such, not just as the max
    pr = line["per_rank"]
    assert len(pr["fps"]) == 2 and pr["frames"] == [[0, 8], [8, 16]]
    assert pr["fps"][0] > 1.5 * pr["fps"][1] > 0 and pr["fps_min"] == min(pr["fps"]) and pr["fps_max"] == max(pr["fps"])
    assert line["value"] <= 2 * pr["fps_min"] * 1.05


def test_bench_refuses_a_world_size_that_contradicts_gpus(tmp_path):
    world = 2
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path), 4), nprocs=world, join=True)
    for k in range(world):
        assert "WORLD_SIZE=2" in json.load(open(tmp_path / f"rank{k}.json"))["exit"]


def test_bench_fans_out_by_itself_without_a_launcher(monkeypatch):
    """`python bench.py --gpus 2 --steps 20 --warmup 5` with no RANK in the environment: the script starts one rank per GPU itself
    (torch.distributed.run on 127.0.0.1), passing its own arguments through, and exits with the job's status."""
    import subprocess
    import bench
    calls = []

    def fake_call(cmd, env=None):
        calls.append((cmd, env))
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert e.value.code == 7 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_self_fan_out_runs_end_to_end_on_cpu():
    """The real thing minus the GPUs: `python bench.py --gpus 2 --selftest` re-executes itself under torch.distributed.run, two ranks
    rendezvous on 127.0.0.1 (gloo), run the timed passes with the pipeline stand-in, and stdout carries exactly one JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest", "--min-seconds", "0.05"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["steps"] == 5 and line["data"].startswith("selftest")
    assert line["value"] > 0


class _OraclePipe:
    """CPU stand-in for FramePipeline in the parity harness: the "product" is the oracle itself (optionally with one corrupted pixel, or with
    frame-loop rays that differ from sample()'s in the last ulp), so the harness's bookkeeping can be checked without a GPU."""

    def __init__(self, hp, sd, seq, torso, corrupt=None, pose_mode_ulp=False):
        import torch
        from geneface_amd import utils
        self.hp, self.sd, self.seq, self.torso, self.corrupt, self.pose_mode_ulp = hp, sd, seq, torso, corrupt, pose_mode_ulp
        self.H, self.W = seq["H"], seq["W"]
        self.poses = torch.from_numpy(seq["poses"]).float()
        self.pose6 = utils.convert_poses(self.poses)
        self.bg = torch.from_numpy(seq["bg_img"]).float().view(1, -1, 3)
        self.bg_coords = utils.get_bg_coords(self.H, self.W, "cpu")
        self.cond = torch.from_numpy(seq["cond_wins"]).float()

    def sample(self, i):
        from geneface_amd import utils
        r = utils.get_rays(self.poses[i:i + 1], [float(v) for v in self.seq["intrinsics"]], self.H, self.W, -1)
        return {"cond_wins": self.cond[i], "rays_o": r["rays_o"].contiguous(), "rays_d": r["rays_d"].contiguous(), "bg_coords": self.bg_coords,
                "pose": self.pose6[i:i + 1], "idx": i, "bg_img": self.bg, "H": self.H, "W": self.W}

    def kernel_sample(self, i):
        import torch
        s = self.sample(i)
        if self.pose_mode_ulp:
            s["rays_d"] = torch.nextafter(s["rays_d"], torch.full_like(s["rays_d"], float("inf")))
        return s

    def run_model(self, smp):
        import bench
        out = bench.oracle_render(self.hp, self.sd, bench.host_inputs(smp), self.torso)
        rgb = out["rgb_map"].clone()
        if self.corrupt is not None:
            rgb.view(-1, 3)[self.corrupt, 1] += 0.25
        return {"rgb_map": rgb}

    def render_frame(self, i):
        import torch
        import bench
        out = bench.oracle_render(self.hp, self.sd, bench.host_inputs(self.kernel_sample(i)), self.torso)
        return (out["rgb_map"].view(self.H, self.W, 3) * 255).to(torch.uint8)

    def wait(self):
        pass


def test_parity_harness_bookkeeping_on_cpu():
    """bench.parity_vs_oracle with the oracle standing in for the product: identical inputs -> exactly zero; a corrupted pixel is found, located
    and NOT excused; the fixture's frame set does not depend on the timing flags; the thread sweep reports the best count; the oracle frame of
    one tier is reused by the next only when the input bits are equal."""
    import bench
    from helpers import model_fixture
    from geneface_amd import synthetic as S
    hp, sd = model_fixture(True)
    seq = S.make_sequence(bench.PARITY_T, 32, 32, hp)
    clock, cache = bench.OracleClock(), {}
    par = bench.parity_vs_oracle(_OraclePipe(hp, sd, seq, True), hp, sd, True, frames=(1, 14), clock=clock, grazing=True, cache=cache)
    assert par["max_abs_rgb"] == 0.0 and par["frames"] == 2 and par["frame_indices"] == [1, 14] and par["uint8_within_1_lsb"] == 1.0
    assert par["pose_mode"]["pixels_off_by_more_than_1_lsb"] == 0 and par["grazing"]["pixels"] >= 0 and len(par["per_frame"]) == 2
    cpu = clock.result("head+torso 32x32")
    assert cpu["value"] > 0 and cpu["cores"] in clock.counts and str(cpu["cores"]) in cpu["s_per_frame_by_threads"] and cpu["kind"] == "port"
    assert set(cache) == {1, 14}
    # a wrong pixel: reported at its place with its size, whatever the grazing probe says
    bad = bench.parity_vs_oracle(_OraclePipe(hp, sd, seq, True, corrupt=37), hp, sd, True, frames=(14,), grazing=False, cache=cache)
    assert abs(bad["max_abs_rgb"] - 0.25) < 1e-3 and bad["worst"]["frame"] == 14 and bad["worst"]["pixel"] == [37 // 32, 37 % 32]
    # frame-loop rays one ulp away from get_rays': whatever moves by more than 1 LSB must be explained by the oracle on those rays
    ulp = bench.parity_vs_oracle(_OraclePipe(hp, sd, seq, True, pose_mode_ulp=True), hp, sd, True, frames=(1, 14), grazing=False)
    assert ulp["pose_mode"]["unexplained_after_oracle_on_kernel_rays"] == 0 and ulp["max_abs_rgb"] == 0.0
    # the fixture, not the flags, picks the frames; a smaller --parity-frames keeps frame 14 (round 3's escape)
    assert bench.PARITY_PRIORITY[0] == 14 and sorted(bench.PARITY_PRIORITY) == sorted(bench.PARITY_FRAMES) and max(bench.PARITY_FRAMES) < bench.PARITY_T - 3
    a, b = S.make_sequence(bench.PARITY_T, 32, 32, hp), S.make_sequence(25, 32, 32, hp)
    import numpy as np
    for i in (1, 14, 21):
        assert np.array_equal(a["poses"][i], b["poses"][i]) and np.array_equal(a["cond_wins"][i], b["cond_wins"][i])


def test_replica_checksum_sees_one_flipped_bit():
    import torch
    import bench
    from helpers import model_fixture
    from geneface_amd.radnerf_torso import RADNeRFTorso
    hp, sd = model_fixture(True)
    m = RADNeRFTorso(hp)
    m.load_state_dict(sd, strict=True)
    a = bench.replica_checksum(m)
    assert a == bench.replica_checksum(m)
    with torch.no_grad():
        w = m.sigma_net.net[1].weight
        w.view(torch.int32)[3, 5] ^= 1
    assert bench.replica_checksum(m) != a
