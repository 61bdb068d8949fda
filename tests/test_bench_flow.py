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


def test_bench_control_flow_world8_gloo(tmp_path):
    """The full-node shape: eight ranks, one line, eight disjoint blocks, eight clocks."""
    world = 8
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path), world), nprocs=world, join=True)
    r = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world)]
    assert [rk["frames"] for rk in r] == [[8 * k, 8 * k + 8] for k in range(world)]
    assert len(r[0]["lines"]) == 1 and all(rk["lines"] == [] for rk in r[1:])
    line = r[0]["lines"][0]
    assert line["n_gpus"] == 8 and line["config"]["rccl_ranks"] == 8 and line["config"]["frames_total"] == 48 and "frame-shard x8" in line["config"]["parallelism"]
    assert len({len(rk["passes"]) for rk in r}) == 1 and all(all(p == list(range(2, 8)) for p in rk["passes"]) for rk in r)
    pr = line["per_rank"]
    assert len(pr["fps"]) == 8 and pr["frames"] == [[8 * k, 8 * k + 8] for k in range(world)]
    assert abs(line["value"] - 8 * 6 / (line["ms_per_step"] * 6 / 1e3)) < 1e-6 * line["value"] and line["value"] <= 8 * pr["fps_min"] * 1.05
    import bench
    assert len(json.dumps(bench.compact_line(line, "x"))) < 3000


def test_a_rank_that_renders_another_picture_voids_the_line():
    """every_rank_parity's verdict is not a footnote: if the ranks' frames of the common fixture frame are not byte-identical, or any of
    them is off the oracle by more than the tolerance, the N > 1 line reports value = null and says why."""
    import torch
    import bench
    ref = torch.rand(64)
    rows = [ref + 1e-6, ref + 1e-6, ref + 1e-6]
    ok = bench.judge_rank_frames(rows, ref, 14)
    assert ok["identical_across_ranks"] is True and ok["max_abs_rgb"] < 1e-4 and len(ok["max_abs_rgb_by_rank"]) == 3
    line = {"value": 123.0}
    bench.apply_rank_parity_guard(line, ok)
    assert line["value"] == 123.0 and "error" not in line
    rows[2] = rows[2].clone()
    rows[2][5] += 3e-7                                   # one float differs in one replica
    bad = bench.judge_rank_frames(rows, ref, 14)
    assert bad["identical_across_ranks"] is False and bad["ranks_differing_from_rank0"] == [2]
    line = {"value": 123.0}
    bench.apply_rank_parity_guard(line, bad)
    assert line["value"] is None and line["value_withheld"] == 123.0 and "identical" in line["error"]
    rows = [ref + 1e-2] * 3                               # identical, but not the reference's picture
    off = bench.judge_rank_frames(rows, ref, 14)
    line = {"value": 5.0}
    bench.apply_rank_parity_guard(line, off)
    assert line["value"] is None and "tolerance" in line["error"]
    line = {"value": 5.0}
    bench.apply_rank_parity_guard(line, {"error": "oracle failed"})
    assert line["value"] is None
    line = {"value": 5.0}
    bench.apply_rank_parity_guard(line, None)             # N = 1 without --rank-parity: nothing to judge
    assert line["value"] == 5.0


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


def test_compact_line_keeps_the_numbers_and_drops_the_lists():
    """the stdout line had grown to 15 KB and the driver's record kept the headline parity only as a key name.  The line is
    now the compact form of the full record (which goes to a side file): `config.parity` carries the headline summary and one row per
    sub-leg, `roofline` / `cpu_baseline` keep the contract's fields, nothing holds a per-frame list."""
    import json
    import os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "round4", "r4o_bench_driver_command.json")))       # a full record of round 4's format
    full["variants"] = {n: {"value": 700.0, "vs_default": 0.92, "roofline_frac": 0.65, "samples_per_frame": 8.6e5, "split_tier_value": 1700.0,
                            "parity": full["parity"], "parity_split_tier": full["split_tier"]["parity"]} for n in bench.VARIANT_NAMES}
    full["variants"]["audio"] = {"error": "RuntimeError: x" * 100}
    line = bench.compact_line(full, "gpurun_out/bench_details.json")
    text = json.dumps(line)
    assert len(text) < 6000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    par = line["config"]["parity"]
    assert abs(par["max_abs_rgb"] - full["parity"]["max_abs_rgb"]) < 1e-7 and par["frames"] == 8 and par["tolerance"] == 1e-4
    assert par["pose_mode_unexplained"] == 0 and par["grazing_pixels"] == full["parity"]["grazing"]["pixels"]
    assert par["legs"]["split_tier"][0] == bench._sig(full["split_tier"]["parity"]["max_abs_rgb"]) and par["legs"]["variant:hash"][1] == 8
    assert par["all_legs_max_abs_rgb"] >= par["max_abs_rgb"] and par["all_legs_unexplained"] == 0
    r = line["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - full["roofline"]["frac"]) < 1e-3 and r["peak"] == 157.3 and r["unit"] == "TFLOP/s" and "traffic" in r
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3 and "kernel_ms_per_frame" in r and "pipelined" in r and "marcher" in r
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert line["variants"]["hash"]["fps"] == 700.0 and "error" in line["variants"]["audio"] and len(line["variants"]["audio"]["error"]) <= 160

    def no_long_lists(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                no_long_lists(v, path + "/" + k)
        elif isinstance(o, list):
            assert len(o) <= 8 and not any(isinstance(e, dict) for e in o), path
    no_long_lists(line)


def test_pmc_traffic_is_marked_stale_when_the_kernels_moved_on(tmp_path, monkeypatch):
    """roofline.traffic is read from a committed PMC summary (counters cannot be collected inside the timed run): the summary carries the
    digest of the kernel sources it was collected on, and the line says `traffic_stale: true` when this tree's digest differs."""
    import json
    import os
    import bench
    from geneface_amd.csrc.build import source_digest
    prof = tmp_path / "profiles" / "round9"
    prof.mkdir(parents=True)
    body = {"k_head_phase": {"fetch_MB_x2": 100.0, "write_MB_raw": 2.0}}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "r9a_pmc_summary.json").write_text(json.dumps(dict(body, _source_digest=source_digest())))
    t, src, stale = bench.pmc_traffic()
    assert t == 102e6 and src.endswith("r9a_pmc_summary.json") and stale is False
    (prof / "r9a_pmc_summary.json").write_text(json.dumps(dict(body, _source_digest="0" * 16)))
    assert bench.pmc_traffic()[2] is True
    (prof / "r9a_pmc_summary.json").write_text(json.dumps(body))          # round 1-4 summaries carry no digest: stale by definition
    assert bench.pmc_traffic()[2] is True


def test_legacy_nerf_baseline_runs_on_the_host_where_a_gpu_is_visible(monkeypatch):
    """bench.py's B2 leg times the reference's own modules.nerfs classes (staged archive) on the HOST cores.  Those modules pick
    `device = "cuda"` at import wherever a GPU is visible (volume_rendering.py:7, ray_samplers.py:8) -- i.e. on every GPU box and never in the
    build container: round 6's first driver-command run on the MI355X died there.  The leg must pin both module globals to the CPU."""
    import torch
    import bench
    archive = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_refpy", "geneface_refpy.zip")
    if not os.path.exists(archive):
        pytest.skip("oracle/_refpy/geneface_refpy.zip not staged")
    # the leg imports the reference's packages from the ARCHIVE (sys.path[0]); other tests of this process import them from the reference tree:
    # every module and path entry this test adds is removed again
    before_modules, before_path = set(sys.modules), list(sys.path)
    ref_pkgs = ("modules", "utils", "tasks", "data_gen", "inference", "data_util")
    stash = {m: sys.modules.pop(m) for m in list(sys.modules) if m.split(".")[0] in ref_pkgs}     # a fresh import, as in a fresh bench process
    try:
        monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
        from geneface_amd import hparams as HP
        from geneface_amd import synthetic as S
        seq = S.make_sequence(1, 64, 64, HP.may_hparams(True))
        rec = bench.legacy_nerf_baseline(seq, rays=64, full_size=8)
        assert rec["kind"] == "reference" and rec["value"] > 0 and rec["whole_frame_64x64"]["finite"]
        import modules.nerfs.commons.ray_samplers as rs
        import modules.nerfs.commons.volume_rendering as vr
        assert vr.device.type == "cpu" and rs.device.type == "cpu"
    finally:
        for m in list(sys.modules):
            if m not in before_modules and m.split(".")[0] in ref_pkgs:
                del sys.modules[m]
        sys.modules.update(stash)
        sys.path[:] = before_path
