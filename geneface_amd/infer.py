"""Frame loop, frame sharding and weight broadcast of the RAD-NeRF inference path.

Mirrors the data-parallel structure of inference/nerfs/base_nerf_infer.py in the reference:
  * contiguous block partition of frames over ranks, last rank takes the remainder (:150-155),
  * every rank holds a full replica; the only collective is a one-off weight broadcast (the DDP
    constructor's implicit broadcast, :126,145) + barriers around the loop (:146,178),
  * per frame: run_model(batch, infer=True) -> rgb*255 -> uint8 on the host (:95-97).
One process per GPU (torchrun / torch.distributed, backend nccl == RCCL over xGMI); no per-frame
communication.  Landmark post-processing lives in lm3d.py, model classes in radnerf*.py.
"""
import os

import numpy as np
import torch

from . import utils


def shard_range(num_frames: int, rank: int, world_size: int):
    """[start, stop) of the frames `rank` renders (base_nerf_infer.py:150-155)."""
    per = num_frames // world_size
    start = rank * per
    stop = (rank + 1) * per if rank != world_size - 1 else num_frames
    return start, stop


def broadcast_model_(model: torch.nn.Module, src: int = 0):
    """Make every replica bit-identical to rank `src` with ONE collective per dtype: parameters and buffers are
    flattened into a single contiguous buffer (~26 MB fp32 for head+torso), broadcast, and scattered back."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return model
    tensors = [t for _, t in sorted(list(model.named_parameters()) + list(model.named_buffers()), key=lambda kv: kv[0])]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for dtype in sorted(by_dtype, key=str):
            group = by_dtype[dtype]
            flat = torch.cat([t.detach().reshape(-1) for t in group])
            dist.broadcast(flat, src=src)
            off = 0
            for t in group:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
    if hasattr(model, "_fused_state"):      # packed copies of the old weights (get_state would notice as well; drop them now)
        from .fused import invalidate
        invalidate(model)
    return model


_STREAMS = {}      # device index -> the frame streams every pipeline of this process shares


def frame_streams(device, n):
    """`n` side streams of `device`, shared by every FramePipeline of the process.  HIP maps streams onto a fixed number of hardware queues
    (GPU_MAX_HW_QUEUES, 8 here) in creation order: a process that keeps creating pipelines (bench.py builds a dozen, one per leg) would hand the
    later ones streams that share hardware queues with each other, and two frames "in flight" on one queue are not concurrent -- measured:
    the split tier's sub-block ran at 1 480-1 510 fps as the thirteenth pipeline of the process against 1 664 fps as the first.  Pipelines
    are used one after the other, so sharing costs nothing.  ONE ACTIVE PIPELINE PER DEVICE is the contract: two used concurrently (a
    writer thread beside a viewer) stay correct -- every stream is in order and slots are per pipeline -- but share streams and lose the
    overlap between their frames."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


class FramePipeline:
    """Device-resident inputs of one rank's frame shard + the per-frame step.

    `seq` is the host dict produced upstream (synthetic.make_sequence or the dataset/landmark loaders):
    cond_wins [T,smo,win,C], poses [T,4,4] (ngp axes, already smoothed), intrinsics [4], bg_img [H*W,3], H, W.
    """

    def __init__(self, model, hp: dict, seq: dict, device, frames=None, impl: str = None, pinned_outputs: int = None, overlap: bool = True,
                 in_flight: int = None):
        self.model, self.hp, self.device = model, hp, torch.device(device)
        self.H, self.W = int(seq["H"]), int(seq["W"])
        self.impl = impl or model.render_impl
        idx = np.arange(len(seq["poses"])) if frames is None else np.arange(frames[0], frames[1])
        self.frame_ids = idx
        dev = self.device
        self.cond_wins = torch.from_numpy(np.ascontiguousarray(seq["cond_wins"][idx])).float().to(dev)
        self.poses = torch.from_numpy(np.ascontiguousarray(seq["poses"][idx])).float().to(dev)
        self.pose6 = utils.convert_poses(self.poses)
        self.intrinsics = [float(v) for v in seq["intrinsics"]]
        self.bg = torch.from_numpy(np.ascontiguousarray(seq["bg_img"])).float().view(1, -1, 3).to(dev)
        self.bg_coords = utils.get_bg_coords(self.H, self.W, dev)
        # frames enqueued concurrently (streams, frame slots, host buffers).  Measured on MI355X (NOTES.md section 5): the strict fp32 head
        # kernel's persistent grid drains slowly, the next frames' workgroups fill the CUs it leaves: 2 / 3 in flight = 690 / 723 fps, and a
        # fourth helps (746) only when it gets a hardware queue of its own (GPU_MAX_HW_QUEUES >= 8, which the package sets by default when it
        # is imported before the HIP runtime starts; with the runtime's default of 4 queues four in flight give 685).  Fast tier: 3 (1 845 ->
        # 1 990 fps over two; more only adds contention).
        # Any count gives the same bytes on both tiers (tests/test_gpu_render.py::test_frames_in_flight_do_not_interfere).
        self._grow_to = 0               # > 0: depth to move to after the first rotation if the scene turns out to be a saturating one
        if in_flight is None:
            try:
                queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
            except ValueError:
                queues = 4
            in_flight = 3
            # The fourth frame only pays when the frames are light on phase 1: a thin-density scene (37 % of the hit rays survive their
            # first max_steps samples, 1.2 M samples per frame) loses 4.5 % with it (536 -> 512 fps), the saturating fixture (6 %) gains
            # 2 %.  So: start with three, look at the first finished frame's survivor count, then decide once.
            if getattr(model, "render_precision", "fp32") in ("fp32", "split") and queues >= 8 and overlap and pinned_outputs is None \
                    and (impl or model.render_impl) == "fused" and self.device.type == "cuda":
                self._grow_to = 4
        self.in_flight = max(1, int(in_flight)) if overlap else 1
        self.max_in_flight = max(self.in_flight, self._grow_to)
        pinned_outputs = pinned_outputs or max(2, self.max_in_flight)
        self._pinned = [torch.empty(self.H, self.W, 3, dtype=torch.uint8).pin_memory() for _ in range(pinned_outputs)] \
            if dev.type == "cuda" else [torch.empty(self.H, self.W, 3, dtype=torch.uint8)]
        self._depth = len(self._pinned) if not self._grow_to else max(2, self.in_flight)   # host slots in rotation
        self._rendered = 0
        self._events = [None] * len(self._pinned)
        self._slot = 0
        # fused path: consecutive frames rotate over `in_flight` side streams (and as many frame slots), so frame i+1 overlaps the
        # tail of frame i; each stream is in order, and the pinned-buffer events order the host reads
        self._streams = None
        if dev.type == "cuda" and self.impl == "fused":
            # overlap=False keeps every frame on ONE side stream: kernels of consecutive frames never share the GPU, which is
            # what per-kernel profiling (rocprofv3 durations, HIP-event timing) wants; throughput runs use two
            self._streams = frame_streams(dev, self.max_in_flight)
            for st in self._streams:
                st.wait_stream(torch.cuda.current_stream(dev))

    def __len__(self):
        return len(self.frame_ids)

    # ---- per-pass batched condition encoder (fused path): cal_cond_feat of tasks/radnerfs/radnerf.py:119-128 for a block of frames at once
    def prepare(self, first: int, stop: int):
        """Encode the landmark windows of frames [first, stop) of this shard in ONE launch (gf_cond_encode_batch) and keep the per-frame
        vectors the frame kernels read (amb_bias [n,128], torso_bias [n,96]).  render_frame(i) then skips its own single-workgroup
        encoder launch (70 us alone, 0.6 ms beside four persistent head grids, and on the critical path of the frame's stream).  Same
        bytes either way (tests/test_gpu_render.py::test_prepared_pass_is_bit_identical).  No-op for the op-by-op path."""
        self._pre = None
        if self.impl != "fused" or self.device.type != "cuda" or stop <= first:
            return
        from .fused import _ver, cond_encode_batch, cond_encode_batch_accepts, get_state, head_aware_coin
        st = get_state(self.model)
        coins = None
        c = st.cond
        if not cond_encode_batch_accepts(st, self.cond_wins):
            return                         # an encoder / window the kernel does not implement (or windows it would refuse: dtype, layout -- the
                                           # SAME predicate the launch applies, checked BEFORE any coin is drawn): every frame runs the torch modules
        if st.head_aware:
            # radnerf_torso.py:175-179 flips a coin per frame that decides what is folded into torso_bias (the encoding of a black, transparent
            # head, or zeros + the per-pixel encoding).  The pass's coins are drawn HERE, in frame order -- the draws, and their order, a
            # frame-by-frame loop makes (one random.random() per rendered frame) -- and each frame later uses its own (prepared_coin).
            bgc = self._fused_bufs.bg_coords if getattr(self, "_fused_bufs", None) is not None else self.bg_coords.reshape(-1, 2)
            coins = [head_aware_coin(self.model, bgc) for _ in range(first, stop)]
        if getattr(self, "_prep_stream", None) is None:
            # one shared stream BEYOND the frame streams of the deepest pipeline this process asks for (in_flight 5 or 6 is swept by
            # bench.py: index 4 would then be frame stream #5 and the batched launch would serialise with that frame's kernels)
            k = max(self.max_in_flight, 4)
            self._prep_stream = frame_streams(self.device, k + 1)[k]
        ps = self._prep_stream
        ps.wait_stream(torch.cuda.current_stream(self.device))
        for fs in self._streams:          # frames of the previous pass may still be reading the rows this launch's buffers replace
            ps.wait_stream(fs)
        with torch.cuda.stream(ps), torch.no_grad():
            p6 = self.pose6[first:stop] if st.has_torso else None
            r = cond_encode_batch(self.model, st, self.cond_wins[first:stop], p6, bool(coins[0]) if coins else False)
            if r is None:
                if coins:                  # (cannot happen after a draw: cond_encode_batch refuses on the encoder's shape, known before)
                    raise RuntimeError("FramePipeline.prepare: the coins of a head-aware pass were drawn but the batched encoder refused")
                return                     # an encoder the kernel does not implement: every frame runs the torch modules as before
            torso = r[2]
            if coins and any(c != coins[0] for c in coins):      # both outcomes occur in the pass: a second launch with the other constant
                r2 = cond_encode_batch(self.model, st, self.cond_wins[first:stop], p6, not coins[0])
                pick = torch.tensor(coins, device=self.device).view(-1, 1)
                torso = torch.where(pick == bool(coins[0]), r[2], r2[2])
            ev = torch.cuda.Event()
            ev.record()
        self._pre = {"first": first, "stop": stop, "amb": r[1], "torso": torso, "stamp": st.stamp, "event": ev, "waited": set(),
                     "inputs": (_ver(self.cond_wins), _ver(self.pose6)), "coins": coins}

    def prepared(self, i: int):
        """(amb_bias [128], torso_bias [96] or None) of frame i from the current pass's batched launch, or None.  Called on the frame's
        stream, which is made to wait for the batch once."""
        pre = getattr(self, "_pre", None)
        if pre is None or not (pre["first"] <= i < pre["stop"]):
            return None
        from .fused import _ver, get_state
        if get_state(self.model).stamp != pre["stamp"] or pre["inputs"] != (_ver(self.cond_wins), _ver(self.pose6)):
            self._pre = None      # the weights, or the windows / poses (edited in place), changed since the batch was encoded
            return None
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream not in pre["waited"]:
            cur.wait_event(pre["event"])
            pre["waited"].add(cur.cuda_stream)
        k = i - pre["first"]
        return pre["amb"][k], (pre["torso"][k] if pre["torso"] is not None else None)

    def prepared_coin(self, i: int):
        """The head-aware coin prepare() drew for frame i (None: no batch covers it, the model has no coin, or the coin has been used).  A
        coin is good for ONE render -- the reference draws once per rendered frame -- so it is consumed here: rendering frame i a second
        time inside the same pass draws a fresh coin and encodes that frame by itself."""
        pre = getattr(self, "_pre", None)
        if pre is None or pre.get("coins") is None or not (pre["first"] <= i < pre["stop"]):
            return None
        c = pre["coins"][i - pre["first"]]
        pre["coins"][i - pre["first"]] = None
        return c

    def sample(self, i: int, rays: bool = True) -> dict:
        """The `sample` dict tasks/radnerfs/radnerf.py:119-126 reads (rays materialised, like the reference's dataset)."""
        s = {"cond_wins": self.cond_wins[i], "bg_coords": self.bg_coords, "pose": self.pose6[i:i + 1], "idx": int(self.frame_ids[i]),
             "bg_img": self.bg, "H": self.H, "W": self.W}
        if rays:
            r = utils.get_rays(self.poses[i:i + 1], self.intrinsics, self.H, self.W, -1)
            s["rays_o"], s["rays_d"] = r["rays_o"], r["rays_d"]
        return s

    def kernel_sample(self, i: int) -> dict:
        """`sample(i)` with the rays the frame loop generates for itself (fused.pinhole_rays: the device function k_frame_init runs), as
        tensors: run_model(kernel_sample(i)) and render_frame(i) see the same ray bits."""
        from .fused import pinhole_rays
        s = self.sample(i, rays=False)
        s["rays_o"], s["rays_d"] = pinhole_rays(self.poses[i], self.intrinsics, self.H, self.W, self.device)
        return s

    def run_model(self, sample: dict) -> dict:
        """RADNeRF(Torso)Task.run_model(sample, infer=True) (tasks/radnerfs/radnerf.py:166-170, radnerf_torso.py:113-117)."""
        bg = sample["bg_torso_img"] if ("bg_torso_img" in sample and not hasattr(self.model, "forward_torso")) else sample["bg_img"]
        return self.model.render(sample["rays_o"], sample["rays_d"], sample["cond_wins"], sample["bg_coords"], sample["pose"],
                                 index=sample["idx"], staged=False, bg_color=bg, perturb=False, force_all_rays=True,
                                 render_impl=self.impl, **self.hp)

    def render_frame(self, i: int) -> torch.Tensor:
        """One step of the frame loop: returns the pinned-host uint8 [H,W,3] RGB frame (valid after `wait(slot)` /
        a stream sync; double-buffered so the D2H copy of frame i overlaps the kernels of frame i+1)."""
        if self._grow_to and self._rendered == self._depth:   # the first rotation is enqueued: its first frame decides the depth
            self._decide_depth()
        self._rendered += 1
        slot = self._slot
        self._slot = (slot + 1) % self._depth
        if self._events[slot] is not None:
            self._events[slot].synchronize()   # the host side of this slot's previous frame has been handed out and may be reused
        if self.impl == "fused":
            from .fused import render_frame_fused
            timing = getattr(self, "frame_timing", None)      # a list: bench.py's latency leg asks for per-frame device times
            with torch.cuda.stream(self._streams[slot % self.in_flight]):
                if timing is not None:
                    t_start = torch.cuda.Event(enable_timing=True)
                    t_start.record()
                rgb8 = render_frame_fused(self, i, slot % max(2, self.in_flight))
                self._pinned[slot].copy_(rgb8, non_blocking=True)
                ev = torch.cuda.Event(enable_timing=timing is not None)
                ev.record()
                self._events[slot] = ev
                if timing is not None:
                    timing.append((i, t_start, ev))
            return self._pinned[slot]
        out = self.run_model(self.sample(i))
        rgb8 = (out["rgb_map"] * 255).view(self.H, self.W, 3).to(torch.uint8)
        self._pinned[slot].copy_(rgb8, non_blocking=True)
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._events[slot] = ev
        return self._pinned[slot]

    def _decide_depth(self):
        """Once per pipeline (in_flight=None on >= 8 hardware queues): a fourth frame in flight if few rays outlive phase 0."""
        grow, self._grow_to = self._grow_to, 0
        from .fused import get_state
        if self._events[0] is not None:
            self._events[0].synchronize()
        ctrl = get_state(self.model).workspace(self.H * self.W, 0)[1].cpu()
        n_hit, n_surv = int(ctrl[1]), int(ctrl[2])
        if n_surv * 100 < 15 * max(n_hit, 1):
            self._slot = self._depth          # the new slot goes next, then the rotation continues over all of them
            self.in_flight = self._depth = grow

    def stream(self, indices):
        """Frame loop with the pipeline kept full: yields (i, uint8 [H,W,3] numpy view of the pinned buffer) once frame i has reached
        host memory, while the following frames are already enqueued.  The view is only valid until the next iteration (the slot is
        reused): consumers copy it or hand it to an encoder that does (png.FrameWriter.submit)."""
        pending = []
        indices = list(indices)
        if indices and indices == list(range(indices[0], indices[-1] + 1)):
            pre = getattr(self, "_pre", None)
            covered = pre is not None and pre["first"] <= indices[0] and indices[-1] < pre["stop"] and self.prepared(indices[0]) is not None
            if covered and pre.get("coins") is not None:
                # a head-aware batch carries one-shot coins (one draw per RENDERED frame, as the reference's loop makes): an explicit
                # prepare(a, b) followed by stream(range(a, b)) uses those draws -- n in all, not 2n -- and only a block one of whose coins
                # has already been consumed is encoded again with fresh ones
                covered = all(pre["coins"][i - pre["first"]] is not None for i in indices)
            if not covered:      # no batch, a stale one, or one that covers only part of the block (the rest would fall back to per-frame launches)
                self.prepare(indices[0], indices[-1] + 1)      # every window of the block is resident: one encoder launch for all of them
        for i in indices:
            buf = self.render_frame(i)
            pending.append((i, buf, self._events[(self._slot - 1) % self._depth]))
            if len(pending) >= self._depth:    # the oldest slot is the next one to be reused: drain it first
                j, b, ev = pending.pop(0)
                if ev is not None:
                    ev.synchronize()
                yield j, b.numpy()
        for j, b, ev in pending:
            if ev is not None:
                ev.synchronize()
            yield j, b.numpy()

    def wait(self, frame: torch.Tensor = None):
        """Block until every enqueued frame (or all work) has reached pinned host memory."""
        for ev in self._events:
            if ev is not None:
                ev.synchronize()
