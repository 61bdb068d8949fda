"""Minimal PNG (8-bit RGB) writer / reader on zlib -- the frame output contract of the reference's loop
(inference/nerfs/base_nerf_infer.py:97-101: `cv2.imwrite(f"{tmp_imgs_dir}/{idx:05d}.png", rgb->bgr)`, i.e. a PNG file holding the
RGB picture) without cv2, which this image does not ship.  The frame loop's writer (FrameWriter) runs on native threads of the library
(csrc/png_writer.cpp); encode_rgb8 / decode_rgb8 are the pure-Python pair the tests and the background-image loader use."""
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_rgb8(img: np.ndarray, level: int = 1) -> bytes:
    """img uint8 [H, W, 3] (RGB) -> PNG bytes (colour type 2, filter 0 on every row)."""
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("encode_rgb8 expects uint8 [H, W, 3]")
    H, W, _ = img.shape
    raw = np.empty((H, 1 + 3 * W), dtype=np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = img.reshape(H, 3 * W)
    return _SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + _chunk(b"IEND", b"")


def decode_rgb8(data: bytes) -> np.ndarray:
    """Inverse of encode_rgb8 for files this module wrote (filter type 0 only)."""
    if data[:8] != _SIG:
        raise ValueError("not a PNG")
    pos, idat, W, H = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            W, H, depth, ctype = struct.unpack(">IIBB", body[:10])
            if depth != 8 or ctype != 2:
                raise ValueError("only 8-bit RGB")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(H, 1 + 3 * W)
    if raw[:, 0].any():
        raise ValueError("only filter type 0 rows")
    return raw[:, 1:].reshape(H, W, 3).copy()


def effective_cpus() -> int:
    """Host cores this PROCESS may actually use: the smallest of os.cpu_count(), the scheduler affinity mask and the cgroup CPU quota
    (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`).  A GPU pod typically reports the node's 256 logical cores while its
    quota is a few dozen: worker pools sized from os.cpu_count() then oversubscribe the quota and every thread is throttled -- measured on
    the MI355X box (tools/png_scale.py, profiles/round6/r6a_png_scale_*): one writer with 8 workers deflates a frame in 4.6 ms, eight writers
    with 32 workers each in 40-54 ms, and the aggregate FALLS from 4 000 to 2 300 frames/s."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:             # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def plan_writer(world_size: int = 1, level=None, cpus: int = None):
    """(workers, zlib level) of ONE rank's FrameWriter when `world_size` ranks of a node write frames at once (base_nerf_infer.py:131-181: every
    rank writes into the same tmp_imgs_dir).  The rank's share of the EFFECTIVE cores decides:
      * share >= 8: Z_RLE level 1 (4.6 ms of deflate per 512x512 frame, 0.66 MB files), one worker per core of the share up to 32 -- a rank
        renders 750-1 800 frames/s, i.e. needs 4-9 deflating cores;
      * fewer: level 0, stored blocks (adler32 + crc32 + a copy: ~0.7-1.1 ms per frame, 0.79 MB files -- 19 % larger; the files are an
        intermediate the reference deletes after the ffmpeg mux, base_nerf_infer.py:307-317), so that eight ranks need ~6 cores in all instead of ~40.
    `level` (hparams['infer_png_zlib_level']) overrides the choice."""
    share = max(1, (cpus if cpus is not None else effective_cpus()) // max(1, int(world_size)))
    if level is None or level == "auto":
        level = 1 if share >= 8 else 0
    level = int(level)
    workers = max(2, min(32, share)) if level > 0 else max(2, min(8, share))
    return workers, level


class FrameWriter:
    """Writes frames as `<dir>/<idx:05d>.png` on NATIVE worker threads (gf_png_writer_* of libgeneface_hip.so: deflate + CRC + file write with
    no interpreter lock anywhere on the path); `close()` waits for all of them.  Round 2 ran zlib.compress on a Python thread pool: the
    workers need the interpreter lock between their zlib calls while the render thread holds it for most of a frame -- 0.75-0.84 of the
    render rate.  strategy: zlib's.  Default 3 = Z_RLE at level 1: measured on rendered 512x512 frames (tools/png_bench.py,
    profiles/round3/png_bench.json) 4.8 ms of deflate per frame and 0.66 MB files against 9.3 ms and 0.40 MB for the default strategy -- 32
    workers encode 3 750 frames/s instead of 1 550, above what one GPU renders on any tier; the files are an intermediate the reference
    deletes after the ffmpeg mux (base_nerf_infer.py:307-317).  strategy=0 gives the smaller files."""

    def __init__(self, out_dir: str, workers: int = 4, level: int = 1, max_pending: int = None, strategy: int = 3):
        os.makedirs(out_dir, exist_ok=True)
        self.out_dir, self.level, self.workers, self.strategy = out_dir, level, workers, strategy
        self._max_pending = max_pending or 4 * workers   # frames copied but not yet encoded: bounds host memory
        self._h, self._shape, self._stats = None, None, None

    def submit(self, idx: int, img: np.ndarray):
        """Takes a private copy of `img` (callers hand in views of reused pinned buffers: FramePipeline.stream) and queues the encode.
        Blocks while `max_pending` frames are waiting, so a renderer faster than the encoders cannot grow the queue without bound."""
        from .lib import check, lib
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("FrameWriter.submit expects uint8 [H, W, 3]")
        if not img.flags["C_CONTIGUOUS"]:
            img = np.ascontiguousarray(img)
        if self._h is None:
            self._shape = img.shape
            self._h = lib().gf_png_writer_create(os.fsencode(self.out_dir), img.shape[0], img.shape[1], self.workers, self.level, self.strategy, self._max_pending)
            if not self._h:
                raise RuntimeError(lib().gf_last_error().decode())
        elif img.shape != self._shape:
            raise ValueError(f"FrameWriter: frame {img.shape} after frames of {self._shape}")
        check(lib().gf_png_writer_submit(self._h, int(idx), img.ctypes.data))

    def close(self):
        if self._h is not None:
            import ctypes as C
            from .lib import lib
            st = (C.c_double * 6)()
            h, self._h = self._h, None
            rc = lib().gf_png_writer_close(h, C.cast(st, C.c_void_p))
            self._stats = list(st)
            if rc != 0:
                raise RuntimeError(lib().gf_last_error().decode())

    def stage_seconds(self):
        """After close(): where the host time went -- seconds copying frames in and waiting for queue room (submit side), deflating and
        writing (summed over the workers), frames and bytes written."""
        if self._stats is None:
            return None
        k = ("copy_in", "wait_for_room", "deflate_sum_over_workers", "write_sum_over_workers", "frames", "bytes")
        return dict(zip(k, self._stats))
