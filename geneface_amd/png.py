"""Minimal PNG (8-bit RGB) writer / reader on zlib -- the frame output contract of the reference's loop
(inference/nerfs/base_nerf_infer.py:97-101: `cv2.imwrite(f"{tmp_imgs_dir}/{idx:05d}.png", rgb->bgr)`, i.e. a PNG file holding the
RGB picture) without cv2, which this image does not ship.  Encoding runs on a small thread pool so it never sits on the render
thread (the reference encodes synchronously between frames)."""
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


class FrameWriter:
    """Writes frames as `<dir>/<idx:05d>.png` on worker threads; `close()` waits for all of them."""

    def __init__(self, out_dir: str, workers: int = 4, level: int = 1, max_pending: int = None):
        os.makedirs(out_dir, exist_ok=True)
        self.out_dir, self.level = out_dir, level
        self._pool = ThreadPoolExecutor(max_workers=workers)
        self._futures = []
        self._max_pending = max_pending or 4 * workers   # frames copied but not yet encoded: bounds host memory

    def _write(self, idx: int, img: np.ndarray):
        with open(os.path.join(self.out_dir, f"{idx:05d}.png"), "wb") as fh:
            fh.write(encode_rgb8(img, self.level))

    def submit(self, idx: int, img: np.ndarray):
        """Takes a private copy of `img` (callers hand in views of reused pinned buffers: FramePipeline.stream) and queues the encode.
        Blocks while `max_pending` frames are waiting, so a renderer faster than the encoders cannot grow the queue without bound."""
        while len(self._futures) >= self._max_pending:
            self._futures.pop(0).result()
        self._futures.append(self._pool.submit(self._write, idx, np.array(img, dtype=np.uint8, order="C", copy=True)))

    def close(self):
        for f in self._futures:
            f.result()
        self._pool.shutdown()
        self._futures = []
