"""Landmark-sequence conditioning: the host-side preprocessing between PostNet's `pred_lm3d/*.npy`
and the renderer's `cond_wins` input (numpy, done once per sequence before frames are sharded).

Behaviour follows inference/nerfs/lm3d_radnerf_infer.py:45-85 (normalise, per-region clamp,
sequential EMA, window gathering) and data_gen/nerf/binarizer.py:24-59 (`get_win_conds`).
"""
import numpy as np


def get_win_conds(conds: np.ndarray, idx: int, smo_win_size: int = 8, pad_option: str = "zero") -> np.ndarray:
    """A centred window of `smo_win_size` rows around `idx` ([idx - w//2, idx + w - w//2)), padded at the
    sequence ends with zeros or with the first / last row ('edge')."""
    T = conds.shape[0]
    idx = min(max(0, idx), T - 1)
    half = smo_win_size // 2
    left, right = idx - half, idx + (smo_win_size - half)
    pad_l, pad_r = max(0, -left), max(0, right - T)
    win = conds[max(left, 0):min(right, T)]
    if pad_option not in ("zero", "edge"):
        raise NotImplementedError
    if pad_l:
        fill = np.zeros_like(conds[:1]) if pad_option == "zero" else conds[:1]
        win = np.concatenate([fill] * pad_l + [win], axis=0)
    if pad_r:
        fill = np.zeros_like(conds[:1]) if pad_option == "zero" else conds[-1:]
        win = np.concatenate([win] + [fill] * pad_r, axis=0)
    assert win.shape[0] == smo_win_size
    return win


# (landmark slice, xy-only half clamp) regions of the 68-point layout
_CLAMP_FULL = [slice(0, 17), slice(27, 36), slice(48, 68)]  # jaw line ("yaw"), nose, mouth
_CLAMP_XY_HALF = [slice(17, 27), slice(36, 48)]              # brows, eyes: x,y at std/2, z at std


def clamp_lm3d(lm: np.ndarray, clamp_std: float) -> np.ndarray:
    """lm [T,68,3] (normalised) -> clamped copy."""
    lm = lm.copy()
    for s in _CLAMP_FULL:
        lm[:, s] = np.clip(lm[:, s], -clamp_std, clamp_std)
    for s in _CLAMP_XY_HALF:
        lm[:, s, 0:2] = np.clip(lm[:, s, 0:2], -clamp_std / 2, clamp_std / 2)
        lm[:, s, 2] = np.clip(lm[:, s, 2], -clamp_std, clamp_std)
    return lm


def ema_lm3d(lm: np.ndarray, lam: float = 0.2) -> np.ndarray:
    """Sequential exponential smoothing  y[i] = lam*y[i-1] + (1-lam)*x[i], y[-1] := x[0]  (all regions use 0.2)."""
    out = lm.copy()
    prev = lm[0].copy()
    one_minus = np.float32(1 - lam)
    lam = np.float32(lam)
    for i in range(out.shape[0]):
        out[i] = lam * prev + one_minus * out[i]
        prev = out[i]
    return out


def normalize_and_smooth(idexp_lm3d: np.ndarray, mean, std, clamp_std: float = 2.5) -> np.ndarray:
    """[T,204] raw id+exp landmarks -> [T,204] float32 normalised, clamped, smoothed."""
    lm = (idexp_lm3d.reshape(-1, 68, 3).astype(np.float32) - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    lm = ema_lm3d(clamp_lm3d(lm, clamp_std))
    return lm.reshape(-1, 204).astype(np.float32)


def cond_windows(lm_norm: np.ndarray, cond_win_size: int = 1, smo_win_size: int = 5) -> np.ndarray:
    """[T,204] -> [T, smo_win, cond_win, 204]: per frame, the `cond_wins` tensor `run_model` consumes."""
    T = lm_norm.shape[0]
    win = np.stack([get_win_conds(lm_norm, i, cond_win_size, "edge") for i in range(T)])  # [T, cond_win, 204]
    return np.stack([get_win_conds(win, i, smo_win_size, "edge") for i in range(T)])         # [T, smo, cond_win, 204]
