"""Interactive-viewer render path, headless: what inference/nerfs/radnerf_gui.py asks of the renderer.

The reference's viewer is a dearpygui window around three pieces that matter to the render path and are mirrored here:
  * `OrbitCamera` (radnerf_gui.py:21-82): pose / intrinsics from radius, fovy, an orientation and a look-at centre, with the
    orbit / scale / pan updates of the mouse handlers;
  * `RADNeRFTask.test_gui_with_editable_data` (tasks/radnerfs/radnerf.py:333-380): render at `downscale` x the window size from a
    free camera, bilinear (image) / nearest (depth) resize back to the window -> {'image' [H,W,3], 'depth' [H,W]} numpy;
  * `NeRFGUI.test_step` (radnerf_gui.py:181-236): sample-per-pixel accumulation of successive renders while the camera rests, reset
    when it moves, and the dynamic-resolution rule (keep a full-resolution frame under 200 ms).
There is no window system in this image; `Viewer.test_step()` returns the buffer the reference would upload as a texture.
Every frame goes through the model's `render()` (fused HIP path on a GPU); nothing here falls back to a CPU renderer.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import utils


class OrbitCamera:
    def __init__(self, W, H, r=2.0, fovy=60.0):
        from scipy.spatial.transform import Rotation
        self._R = Rotation
        self.W, self.H = W, H
        self.radius, self.fovy = r, fovy
        self.center = np.zeros(3, dtype=np.float32)
        self.rot = Rotation.from_matrix([[0, -1, 0], [0, 0, -1], [1, 0, 0]])       # ngp axes (radnerf_gui.py:27)
        self.up = np.array([1, 0, 0], dtype=np.float32)

    @property
    def pose(self):
        res = np.eye(4, dtype=np.float32)
        res[2, 3] -= self.radius
        rot = np.eye(4, dtype=np.float32)
        rot[:3, :3] = self.rot.as_matrix()
        res = rot @ res
        res[:3, 3] -= self.center
        return res

    def update_pose(self, pose):
        self.radius = float(np.linalg.norm(pose[:3, 3]))
        T = np.eye(4)
        T[2, 3] = -self.radius
        self.rot = self._R.from_matrix((pose @ np.linalg.inv(T))[:3, :3])

    def update_intrinsics(self, intrinsics):
        _, fl_y, cx, cy = intrinsics
        self.W, self.H = int(cx * 2), int(cy * 2)
        self.fovy = float(np.rad2deg(2 * np.arctan2(self.H, 2 * fl_y)))

    @property
    def intrinsics(self):
        focal = self.H / (2 * np.tan(np.deg2rad(self.fovy) / 2))
        return np.array([focal, focal, self.W // 2, self.H // 2])

    def orbit(self, dx, dy):
        side = self.rot.as_matrix()[:3, 0]
        self.rot = self._R.from_rotvec(self.up * np.radians(-0.01 * dx)) * self._R.from_rotvec(side * np.radians(-0.01 * dy)) * self.rot

    def scale(self, delta):
        self.radius *= 1.1 ** (-delta)

    def pan(self, dx, dy, dz=0):
        self.center += (0.0001 * self.rot.as_matrix()[:3, :3] @ np.array([dx, dy, dz])).astype(np.float32)


def test_gui_with_editable_data(model, hparams, pose, intrinsics, W, H, cond_wins, index=0, bg_color=None, spp=1, downscale=1, device=None):
    """tasks/radnerfs/radnerf.py:333-380.  `bg_color`: [1, H*W, 3] image (resampled to the render size when downscale != 1),
    an RGB triple, or None (white).  Returns {'image': float32 [H,W,3], 'depth': float32 [H,W]} (numpy)."""
    device = torch.device(device) if device is not None else next(model.parameters()).device
    rH, rW = int(H * downscale), int(W * downscale)
    intr = np.asarray(intrinsics, dtype=np.float64) * downscale
    pose_t = torch.from_numpy(np.asarray(pose, dtype=np.float32)).unsqueeze(0).to(device)
    rays = utils.get_rays(pose_t, intr, rH, rW, -1)
    bg_coords = utils.get_bg_coords(rH, rW, device)
    if bg_color is not None:
        bg_color = torch.as_tensor(bg_color, dtype=torch.float32, device=device)
        if bg_color.numel() == H * W * 3 and (rH, rW) != (H, W):
            bg_color = F.interpolate(bg_color.view(1, H, W, 3).permute(0, 3, 1, 2), size=(rH, rW), mode="bilinear").permute(0, 2, 3, 1).reshape(1, -1, 3)
        elif bg_color.numel() == 3:
            bg_color = bg_color.view(1, 1, 3).expand(1, rH * rW, 3).contiguous()
    model.eval()
    with torch.no_grad():
        out = model.render(rays["rays_o"], rays["rays_d"], cond_wins.to(device), bg_coords, utils.convert_poses(pose_t), index=index, staged=False,
                           bg_color=bg_color, perturb=False, force_all_rays=True, **hparams)      # run_model(infer=True), radnerf.py:169
    preds = out["rgb_map"].reshape(1, rH, rW, 3)
    depth = out["depth_map"].reshape(1, rH, rW)
    if downscale != 1:
        preds = F.interpolate(preds.permute(0, 3, 1, 2), size=(H, W), mode="bilinear").permute(0, 2, 3, 1).contiguous()
        depth = F.interpolate(depth.unsqueeze(1), size=(H, W), mode="nearest").squeeze(1)
    return {"image": preds[0].cpu().numpy(), "depth": depth[0].cpu().numpy()}


class Viewer:
    """The state machine of NeRFGUI without the window: a free camera, spp accumulation, dynamic resolution."""

    def __init__(self, model, hparams, W=None, H=None, cond_features=None, bg_color=None, device=None, max_frame_ms=200.0):
        self.model, self.hparams = model, hparams
        self.W, self.H = W or hparams.get("gui_w", 512), H or hparams.get("gui_h", 512)
        self.cam = OrbitCamera(self.W, self.H, r=hparams.get("gui_radius", 3.35), fovy=hparams.get("gui_fovy", 21.24))
        self.cond_features, self.cond_idx, self.ind_index = cond_features, 0, 0
        self.bg_color = bg_color
        self.device = device
        self.render_buffer = np.zeros((self.H, self.W, 3), dtype=np.float32)
        self.need_update, self.spp, self.mode = True, 1, "image"
        self.downscale, self.dynamic_resolution, self.max_frame_ms = 1.0, False, max_frame_ms
        self.max_spp = hparams.get("gui_max_spp", 1)
        self.last_ms = None

    def prepare_buffer(self, outputs):
        return outputs["image"] if self.mode == "image" else np.expand_dims(outputs["depth"], -1).repeat(3, -1)

    def conds(self):
        if self.cond_features is None:
            return None
        return utils.get_audio_features(self.cond_features, 2, self.cond_idx, self.hparams["smo_win_size"])

    def test_step(self):
        if not (self.need_update or self.spp < self.max_spp):
            return self.render_buffer
        timed = torch.cuda.is_available()
        if timed:
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
        outputs = test_gui_with_editable_data(self.model, self.hparams, self.cam.pose, self.cam.intrinsics, self.W, self.H, self.conds(), self.ind_index,
                                              self.bg_color, self.spp, self.downscale, self.device)
        if timed:
            end.record()
            torch.cuda.synchronize()
            self.last_ms = start.elapsed_time(end)
            if self.dynamic_resolution:
                full_t = self.last_ms / (self.downscale ** 2)
                downscale = min(1.0, max(0.25, math.sqrt(self.max_frame_ms / full_t)))
                if downscale > self.downscale * 1.2 or downscale < self.downscale * 0.8:
                    self.downscale = downscale
        if self.need_update:
            self.render_buffer, self.spp, self.need_update = self.prepare_buffer(outputs), 1, False
        else:
            self.render_buffer = (self.render_buffer * self.spp + self.prepare_buffer(outputs)) / (self.spp + 1)
            self.spp += 1
        return self.render_buffer
