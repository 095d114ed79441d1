"""Camera matrices consumed by the rasterizer (host side, once per camera).

Mirrors the conventions of the reference so its callers drop in unchanged:
  * world_view_transform = W2C(R, T)^T, row-vector convention   [REF scene/cameras.py:59,
    utils/graphics_utils.py:38-49]
  * projection_matrix = P(znear=0.01, zfar=100, fovX, fovY)^T   [REF scene/cameras.py:53-54,60,
    utils/graphics_utils.py:51-71]
  * full_proj_transform = view @ proj, camera_center = inverse(view)[3, :3]   [REF scene/cameras.py:61-62]
A row-major [4,4] tensor holding M^T has the memory image of column-major M, which is what the HIP
kernels index (`m[col*4+row]`).
"""
from __future__ import annotations

import math

import numpy as np
import torch

ZNEAR = 0.01
ZFAR = 100.0


def world_to_view(R: np.ndarray, t: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """4x4 world->camera matrix. `R` is stored transposed (camera-to-world rotation), `t` is the
    world->camera translation, as in the reference's loaders [REF utils/graphics_utils.py:38-49]."""
    Rt = np.eye(4, dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    if scale != 1.0 or any(float(v) != 0.0 for v in translate):
        c2w = np.linalg.inv(Rt)
        c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
        Rt = np.linalg.inv(c2w)
    else:
        # the reference always round-trips through two inversions; do the same so float32 bits agree
        Rt = np.linalg.inv(np.linalg.inv(Rt))
    return Rt.astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """OpenGL-style perspective with z in [0,1], w = +z  [REF utils/graphics_utils.py:51-71]."""
    tx = math.tan(fovX / 2)
    ty = math.tan(fovY / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov: float, pixels: int) -> float:
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal: float, pixels: int) -> float:
    return 2 * math.atan(pixels / (2 * focal))


class Camera:
    """Minimal stand-in for the reference's `scene.cameras.Camera` carrying exactly the attributes
    `gaussian_renderer.render()` reads [REF gaussian_renderer/__init__.py:34-46]."""

    def __init__(self, R, T, FoVx, FoVy, width, height, time=0.0, device="cpu", uid=0,
                 trans=(0.0, 0.0, 0.0), scale=1.0, image=None):
        self.uid = uid
        self.R = np.asarray(R, dtype=np.float64)
        self.T = np.asarray(T, dtype=np.float64)
        self.FoVx = float(FoVx)
        self.FoVy = float(FoVy)
        self.image_width = int(width)
        self.image_height = int(height)
        self.time = np.asarray([time], dtype=np.float32)
        self.znear, self.zfar = ZNEAR, ZFAR
        self.original_image = image
        wv = torch.tensor(world_to_view(self.R, self.T, trans, scale)).transpose(0, 1)
        pj = projection_matrix(self.znear, self.zfar, self.FoVx, self.FoVy).transpose(0, 1)
        self.world_view_transform = wv.contiguous().to(device)
        self.projection_matrix = pj.contiguous().to(device)
        self.full_proj_transform = (wv.unsqueeze(0).bmm(pj.unsqueeze(0))).squeeze(0).contiguous().to(device)
        self.camera_center = wv.inverse()[3, :3].contiguous().to(device)

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        if self.original_image is not None:
            self.original_image = self.original_image.to(device)
        return self


def look_at_camera(eye, target, up, FoVx, width, height, time=0.0, device="cpu", uid=0) -> Camera:
    """Camera at `eye` looking at `target` (+z forward, +y down, the 3DGS/COLMAP convention)."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w_R = np.stack([right, down, fwd], axis=1)      # columns = camera axes in world
    w2c_R = c2w_R.T
    T = -w2c_R @ eye
    FoVy = focal2fov(fov2focal(FoVx, width), height)
    return Camera(R=c2w_R, T=T, FoVx=FoVx, FoVy=FoVy, width=width, height=height, time=time, device=device, uid=uid)


def orbit_cameras(n, radius, FoVx, width, height, elevation_deg=20.0, arc_deg=360.0, target=(0, 0, 0),
                  device="cpu"):
    """n cameras on an orbit (D-NeRF-like) or an arc (HyperNeRF-like), times spread over [0,1]."""
    cams = []
    el = math.radians(elevation_deg)
    for i in range(n):
        az = math.radians(arc_deg) * (i / max(n, 1)) - (math.radians(arc_deg) / 2 if arc_deg < 360 else 0.0)
        eye = np.array([radius * math.cos(el) * math.sin(az), -radius * math.sin(el), -radius * math.cos(el) * math.cos(az)])
        eye = eye + np.asarray(target, dtype=np.float64)
        t = i / (n - 1) if n > 1 else 0.0
        cams.append(look_at_camera(eye, target, (0.0, -1.0, 0.0), FoVx, width, height, time=t, device=device, uid=i))
    return cams
