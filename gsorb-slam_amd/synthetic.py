"""Synthetic scenes and cameras for the parity tests and bench.py (numpy only).

Cameras follow the reference's construction exactly
(/root/reference/src/Camera.cc:7-47, scripts/replay.py:91-120): tanfov = size /
(2 f), near 0.01, far 100, symmetric frustum, and the 4x4 matrices are handed
to the rasterizer transposed (the kernels read them column-major).

Splats are generated in the camera frame (the reference moves means into the
camera frame with a bmm before the op and renders with viewmatrix = I,
/root/reference/src/Render.cc:750-752) with the SinglePixel scale init of
/root/reference/src/Gaussian.cc:70-74 and logit-opacity 1 (:55).

Randomness is a counter-based splitmix64 hash so a scene depends only on
(seed, index, field).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(seed: int, fld: int, n: int) -> np.ndarray:
    """U[0,1) float64, one value per index, counter based."""
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64)
        key = (np.uint64(seed) << np.uint64(48)) ^ (np.uint64(fld) << np.uint64(40)) ^ i
        h = _splitmix(_splitmix(key))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed: int, fld: int, n: int) -> np.ndarray:
    u1 = uniform(seed, 2 * fld + 1000, n)
    u2 = uniform(seed, 2 * fld + 1001, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


@dataclass
class Camera:
    """What GaussianRasterizationSettings carries (include/Rasterizer.cuh:79-91)."""
    width: int
    height: int
    fx: float
    fy: float
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray   # [4,4] float32, = Tcw^T (row-major storage)
    projmatrix: np.ndarray   # [4,4] float32, = (P @ Tcw)^T
    campos: np.ndarray       # [3]
    bg: np.ndarray = field(default_factory=lambda: np.zeros(3, np.float32))
    scale_modifier: float = 1.0
    sh_degree: int = 0


def make_camera(width: int, height: int, fx: float, fy: float, Tcw: np.ndarray | None = None,
                near: float = 0.01, far: float = 100.0, bg=(0.0, 0.0, 0.0)) -> Camera:
    """src/Camera.cc:7-47 (and scripts/replay.py:91-120)."""
    tanfovx = np.float32(width / (2 * np.float32(fx)))
    tanfovy = np.float32(height / (2 * np.float32(fy)))
    top = tanfovy * np.float32(near)
    bottom = -top
    right = tanfovx * np.float32(near)
    left = -right
    n, f = np.float32(near), np.float32(far)
    Pm = np.array([[2 * n / (right - left), 0.0, (right + left) / (right - left), 0.0],
                   [0.0, 2 * n / (top - bottom), (top + bottom) / (top - bottom), 0.0],
                   [0.0, 0.0, f / (f - n), -(f * n) / (f - n)],
                   [0.0, 0.0, 1.0, 0.0]], dtype=np.float32)
    if Tcw is None:
        Tcw = np.eye(4, dtype=np.float32)
    Tcw = np.asarray(Tcw, dtype=np.float32)
    view = np.ascontiguousarray(Tcw.T)                      # _viewmatrix = Tcw^T
    full = np.ascontiguousarray((view @ Pm.T).astype(np.float32))  # view.bmm(proj^T)
    campos = np.linalg.inv(Tcw.astype(np.float64))[:3, 3].astype(np.float32)
    return Camera(width, height, float(fx), float(fy), float(tanfovx), float(tanfovy), view, full,
                  campos, np.asarray(bg, np.float32))


REPLICA = dict(width=1200, height=680, fx=600.0, fy=600.0)          # Examples/RGB-D/replica.yaml:12-17
TUM1 = dict(width=640, height=480, fx=517.306408, fy=516.469215)    # Examples/RGB-D/tum/TUM1.yaml:13-16
SCANNET = dict(width=640, height=480, fx=577.590698, fy=578.729797)  # Examples/RGB-D/scannet.yaml:11-14
CAMERAS = dict(replica=REPLICA, tum=TUM1, scannet=SCANNET)


@dataclass
class Scene:
    cam: Camera
    means3D: np.ndarray      # [P,3] camera-frame (or world if cam has a pose)
    scales: np.ndarray       # [P,3]  (activated: exp(log_scale))
    rotations: np.ndarray    # [P,4]  (normalised r,x,y,z)
    opacities: np.ndarray    # [P,1]  (activated: sigmoid)
    colors: np.ndarray | None  # [P,3] or None when shs is used
    shs: np.ndarray | None = None  # [P,M,3]
    dL_dpix: np.ndarray | None = None  # [3,H,W] upstream gradient

    @property
    def P(self) -> int:
        return int(self.means3D.shape[0])


def make_scene(P: int, cam: Camera, seed: int = 0, scale_mult: float = 1.0,
               color_mode: str = "rgb", sh_degree: int | None = None,
               z_range=(0.5, 6.0), frac_behind: float = 0.0, frac_offscreen: float = 0.0) -> Scene:
    """Random splats seen by `cam`.

    color_mode: "rgb" (colours ~U[0,1]) | "depth" (colours = [z,1,0], the
    reference's depth/silhouette render, src/Render.cc:949-981) | "sh".
    """
    W, H = cam.width, cam.height
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    u = uniform(seed, 0, P) * W
    v = uniform(seed, 1, P) * H
    if frac_offscreen > 0:  # push some splats outside the image (still binned if radius reaches in)
        off = uniform(seed, 20, P) < frac_offscreen
        u = np.where(off, u * 1.6 - 0.3 * W, u)
        v = np.where(off, v * 1.6 - 0.3 * H, v)
    z = z_range[0] + uniform(seed, 2, P) * (z_range[1] - z_range[0])
    if frac_behind > 0:     # some behind the near cull plane (z <= 0.2)
        beh = uniform(seed, 21, P) < frac_behind
        z = np.where(beh, uniform(seed, 22, P) * 0.4 - 0.1, z)
    means_c = np.stack([(u - cx) * z / cam.fx, (v - cy) * z / cam.fy, z], 1)
    # camera frame -> the frame the camera's viewmatrix expects
    Tcw = cam.viewmatrix.T.astype(np.float64)
    Twc = np.linalg.inv(Tcw)
    means = (means_c @ Twc[:3, :3].T + Twc[:3, 3]).astype(np.float32)

    fbar = 0.5 * (cam.fx + cam.fy)
    s0 = z / fbar                                               # src/Gaussian.cc:70-74
    scales = np.stack([s0 * np.exp(0.3 * normal(seed, 3 + k, P)) for k in range(3)], 1) * scale_mult
    q = np.stack([1.0 + 0.1 * normal(seed, 6, P)] + [0.1 * normal(seed, 7 + k, P) for k in range(3)], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-(1.0 + normal(seed, 10, P))))   # logit init 1, src/Gaussian.cc:55

    colors = shs = None
    if color_mode == "rgb":
        colors = np.stack([uniform(seed, 11 + k, P) for k in range(3)], 1).astype(np.float32)
    elif color_mode == "depth":
        colors = np.stack([z, np.ones(P), np.zeros(P)], 1).astype(np.float32)
    elif color_mode == "sh":
        deg = 3 if sh_degree is None else sh_degree
        M = (deg + 1) ** 2
        sh = np.stack([normal(seed, 30 + k, P) for k in range(M * 3)], 1).reshape(P, M, 3) * 0.3
        sh[:, 0, :] += 1.0
        shs = sh.astype(np.float32)
        cam.sh_degree = deg
    else:
        raise ValueError(color_mode)

    g = np.stack([normal(seed, 100 + k, W * H) for k in range(3)], 0).reshape(3, H, W).astype(np.float32)
    return Scene(cam, np.ascontiguousarray(means), scales.astype(np.float32), q.astype(np.float32),
                 opac.astype(np.float32).reshape(P, 1), colors, shs, g)
