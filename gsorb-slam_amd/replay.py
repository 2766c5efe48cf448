"""Replay / evaluation harness and map I/O.

Counterpart of the reference's only Python caller, scripts/replay.py: load the map GSORB-SLAM
saved (`GaussianModel.ply`, written by src/Utils.cc:182-280) and the estimated trajectory
(`CarameTrajectory.txt`, src/System.cc:598-726), re-render every frame through the Python
operator (two renders per frame: colours, then depth colours) and report PSNR / depth-L1.
MS-SSIM and LPIPS need packages that are not available offline and are left out; image files
are handed in as arrays by the caller (no OpenCV dependency here).

PLY format (the reference writes it with tinyply, an empty submodule there, and reads it with
plyfile — parity unpinned by any reference test): binary little-endian, one element `vertex`,
float32 properties `x y z rgb_0 rgb_1 rgb_2 opacity scale_0 scale_1 scale_2 rot_0 rot_1 rot_2 rot_3`,
holding the RAW parameters (logit opacity, log scale, un-normalised quaternion).
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

PLY_PROPS = ["x", "y", "z", "rgb_0", "rgb_1", "rgb_2", "opacity", "scale_0", "scale_1", "scale_2",
             "rot_0", "rot_1", "rot_2", "rot_3"]


@dataclass
class GaussianModel:
    """Raw map parameters, as saved (scripts/replay.py:38-83)."""
    xyz: np.ndarray        # [P,3]
    rgb: np.ndarray        # [P,3]
    opacity: np.ndarray    # [P,1] logit
    scaling: np.ndarray    # [P,3] log
    rotation: np.ndarray   # [P,4] un-normalised (r,x,y,z)


def write_ply(path: str, m: GaussianModel) -> None:
    """src/Utils.cc:211-280 (ConstructListAttributes + WriteOutputPly)."""
    P = m.xyz.shape[0]
    data = np.concatenate([m.xyz, m.rgb, m.opacity.reshape(P, 1), m.scaling, m.rotation], 1).astype("<f4")
    assert data.shape[1] == len(PLY_PROPS)
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    hdr += "".join("property float %s\n" % n for n in PLY_PROPS) + "end_header\n"
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(np.ascontiguousarray(data).tobytes())


def read_ply(path: str) -> GaussianModel:
    """scripts/replay.py:38-83; properties are looked up by name, scale_*/rot_* sorted by index."""
    with open(path, "rb") as f:
        raw = f.read()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    lines = raw[:end].decode("ascii").split("\n")
    if not lines[0].startswith("ply") or "binary_little_endian" not in lines[1]:
        raise ValueError("only binary little-endian PLY is supported")
    n, props, in_vertex = 0, [], False
    for ln in lines:
        t = ln.split()
        if t[:2] == ["element", "vertex"]:
            n, in_vertex = int(t[2]), True
        elif t[:1] == ["element"]:
            in_vertex = False
        elif t[:1] == ["property"] and in_vertex:
            if t[1] not in ("float", "float32"):
                raise ValueError("vertex properties must be float32")
            props.append(t[2])
    a = np.frombuffer(raw, "<f4", n * len(props), end).reshape(n, len(props))
    col = {p: i for i, p in enumerate(props)}
    pick = lambda names: np.stack([a[:, col[k]] for k in names], 1).astype(np.float32)
    by_idx = lambda pre: sorted([p for p in props if p.startswith(pre)], key=lambda s: int(s.split("_")[-1]))
    return GaussianModel(pick(["x", "y", "z"]), pick(["rgb_0", "rgb_1", "rgb_2"]), pick(["opacity"]),
                         pick(by_idx("scale_")), pick(by_idx("rot_")))


def _quat_to_rot(q):  # (qx, qy, qz, qw), scipy convention used by scripts/replay.py:172-175
    x, y, z, w = q
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def read_trajectory(path: str, kind: str):
    """Poses as 4x4 float32 arrays. kind: 'replica' (16 numbers per line), 'scannet' (index + 16),
    'tum' (timestamp tx ty tz qx qy qz qw)  — scripts/replay.py:163-232."""
    poses, stamps = [], []
    for ln in open(path, "r", encoding="utf-8"):
        if not ln.strip() or ln.strip().startswith("#"):
            continue
        t = ln.split()
        T = np.eye(4, dtype=np.float32)
        if kind == "replica":
            M = np.array(t[:16], np.float32).reshape(4, 4)
            T[:3, :3], T[:3, 3] = M[:3, :3], M[:3, 3]
        elif kind == "scannet":
            M = np.array(t[1:17], np.float32).reshape(4, 4)
            T[:3, :3], T[:3, 3] = M[:3, :3], M[:3, 3]
            stamps.append(t[0])
        elif kind == "tum":
            T[:3, :3] = _quat_to_rot([float(v) for v in t[4:8]])
            T[:3, 3] = np.array(t[1:4], np.float32)
            stamps.append(float(t[0]))
        else:
            raise ValueError(kind)
        poses.append(T)
    return poses, stamps


def read_trajectory_matrices(path: str) -> np.ndarray:
    """scripts/eval_ate.py:35-54: one 4x4 matrix per line (16 numbers, or 17 with a leading timestamp);
    '#' lines and lines of any other length are skipped. Returns [N,4,4] float64."""
    out = []
    with open(path, "r", encoding="utf-8") as f:
        for ln in f:
            if ln.startswith("#"):
                continue
            v = [float(x) for x in ln.strip().split()]
            if len(v) == 17:
                v = v[1:]
            if len(v) != 16:
                continue
            out.append(np.array(v).reshape(4, 4))
    return np.array(out)


def align_umeyama(model: np.ndarray, data: np.ndarray):
    """scripts/eval_ate.py:6-33: rigid (rotation + translation, no scale) least-squares alignment of the
    3xN point set `model` onto `data` by SVD of the cross-covariance. Returns (R [3,3], t [3,1],
    per-point translation error [N])."""
    mm, dm = model.mean(1, keepdims=True), data.mean(1, keepdims=True)
    Wm = (model - mm) @ (data - dm).T                       # sum of outer products
    U, _, Vh = np.linalg.svd(Wm.T, full_matrices=False)
    S = np.identity(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    R = U @ S @ Vh
    t = dm - R @ mm
    err = R @ model + t - data
    return R, t, np.sqrt((err ** 2).sum(0))


def ate_rmse(gt_traj, est_traj) -> float:
    """scripts/eval_ate.py:56-82 (`evaluate_ate`): translation parts of the first min(len) pose pairs without
    an inf entry, ground truth aligned onto the estimate, and — like the reference, whose result line calls
    it "ATE RMSE" — the MEAN of the per-pose translation errors (metres)."""
    gt_traj, est_traj = np.asarray(gt_traj, np.float64), np.asarray(est_traj, np.float64)
    n = min(len(gt_traj), len(est_traj))
    if n == 0:
        raise ValueError("Empty trajectory input.")
    ok = [i for i in range(n) if not np.any(np.isinf(gt_traj[i])) and not np.any(np.isinf(est_traj[i]))]
    if not ok:
        raise ValueError("No valid trajectory point pairs found.")
    g = np.array([gt_traj[i][:3, 3] for i in ok]).T
    e = np.array([est_traj[i][:3, 3] for i in ok]).T
    return float(np.mean(align_umeyama(g, e)[2]))


def _dgr():
    pkg = os.path.dirname(os.path.abspath(__file__))
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    import diff_gaussian_rasterization as dgr
    return dgr


def setup_camera(w, h, k, w2c, near=0.01, far=100.0, device="cuda"):
    """scripts/replay.py:91-120 -> GaussianRasterizationSettings."""
    dgr = _dgr()
    fx, fy = float(k[0][0]), float(k[1][1])
    w2c_t = torch.as_tensor(np.asarray(w2c), dtype=torch.float32, device=device)
    cam_center = torch.inverse(w2c_t)[:3, 3]
    w2c_b = w2c_t.unsqueeze(0).transpose(1, 2)
    tanfovx, tanfovy = w / (2 * fx), h / (2 * fy)
    top = tanfovy * near
    right = tanfovx * near
    proj = torch.tensor([[2 * near / (2 * right), 0.0, 0.0, 0.0], [0.0, 2 * near / (2 * top), 0.0, 0.0],
                         [0.0, 0.0, far / (far - near), -(far * near) / (far - near)], [0.0, 0.0, 1.0, 0.0]],
                        dtype=torch.float32, device=device).unsqueeze(0).transpose(1, 2)
    full = w2c_b.bmm(proj)
    return dgr.GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=tanfovx, tanfovy=tanfovy,
        bg=torch.zeros(3, dtype=torch.float32, device=device), scale_modifier=1.0, viewmatrix=w2c_b[0].contiguous(),
        projmatrix=full[0].contiguous(), sh_degree=0, campos=cam_center, prefiltered=False)


def calc_psnr(a, b):
    """scripts/replay.py:245-247."""
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


class Replayer:
    def __init__(self, model: GaussianModel, intrinsics, width: int, height: int, device="cuda"):
        self.dgr = _dgr()
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
        self.xyz, self.rgb = t(model.xyz), t(model.rgb)
        self.opac, self.scal, self.rot = t(model.opacity), t(model.scaling), t(model.rotation)
        self.cam = setup_camera(width, height, intrinsics, np.eye(4, dtype=np.float32), device=device)
        self.device = device

    @torch.no_grad()
    def render(self, w2c):
        """Two renders of one frame (scripts/replay.py:314-325): (rgb [3,H,W], surface depth [1,H,W],
        alpha-blended depth [1,H,W])."""
        w2c = torch.as_tensor(np.asarray(w2c), dtype=torch.float32, device=self.device)
        ones = torch.ones(self.xyz.shape[0], 1, device=self.device)
        pts = (w2c @ torch.cat((self.xyz, ones), 1).T).T[:, :3].contiguous()      # transform_to_frame, :234-243
        common = dict(means3D=pts, rotations=F.normalize(self.rot), opacities=torch.sigmoid(self.opac),
                      scales=torch.exp(self.scal), means2D=torch.zeros_like(pts))
        r = self.dgr.GaussianRasterizer(raster_settings=self.cam)
        rgb, _, surf = r(colors_precomp=self.rgb, **common)
        dcol = torch.zeros_like(pts)
        dcol[:, 0] = pts[:, 2]                                                     # replay.py:136-152: [z, 0, 0]
        dep, _, _ = r(colors_precomp=dcol, **common)
        return rgb, surf, dep[0:1]

    def evaluate(self, poses, colors=None, depths=None):
        """poses: list of w2c 4x4 as saved; the first frame defines the world (replay.py:308-312).
        colors[i] [3,H,W] in 0..1 and depths[i] [1,H,W] in metres are optional ground truth."""
        world_center = torch.eye(4)
        out = dict(psnr=[], depth_l1=[], frames=len(poses))
        for i, Tp in enumerate(poses):
            w2c_raw = torch.as_tensor(np.asarray(Tp), dtype=torch.float32)
            if i == 0:
                world_center = torch.inverse(w2c_raw)
            w2c = torch.inverse(world_center @ w2c_raw)
            rgb, surf, _ = self.render(w2c.numpy())
            if colors is not None:
                gt = torch.as_tensor(colors[i], dtype=torch.float32, device=self.device)
                mask = torch.ones_like(gt[:1], dtype=torch.bool)
                if depths is not None:
                    d = torch.as_tensor(depths[i], dtype=torch.float32, device=self.device)
                    mask = d > 0
                    out["depth_l1"].append(float(torch.abs((surf - d)[mask]).mean()))
                out["psnr"].append(float(calc_psnr(rgb * mask, gt * mask).mean()))
        out["mean_psnr"] = float(np.mean(out["psnr"])) if out["psnr"] else None
        out["mean_depth_l1"] = float(np.mean(out["depth_l1"])) if out["depth_l1"] else None
        return out
