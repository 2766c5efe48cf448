"""Generates tests/golden/ref_utils.npz by IMPORTING the reference's own Python helpers.

Run only in the build container (needs /root/reference, which does not exist on the GPU
box); the .npz it writes is committed. Nothing of the reference's source is copied: the
fixture holds inputs and the outputs the reference functions returned for them.

Reference functions called (Thirdparty/diff_gaussian_rasterization/utils/):
  sh_utils.eval_sh(deg, sh, dirs)                      sh_utils.py:57-118
  graphics_utils.getProjectionMatrix(znear, zfar, fovX, fovY)   graphics_utils.py:51-71
  graphics_utils.geom_transform_points(points, transf_matrix)   graphics_utils.py:22-29
  graphics_utils.focal2fov / fov2focal                 graphics_utils.py:73-77
  image_utils.psnr                                     image_utils.py:17-19
These are the only parts of the rasterizer path the reference ships in runnable
(non-CUDA) form; they pin the oracle's SH basis, projection convention and PSNR.
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference/Thirdparty/diff_gaussian_rasterization/utils"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    sh_utils, gfx, img = _load("sh_utils"), _load("graphics_utils"), _load("image_utils")
    rng = np.random.default_rng(1234)
    P = 96
    out = {}
    # --- SH evaluation, degrees 0..3 (reference layout: sh[..., C, coeffs]) ---
    sh = (rng.standard_normal((P, 16, 3)) * 0.4).astype(np.float32)   # rasterizer layout [P,M,3]
    dirs = rng.standard_normal((P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out["sh"] = sh
    out["dirs"] = dirs
    for deg in range(4):
        r = sh_utils.eval_sh(deg, torch.from_numpy(sh).transpose(1, 2), torch.from_numpy(dirs))
        out[f"eval_sh_deg{deg}"] = r.numpy().astype(np.float32)
    # --- projection matrix + point transform ---
    W, H, fx, fy = 640, 480, 517.306408, 516.469215
    fovx, fovy = gfx.focal2fov(fx, W), gfx.focal2fov(fy, H)
    Pm = gfx.getProjectionMatrix(0.01, 100.0, fovx, fovy)            # [4,4], row-major math
    out["cam"] = np.array([W, H, fx, fy, 0.01, 100.0], np.float64)
    out["fov"] = np.array([fovx, fovy, gfx.fov2focal(fovx, W), gfx.fov2focal(fovy, H)], np.float64)
    out["proj"] = Pm.numpy().astype(np.float32)
    th = 0.25
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = [[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]]
    Tcw[:3, 3] = [0.05, -0.1, 0.2]
    out["Tcw"] = Tcw
    view_t = torch.from_numpy(Tcw).T                                  # world_view_transform (transposed)
    full_t = (view_t.unsqueeze(0).bmm(Pm.T.unsqueeze(0))).squeeze(0)  # full_proj_transform (transposed)
    pts = np.stack([rng.uniform(-1.5, 1.5, P), rng.uniform(-1.0, 1.0, P), rng.uniform(0.6, 5.0, P)], 1).astype(np.float32)
    out["points"] = pts
    out["full_proj_t"] = full_t.numpy().astype(np.float32)
    out["ndc"] = gfx.geom_transform_points(torch.from_numpy(pts), full_t).numpy().astype(np.float32)
    # --- psnr ---
    a = rng.uniform(0, 1, (2, 3, 8, 8)).astype(np.float32)
    b = np.clip(a + rng.normal(0, 0.05, a.shape), 0, 1).astype(np.float32)
    out["psnr_a"], out["psnr_b"] = a, b
    out["psnr"] = img.psnr(torch.from_numpy(a), torch.from_numpy(b)).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_utils.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_utils.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
