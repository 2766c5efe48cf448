"""Generates tests/golden/ref_loss.npz by IMPORTING the reference's own Python loss / camera helpers
(Thirdparty/diff_gaussian_rasterization/utils/). Run only in the build container (needs /root/reference); the .npz it writes is
committed. Nothing of the reference's source is copied: the fixture holds inputs and what the reference functions returned.

  loss_utils._ssim(img1, img2, window, 11, channel)   loss_utils.py:53-74 — fed the window GSORB-SLAM's C++ side builds
      (src/Utils.cc:67-74: exp(-floor((x - 11)/2)^2 / (2 sigma^2)), normalised: asymmetric), i.e. the function gsr_ssim_forward and
      harness.ssim_torch restate; also with the reference's own symmetric create_window, and the image gradient through autograd
  loss_utils.l1_loss                                   loss_utils.py:20-21
  graphics_utils.getWorld2View2(R, t, translate, scale)   graphics_utils.py:38-49 — the view-matrix convention
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


def gsorb_window(channel):
    g = torch.tensor([math.exp(-(math.floor((x - 11) / 2.0) ** 2) / (2.0 * 1.5 * 1.5)) for x in range(11)], dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).unsqueeze(0).unsqueeze(0).expand(channel, 1, 11, 11).contiguous()


def main():
    lu, gfx = _load("loss_utils"), _load("graphics_utils")
    g = torch.Generator().manual_seed(77)
    out = {}
    for name, shape in (("a", (3, 37, 53)), ("b", (1, 24, 40))):
        img1 = torch.rand(shape, generator=g)
        img2 = (img1 + 0.15 * torch.randn(shape, generator=g)).clamp(0, 1)
        x = img1.clone().requires_grad_(True)
        v = lu._ssim(x, img2, gsorb_window(shape[0]), 11, shape[0], True)
        v.backward()
        out[f"{name}_img1"], out[f"{name}_img2"] = img1.numpy(), img2.numpy()
        out[f"{name}_ssim_gsorb_window"] = np.float32(v.item())
        out[f"{name}_dssim_dimg1_gsorb_window"] = x.grad.numpy()
        out[f"{name}_ssim_symmetric_window"] = np.float32(lu.ssim(img1, img2).item())
        out[f"{name}_l1"] = np.float32(lu.l1_loss(img1, img2).item())
    rng = np.random.default_rng(5)
    th = 0.4
    R = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]], np.float64)
    t = rng.standard_normal(3)
    out["w2v_R"], out["w2v_t"] = R, t
    out["w2v"] = gfx.getWorld2View2(R, t)
    out["w2v_translated"] = gfx.getWorld2View2(R, t, np.array([0.1, -0.2, 0.3]), 1.5)
    np.savez(os.path.join(HERE, "ref_loss.npz"), **out)
    print("written", sorted(out))


if __name__ == "__main__":
    main()
