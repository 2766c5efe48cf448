"""Shared helpers of the parity tests."""
import numpy as np


def rel_err(a, b):
    """max |a-b| normalised by the tensor's own scale (max |b|): the '1e-4 rel fp32' bar of
    BASELINE.json is applied per output tensor, not per element (elements that cancel to ~0 have
    no meaningful element-wise relative error)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def mixed_err(a, b, rtol=1e-4, afloor=1e-6):
    """Element-wise bar: |a-b| <= rtol*|b| + afloor*max|b| for EVERY element. Returns the worst ratio
    |a-b| / (rtol*|b| + afloor*max|b|): <= 1 passes. The absolute floor (1e-6 of the tensor's largest element, i.e.
    ~10 fp32 ulps of it) is what a sum of thousands of fp32 terms of either sign can resolve at all; above it every
    element must hold 1e-4 relative to ITSELF, which max-normalised rel_err does not demand."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    bound = rtol * np.abs(b) + afloor * max(np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) / bound).max())


def pose(theta=0.3, t=(0.1, -0.2, 0.3)):
    T = np.eye(4)
    T[:3, :3] = [[np.cos(theta), 0, np.sin(theta)], [0, 1, 0], [-np.sin(theta), 0, np.cos(theta)]]
    T[:3, 3] = t
    return T


def run_oracle(scene, oracle_mod, backward=True, cov3D=None):
    o = oracle_mod.Oracle()
    f = o.forward(means3D=scene.means3D, opacities=scene.opacities, cam=scene.cam, colors=scene.colors,
                  shs=scene.shs, scales=None if cov3D is not None else scene.scales,
                  rotations=None if cov3D is not None else scene.rotations, cov3D_precomp=cov3D)
    b = o.backward(scene.dL_dpix) if backward else None
    return o, f, b
