"""CPU: the C oracle against the independent fp64 autograd statement of the same maths
(tests/spec_fp64.py). Covers both colour modes, SH, a camera pose, background, fat splats
(fov clamp + saturation), near-culling and off-screen splats."""
import numpy as np
import pytest

import spec_fp64
from oracle import oracle
from util import pose, rel_err

CASES = [
    dict(mode="rgb", mult=1.0, Tcw=None, bg=(0, 0, 0)),
    dict(mode="rgb", mult=4.0, Tcw=pose(), bg=(0.3, 0.5, 0.7)),
    dict(mode="depth", mult=2.0, Tcw=None, bg=(0, 0, 0)),
    dict(mode="sh", mult=3.0, Tcw=pose(), bg=(0.1, 0.2, 0.3)),
    dict(mode="rgb", mult=12.0, Tcw=pose(0.2), bg=(0.3, 0.5, 0.7)),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['mode']}-x{c['mult']}")
def test_oracle_matches_fp64_autograd(syn, case):
    cam = syn.make_camera(80, 56, 60.0, 55.0, Tcw=case["Tcw"], bg=case["bg"])
    sc = syn.make_scene(300, cam, seed=3, scale_mult=case["mult"], color_mode=case["mode"],
                        frac_behind=0.1, frac_offscreen=0.3)
    o, f = oracle.forward_scene(sc)
    b = o.backward(sc.dL_dpix)
    s = spec_fp64.render(sc, f.radii, f.stages["means2D"], sc.dL_dpix)
    tol = 2e-5  # fp32 pipeline vs fp64 truth
    assert rel_err(f.color, s["color"]) < tol
    assert rel_err(f.stages["final_T"].reshape(56, 80), s["final_T"]) < tol
    assert np.abs(f.depth[0] - s["depth"]).max() < 1e-5
    g = s["grads"]
    assert rel_err(b.dL_dmeans3D, g["means3D"]) < tol
    assert rel_err(b.dL_dscales, g["scales"]) < tol
    assert rel_err(b.dL_drotations, g["rotations"]) < tol
    assert rel_err(b.dL_dopacity.ravel(), g["opacities"]) < tol
    if case["mode"] == "sh":
        assert rel_err(b.dL_dsh, g["shs"]) < tol
    else:
        assert rel_err(b.dL_dcolors, g["colors"]) < tol
