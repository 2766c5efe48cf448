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
    dict(mode="sh", mult=2.0, Tcw=pose(-0.4, (0.2, 0.1, -0.1)), bg=(0.2, 0.0, 0.4), sh_degree=2),
    dict(mode="sh", mult=2.0, Tcw=None, bg=(0, 0, 0), sh_degree=1),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['mode']}{c.get('sh_degree', '')}-x{c['mult']}")
def test_oracle_matches_fp64_autograd(syn, case):
    cam = syn.make_camera(80, 56, 60.0, 55.0, Tcw=case["Tcw"], bg=case["bg"])
    sc = syn.make_scene(300, cam, seed=3, scale_mult=case["mult"], color_mode=case["mode"],
                        frac_behind=0.1, frac_offscreen=0.3, **({"sh_degree": case["sh_degree"]} if "sh_degree" in case else {}))
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


def test_oracle_cov3d_precomp_path_matches_fp64_autograd(syn):
    """cov3D_precomp (forward.cu:94-101 taken as input, backward.cu:144-274 returns dL_dcov3D): the covariances are
    random SPD matrices, not products of the scene's scales / rotations."""
    cam = syn.make_camera(80, 56, 60.0, 55.0, Tcw=pose(0.25, (0.1, 0.0, -0.1)), bg=(0.4, 0.1, 0.2))
    sc = syn.make_scene(300, cam, seed=8, scale_mult=2.5, frac_behind=0.1, frac_offscreen=0.2)
    rng = np.random.default_rng(5)
    A = rng.standard_normal((sc.P, 3, 3)) * sc.scales.mean(1)[:, None, None]
    S = A @ A.transpose(0, 2, 1) + (0.2 * sc.scales.mean(1) ** 2)[:, None, None] * np.eye(3)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    o = oracle.Oracle()
    f = o.forward(means3D=sc.means3D, opacities=sc.opacities, cam=sc.cam, colors=sc.colors, cov3D_precomp=cov)
    b = o.backward(sc.dL_dpix)
    s = spec_fp64.render(sc, f.radii, f.stages["means2D"], sc.dL_dpix, cov3D_precomp=cov)
    tol = 2e-5
    assert rel_err(f.color, s["color"]) < tol
    assert np.abs(f.depth[0] - s["depth"]).max() < 1e-5
    g = s["grads"]
    # the symmetric-matrix convention: off-diagonal entries of the 6-vector feed two matrix elements each
    assert rel_err(b.dL_dcov3D, g["cov3D"]) < tol
    assert rel_err(b.dL_dmeans3D, g["means3D"]) < tol
    assert rel_err(b.dL_dopacity.ravel(), g["opacities"]) < tol
    assert rel_err(b.dL_dcolors, g["colors"]) < tol
    assert float(np.abs(b.dL_dscales).max()) == 0.0 and float(np.abs(b.dL_drotations).max()) == 0.0


def test_oracle_matches_fp64_autograd_on_a_2000_splat_odd_frame(syn):
    """203x149 (ragged edge tiles), 2000 splats with a pose, background and off-screen / near-culled splats:
    lists of ~100 entries per tile, early termination active."""
    cam = syn.make_camera(203, 149, 150.0, 152.0, Tcw=pose(0.1, (0.05, -0.05, 0.1)), bg=(0.3, 0.5, 0.7))
    sc = syn.make_scene(2000, cam, seed=13, scale_mult=3.0, frac_behind=0.1, frac_offscreen=0.3)
    o, f = oracle.forward_scene(sc)
    b = o.backward(sc.dL_dpix)
    s = spec_fp64.render(sc, f.radii, f.stages["means2D"], sc.dL_dpix, chunk=2048)
    tol = 2e-5
    assert rel_err(f.color, s["color"]) < tol
    assert rel_err(f.stages["final_T"].reshape(149, 203), s["final_T"]) < tol
    mism = np.abs(f.depth[0] - s["depth"]) > 1e-5        # median depth flips only where T sits on 0.5 to fp32 rounding
    assert mism.mean() < 2e-4
    g = s["grads"]
    for a, r in ((b.dL_dmeans3D, g["means3D"]), (b.dL_dscales, g["scales"]), (b.dL_drotations, g["rotations"]),
                 (b.dL_dopacity.ravel(), g["opacities"]), (b.dL_dcolors, g["colors"])):
        assert rel_err(a, r) < tol
