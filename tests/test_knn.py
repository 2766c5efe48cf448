"""simple_knn / distCUDA2 (reference src/simple_knn.cu, src/spatial.cu).

CPU: the oracle's brute-force restatement against scipy's cKDTree (independent exact k-NN).
GPU: the HIP path (gsr_dist2 through the C ABI, and the libtorch/Python distCUDA2 wrappers)
against the oracle. Squared distances are sums of three products: 1e-6 relative covers the
contraction/association freedom; everything else is exact (it is a selection problem)."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle


def _cloud(kind, P, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-3, 3, (P, 3)).astype(np.float32)
    if kind == "surface":        # what a SLAM map looks like: points on a few surfaces, uneven density
        u, v = rng.uniform(-2, 2, (2, P))
        z = np.where(rng.random(P) < 0.5, 2.0 + 0.01 * rng.standard_normal(P), 0.5 * u + 4.0)
        return np.stack([u * rng.choice([1.0, 0.1], P), v, z], 1).astype(np.float32)
    if kind == "dupes":          # duplicates and a far outlier
        p = rng.uniform(0, 1, (P, 3)).astype(np.float32)
        p[1::7] = p[0::7][: len(p[1::7])]
        p[-1] = [50, -40, 30]
        return p
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "surface", "dupes"])
def test_oracle_matches_kdtree(kind):
    from scipy.spatial import cKDTree
    pts = _cloud(kind, 3000, 1)
    d = oracle.dist2(pts)
    dd, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    ref = (dd[:, 1:] ** 2).mean(1)
    np.testing.assert_allclose(d, ref, rtol=2e-5, atol=1e-12)


def test_oracle_small_counts():
    big = np.float32(3.402823466e+38)
    with np.errstate(over="ignore"):
        assert np.isinf(oracle.dist2(np.zeros((1, 3), np.float32))[0])          # (FLT_MAX*3)/3 overflows like the reference
        d = oracle.dist2(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32))
    assert np.isinf(d).all() or (d > big / 4).all()                              # fewer than 3 neighbours: FLT_MAX terms
    d = oracle.dist2(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]], np.float32))
    np.testing.assert_allclose(d, [(1 + 4 + 9) / 3, (1 + 5 + 10) / 3, (4 + 5 + 13) / 3, (9 + 10 + 13) / 3], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,P", [("uniform", 20000), ("surface", 30000), ("dupes", 5000), ("uniform", 700),
                                      ("surface", 5)])
def test_hip_dist2_matches_oracle(gsr, kind, P):
    pts = _cloud(kind, P, 2)
    d = gsr.dist2(pts).cpu().numpy()
    np.testing.assert_allclose(d, oracle.dist2(pts), rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_hip_dist2_edge_counts(gsr):
    import torch
    assert gsr.dist2(np.zeros((0, 3), np.float32)).numel() == 0
    with np.errstate(over="ignore"):
        for n in (1, 2, 3):
            pts = _cloud("uniform", n, 3)
            d = gsr.dist2(pts).cpu().numpy()
            assert (d > 1e37).all()
    pts = _cloud("uniform", 4, 3)
    np.testing.assert_allclose(gsr.dist2(pts).cpu().numpy(), oracle.dist2(pts), rtol=1e-6)


@pytest.mark.gpu
def test_hip_dist2_million_points_properties(gsr):
    """1 M points (the BASELINE scale): properties only — every value is the mean of 3 real neighbour
    distances: compare a random sample with an exact kd-tree."""
    from scipy.spatial import cKDTree
    pts = _cloud("surface", 1_000_000, 5)
    d = gsr.dist2(pts).cpu().numpy()
    assert np.isfinite(d).all() and (d >= 0).all()
    sel = np.random.default_rng(0).choice(len(pts), 2000, replace=False)
    dd, _ = cKDTree(pts.astype(np.float64)).query(pts[sel].astype(np.float64), k=4)
    np.testing.assert_allclose(d[sel], (dd[:, 1:] ** 2).mean(1), rtol=5e-5, atol=1e-12)


@pytest.mark.gpu
def test_python_and_cpp_distCUDA2(syn):
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gsorb-slam_amd"))
    import diff_gaussian_rasterization as dgr
    pts = _cloud("surface", 8000, 4)
    out = dgr._C.distCUDA2(torch.tensor(pts, device="cuda"))
    assert out.shape == (8000,) and out.dtype == torch.float32
    np.testing.assert_allclose(out.cpu().numpy(), oracle.dist2(pts), rtol=1e-6)


@pytest.mark.gpu
def test_densification_at_config5_size_feeds_the_rasterizer(gsr, syn):
    """BASELINE.json config 5 on one GPU: 2 M points -> distCUDA2 (src/Gaussian.cc:59-69, InitScalarMethod Distance:
    scale = sqrt(max(dist2, 1e-7))) -> forward + backward on the ScanNet camera. Size-independent properties on the
    full set; a 50k-point subsample is run through the SAME pipeline and checked against the oracle (brute-force
    3-NN and the rasterizer restatement)."""
    import torch
    from util import rel_err
    cam = syn.make_camera(**syn.SCANNET)
    P = 2_000_000
    sc = syn.make_scene(P, cam, seed=4)
    pts = torch.tensor(sc.means3D, device="cuda")
    d2 = gsr.dist2(pts)
    assert d2.shape == (P,) and bool(torch.isfinite(d2).all()) and float(d2.min()) >= 0.0
    # every mean-of-3-NN distance is bounded by the distance to ANY three other points (here: index neighbours)
    with torch.no_grad():
        nb = sum(((pts - torch.roll(pts, k, 0)) ** 2).sum(1) for k in (1, 2, 3)) / 3.0
    assert bool((d2 <= nb * (1 + 1e-5)).all())
    scales = torch.sqrt(torch.clamp_min(d2, 1e-7)).unsqueeze(-1).repeat(1, 3)
    s = gsr.capi.Settings.from_camera(cam)
    kw = dict(colors=sc.colors, scales=scales, rotations=sc.rotations)
    st = gsr.forward(s, sc.means3D, sc.opacities, **kw)
    d = gsr.debug_export(st)
    assert st.num_rendered == int(d["tiles_touched"].sum()) and st.num_rendered > P
    r = d["ranges"].astype(np.int64)
    assert int((r[:, 1] - r[:, 0]).sum()) == st.num_rendered
    assert bool(torch.isfinite(st.color).all()) and float(st.color.max()) <= 1.0 + 1e-4
    g = gsr.backward(st, sc.dL_dpix)
    for n in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dcolors"):
        assert bool(torch.isfinite(getattr(g, n)).all()), n
    assert float(g.dL_dmeans3D[st.radii == 0].abs().sum()) == 0.0
    # ---- the same pipeline on a 50k subsample, against the oracle end to end
    idx = np.arange(0, P, P // 50_000)[:50_000]
    sub = sc.means3D[idx]
    d2s = gsr.dist2(sub).cpu().numpy()
    np.testing.assert_allclose(d2s, oracle.dist2(sub), rtol=1e-6, atol=0)
    sc_s = np.sqrt(np.maximum(d2s, 1e-7)).astype(np.float32)[:, None].repeat(3, 1)
    o = oracle.Oracle(True)
    f = o.forward(means3D=sub, opacities=sc.opacities[idx], cam=cam, colors=sc.colors[idx], scales=sc_s, rotations=sc.rotations[idx])
    mc, _ = o.margins(f)
    ok = mc >= 1e-5
    st2 = gsr.forward(s, sub, sc.opacities[idx], colors=sc.colors[idx], scales=sc_s, rotations=sc.rotations[idx])
    np.testing.assert_array_equal(st2.radii.cpu().numpy(), f.radii)
    assert st2.num_rendered == f.num_rendered
    assert np.abs(st2.color.cpu().numpy() - f.color)[:, ok].max() <= 1e-4
    gin = sc.dL_dpix * ok[None]
    b = o.backward(gin)
    g2 = gsr.backward(st2, gin)
    for n in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dcolors"):
        assert rel_err(getattr(g2, n).cpu().numpy(), getattr(b, n)) <= 1e-4, n


@pytest.mark.gpu
@pytest.mark.parametrize("kind,P", [("uniform", 20000), ("surface", 30000), ("dupes", 5000), ("uniform", 700), ("surface", 5), ("uniform", 300000)])
def test_oracle_and_hip_dist2_against_the_reference_knn(gsr, kind, P):
    """SimpleKNN::knn itself (src/simple_knn.cu, translated by hipify-perl at build time and compiled by hipcc: oracle/build_ref.sh) on this GPU: the oracle's
    brute force and the library's grid search select the same three neighbours; the squared distances agree to the association freedom of three products."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref is not built")
    pts = _cloud(kind, P, 2)
    want = ref.dist2(pts)
    with np.errstate(over="ignore"):
        if P <= 30000:
            np.testing.assert_allclose(oracle.dist2(pts), want, rtol=1e-6, atol=0)
        np.testing.assert_allclose(gsr.dist2(pts).cpu().numpy(), want, rtol=1e-6, atol=0)
