"""GPU (-m gpu): the drop-in Python operator (gsorb-slam_amd/diff_gaussian_rasterization, the
twin of the reference package) and its libtorch host layer, used the way the reference's
only Python caller uses it (scripts/replay.py:122-161,299-330): camera-frame means, two
renders per frame (RGB and depth colours), autograd backward."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import oracle
from util import mixed_err, pose, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gsorb-slam_amd"))


def _settings(dgr, cam):
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda")
    return dgr.GaussianRasterizationSettings(
        image_height=cam.height, image_width=cam.width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=t(cam.bg),
        scale_modifier=1.0, viewmatrix=t(cam.viewmatrix), projmatrix=t(cam.projmatrix), sh_degree=cam.sh_degree,
        campos=t(cam.campos), prefiltered=False)


def test_module_surface():
    import diff_gaussian_rasterization as dgr
    assert {"rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"} <= set(dir(dgr._C))
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered")
    r = dgr.GaussianRasterizer(None)
    z = torch.zeros(4, 3, device="cuda")
    with pytest.raises(Exception):      # exactly one of SHs / colours
        r(means3D=z, means2D=z, opacities=z[:, :1])
    with pytest.raises(Exception):      # exactly one of scale+rot / cov3D
        r(means3D=z, means2D=z, opacities=z[:, :1], colors_precomp=z)


@pytest.mark.parametrize("mode", ["rgb", "depth", "sh"])
def test_replay_style_render_and_autograd(syn, mode):
    import diff_gaussian_rasterization as dgr
    cam = syn.make_camera(320, 240, 260.0, 258.0, bg=(0.0, 0.0, 0.0) if mode != "sh" else (0.2, 0.3, 0.1),
                          Tcw=pose() if mode == "sh" else None)
    sc = syn.make_scene(6000, cam, seed=2, scale_mult=2.0, color_mode=mode)
    o, f = oracle.forward_scene(sc)
    mc, md = o.margins(f)
    ok = mc >= 1e-5
    g_in = sc.dL_dpix * ok[None]
    b = o.backward(g_in)

    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True)
    means3D, opac, scales, rots = t(sc.means3D), t(sc.opacities), t(sc.scales), t(sc.rotations)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    kw = dict(means3D=means3D, means2D=means2D, opacities=opac, scales=scales, rotations=rots)
    if mode == "sh":
        shs = t(sc.shs)
        kw["shs"] = shs
    else:
        cols = t(sc.colors)
        kw["colors_precomp"] = cols
    im, radius, depth = dgr.GaussianRasterizer(raster_settings=_settings(dgr, cam))(**kw)
    assert im.shape == (3, cam.height, cam.width) and depth.shape == (1, cam.height, cam.width)
    assert radius.dtype == torch.int32 and not radius.requires_grad and not depth.requires_grad
    np.testing.assert_array_equal(radius.cpu().numpy(), f.radii)
    assert np.abs(im.detach().cpu().numpy() - f.color)[:, ok].max() <= 1e-4 * max(1.0, np.abs(f.color).max())
    dok = md >= 1e-5
    assert np.array_equal(depth[0].cpu().numpy()[dok], f.depth[0][dok])

    (im * torch.tensor(g_in, device="cuda")).sum().backward()
    assert rel_err(means3D.grad.cpu().numpy(), b.dL_dmeans3D) <= 1e-4
    assert rel_err(means2D.grad.cpu().numpy(), b.dL_dmeans2D) <= 1e-4     # retained by GSORB (src/Render.cc:763)
    assert rel_err(opac.grad.cpu().numpy(), b.dL_dopacity) <= 1e-4
    assert rel_err(scales.grad.cpu().numpy(), b.dL_dscales) <= 1e-4
    assert rel_err(rots.grad.cpu().numpy(), b.dL_drotations) <= 1e-4
    if mode == "sh":
        assert rel_err(shs.grad.cpu().numpy(), b.dL_dsh) <= 1e-4
    else:
        assert rel_err(cols.grad.cpu().numpy(), b.dL_dcolors) <= 1e-4

    vis = dgr.GaussianRasterizer(raster_settings=_settings(dgr, cam)).markVisible(means3D.detach())
    np.testing.assert_array_equal(vis.cpu().numpy(), oracle.mark_visible(sc.means3D, cam))


def test_cov3d_precomp_and_no_grad_paths(syn):
    import diff_gaussian_rasterization as dgr
    cam = syn.make_camera(160, 120, 120.0, 118.0)
    sc = syn.make_scene(1500, cam, seed=4, scale_mult=2.0)
    _, f0 = oracle.forward_scene(sc)
    cov = torch.tensor(f0.stages["cov3D"].reshape(-1, 6), device="cuda", requires_grad=True)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    r = dgr.GaussianRasterizer(raster_settings=_settings(dgr, cam))
    im, radius, depth = r(means3D=t(sc.means3D), means2D=torch.zeros(sc.P, 3, device="cuda"), opacities=t(sc.opacities),
                          colors_precomp=t(sc.colors), cov3D_precomp=cov)
    np.testing.assert_array_equal(radius.cpu().numpy(), f0.radii)
    im.sum().backward()
    assert cov.grad is not None and torch.isfinite(cov.grad).all()
    with torch.no_grad():
        im2, _, _ = r(means3D=t(sc.means3D), means2D=torch.zeros(sc.P, 3, device="cuda"), opacities=t(sc.opacities),
                      colors_precomp=t(sc.colors), scales=t(sc.scales), rotations=t(sc.rotations))
    assert torch.allclose(im2, im.detach(), atol=1e-5)


def test_backward_twice_with_retain_graph(syn):
    """The autograd node's first backward skips the re-zero of the per-splat accumulators; a second backward
    on the retained graph must start with a clear and reproduce the first gradients. The stateless
    _C.rasterize_gaussians_backward may be called any number of times."""
    import diff_gaussian_rasterization as dgr
    cam = syn.make_camera(192, 128, 150.0, 150.0)
    sc = syn.make_scene(6000, cam, seed=12, scale_mult=2.0)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    r = dgr.GaussianRasterizer(_settings(dgr, cam))
    means, op, col = t(sc.means3D).requires_grad_(True), t(sc.opacities).requires_grad_(True), t(sc.colors).requires_grad_(True)
    scl, rot = t(sc.scales).requires_grad_(True), t(sc.rotations).requires_grad_(True)
    im, _, _ = r(means3D=means, means2D=torch.zeros_like(means, requires_grad=True), opacities=op, colors_precomp=col,
                 scales=scl, rotations=rot)
    loss = (im * t(sc.dL_dpix)).sum()
    g1 = torch.autograd.grad(loss, [means, op, col, scl, rot], retain_graph=True)
    g2 = torch.autograd.grad(loss, [means, op, col, scl, rot], retain_graph=True)
    g3 = torch.autograd.grad(loss, [means, op, col, scl, rot])
    for a, b, c in zip(g1, g2, g3):
        scale = float(a.abs().max()) + 1e-30
        assert float((a - b).abs().max()) / scale < 1e-5 and float((a - c).abs().max()) / scale < 1e-5


def test_reference_named_backward_returns_what_the_reference_returns(syn):
    """_C.rasterize_gaussians_backward / ORB_SLAM2::RasterizeGaussiansBackwardCUDA (the reference-named free functions,
    DGR/rasterize_points.cu:117-199, src/Rasterizer.cu:217-305) return the reference's eight tensors with the reference's
    shapes — dL_dcov3D is [P,6] and FILLED on the scales + rotations path too (there it is computeCov2DCUDA's output,
    backward.cu:144-274) — and may be called any number of times per forward. The lean form ([0,6]: nothing consumes it)
    is what the autograd nodes opt into through rasterize_gaussians_backward_staged."""
    import diff_gaussian_rasterization as dgr
    cam = syn.make_camera(320, 240, 260.0, 258.0, bg=(0.1, 0.0, 0.2), Tcw=pose())
    sc = syn.make_scene(6000, cam, seed=21, scale_mult=2.0)
    o, f = oracle.forward_scene(sc)
    mc, _ = o.margins(f)
    g_in = sc.dL_dpix * (mc >= 1e-5)[None]
    b = o.backward(g_in)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    e = torch.empty(0, device="cuda")
    bg, m3, col, op, scl, rot = t(cam.bg), t(sc.means3D), t(sc.colors), t(sc.opacities), t(sc.scales), t(sc.rotations)
    vm, pm, cp = t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos)
    R, color, radii, geom, binning, img, depth = dgr._C.rasterize_gaussians(
        bg, m3, col, op, scl, rot, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, cam.height, cam.width, e, 0, cp, False)
    assert R == f.num_rendered
    args = (bg, m3, radii, col, scl, rot, 1.0, e, vm, pm, cam.tanfovx, cam.tanfovy, t(g_in), e, 0, cp, geom, R, binning, img)
    for _ in range(2):      # stateless: a second call on the same forward gives the same tensors
        d2, dcol, dop, d3, dcov, dsh, dscl, drot = dgr._C.rasterize_gaussians_backward(*args)
        assert tuple(dcov.shape) == (sc.P, 6) and tuple(d2.shape) == (sc.P, 3) and tuple(dop.shape) == (sc.P, 1)
        assert tuple(dsh.shape) == (sc.P, 0, 3) and tuple(dscl.shape) == (sc.P, 3) and tuple(drot.shape) == (sc.P, 4)
        assert rel_err(dcov.cpu().numpy(), b.dL_dcov3D) <= 1e-4
        assert mixed_err(dcov.cpu().numpy(), b.dL_dcov3D) <= 1.0
        for a, r in ((d2, b.dL_dmeans2D), (dcol, b.dL_dcolors), (dop, b.dL_dopacity), (d3, b.dL_dmeans3D),
                     (dscl, b.dL_dscales), (drot, b.dL_drotations)):
            assert rel_err(a.cpu().numpy().reshape(r.shape), r) <= 1e-4
    # the opt-in lean form: same gradients, no dL_dcov3D on this parameterisation
    lean = dgr._C.rasterize_gaussians_backward_staged(*args, 0)
    assert tuple(lean[4].shape) == (0, 6)
    assert rel_err(lean[6].cpu().numpy(), b.dL_dscales) <= 1e-4
    # cov3D_precomp path: [P,6] in both forms
    cov = t(f.stages["cov3D"].reshape(-1, 6))
    R2, _, radii2, geom2, binning2, img2, _ = dgr._C.rasterize_gaussians(
        bg, m3, col, op, e, e, 1.0, cov, vm, pm, cam.tanfovx, cam.tanfovy, cam.height, cam.width, e, 0, cp, False)
    out = dgr._C.rasterize_gaussians_backward(bg, m3, radii2, col, e, e, 1.0, cov, vm, pm, cam.tanfovx, cam.tanfovy, t(g_in), e, 0, cp,
                                              geom2, R2, binning2, img2)
    assert R2 == R and tuple(out[4].shape) == (sc.P, 6)
    assert rel_err(out[4].cpu().numpy(), b.dL_dcov3D) <= 1e-4


@pytest.mark.parametrize("detach", [False, True])
def test_forward_pair_equals_the_two_renders_of_the_op(syn, detach):
    """GaussianRasterizer.forward_pair (one pass: colours + [view depth, 1]) against two calls of the op itself the way
    GSORB-SLAM makes them (src/Render.cc:927-981: camera-frame means, identity view matrix, second render with
    colors_precomp = [z, 1, 0] built from the means — detached in tracking iterations), through autograd."""
    import diff_gaussian_rasterization as dgr
    cam = syn.make_camera(320, 240, 260.0, 258.0)
    sc = syn.make_scene(8000, cam, seed=3, scale_mult=2.5)
    rast = dgr.GaussianRasterizer(_settings(dgr, cam))
    g = torch.Generator().manual_seed(1)
    gA = torch.randn((3, 240, 320), generator=g).cuda()
    gB = (torch.randn((2, 240, 320), generator=g) * torch.tensor([0.3, 1.0]).reshape(2, 1, 1)).cuda()

    def leaves():
        t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda", requires_grad=True)
        return t(sc.means3D), t(sc.opacities), t(sc.colors), t(sc.scales), t(sc.rotations)

    # two passes
    m, o, c, s, r = leaves()
    z = m[:, 2:3]
    col2 = torch.cat([z, torch.ones_like(z), torch.zeros_like(z)], 1)
    if detach:
        col2 = col2.detach()
    imA, _, surA = rast(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, colors_precomp=c, scales=s, rotations=r)
    imB, _, _ = rast(means3D=m, means2D=torch.zeros_like(m, requires_grad=True), opacities=o, colors_precomp=col2, scales=s, rotations=r)
    ((imA * gA).sum() + (imB[0:2] * gB).sum()).backward()
    ref = [x.grad.clone() for x in (m, o, c, s, r)]
    # one pass
    m2, o2, c2, s2, r2 = leaves()
    im, ds, radii, sur = rast.forward_pair(means3D=m2, means2D=torch.zeros_like(m2, requires_grad=True), opacities=o2, colors_precomp=c2,
                                           scales=s2, rotations=r2, detach_depth_color=detach)
    assert torch.equal(im, imA) and torch.equal(sur, surA)
    assert float((ds - imB[0:2]).abs().max()) <= 1e-5 * max(1.0, float(imB[0].abs().max()))
    ((im * gA).sum() + (ds * gB).sum()).backward()
    for name, a, b in zip(("means3D", "opacities", "colors", "scales", "rotations"), (m2, o2, c2, s2, r2), ref):
        err = float((a.grad - b).abs().max() / (b.abs().max() + 1e-30))
        assert err <= 2e-5, (name, err)          # the same terms summed in another order (one accumulation instead of two)
    # an output the loss does not use: its gradient is zero, not an error
    m3, o3, c3, s3, r3 = leaves()
    im3, ds3, _, _ = rast.forward_pair(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), opacities=o3, colors_precomp=c3, scales=s3, rotations=r3)
    (ds3 * gB).sum().backward()
    assert float(c3.grad.abs().max()) == 0.0 and float(m3.grad.abs().max()) > 0.0
