"""GPU (-m gpu): the fused colour + depth / silhouette render (include/gsr.h: gsr_forward_args.out_ds,
gsr_backward_args.dL_dds) against what it replaces — TWO renders of the same geometry by the CPU oracle, one with the
colours and one with colors_precomp = [z, 1, 0] (src/Render.cc:927-981), and the sum of their backward passes, the second
one's dL/dcolour[:, 0] taken back to the mean through z = (view matrix row 2) . mean, as autograd does in the reference."""
import numpy as np
import pytest
import torch

from oracle import oracle
from util import mixed_err, pose, rel_err

pytestmark = pytest.mark.gpu
TOL, EPS_MARGIN = 1e-4, 1e-5


def _two_oracle_renders(sc, gA, gB):
    o, fA = oracle.forward_scene(sc, omp=True)
    mc, _ = o.margins(fA)
    ok = mc >= EPS_MARGIN
    bA = o.backward(gA * ok[None])
    z = fA.stages["depths"].astype(np.float32)                 # view-space depth of every splat, as preprocess computes it
    colB = np.stack([z, np.ones_like(z), np.zeros_like(z)], 1).astype(np.float32)
    o2 = oracle.Oracle(omp=True)
    bg0 = sc.cam.bg
    sc.cam.bg = (0.0, 0.0, 0.0)                                # the depth / silhouette channels have background 0
    try:
        fB = o2.forward(means3D=sc.means3D, opacities=sc.opacities, cam=sc.cam, colors=colB, scales=sc.scales, rotations=sc.rotations)
        bB = o2.backward(gB * ok[None])
    finally:
        sc.cam.bg = bg0
    vm = np.asarray(sc.cam.viewmatrix, np.float64).reshape(-1)   # column-major 4x4 as the kernels read it
    row2 = np.array([vm[2], vm[6], vm[10]])
    tot = {n: np.asarray(getattr(bA, n), np.float64) + np.asarray(getattr(bB, n), np.float64)
           for n in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D")}
    tot["dL_dmeans3D"] = tot["dL_dmeans3D"] + np.asarray(bB.dL_dcolors, np.float64)[:, 0:1] * row2[None]
    tot["dL_dcolors"] = np.asarray(bA.dL_dcolors, np.float64)
    return fA, fB, tot, ok


@pytest.mark.parametrize("name", ["tum-10k", "odd-pose-bg-fat", "replica-300k"])
def test_fused_pair_equals_two_renders(gsr, syn, name):
    cases = {
        "tum-10k": dict(P=10000, cam=syn.TUM1, mult=2.0),
        "odd-pose-bg-fat": dict(P=3000, cam=dict(width=203, height=149, fx=150.0, fy=152.0), mult=4.0, Tcw=pose(), bg=(0.3, 0.5, 0.7),
                                frac_behind=0.1, frac_offscreen=0.3),
        "replica-300k": dict(P=300000, cam=syn.REPLICA),
    }
    kw = dict(cases[name])
    cam = syn.make_camera(**kw.pop("cam"), Tcw=kw.pop("Tcw", None), bg=kw.pop("bg", (0, 0, 0)))
    sc = syn.make_scene(kw.pop("P"), cam, seed=5, scale_mult=kw.pop("mult", 1.0), **kw)
    rng = np.random.default_rng(7)
    H, W = cam.height, cam.width
    gA = sc.dL_dpix
    gB = rng.standard_normal((3, H, W)).astype(np.float32)
    gB[0] *= 0.3                                               # depth residuals are small next to colour residuals in the SLAM losses
    gB[2] = 0.0
    fA, fB, tot, ok = _two_oracle_renders(sc, gA, gB)

    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, dual=True)
    plain = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    assert torch.equal(st.color, plain.color) and torch.equal(st.depth, plain.depth) and torch.equal(st.radii, plain.radii)   # the colour pass is untouched
    ds = st.ds.cpu().numpy()
    scale = max(1.0, float(np.abs(fB.color[0]).max()))
    assert np.abs(ds[0] - fB.color[0])[ok].max() <= TOL * scale
    assert np.abs(ds[1] - fB.color[1])[ok].max() <= TOL
    assert np.abs(st.color.cpu().numpy() - fA.color)[:, ok].max() <= TOL * max(1.0, float(np.abs(fA.color).max()))

    gr = gsr.backward(st, gA * ok[None], dL_dds=(gB * ok[None])[0:2])
    torch.cuda.synchronize()
    worst = {}
    for n, ref in tot.items():
        got = getattr(gr, n).cpu().numpy()
        e, m = rel_err(got, ref), mixed_err(got, ref, afloor=2e-6)
        worst[n] = (e, m)
        assert e <= TOL, (n, e)
        assert m <= 1.0, (n, m)
    print("\n%s: fused pair vs two oracle renders (tensor-scale rel_err, element-wise ratio):" % name, {k: "%.1e / %.2f" % v for k, v in worst.items()})
    # without dL_dds the backward is the plain one
    g0 = gsr.backward(plain, gA * ok[None])
    g1 = gsr.backward(st, gA * ok[None])
    assert rel_err(g1.dL_dmeans3D.cpu().numpy(), g0.dL_dmeans3D.cpu().numpy()) < 1e-5


def test_pose_only_backward_without_the_colour_sums_gives_the_same_mean_gradient(gsr, syn):
    """A tracking iteration hands gsr_backward a buffer for dL/dmean3D only, with the depth channel's colour detached: the blend stage then
    runs without its colour sums (K_blend_bwd<..., COLORS = false>) and the per-splat stage skips the scale / rotation chain. The mean
    gradient must be the one of the full backward (the sums it needs are accumulated by the same arithmetic; float atomics order aside)."""
    import torch
    cam = syn.make_camera(373, 251, 300.0, 305.0)
    sc = syn.make_scene(20000, cam, seed=9, scale_mult=2.0)
    s = gsr.capi.Settings.from_camera(cam)
    rng = np.random.default_rng(1)
    gA = rng.standard_normal((3, 251, 373)).astype(np.float32); gB = rng.standard_normal((2, 251, 373)).astype(np.float32)
    gB[1] = 0.0   # (the loops never differentiate the silhouette: it is a detached mask)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, dual=True)
    full = gsr.backward(st, gA, dL_dds=gB, detach_depth_color=True)
    ref = full.dL_dmeans3D.clone()
    only = gsr.capi.Grads(None, None, None, None, torch.empty_like(ref), None, None, None, None)
    st2 = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, dual=True)
    gsr.backward(st2, gA, grads=only, dL_dds=gB, detach_depth_color=True, once=True)
    scale = float(ref.abs().max())
    assert scale > 0 and float((only.dL_dmeans3D - ref).abs().max()) <= 2e-6 * scale
    # ... and with the depth plane alone (gsr_backward_args.dds_depth_only: the silhouette's recursion leaves the loop), with and without the colour sums
    for grads in (gsr.capi.Grads(None, None, None, None, torch.empty_like(ref), None, None, None, None), None):
        st3 = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, dual=True)
        out = gsr.backward(st3, gA, grads=grads, dL_dds=gB[0:1].copy(), detach_depth_color=True, once=True, dds_depth_only=True)
        assert float((out.dL_dmeans3D - ref).abs().max()) <= 2e-6 * scale
        if grads is None:
            for n in ("dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations"):
                a, b = getattr(out, n), getattr(full, n)
                assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), n


@pytest.mark.parametrize("name", ["tum-10k", "odd-pose-fat"])
def test_silhouette_from_the_plain_forwards_transmittance(gsr, syn, name):
    """Round 6: tracking on the surface depth renders the 3 colour channels only; its loss masks with 1 - final_T, which the plain forward keeps per pixel
    anyway (forward.cu:455 final_T; include/gsr.h: gsr_transmittance_view, gsr_track_loss_rows' sil_is_transmittance). Pinned: 1 - T is the silhouette
    channel of the fused pair to rounding (the channel is the sum of alpha * T the same pass accumulates, Render.cc:963-975 renders it as colour 1), and
    the tracking loss on T equals the tracking loss on the silhouette wherever no pixel sits within that rounding of the 0.99 threshold."""
    import ctypes as C
    kw = dict(P=10000, cam=syn.TUM1, mult=2.0) if name == "tum-10k" else dict(P=3000, cam=dict(width=203, height=149, fx=150.0, fy=152.0), mult=4.0, Tcw=pose())
    cam = syn.make_camera(**kw["cam"], Tcw=kw.get("Tcw"))
    sc = syn.make_scene(kw["P"], cam, seed=9, scale_mult=kw["mult"])
    H, W = cam.height, cam.width
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, dual=True)
    plain = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    T = gsr.capi.transmittance_view(plain)
    sil = st.ds[1]
    assert T.shape == (H, W) and float(T.min()) >= 0.0 and float(T.max()) <= 1.0
    assert ((1.0 - T) - sil).abs().max() <= 2e-6
    assert torch.equal(gsr.capi.transmittance_view(st), T)                       # the fused pair keeps the same plane
    # gsr_forward_args.out_sil (ABI 10): the plain forward stores 1 - T itself; same colours, same radii; refused together with the pair
    sil_out = torch.full((H, W), float("nan"), device="cuda")
    plain2 = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, out_sil=sil_out)
    assert torch.equal(plain2.color, plain.color) and torch.equal(plain2.radii, plain.radii)
    assert torch.equal(sil_out, 1.0 - gsr.capi.transmittance_view(plain2))
    with pytest.raises(gsr.GsrError):
        gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, dual=True, out_sil=sil_out)
    # ... and the silhouette-only backward (dds_depth_only = 2) gives the same gradients behind either forward
    gS = torch.rand((H, W), generator=torch.Generator().manual_seed(5)).cuda()
    gpix = torch.tensor(sc.dL_dpix, device="cuda")
    def nocol():                                                              # (the form exists without colour sums only: a tracking iteration)
        gr = gsr.capi.alloc_grads(sc.P, 0, "cuda", intermediates=False)
        gr.dL_dcolors = None; gr.dL_dsh = None
        return gr
    ga = gsr.backward(st, gpix, grads=nocol(), dL_dds=gS.reshape(1, H, W).contiguous(), detach_depth_color=True, dds_depth_only=2)
    gb = gsr.backward(plain2, gpix, grads=nocol(), dL_dds=gS.reshape(1, H, W).contiguous(), detach_depth_color=True, dds_depth_only=2)
    torch.cuda.synchronize()
    for n in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations"):
        assert rel_err(getattr(gb, n).cpu().numpy(), getattr(ga, n).cpu().numpy()) <= 2e-6, n
    # the tracking loss, masked by either plane
    g = torch.Generator().manual_seed(3)
    frgb = torch.rand((3, H, W), generator=g).cuda(); fd = (0.5 + 3 * torch.rand((H, W), generator=g)); fd[::5, ::3] = 0.0; fd = fd.cuda()
    L = gsr.lib(); p = lambda t: C.c_void_p(t.data_ptr()); chk = gsr.capi._check
    wt = (C.c_float * 3)(0.7, 1.0, 0.0)
    near = ((sil - 0.99).abs() <= 4e-6)
    res = []
    for plane, flag in ((sil.contiguous(), 0), (T, 1)):
        part = torch.empty((1024 * 5,), device="cuda"); sums = torch.empty((8,), device="cuda")
        di = torch.empty((3, H, W), device="cuda"); dd = torch.empty((H, W), device="cuda")
        chk(L.gsr_track_loss_rows(p(plain.color), p(st.ds[0].contiguous()), None, p(plane), p(frgb), p(fd), H, W, 0.99, wt, p(part), p(sums), p(di), p(dd), None, 0, H, flag, None))
        res.append((sums.clone(), di, dd))
    keep = ~near
    assert torch.equal(res[0][1][:, keep], res[1][1][:, keep]) and torch.equal(res[0][2][keep], res[1][2][keep])
    if not bool(near.any()):
        assert (res[0][0][:6] - res[1][0][:6]).abs().max() <= 1e-6 * float(res[0][0][:6].abs().max())
    assert int(keep.sum()) > 0.99 * H * W
