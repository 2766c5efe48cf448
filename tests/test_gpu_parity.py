"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.

Bars (BASELINE.json north_star): tile/sort indices BIT-EXACT (radii, tiles_touched,
num_rendered, sorted point_list, keys, ranges); rendered RGB/depth and all gradients within
1e-4 relative (fp32). Two metrics are asserted for every gradient tensor (tests/util.py):
  rel_err   max|a-b| / max|b|                      <= 1e-4   (tensor scale)
  mixed_err |a-b| <= 1e-4*|b| + 1e-6*max|b|  for EVERY element (element-wise, with the absolute floor
            a sum of thousands of fp32 terms can resolve); images use |a-b| <= 1e-4 * max(1, max|ref|).

exp() is not bit-reproducible between glibc and the GPU, so a pixel that sits within ~1e-7
of one of the blend's branch thresholds (alpha<1/255, power>0, T(1-alpha)<1e-4, T>0.5) may
legitimately take the other branch — 1 pixel in 816 000 in the 300k-splat scene. The
oracle reports each pixel's smallest branch margin (gsro_render_margins); pixels with margin
< 1e-5 are excluded from the image comparison and their upstream gradient is zeroed for
BOTH implementations, instead of loosening any tolerance. The fraction of such pixels is
asserted to stay tiny.
"""
import numpy as np
import pytest
import torch

from oracle import oracle
from util import mixed_err, pose, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4
EPS_MARGIN = 1e-5

SMALL = dict(width=160, height=120, fx=120.0, fy=118.0)
ODD = dict(width=203, height=149, fx=150.0, fy=152.0)   # not a multiple of 16: ragged edge tiles


def _cases(syn):
    return {
        "small-rgb": dict(P=2000, cam=SMALL),
        "tum-10k-rgb": dict(P=10000, cam=syn.TUM1),                                      # BASELINE configs[0] shape
        "tum-10k-depth-fat": dict(P=10000, cam=syn.TUM1, mode="depth", mult=4.0),        # [z,1,0] colours, R/P ~ 8
        "odd-sh3-pose-bg": dict(P=3000, cam=ODD, mode="sh", mult=3.0, Tcw=pose(), bg=(0.3, 0.5, 0.7),
                                frac_behind=0.1, frac_offscreen=0.3),
        "odd-sh1": dict(P=1500, cam=ODD, mode="sh", sh_degree=1, mult=2.0, bg=(0.2, 0.2, 0.2)),
        "fat-clamped": dict(P=1500, cam=SMALL, mult=14.0, Tcw=pose(0.2), frac_offscreen=0.4),   # fov clamp, long lists
        "dense-long-lists": dict(P=200000, cam=SMALL, mult=1.2),                           # > 256 and > 4096 entries per tile
        "replica-300k": dict(P=300000, cam=syn.REPLICA),                                 # BASELINE configs[1]
        # ~30 overlapping layers of near-identical colours per pixel, lists of ~10k entries: the case in which a
        # reformulated accum_rec recursion once lost accuracy (scripts/fuzz_parity.py found it)
        "deep-stack-depth": dict(P=40000, cam=dict(width=48, height=448, fx=40.0, fy=42.0), mode="depth", mult=8.0),
        # the sizes bench.py quotes (BASELINE.json metric / configs 4 and 5), full oracle comparison (OpenMP build of the oracle)
        "replica-1M-rgb": dict(P=1_000_000, cam=syn.REPLICA),
        "replica-1M-depth": dict(P=1_000_000, cam=syn.REPLICA, mode="depth"),             # colours [z,1,0]: half of every SLAM iteration
        "scannet-2M-rgb": dict(P=2_000_000, cam=syn.CAMERAS["scannet"]),
    }


def _build(syn, P, cam, mode="rgb", mult=1.0, Tcw=None, bg=(0, 0, 0), seed=0, **kw):
    c = syn.make_camera(**cam, Tcw=Tcw, bg=bg)
    return syn.make_scene(P, c, seed=seed, scale_mult=mult, color_mode=mode, **kw)


CASE_NAMES = ["small-rgb", "tum-10k-rgb", "tum-10k-depth-fat", "odd-sh3-pose-bg", "odd-sh1", "fat-clamped",
              "dense-long-lists", "replica-300k", "deep-stack-depth", "replica-1M-rgb", "replica-1M-depth", "scannet-2M-rgb"]
BIG = ("replica-1M-rgb", "replica-1M-depth", "scannet-2M-rgb")


@pytest.mark.parametrize("name", CASE_NAMES)
def test_forward_backward_parity(gsr, syn, name):
    sc = _build(syn, **_cases(syn)[name])
    o, f = oracle.forward_scene(sc, omp=name in BIG)
    mc, md = o.margins(f)
    ok_c, ok_d = mc >= EPS_MARGIN, md >= EPS_MARGIN
    # knife-edge pixels (margin < 1e-5: candidates, not flips) stay rare; the budget is what the oracle itself
    # reports for these scenes x ~1.3 (observed: colour <= 1.6e-3, depth <= 2.4e-3; 2.3e-3 / 3.8e-3 on the
    # 200k-splat 160x120 scene whose pixels see ~600 list entries each)
    # (the bench-size scenes, observed: 1 M 1.9e-3 / 3.0e-3, 2 M ScanNet 2.1e-3 / 3.3e-3)
    budget_c, budget_d = (3e-3, 5e-3) if name in ("dense-long-lists",) + BIG else (2e-3, 5e-3)
    print("\n%s: knife-edge pixels colour %.2e depth %.2e" % (name, (~ok_c).mean(), (~ok_d).mean()))
    assert (~ok_c).mean() <= budget_c and (~ok_d).mean() <= budget_d
    g_in = sc.dL_dpix * ok_c[None]
    b = o.backward(g_in)

    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, shs=sc.shs, scales=sc.scales,
                     rotations=sc.rotations)
    d = gsr.debug_export(st)

    # ---- integer stages: bit exact ----
    assert st.num_rendered == f.num_rendered
    np.testing.assert_array_equal(st.radii.cpu().numpy(), f.radii)
    np.testing.assert_array_equal(d["tiles_touched"], f.stages["tiles_touched"])
    np.testing.assert_array_equal(d["point_list"], f.stages["point_list"])
    np.testing.assert_array_equal(d["point_list_keys"], f.stages["keys_sorted"])
    np.testing.assert_array_equal(d["ranges"], f.stages["ranges"])
    # per-splat projected geometry: same arithmetic, same bits
    np.testing.assert_array_equal(d["means2D"], f.stages["means2D"])
    np.testing.assert_array_equal(d["depths"], f.stages["depths"])
    np.testing.assert_array_equal(d["conic_opacity"], f.stages["conic_opacity"])
    if sc.shs is not None:
        vis = f.radii > 0
        np.testing.assert_allclose(d["rgb"][vis], f.stages["rgb"].reshape(-1, 3)[vis], atol=2e-6)

    # ---- images ----
    H, W = sc.cam.height, sc.cam.width
    col = st.color.cpu().numpy()
    scale = max(1.0, float(np.abs(f.color).max()))
    assert np.abs(col - f.color)[:, ok_c].max() <= TOL * scale
    dep = st.depth.cpu().numpy()[0]
    assert np.array_equal(dep[ok_d], f.depth[0][ok_d])           # median depth is a copied input value
    fT = d["final_T"].reshape(H, W)
    assert np.abs(fT - f.stages["final_T"].reshape(H, W))[ok_c].max() <= TOL
    assert np.array_equal(d["n_contrib"].reshape(H, W)[ok_c], f.stages["n_contrib"].reshape(H, W)[ok_c])

    # ---- gradients ----
    gr = gsr.backward(st, g_in)
    torch.cuda.synchronize()
    worst = {}
    # absolute floor of the element-wise bar: 1e-6 of the tensor's largest element (~10 ulps of it); observed worst ratio
    # on eight scenes 0.28. Only deep-stack-depth (R/P = 26: every splat sums ~1e5 signed fp32 terms of near-identical
    # colours, atomics in any order, against the oracle's double accumulators) needs 4e-5 (observed 5e-6 .. 2e-5 from run to run).
    afloor = 4e-5 if name == "deep-stack-depth" else 1e-6
    for n in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh",
              "dL_dscales", "dL_drotations"):
        got, ref = getattr(gr, n).cpu().numpy(), getattr(b, n)
        e, m = rel_err(got, ref), mixed_err(got, ref, afloor=afloor)
        worst[n] = (e, m)
        assert e <= TOL, (n, e)
        assert m <= 1.0, (n, "element-wise bar |a-b| <= 1e-4|b| + 1e-6 max|b| exceeded by x%.2f" % m)
    print("%s: gradients (tensor-scale rel_err, element-wise ratio):" % name, {k: "%.1e / %.2f" % v for k, v in worst.items()})


def test_cov3d_precomp_path(gsr, syn):
    sc = _build(syn, 3000, ODD, mult=2.0, bg=(0.1, 0.1, 0.4))
    o0, f0 = oracle.forward_scene(sc)
    cov = f0.stages["cov3D"].reshape(-1, 6)
    o = oracle.Oracle()
    f = o.forward(means3D=sc.means3D, opacities=sc.opacities, cam=sc.cam, colors=sc.colors, cov3D_precomp=cov)
    mc, _ = o.margins(f)
    g_in = sc.dL_dpix * (mc >= EPS_MARGIN)[None]
    b = o.backward(g_in)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, cov3D_precomp=cov)
    np.testing.assert_array_equal(st.radii.cpu().numpy(), f.radii)
    assert np.abs(st.color.cpu().numpy() - f.color)[:, mc >= EPS_MARGIN].max() <= TOL
    gr = gsr.backward(st, g_in)
    assert rel_err(gr.dL_dcov3D.cpu().numpy(), b.dL_dcov3D) <= TOL
    assert rel_err(gr.dL_dmeans3D.cpu().numpy(), b.dL_dmeans3D) <= TOL
    assert float(gr.dL_dscales.abs().max()) == 0.0 and float(gr.dL_drotations.abs().max()) == 0.0


def test_edge_cases(gsr, syn):
    cam = syn.make_camera(64, 48, 50.0, 50.0, bg=(0.1, 0.2, 0.3))
    s = gsr.capi.Settings.from_camera(cam)
    z = lambda *sh: torch.zeros(sh, device="cuda")
    # P == 0: zero images, no background (src/Rasterizer.cu:183)
    st = gsr.forward(s, z(0, 3), z(0, 1), colors=z(0, 3), scales=z(0, 3), rotations=z(0, 4))
    assert st.num_rendered == 0 and float(st.color.abs().max()) == 0.0
    gsr.backward(st, z(3, 48, 64))
    # everything culled: background only, R == 0, zero gradients
    P = 10
    m = torch.tensor([[0, 0, -1.0]], device="cuda").repeat(P, 1)
    q = torch.tensor([[1.0, 0, 0, 0]], device="cuda").repeat(P, 1)
    st = gsr.forward(s, m, torch.full((P, 1), .5, device="cuda"), colors=z(P, 3), scales=torch.full((P, 3), .1, device="cuda"), rotations=q)
    assert st.num_rendered == 0 and int(st.radii.abs().sum()) == 0
    exp = torch.tensor(cam.bg, device="cuda")[:, None, None].expand(3, 48, 64)
    assert torch.equal(st.color, exp)
    gr = gsr.backward(st, torch.ones(3, 48, 64, device="cuda"))
    assert float(gr.dL_dmeans3D.abs().max()) == 0.0 and float(gr.dL_dopacity.abs().max()) == 0.0
    assert not gsr.mark_visible(m, s.viewmatrix, s.projmatrix).any()
    # argument errors surface as GsrError / ValueError, not crashes
    with pytest.raises(gsr.GsrError):
        gsr.forward(s, m, torch.full((P, 1), .5, device="cuda"), colors=z(P, 3), shs=z(P, 16, 3), scales=z(P, 3), rotations=q)
    with pytest.raises(gsr.GsrError):
        gsr.forward(s, m, torch.full((P, 1), .5, device="cuda"), colors=z(P, 3))
    with pytest.raises(ValueError):
        gsr.forward(s, z(P, 4), torch.full((P, 1), .5, device="cuda"), colors=z(P, 3), scales=z(P, 3), rotations=q)


def test_prefiltered_flag_skips_culled_splats_instead_of_trapping(gsr, syn):
    """Reference auxiliary.h:156-160: with prefiltered=true a splat that fails the frustum test makes the kernel
    printf + __trap(), i.e. the caller promises it culled already and the process dies if it did not. The C ABI
    accepts the flag for signature parity and SKIPS such a splat (include/gsr.h): same frame, radii
    and gradients as prefiltered=false, the context stays alive."""
    sc = _build(syn, 4000, SMALL, mult=2.0, frac_behind=0.3, frac_offscreen=0.3)
    s0 = gsr.capi.Settings.from_camera(sc.cam)
    s1 = gsr.capi.Settings.from_camera(sc.cam)
    s1.prefiltered = True
    kw = dict(colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    a = gsr.forward(s0, sc.means3D, sc.opacities, **kw)
    b = gsr.forward(s1, sc.means3D, sc.opacities, **kw)
    torch.cuda.synchronize()
    assert int((a.radii == 0).sum()) > 500                      # there ARE culled splats
    assert torch.equal(a.radii, b.radii) and torch.equal(a.color, b.color) and torch.equal(a.depth, b.depth)
    assert a.num_rendered == b.num_rendered
    ga, gb = gsr.backward(a, sc.dL_dpix), gsr.backward(b, sc.dL_dpix)
    assert rel_err(gb.dL_dmeans3D.cpu().numpy(), ga.dL_dmeans3D.cpu().numpy()) < 1e-5
    r0 = gsr.visible_filter(s0, sc.means3D, sc.scales, sc.rotations)
    r1 = gsr.visible_filter(s1, sc.means3D, sc.scales, sc.rotations)
    assert torch.equal(r0, r1)
    _, f = oracle.forward_scene(sc)
    np.testing.assert_array_equal(b.radii.cpu().numpy(), f.radii)


def test_mark_visible_and_visible_filter(gsr, syn):
    sc = _build(syn, 20000, syn.TUM1, mult=2.0, Tcw=pose(), frac_behind=0.3, frac_offscreen=0.3)
    s = gsr.capi.Settings.from_camera(sc.cam)
    vis = gsr.mark_visible(sc.means3D, s.viewmatrix, s.projmatrix).cpu().numpy()
    np.testing.assert_array_equal(vis, oracle.mark_visible(sc.means3D, sc.cam))
    # GSORB renders the filter pass on a 1.2x enlarged image (src/Render.cc:794-795)
    W2, H2 = int(sc.cam.width * 1.2), int(sc.cam.height * 1.2)
    r = gsr.visible_filter(s, sc.means3D, sc.scales, sc.rotations, W2, H2).cpu().numpy()
    np.testing.assert_array_equal(r, oracle.filter_radii(sc.means3D, sc.scales, sc.rotations, sc.cam, W2, H2))


def test_workspace_path_matches_callback_path_and_flags_overflow(gsr, syn):
    sc = _build(syn, 8000, SMALL, mult=2.0)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    gr = gsr.backward(st, sc.dL_dpix)
    ws = gsr.capi.Workspace(sc.P, sc.cam.width, sc.cam.height, max_rendered=st.num_rendered + 100)
    st2 = gsr.forward_ws(s, ws, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    n, ovf = ws.status()
    assert n == st.num_rendered and not ovf
    assert torch.equal(st2.color, st.color) and torch.equal(st2.depth, st.depth) and torch.equal(st2.radii, st.radii)
    gr2 = gsr.backward(st2, sc.dL_dpix)
    assert rel_err(gr2.dL_dmeans3D.cpu().numpy(), gr.dL_dmeans3D.cpu().numpy()) < 1e-5   # atomic order only
    small = gsr.capi.Workspace(sc.P, sc.cam.width, sc.cam.height, max_rendered=st.num_rendered // 2)
    gsr.forward_ws(s, small, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    n, ovf = small.status()
    assert n == st.num_rendered and ovf


@pytest.mark.parametrize("P,camname", [(1_000_000, "replica"), (2_000_000, "scannet")])
def test_properties_at_baseline_size(gsr, syn, P, camname):
    """BASELINE.json metric size (1M splats, 1200x680) and the config-5 size (2M+ splats on the ScanNet camera,
    640x480: ~6.5 splats per pixel-area unit more than the headline): size-independent properties only."""
    camd = syn.CAMERAS[camname]
    Hh, Ww = camd["height"], camd["width"]
    sc = _build(syn, P, camd)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    assert st.num_rendered == int(d["tiles_touched"].sum())
    keys = d["point_list_keys"]
    assert np.all(np.diff(keys.astype(np.uint64)) >= 0) or np.all(keys[1:] >= keys[:-1])   # sorted by (tile, depth)
    same = keys[1:] == keys[:-1]
    assert np.all(d["point_list"][1:][same] > d["point_list"][:-1][same])                   # stable
    r = d["ranges"].astype(np.int64)
    assert int((r[:, 1] - r[:, 0]).sum()) == st.num_rendered
    assert np.array_equal(np.sort(np.unique(d["point_list"])), np.nonzero(st.radii.cpu().numpy() > 0)[0])
    T0 = d["final_T"].reshape(Hh, Ww)
    assert np.isfinite(st.color.cpu().numpy()).all() and T0.min() >= 0 and T0.max() <= 1
    # background linearity: C(bg) = C(0) + T_final * bg
    bg = np.array([0.25, 0.5, 0.75], np.float32)
    s2 = gsr.capi.Settings.from_camera(sc.cam)
    s2.bg = torch.tensor(bg, device="cuda")
    st2 = gsr.forward(s2, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    exp = st.color.cpu().numpy() + T0[None] * bg[:, None, None]
    assert np.abs(st2.color.cpu().numpy() - exp).max() < 1e-5
    # gradient linearity in the upstream gradient, and run-to-run agreement (atomics reorder only)
    g1 = gsr.backward(st, sc.dL_dpix)
    a = g1.dL_dmeans3D.clone()
    g2 = gsr.backward(st, 2.0 * torch.tensor(sc.dL_dpix, device="cuda"))
    assert rel_err(g2.dL_dmeans3D.cpu().numpy(), 2.0 * a.cpu().numpy()) < 1e-5
    inv = st.radii == 0
    assert float(g2.dL_dmeans3D[inv].abs().sum()) == 0.0


# ---------------------------------------------------------------------------------------
# tile-band sharding (multi-GPU scheme A) simulated on one GPU: the bands of a frame rendered one
# after the other must reproduce the one-call render bit for bit, and the band accumulators summed
# (what the all-reduce does) must give the one-call gradients.
@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_tile_bands_equal_full_frame(gsr, syn, world):
    import torch
    capi = gsr.capi
    sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
    cam = syn.make_camera(320, 200, 250.0, 250.0)   # 13 tile rows, the last one partial (200 = 12*16 + 8)
    sc = syn.make_scene(30000, cam, seed=21, scale_mult=2.0)
    dev = torch.device("cuda:0")
    s = capi.Settings.from_camera(cam, dev)
    t = lambda a: torch.tensor(a, device=dev)
    kw = dict(means3D=t(sc.means3D), opacities=t(sc.opacities), colors=t(sc.colors), scales=t(sc.scales),
              rotations=t(sc.rotations))
    full = capi.forward(s, **kw)
    dpix = t(sc.dL_dpix)
    gfull = capi.backward(full, dpix)
    img = torch.full((4, cam.height, cam.width), float("nan"), device=dev)
    bands = sharded.band_rows((cam.height + 15) // 16, world)
    assert bands[0][0] == 0 and bands[-1][1] == 13 and all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
    acc_sum, states, R = None, [], 0
    for b in bands:
        st = capi.forward(s, band=b, out=(img[0:3], img[3:4]), **kw)
        assert torch.equal(st.radii, full.radii)            # radii cover every splat on every rank
        R += st.num_rendered
        grads = capi.alloc_grads(st.P, st.M, dev)
        capi.backward(st, dpix, grads=grads, stages=1 | 2)
        a = capi.acc_view(st)
        acc_sum = a.clone() if acc_sum is None else acc_sum + a
        states.append((st, grads))
    assert R == full.num_rendered                             # the bands partition the (splat, tile) pairs
    assert torch.equal(img[0:3], full.color) and torch.equal(img[3:4], full.depth)   # bit-exact
    st, grads = states[-1]
    capi.acc_view(st).copy_(acc_sum)
    g = capi.backward(st, dpix, grads=grads, stages=4)
    for name in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
        a, b = getattr(g, name), getattr(gfull, name)
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        assert err < 1e-5, (name, err)   # same terms, different float summation order (atomics) only


@pytest.mark.gpu
def test_backward_twice_on_one_forward_state(gsr, syn):
    """The per-splat accumulators are zeroed by the forward and re-zeroed by the per-splat stage, not by a
    memset in gsr_backward: a second backward on the same state must reproduce the first one, and an
    explicit GSR_STAGE_CLEAR must discard a blend stage that was not consumed."""
    import torch
    capi = gsr.capi
    cam = syn.make_camera(256, 160, 200.0, 200.0)
    sc = syn.make_scene(20000, cam, seed=8, scale_mult=2.0, frac_behind=0.1)
    dev = torch.device("cuda:0")
    s = capi.Settings.from_camera(cam, dev)
    t = lambda a: torch.tensor(a, device=dev)
    st = capi.forward(s, means3D=t(sc.means3D), opacities=t(sc.opacities), colors=t(sc.colors), scales=t(sc.scales),
                      rotations=t(sc.rotations))
    dpix = t(sc.dL_dpix)
    g1 = capi.backward(st, dpix)
    ref = {n: getattr(g1, n).clone() for n in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales")}
    assert float(capi.acc_view(st).abs().max()) == 0.0
    g2 = capi.backward(st, dpix)
    for n, r in ref.items():
        err = float((getattr(g2, n) - r).abs().max() / (r.abs().max() + 1e-30))
        assert err < 1e-5, (n, err)
    capi.backward(st, dpix, stages=2)                      # blend only: accumulators now hold one frame's sums
    assert float(capi.acc_view(st).abs().max()) > 0.0
    g3 = capi.backward(st, dpix, stages=1 | 2 | 4)         # explicit clear discards them
    for n, r in ref.items():
        err = float((getattr(g3, n) - r).abs().max() / (r.abs().max() + 1e-30))
        assert err < 1e-5, (n, err)


@pytest.mark.gpu
def test_forward_capacity_guess_too_small_reruns_the_tail(gsr, syn):
    """gsr_forward sizes the binning blob from a per-thread guess and enqueues the tail before it knows
    num_rendered; a fresh thread starts with the floor 4P+4096, which a fat scene (R/P ~ 8) exceeds: the
    second request + second tail must give exactly what a call with a sufficient guess gives."""
    import threading
    import torch
    capi = gsr.capi
    cam = syn.make_camera(**syn.TUM1)
    sc = syn.make_scene(10000, cam, seed=3, scale_mult=4.0, color_mode="depth")
    dev = torch.device("cuda:0")
    s = capi.Settings.from_camera(cam, dev)
    t = lambda a: torch.tensor(a, device=dev)
    kw = dict(means3D=t(sc.means3D), opacities=t(sc.opacities), colors=t(sc.colors), scales=t(sc.scales), rotations=t(sc.rotations))
    out = {}

    def run():
        with torch.cuda.device(dev):
            a = capi.forward(s, **kw)          # fresh thread: floor guess, too small
            ga = capi.backward(a, t(sc.dL_dpix))
            b = capi.forward(s, **kw)          # guess from the previous frame: large enough
            out.update(a=a, b=b, ga=ga, da=capi.debug_export(a), db=capi.debug_export(b))
            torch.cuda.synchronize()

    th = threading.Thread(target=run)
    th.start()
    th.join()
    a, b = out["a"], out["b"]
    assert a.num_rendered == b.num_rendered and a.num_rendered > 4 * 10000 + 4096
    assert a.binning.numel() < b.binning.numel()          # exact size after the re-run, guess (R*1.25) afterwards
    assert torch.equal(a.color, b.color) and torch.equal(a.depth, b.depth)
    np.testing.assert_array_equal(out["da"]["point_list"], out["db"]["point_list"])
    o, f = oracle.forward_scene(sc)
    assert a.num_rendered == f.num_rendered
    np.testing.assert_array_equal(out["da"]["point_list"], f.stages["point_list"])
    gb = capi.backward(b, t(sc.dL_dpix))
    for n in ("dL_dmeans3D", "dL_dopacity", "dL_dscales"):
        x, y = getattr(out["ga"], n), getattr(gb, n)
        assert float((x - y).abs().max() / (y.abs().max() + 1e-30)) < 1e-5


@pytest.mark.gpu
def test_large_frame_more_than_8192_tiles(gsr, syn):
    """3840x2160 = 240x135 = 32400 tiles: the one-block scan works through more than eight tiles per thread
    and tile coordinates no longer fit a byte. Integer stages bit-exact, image within tolerance."""
    cam = syn.make_camera(3840, 2160, 2000.0, 2000.0)
    sc = syn.make_scene(60000, cam, seed=4, scale_mult=6.0, frac_offscreen=0.1)
    o, f = oracle.forward_scene(sc, omp=True)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    assert st.num_rendered == f.num_rendered
    np.testing.assert_array_equal(st.radii.cpu().numpy(), f.radii)
    np.testing.assert_array_equal(d["ranges"], f.stages["ranges"])
    np.testing.assert_array_equal(d["point_list"], f.stages["point_list"])
    mc, md = o.margins(f)
    ok = mc >= EPS_MARGIN
    assert (~ok).mean() <= 2e-3
    col = st.color.cpu().numpy()
    assert np.abs(col - f.color)[:, ok].max() <= TOL * max(1.0, float(np.abs(f.color).max()))
    g_in = sc.dL_dpix * ok[None]
    b = o.backward(g_in)
    gr = gsr.backward(st, g_in)
    worst, name = {}, "4k-frame"
    for n in ("dL_dmeans3D", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations"):
        got, ref = getattr(gr, n).cpu().numpy(), getattr(b, n)
        e, m = rel_err(got, ref), mixed_err(got, ref)
        worst[n] = (e, m)
        assert e <= TOL, (n, e)
        assert m <= 1.0, (n, "element-wise bar |a-b| <= 1e-4|b| + 1e-6 max|b| exceeded by x%.2f" % m)
    print("%s: gradients (tensor-scale rel_err, element-wise ratio):" % name, {k: "%.1e / %.2f" % v for k, v in worst.items()})


@pytest.mark.gpu
def test_frame_of_exactly_one_bin_window(gsr, syn):
    """2048x2048 = 128x128 = 16384 tiles = exactly GSR_BIN_WINDOW: the count pass's window plus its two words of slack must
    still fit 64 KB of LDS (it asked for 65544 bytes before the window was shortened by the slack)."""
    cam = syn.make_camera(2048, 2048, 1500.0, 1500.0)
    sc = syn.make_scene(40000, cam, seed=6, scale_mult=6.0, frac_offscreen=0.1)
    o, f = oracle.forward_scene(sc, omp=True)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    assert st.num_rendered == f.num_rendered
    np.testing.assert_array_equal(d["ranges"], f.stages["ranges"])
    np.testing.assert_array_equal(d["point_list"], f.stages["point_list"])
    mc, _ = o.margins(f)
    ok = mc >= EPS_MARGIN
    assert np.abs(st.color.cpu().numpy() - f.color)[:, ok].max() <= TOL * max(1.0, float(np.abs(f.color).max()))


@pytest.mark.gpu
def test_huge_frame_more_than_131072_tiles(gsr, syn):
    """8192x4112 = 512x257 = 131584 tiles: the count pass histograms the frame in nine LDS windows and the fill pass in
    sixteen (two per XCD), with splats from one tile to thousands of tiles wide. Integer stages bit-exact."""
    cam = syn.make_camera(8192, 4112, 4000.0, 4000.0)
    sc = syn.make_scene(30000, cam, seed=11, scale_mult=5.0, frac_offscreen=0.1)
    sc.scales[::97] *= 40.0  # a few splats that cover hundreds of tile rows
    o, f = oracle.forward_scene(sc, omp=True)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    assert st.num_rendered == f.num_rendered
    np.testing.assert_array_equal(st.radii.cpu().numpy(), f.radii)
    np.testing.assert_array_equal(d["ranges"], f.stages["ranges"])
    np.testing.assert_array_equal(d["point_list"], f.stages["point_list"])
    assert (f.stages["ranges"][:, 1] - f.stages["ranges"][:, 0]).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("frame", ["lists-to-4096", "lists-over-4096"])
@pytest.mark.parametrize("layout", ["uniform", "planes", "wall+outlier", "two-slabs", "all-equal"])
def test_sort_order_with_depth_ties_and_crowded_bins(gsr, syn, layout, frame):
    """The tile sort bins keys over the tile's own depth range and ranks inside a bin; lists whose depths crowd into few
    bins are binned again with equalised bins, and only then take the bitonic network. Exact depth ties (order by splat
    id, like the reference's stable radix sort), a thin wall plus one far outlier (everything in two bins), two 2-cm
    slabs (the equalised second binning succeeds) and a single depth for the whole map, on a frame whose lists use the
    one-wave and the 256-thread LDS sorts and on one whose lists exceed 4096 entries (bucket sort through global
    scratch): point_list bit-exact."""
    cam = syn.make_camera(320, 240, 240.0, 240.0) if frame == "lists-to-4096" else syn.make_camera(128, 96, 96.0, 96.0)
    sc = syn.make_scene(60000, cam, seed=21, scale_mult=1.5)
    rng = np.random.default_rng(5)
    z = sc.means3D[:, 2].copy()
    if layout == "uniform":
        pass
    elif layout == "planes":
        z = np.float32(1.0) + np.float32(0.5) * rng.integers(0, 3, len(z)).astype(np.float32)
    elif layout == "wall+outlier":
        z = (2.0 + 1e-4 * rng.random(len(z))).astype(np.float32)
        z[::50] = 40.0
    elif layout == "two-slabs":
        z = (np.where(rng.random(len(z)) < 0.5, 1.5, 4.0) + 0.02 * rng.random(len(z))).astype(np.float32)
    else:
        z[:] = 1.5
    scale = z / sc.means3D[:, 2]  # keep every splat on its pixel ray: same tiles, new depth
    sc.means3D = (sc.means3D * scale[:, None]).astype(np.float32)
    sc.means3D[:, 2] = z
    o, f = oracle.forward_scene(sc, omp=True)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    assert st.num_rendered == f.num_rendered
    lens = f.stages["ranges"][:, 1] - f.stages["ranges"][:, 0]
    if layout == "uniform":
        pass  # control: the scene's own depths (its lists are shorter: splat size follows depth)
    elif frame == "lists-to-4096":
        assert lens.max() > 1024 and (lens[lens > 0] <= 1024).any()  # both LDS sort kernels run
    else:
        assert (lens > 4096).sum() > 8
    np.testing.assert_array_equal(d["ranges"], f.stages["ranges"])
    np.testing.assert_array_equal(d["point_list"], f.stages["point_list"])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["uniform", "all-equal"])
def test_one_tile_with_a_hundred_thousand_entries(gsr, syn, layout):
    """A 16x16 frame: every splat lands in the single tile. 120 k list entries go through the global-scratch bucket sort
    (15 keys per bin on average), and with one depth for the whole map through the bitonic network in global memory."""
    cam = syn.make_camera(16, 16, 12.0, 12.0)
    sc = syn.make_scene(120000, cam, seed=8, scale_mult=2.0)
    if layout == "all-equal":
        sc.means3D = (sc.means3D * (np.float32(1.5) / sc.means3D[:, 2])[:, None]).astype(np.float32)
        sc.means3D[:, 2] = 1.5
    o, f = oracle.forward_scene(sc, omp=True)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    assert st.num_rendered == f.num_rendered and f.num_rendered > 100000
    np.testing.assert_array_equal(d["ranges"], f.stages["ranges"])
    np.testing.assert_array_equal(d["point_list"], f.stages["point_list"])
