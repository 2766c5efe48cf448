"""GPU (-m gpu): the REFERENCE's own rasterizer run on this GPU — oracle/_ref/libgsr_ref.so, built by oracle/build_ref.sh from the reference's
sources where they lie (Thirdparty/diff_gaussian_rasterization/cuda_rasterizer/*.cu, *.h translated by ROCm's hipify-perl at build time and compiled by hipcc
with -ffp-contract=off; oracle/ref_shim.hip only moves arrays) — against (A) the CPU oracle and (B) the HIP library, on the same seeded inputs.

(A) is what pins the oracle: its restatement of preprocessCUDA / computeCov2D / duplicateWithKeys / the radix sort / identifyTileRanges / renderCUDA and the
whole backward (forward.cu:74-401, backward.cu:144-557, rasterizer_impl.cu:71-345) is held to values the reference's own kernels computed — radii,
tiles_touched, offsets, sorted keys, point_list, ranges and the projected geometry BIT-EXACT, images and gradients inside the 1e-4 bars (observed: colour
4e-7, gradients 4e-7). (B) holds the shipped library to the same values directly. What this does NOT pin: nvcc's own code generation (its contraction
choices, CUDA's expf against ROCm's) — test_contracted_reference_build_census counts what hipcc's default contraction moves, the nearest thing to it that
can be run here.

The bars are test_gpu_parity.py's; the knife-edge pixels (a blend branch within 1e-5 of its threshold: exp() differs between glibc, CUDA and ROCm in the
last bit) come from the oracle's margins, as there."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle, ref
from test_gpu_parity import BIG, CASE_NAMES, EPS_MARGIN, ODD, SMALL, TOL, _build, _cases
from util import mixed_err, rel_err

pytestmark = pytest.mark.gpu

GRADS = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
INT_STAGES = ("tiles_touched", "point_offsets", "keys_unsorted", "values_unsorted", "keys_sorted", "point_list", "ranges")
GEOM_STAGES = ("means2D", "depths", "conic_opacity", "cov3D")


@pytest.fixture(scope="module", autouse=True)
def _needs_the_reference_build():
    if not ref.available():
        pytest.skip("opt-in: oracle/_ref is built and used only with GSR_REFERENCE_BUILD=1 (oracle/build_ref.sh: a human's decision)")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], -1) if a.ndim else a


def _compare_forward(name, what, got_stages, got_radii, got_R, got_color, got_depth, fr, ok_c, ok_d, H, W, sh):
    """`got` (the oracle, or the HIP library re-expressed by gsr_debug_export) against the reference's forward `fr`"""
    assert got_R == fr.num_rendered, (name, what)
    np.testing.assert_array_equal(got_radii, fr.radii, err_msg="%s %s radii" % (name, what))
    vis = fr.radii > 0
    for k in INT_STAGES:
        if k in got_stages:
            np.testing.assert_array_equal(got_stages[k], fr.stages[k], err_msg="%s %s %s" % (name, what, k))
    for k in GEOM_STAGES:       # (a culled splat's entries are never written by the reference: forward.cu:196-200 returns early)
        if k in got_stages:
            a, b = got_stages[k].reshape(len(vis), -1)[vis], fr.stages[k].reshape(len(vis), -1)[vis]
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s %s %s: %d of %d values differ" % (name, what, k, int((a.view(np.uint32) != b.view(np.uint32)).sum()), a.size)
    if sh and "rgb" in got_stages:
        np.testing.assert_allclose(got_stages["rgb"].reshape(-1, 3)[vis], fr.stages["rgb"].reshape(-1, 3)[vis], atol=2e-6)
    scale = max(1.0, float(np.abs(fr.color).max()))
    e_col = float(np.abs(got_color - fr.color)[:, ok_c].max())
    assert e_col <= TOL * scale, (name, what, e_col)
    assert np.array_equal(got_depth.reshape(H, W)[ok_d], fr.depth.reshape(H, W)[ok_d]), (name, what, "median depth")
    e_T = float(np.abs(got_stages["final_T"].reshape(H, W) - fr.stages["final_T"].reshape(H, W))[ok_c].max())
    assert e_T <= TOL, (name, what, e_T)
    assert np.array_equal(got_stages["n_contrib"].reshape(H, W)[ok_c], fr.stages["n_contrib"].reshape(H, W)[ok_c]), (name, what, "n_contrib")
    return e_col, e_T


def _compare_grads(name, what, got, want, afloor):
    worst = {}
    for n in GRADS:
        a, b = np.asarray(getattr(got, n)), np.asarray(getattr(want, n))
        if b.size == 0:
            continue
        e, m = rel_err(a, b), mixed_err(a, b, afloor=afloor)
        worst[n] = (e, m)
        assert e <= TOL, (name, what, n, e)
        assert m <= 1.0, (name, what, n, "element-wise bar exceeded by x%.2f" % m)
    return worst


@pytest.mark.parametrize("name", CASE_NAMES)
def test_oracle_and_hip_library_against_the_reference_kernels(gsr, syn, name):
    sc = _build(syn, **_cases(syn)[name])
    H, W = sc.cam.height, sc.cam.width
    r, fr = ref.forward_scene(sc)                                  # the reference's kernels
    o, fo = oracle.forward_scene(sc, omp=name in BIG)              # the CPU restatement
    mc, md = o.margins(fo)
    ok_c, ok_d = mc >= EPS_MARGIN, md >= EPS_MARGIN
    g_in = sc.dL_dpix * ok_c[None]
    br = r.backward(g_in)
    # both sides of (A) accumulate in float here (the reference's atomicAdd; the oracle's double accumulators are the parity tests' choice): the floor of the
    # element-wise bar is test_gpu_parity's
    afloor = 4e-5 if name == "deep-stack-depth" else 1e-6

    # ---- (A) the oracle against the reference
    ea = _compare_forward(name, "oracle", fo.stages, fo.radii, fo.num_rendered, fo.color, fo.depth, fr, ok_c, ok_d, H, W, sc.shs is not None)
    wa = _compare_grads(name, "oracle", o.backward(g_in), br, afloor)

    # ---- (B) the HIP library against the reference
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    stages = dict(d)
    stages["keys_sorted"] = d["point_list_keys"]
    eb = _compare_forward(name, "hip", stages, st.radii.cpu().numpy(), st.num_rendered, st.color.cpu().numpy(), st.depth.cpu().numpy(), fr, ok_c, ok_d, H, W,
                          sc.shs is not None)
    gr = gsr.backward(st, g_in)
    torch.cuda.synchronize()

    class _G:
        pass
    gg = _G()
    for n in GRADS:
        setattr(gg, n, getattr(gr, n).cpu().numpy())
    wb = _compare_grads(name, "hip", gg, br, afloor)
    print("\n%s against the reference's kernels: oracle colour %.1e final_T %.1e, worst gradient %.1e | HIP library colour %.1e final_T %.1e, worst gradient %.1e"
          % (name, ea[0], ea[1], max(v[0] for v in wa.values()), eb[0], eb[1], max(v[0] for v in wb.values())))


def test_reference_mark_visible_and_culled_scene(gsr, syn):
    """markVisible (rasterizer_impl.cu:140-160) and a frame with splats behind the camera / off screen: the three implementations agree on who is visible."""
    sc = _build(syn, 4000, SMALL, mult=2.0, frac_behind=0.3, frac_offscreen=0.3)
    s = gsr.capi.Settings.from_camera(sc.cam)
    m_ref = ref.mark_visible(sc.means3D, sc.cam)
    m_ora = oracle.mark_visible(sc.means3D, sc.cam)
    m_hip = gsr.mark_visible(torch.tensor(sc.means3D, device="cuda"), s.viewmatrix, s.projmatrix).cpu().numpy()
    assert 0 < int(m_ref.sum()) < len(m_ref)
    assert np.array_equal(m_ref, m_ora) and np.array_equal(m_ref, m_hip)
    _, fr = ref.forward_scene(sc)
    _, fo = oracle.forward_scene(sc)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    assert int((fr.radii == 0).sum()) > 500
    np.testing.assert_array_equal(fo.radii, fr.radii)
    np.testing.assert_array_equal(st.radii.cpu().numpy(), fr.radii)


def test_visible_filter_against_the_reference(gsr, syn):
    """Rasterizer::visible_filter on the 1.2x enlarged image of Render.cc:784-831"""
    sc = _build(syn, 6000, SMALL, mult=3.0, frac_behind=0.2, frac_offscreen=0.4)
    s = gsr.capi.Settings.from_camera(sc.cam)
    W2, H2 = int(sc.cam.width * 1.2), int(sc.cam.height * 1.2)
    want = ref.filter_radii(sc.means3D, sc.scales, sc.rotations, sc.cam, W2, H2)
    assert 0 < int((want > 0).sum()) < len(want)
    np.testing.assert_array_equal(oracle.filter_radii(sc.means3D, sc.scales, sc.rotations, sc.cam, W2, H2), want)
    np.testing.assert_array_equal(gsr.visible_filter(s, sc.means3D, sc.scales, sc.rotations, W2, H2).cpu().numpy(), want)


def test_cov3d_precomp_path_against_the_reference(gsr, syn):
    sc = _build(syn, 3000, ODD, mult=2.0, bg=(0.1, 0.1, 0.4))
    _, f0 = ref.forward_scene(sc)
    cov = f0.stages["cov3D"].reshape(-1, 6)                        # the reference's own computeCov3D
    r = ref.Reference()
    fr = r.forward(means3D=sc.means3D, opacities=sc.opacities, cam=sc.cam, colors=sc.colors, cov3D_precomp=cov)
    o = oracle.Oracle()
    fo = o.forward(means3D=sc.means3D, opacities=sc.opacities, cam=sc.cam, colors=sc.colors, cov3D_precomp=cov)
    mc, _ = o.margins(fo)
    g_in = sc.dL_dpix * (mc >= EPS_MARGIN)[None]
    br, bo = r.backward(g_in), o.backward(g_in)
    np.testing.assert_array_equal(fo.radii, fr.radii)
    np.testing.assert_array_equal(fo.stages["point_list"], fr.stages["point_list"])
    assert rel_err(bo.dL_dcov3D, br.dL_dcov3D) <= TOL and rel_err(bo.dL_dmeans3D, br.dL_dmeans3D) <= TOL
    assert float(np.abs(br.dL_dscales).max()) == 0.0 and float(np.abs(br.dL_drotations).max()) == 0.0
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, cov3D_precomp=cov)
    np.testing.assert_array_equal(st.radii.cpu().numpy(), fr.radii)
    gr = gsr.backward(st, g_in)
    assert rel_err(gr.dL_dcov3D.cpu().numpy(), br.dL_dcov3D) <= TOL and rel_err(gr.dL_dmeans3D.cpu().numpy(), br.dL_dmeans3D) <= TOL


@pytest.mark.parametrize("name", ["replica-1M-rgb", "scannet-2M-rgb", "odd-sh3-pose-bg", "fat-clamped"])
def test_contracted_reference_build_census(syn, name):
    """What floating-point contraction moves in the reference's OWN kernels: libgsr_ref_fma.so (hipcc's default -ffp-contract=fast, the counterpart of nvcc's
    default --fmad=true that the reference's CMake builds with) against libgsr_ref.so (-ffp-contract=off, the statement the oracle and the library hold). The
    GPU-code counterpart of tests/test_oracle_fma_census.py (gcc's contractions of the C restatement). hipcc's choice of contractions is not nvcc's: the size of
    the effect, not a prediction of which entries move. Asserted: the INDEX stages barely move (a radius needs a covariance eigenvalue within an ulp of an
    integer boundary), images stay inside the 1e-4 bar."""
    sc = _build(syn, **_cases(syn)[name])
    _, a = ref.forward_scene(sc)
    _, b = ref.forward_scene(sc, fma=True)
    vis = a.radii > 0
    P = len(vis)
    out = {"scene": name, "P": P, "visible": int(vis.sum()), "num_rendered": [a.num_rendered, b.num_rendered],
           "radii_differ": int((a.radii != b.radii).sum()), "tiles_touched_differ": int((a.stages["tiles_touched"] != b.stages["tiles_touched"]).sum())}
    if a.num_rendered == b.num_rendered:
        out["point_list_positions_that_differ"] = int((a.stages["point_list"] != b.stages["point_list"]).sum())
        out["ranges_differ"] = int((a.stages["ranges"] != b.stages["ranges"]).any(1).sum())
    for k in GEOM_STAGES:
        x, y = a.stages[k].reshape(P, -1)[vis], b.stages[k].reshape(P, -1)[vis]
        out[k + "_values_that_differ"] = int((x.view(np.uint32) != y.view(np.uint32)).sum())
        out[k + "_max_rel"] = float((np.abs(x - y) / np.maximum(np.abs(x), 1e-30)).max()) if x.size else 0.0
    out["color_max_abs_diff"] = float(np.abs(a.color - b.color).max())
    out["color_pixels_beyond_1e-4"] = int((np.abs(a.color - b.color).max(0) > 1e-4).sum())
    out["n_contrib_pixels_differ"] = int((a.stages["n_contrib"] != b.stages["n_contrib"]).sum())
    print("\ncontraction census, the reference's kernels:", json.dumps(out))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "ref_fma_census_%s.json" % name), "w") as f:
            json.dump(out, f, indent=1)
    assert out["radii_differ"] <= max(2, P // 100000)
    assert out["tiles_touched_differ"] <= max(2, P // 100000)
    assert out["color_pixels_beyond_1e-4"] <= max(4, a.color[0].size // 20000)


@pytest.mark.parametrize("mode", ["rgb", "sh"])
def test_python_operator_binding_argument_for_argument(gsr, syn, mode):
    """The `_C` module of the Python operator: the reference's own binding (Thirdparty/diff_gaussian_rasterization/ext.cpp:15-19, rasterize_points.cu — built
    against this image's libtorch as oracle/_ref/gsr_ref_C.so) and the library's (`gsorb-slam_amd/diff_gaussian_rasterization/_C.so`) are called with the SAME
    positional arguments — rasterize_gaussians (18 of them), rasterize_gaussians_backward (20), mark_visible (3) — and return tuples of the same arity, shapes and
    dtypes with the same values: num_rendered, radii, the median depth equal; colour and the eight gradient tensors inside the 1e-4 bars. (The three state
    buffers are each implementation's own layout: what crosses forward -> backward is passed back to the module that made it.)"""
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "oracle", "_ref"))
    sys.path.insert(0, os.path.join(here, "gsorb-slam_amd"))
    if not ref.available() or not os.path.exists(os.path.join(here, "oracle", "_ref", "gsr_ref_C.so")):
        pytest.skip("oracle/_ref/gsr_ref_C.so is not built")
    import gsr_ref_C
    from diff_gaussian_rasterization import _C
    kw = dict(P=3000, cam=ODD, mode="sh", mult=3.0, bg=(0.3, 0.5, 0.7), frac_behind=0.1, frac_offscreen=0.3) if mode == "sh" else dict(P=10000, cam=syn.TUM1, mult=2.0)
    sc = _build(syn, **kw)
    o, fo = oracle.forward_scene(sc)
    mc, _ = o.margins(fo)
    ok = mc >= EPS_MARGIN
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    empty = torch.empty(0, device="cuda")
    cam = sc.cam
    sh = t(sc.shs) if sc.shs is not None else empty
    colors = empty if sc.shs is not None else t(sc.colors)
    fwd_args = (t(cam.bg), t(sc.means3D), colors, t(sc.opacities), t(sc.scales), t(sc.rotations), float(cam.scale_modifier), empty, t(cam.viewmatrix), t(cam.projmatrix),
                float(cam.tanfovx), float(cam.tanfovy), int(cam.height), int(cam.width), sh, int(cam.sh_degree), t(cam.campos), False)
    a, b = _C.rasterize_gaussians(*fwd_args), gsr_ref_C.rasterize_gaussians(*fwd_args)
    torch.cuda.synchronize()
    assert len(a) == len(b) == 7 and a[0] == b[0] == fo.num_rendered
    for i in (1, 2, 6):                                           # colour [3,H,W], radii [P] int32, depth [1,H,W]
        assert a[i].shape == b[i].shape and a[i].dtype == b[i].dtype and a[i].device == b[i].device, i
    for i in (3, 4, 5):                                           # the three state buffers: byte tensors on the device
        assert a[i].dtype == b[i].dtype == torch.uint8 and a[i].is_cuda and a[i].dim() == b[i].dim() == 1
    assert torch.equal(a[2], b[2])
    okt = torch.tensor(ok, device="cuda")
    assert float((a[1] - b[1]).abs()[:, okt].max()) <= TOL * max(1.0, float(b[1].abs().max()))
    assert torch.equal(a[6][0][okt], b[6][0][okt])
    g = t(sc.dL_dpix * ok[None])
    bwd = lambda out: (fwd_args[0], fwd_args[1], out[2], fwd_args[2], fwd_args[4], fwd_args[5], fwd_args[6], fwd_args[7], fwd_args[8], fwd_args[9], fwd_args[10],
                       fwd_args[11], g, fwd_args[14], fwd_args[15], fwd_args[16], out[3], out[0], out[4], out[5])
    ga, gb = _C.rasterize_gaussians_backward(*bwd(a)), gsr_ref_C.rasterize_gaussians_backward(*bwd(b))
    torch.cuda.synchronize()
    assert len(ga) == len(gb) == 8
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    worst = {}
    for n, x, y in zip(names, ga, gb):
        assert x.shape == y.shape and x.dtype == y.dtype, (n, x.shape, y.shape)
        if y.numel() == 0:
            continue
        e, m = rel_err(x.cpu().numpy(), y.cpu().numpy()), mixed_err(x.cpu().numpy(), y.cpu().numpy())
        worst[n] = e
        assert e <= TOL and m <= 1.0, (n, e, m)
    va, vb = _C.mark_visible(fwd_args[1], fwd_args[8], fwd_args[9]), gsr_ref_C.mark_visible(fwd_args[1], fwd_args[8], fwd_args[9])
    assert va.dtype == vb.dtype and torch.equal(va, vb)
    print("\n_C against the reference's _C (%s): worst gradient %.1e" % (mode, max(worst.values())))


def _read_blocks(path):
    import struct
    out, b = {}, open(path, "rb").read()
    i = 0
    while i < len(b):
        (ln,) = struct.unpack_from("<i", b, i); i += 4
        name = b[i:i + ln].decode(); i += ln
        kind, n = struct.unpack_from("<iq", b, i); i += 12
        out[name] = np.frombuffer(b, dtype=np.float32 if kind == 0 else np.int32, count=n, offset=i).copy(); i += 4 * n
    return out


@pytest.mark.parametrize("mode", ["rgb", "sh"])
def test_one_cpp_caller_against_both_host_layers(gsr, syn, tmp_path, mode):
    """tests/cpp/dropin_main.cpp is written once against the reference's C++ API as src/Render.cc uses it (GaussianRasterizationSettings,
    GaussianRasterizer::forward / mark_visible / Visable, autograd through the render, distCUDA2; include/Rasterizer.cuh:76-382) and compiled twice without an
    #ifdef: against this repository's host layer (tests/cpp/dropin_hip.bin) and against the reference's own src/Rasterizer.cu + spatial.cu + rasterizer
    (oracle/_ref/dropin_ref.bin). Same scene file in, the two output files compared: radii, median depth, visibility, filter radii equal; colour, the gradients
    autograd hands back for every leaf (means3D, means2D, opacities, scales, rotations, colours or SH) and distCUDA2 inside the bars; the same
    std::invalid_argument for the same bad argument combinations."""
    import struct
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe_hip, exe_ref = os.path.join(here, "tests", "cpp", "dropin_hip.bin"), os.path.join(here, "oracle", "_ref", "dropin_ref.bin")
    if not ref.available() or not os.path.exists(exe_ref):
        pytest.skip("oracle/_ref/dropin_ref.bin is not built")
    assert os.path.exists(exe_hip), "tests/cpp/dropin_hip.bin is missing: run __graft_entry__.build()"
    kw = dict(P=3000, cam=ODD, mode="sh", mult=3.0, Tcw=None, bg=(0.3, 0.5, 0.7), frac_behind=0.1, frac_offscreen=0.3) if mode == "sh" else dict(P=10000, cam=syn.TUM1, mult=2.0)
    sc = _build(syn, **kw)
    cam = sc.cam
    o, fo = oracle.forward_scene(sc)
    mc, md = o.margins(fo)
    ok = mc >= EPS_MARGIN
    G = (sc.dL_dpix * ok[None]).astype(np.float32)
    P = len(sc.means3D)
    M = 0 if sc.shs is None else sc.shs.shape[1]
    scene = str(tmp_path / "scene.bin")
    with open(scene, "wb") as f:
        f.write(struct.pack("<5i3f", P, M, cam.width, cam.height, cam.sh_degree, cam.tanfovx, cam.tanfovy, cam.scale_modifier))
        for a in (cam.bg, cam.viewmatrix, cam.projmatrix, cam.campos, sc.means3D, sc.opacities, sc.scales, sc.rotations, sc.colors if M == 0 else sc.shs, G, sc.means3D):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    outs = {}
    for name, exe in (("hip", exe_hip), ("ref", exe_ref)):
        out = str(tmp_path / (name + ".out"))
        r = subprocess.run([exe, scene, out], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stdout[-500:], r.stderr[-2000:])
        outs[name] = _read_blocks(out)
    a, b = outs["hip"], outs["ref"]
    assert set(a) == set(b)
    for k in ("radii", "visible", "filter_radii", "invalid_argument_checks"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert int(b["invalid_argument_checks"][0]) == 7                       # all three bad combinations throw, on both sides
    H, W = cam.height, cam.width
    assert np.abs(a["color"] - b["color"]).reshape(3, H, W)[:, ok].max() <= TOL * max(1.0, float(np.abs(b["color"]).max()))
    okd = md >= EPS_MARGIN
    assert np.array_equal(a["depth"].reshape(H, W)[okd], b["depth"].reshape(H, W)[okd])
    worst = {}
    for k in sorted(a):
        if k.startswith("d_") or k == "dist2":
            assert a[k].shape == b[k].shape, k
            with np.errstate(over="ignore", invalid="ignore"):
                fin = np.isfinite(b[k])
                assert np.array_equal(fin, np.isfinite(a[k])), k
                e, m = rel_err(a[k][fin], b[k][fin]), mixed_err(a[k][fin], b[k][fin])
            worst[k] = e
            assert e <= TOL and m <= 1.0, (k, e, m)
    print("\none C++ caller, two host layers (%s): %s" % (mode, {k: "%.1e" % v for k, v in worst.items()}))


def test_random_frames_against_the_reference_kernels(gsr, syn):
    """scripts/fuzz_ref.py as a test: 200 random frames (sizes, splat counts, scales, colour modes, poses, culling as tests/test_gpu_fuzz.py draws them; another
    seed) through the HIP library and through the reference's own kernels, no CPU oracle in the loop. Index stages and projected geometry bit-exact in every
    frame (asserted inside run_case); images within 1e-4 on the pixels where both renders took the same branches; gradients within 1e-4 — or, on the one
    documented kind of ill-conditioned frame (depth colours at scale x16, lists of tens of thousands of entries: tests/test_gpu_fuzz.py), no further from the
    EXACT value of the reference's formulas than the reference's own fp32 kernels are (x1.25) or than 1e-4. The 1 000-frame sweep of the same script:
    profiles/r06_fuzz_ref_1000.json."""
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "scripts"))
    import fuzz_ref
    seed, ill = 77, []
    rows = [fuzz_ref.run_case(gsr, syn, c, seed) for c in fuzz_ref.configs(200, seed)]
    for x in rows:
        assert x["image"] <= TOL, x
        assert x["depth_pixels_differ"] <= 2, x
        assert x["branch_pixels"] <= 2e-3, x
        if x["worst_grad"] > TOL:
            c = x["ill_conditioned"]
            assert c["hip_vs_exact"] <= max(TOL, 1.25 * c["reference_vs_exact"]), x
            ill.append((x["it"], x["worst_grad"], c))
    print("\n200 random frames against the reference's kernels: %d tile instances, image worst %.1e, gradient worst %.1e, ill-conditioned frames %s"
          % (sum(x["R"] for x in rows), max(x["image"] for x in rows), max(x["worst_grad"] for x in rows), ill))
    assert len(ill) <= 3
