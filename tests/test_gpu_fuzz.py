"""GPU (-m gpu): randomised parity sweep of the HIP path against the CPU oracle (scripts/fuzz_parity.py as a test).
120 seeded frames — 17..700 x 17..500 pixels, 1..150 000 splats, scale x0.5..x16, RGB / depth / SH colours, random poses and
backgrounds, splats behind the camera and off screen: integer stages (radii, ranges, sorted point_list, num_rendered, n_contrib)
BIT-EXACT in every case; image and all nine gradient tensors within 1e-4 of their scale with the knife-edge protocol of
tests/test_gpu_parity.py (pixels whose smallest branch margin is below 1e-5 are left out on both sides: exp() is not
bit-reproducible between libm and the GPU).
Where a gradient tensor is further than 1e-4 from the fp32 oracle, the case is only accepted if the REFERENCE'S OWN fp32 arithmetic is
that ill-conditioned there: the oracle is run a second time with the per-pixel state of backward.cu:470-530 (T, accum_rec, dL/dalpha)
in double — the exact value of the reference's formulas for the alphas it blended with — and the HIP result must be within 1e-4 of
THAT and at least as close to it as the fp32 oracle is (x1.25). Seed 31's case 19 is the one such frame in 200 (597 x 30 pixels,
150 000 splats at scale x16, depth colours, 27 000-entry lists): its 15-24 deviating splats are the front-most entries of every list
(view depth 0.2000-0.2005), whose colour z equals what is accumulated behind them to 1e-3, so that (c - accum_rec) cancels; the fp32
reference formulas are 1.5e-4 / 1.1e-4 / 1.07e-4 off their exact value there (dL_dconic / dL_dopacity / dL_dscales), the HIP kernels
5.7e-5 / 3.0e-5 / 7.7e-5 (it forms c - S per channel before the contraction, csrc/gsr_blend.h). Widening the knife-edge mask tenfold
does not change those numbers: no threshold flip is involved (round 3's explanation was wrong; scripts/fuzz_case.py)."""
import numpy as np
import pytest
import torch

from util import pose, rel_err

pytestmark = pytest.mark.gpu
SEED0, CASES, CHUNKS = 31, 120, 6
GRADS = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")


def _configs():
    rng = np.random.default_rng(SEED0)          # one stream for all cases, as scripts/fuzz_parity.py draws them
    out = []
    for it in range(CASES):
        W = int(rng.integers(17, 700)); H = int(rng.integers(17, 500))
        fx = float(rng.uniform(0.4, 1.5) * W); fy = float(fx * rng.uniform(0.9, 1.1))
        P = int(rng.choice([1, 7, 300, 5000, 40000, 150000]))
        mult = float(rng.choice([0.5, 1.0, 2.0, 4.0, 8.0, 16.0]))
        mode = str(rng.choice(["rgb", "depth", "sh"]))
        kw = dict(frac_behind=float(rng.choice([0.0, 0.2])), frac_offscreen=float(rng.choice([0.0, 0.3])))
        if mode == "sh":
            kw["sh_degree"] = int(rng.integers(0, 4))
        Tcw = pose(float(rng.uniform(0, 0.3))) if rng.random() < 0.5 else None
        bg = tuple(float(x) for x in rng.uniform(0, 1, 3)) if rng.random() < 0.5 else (0, 0, 0)
        out.append(dict(it=it, W=W, H=H, fx=fx, fy=fy, P=P, mult=mult, mode=mode, kw=kw, Tcw=Tcw, bg=bg))
    return out


def _run_case(gsr, syn, oracle, c, margin):
    cam = syn.make_camera(c["W"], c["H"], c["fx"], c["fy"], Tcw=c["Tcw"], bg=c["bg"])
    sc = syn.make_scene(c["P"], cam, seed=SEED0 * 1000 + c["it"], scale_mult=c["mult"], color_mode=c["mode"], **c["kw"])
    o, f = oracle.forward_scene(sc, omp=True)
    mc, _ = o.margins(f)
    ok = mc >= margin
    g_in = sc.dL_dpix * ok[None]
    b = o.backward(g_in)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    # integer stages: bit-exact, always
    assert st.num_rendered == f.num_rendered, (c["it"], st.num_rendered, f.num_rendered)
    np.testing.assert_array_equal(st.radii.cpu().numpy(), f.radii)
    np.testing.assert_array_equal(d["ranges"], f.stages["ranges"])
    np.testing.assert_array_equal(d["point_list"], f.stages["point_list"])
    H, W = c["H"], c["W"]
    assert np.array_equal(d["n_contrib"].reshape(H, W)[ok], f.stages["n_contrib"].reshape(H, W)[ok])
    col = st.color.cpu().numpy()
    e_img = float(np.abs(col - f.color)[:, ok].max() / max(1.0, float(np.abs(f.color).max()))) if ok.any() else 0.0
    gr = gsr.backward(st, g_in)
    errs = {n: rel_err(getattr(gr, n).cpu().numpy(), getattr(b, n)) for n in GRADS}
    cond = {}
    if max(errs.values()) > 1e-4:   # how far is the reference's own fp32 arithmetic from the exact value of its formulas here?
        ex = o.backward(g_in, accum_double=3)
        for n in GRADS:
            if errs[n] > 1e-4:
                cond[n] = (rel_err(getattr(gr, n).cpu().numpy(), getattr(ex, n)), rel_err(getattr(b, n), getattr(ex, n)))
    return e_img, errs, float((~ok).mean()), f.num_rendered, cond


@pytest.mark.parametrize("chunk", range(CHUNKS))
def test_random_frames_match_the_oracle(gsr, syn, chunk):
    from oracle import oracle
    cfgs = _configs()[chunk * (CASES // CHUNKS):(chunk + 1) * (CASES // CHUNKS)]
    ill = []
    for c in cfgs:
        e_img, errs, knife, R, cond = _run_case(gsr, syn, oracle, c, 1e-5)
        line = f"[{c['it']}] {c['W']}x{c['H']} P={c['P']} x{c['mult']} {c['mode']} R={R} knife={knife:.4f} img={e_img:.1e} grad={max(errs.values()):.1e}"
        assert e_img <= 1e-4, line
        for n, e in errs.items():
            if e <= 1e-4:
                continue
            hip_ex, ref_ex = cond[n]
            line += f"  {n}: {e:.2e} from the fp32 oracle; from the exactly evaluated formulas: HIP {hip_ex:.2e}, fp32 oracle {ref_ex:.2e}"
            assert e <= 2e-4 and hip_ex <= 1e-4 and hip_ex <= 1.25 * ref_ex, line
        if cond:
            ill.append(c["it"])
        print(line, flush=True)
    assert len(ill) <= 1, ill
