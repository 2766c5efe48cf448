"""Randomised parity sweep (GPU box): many scene shapes against the CPU oracle, integer stages bit-exact,
image and gradients within 1e-4 away from knife-edge pixels. Usage: python scripts/fuzz_parity.py [n] [seed0]"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
from util import pose, rel_err
from oracle import oracle
gsr = load_package(); syn = gsr.synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed0)
worst = 0.0
fails = []
for it in range(n):
    W = int(rng.integers(17, 700)); H = int(rng.integers(17, 500))
    fx = float(rng.uniform(0.4, 1.5) * W); fy = float(fx * rng.uniform(0.9, 1.1))
    P = int(rng.choice([1, 7, 300, 5000, 40000, 150000]))
    mult = float(rng.choice([0.5, 1.0, 2.0, 4.0, 8.0, 16.0]))
    mode = str(rng.choice(["rgb", "depth", "sh"]))
    kw = dict(frac_behind=float(rng.choice([0.0, 0.2])), frac_offscreen=float(rng.choice([0.0, 0.3])))
    if mode == "sh": kw["sh_degree"] = int(rng.integers(0, 4))
    Tcw = pose(float(rng.uniform(0, 0.3))) if rng.random() < 0.5 else None
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3)) if rng.random() < 0.5 else (0, 0, 0)
    cam = syn.make_camera(W, H, fx, fy, Tcw=Tcw, bg=bg)
    sc = syn.make_scene(P, cam, seed=seed0 * 1000 + it, scale_mult=mult, color_mode=mode, **kw)
    o, f = oracle.forward_scene(sc, omp=True)
    mc, md = o.margins(f)
    ok = mc >= 1e-5
    g_in = sc.dL_dpix * ok[None]
    b = o.backward(g_in)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    assert st.num_rendered == f.num_rendered, (it, st.num_rendered, f.num_rendered)
    np.testing.assert_array_equal(st.radii.cpu().numpy(), f.radii)
    np.testing.assert_array_equal(d["ranges"], f.stages["ranges"])
    np.testing.assert_array_equal(d["point_list"], f.stages["point_list"])
    col = st.color.cpu().numpy()
    e_img = float(np.abs(col - f.color)[:, ok].max() / max(1.0, float(np.abs(f.color).max()))) if ok.any() else 0.0
    assert np.array_equal(d["n_contrib"].reshape(H, W)[ok], f.stages["n_contrib"].reshape(H, W)[ok])
    gr = gsr.backward(st, g_in)
    errs = {nme: rel_err(getattr(gr, nme).cpu().numpy(), getattr(b, nme)) for nme in
            ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")}
    e = max([e_img] + list(errs.values()))
    worst = max(worst, e)
    flag = "" if e <= 1e-4 else "  <-- FAIL"
    print(f"[{it}] {W}x{H} P={P} x{mult} {mode} R={f.num_rendered} knife={(~ok).mean():.4f} img={e_img:.1e} grad={max(errs.values()):.1e}{flag}", flush=True)
    if e > 1e-4: fails.append((it, {k: float("%.2e" % v) for k, v in errs.items() if v > 1e-4}))
print("worst", worst, "failures", fails)
assert not fails
