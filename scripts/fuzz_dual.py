"""Randomised sweep of the FUSED PAIR (colour + depth / silhouette in one pass, out_ds / dL_dds with a non-zero silhouette gradient) against two renders of the
CPU oracle that use the reference's recursion (tests/test_gpu_dual.py's comparison on random frames): python scripts/fuzz_dual.py [n] [seed]   (GPU box)"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
from util import pose, rel_err, mixed_err
import test_gpu_dual as td
gsr = load_package(); syn = gsr.synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed0)
worst, fails = 0.0, []
for it in range(n):
    W = int(rng.integers(33, 500)); H = int(rng.integers(33, 400))
    fx = float(rng.uniform(0.5, 1.4) * W); fy = float(fx * rng.uniform(0.9, 1.1))
    P = int(rng.choice([7, 300, 5000, 40000]))
    mult = float(rng.choice([0.5, 1.0, 2.0, 4.0, 8.0]))
    Tcw = pose(float(rng.uniform(0, 0.3))) if rng.random() < 0.5 else None
    bg = tuple(float(x) for x in rng.uniform(0, 1, 3)) if rng.random() < 0.5 else (0, 0, 0)
    cam = syn.make_camera(W, H, fx, fy, Tcw=Tcw, bg=bg)
    sc = syn.make_scene(P, cam, seed=seed0 * 1000 + it, scale_mult=mult, frac_behind=float(rng.choice([0.0, 0.2])), frac_offscreen=float(rng.choice([0.0, 0.3])))
    gA = sc.dL_dpix
    gB = rng.standard_normal((3, H, W)).astype(np.float32); gB[0] *= 0.3; gB[2] = 0.0
    fA, fB, tot, ok = td._two_oracle_renders(sc, gA, gB)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, dual=True)
    ds = st.ds.cpu().numpy()
    e_img = max(float(np.abs(ds[0] - fB.color[0])[ok].max(initial=0)) / max(1.0, float(np.abs(fB.color[0]).max())), float(np.abs(ds[1] - fB.color[1])[ok].max(initial=0)))
    gr = gsr.backward(st, gA * ok[None], dL_dds=(gB * ok[None])[0:2])
    torch.cuda.synchronize()
    errs = {k: rel_err(getattr(gr, k).cpu().numpy(), ref) for k, ref in tot.items()}
    e = max(errs.values())
    worst = max(worst, e, e_img)
    bad = e > 1e-4 or e_img > 1e-4
    if bad: fails.append((it, {k: round(v, 7) for k, v in errs.items() if v > 1e-4}))
    print(f"[{it}] {W}x{H} P={P} x{mult} R={st.num_rendered} ds={e_img:.1e} grad={e:.1e}" + ("  <-- FAIL" if bad else ""), flush=True)
print("worst", worst, "failures", fails)
print("cases:", n, "beyond 1e-4:", len(fails))
