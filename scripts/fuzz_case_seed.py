"""One case of scripts/fuzz_parity.py (any seed) against the exact-state oracle: for every gradient tensor, the HIP kernels' and the fp32 oracle's
distance from the reference's formulas evaluated with the per-pixel state in double (oracle mode accum_double = 3).
    python scripts/fuzz_case_seed.py <seed0> <case>"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
from util import pose, rel_err
from oracle import oracle
gsr = load_package(); syn = gsr.synthetic
seed0, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0)
for it in range(case + 1):   # the generator of fuzz_parity.py, draw for draw
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
sc = syn.make_scene(P, cam, seed=seed0 * 1000 + case, scale_mult=mult, color_mode=mode, **kw)
print(f"seed {seed0} case {case}: {W}x{H} P={P} x{mult} {mode}")
o, f = oracle.forward_scene(sc, omp=True)
mc, _ = o.margins(f)
ok = mc >= 1e-5
g_in = sc.dL_dpix * ok[None]
b64 = o.backward(g_in)                       # the parity oracle (double accumulators per splat)
b32 = o.backward(g_in, accum_double=False)   # the reference's own arithmetic: everything in fp32
bex = o.backward(g_in, accum_double=3)       # the reference's formulas with the per-pixel state in double: their exact value for the same alphas
s = gsr.capi.Settings.from_camera(sc.cam)
st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
gr = gsr.backward(st, g_in)
for name in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
    hip = getattr(gr, name).cpu().numpy()
    print("%-14s HIP vs parity oracle %.2e | vs exact state: HIP %.2e, fp32 reference arithmetic %.2e" %
          (name, rel_err(hip, getattr(b64, name)), rel_err(hip, getattr(bex, name)), rel_err(np.asarray(getattr(b32, name)), getattr(bex, name))))
