"""One case of the randomised sweep (tests/test_gpu_fuzz.py) looked at closely: which splats deviate, by how much, what they look like.
    python scripts/fuzz_case.py 19"""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
import test_gpu_fuzz as tf
from oracle import oracle
gsr = load_package(); syn = gsr.synthetic
it = int(sys.argv[1]) if len(sys.argv) > 1 else 19
c = tf._configs()[it]
cam = syn.make_camera(c["W"], c["H"], c["fx"], c["fy"], Tcw=c["Tcw"], bg=c["bg"])
sc = syn.make_scene(c["P"], cam, seed=tf.SEED0 * 1000 + c["it"], scale_mult=c["mult"], color_mode=c["mode"], **c["kw"])
o, f = oracle.forward_scene(sc, omp=True)
mc, _ = o.margins(f)
ok = mc >= 1e-5
g_in = sc.dL_dpix * ok[None]
b = o.backward(g_in)
b32 = o.backward(g_in, accum_double=False)
bex = o.backward(g_in, accum_double=3)     # the reference's formulas with the per-pixel state in double: their exact value for the same alphas
s = gsr.capi.Settings.from_camera(sc.cam)
st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
gr = gsr.backward(st, g_in)
d = gsr.debug_export(st)
for name in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
    a = getattr(gr, name).cpu().numpy().astype(np.float64).reshape(c["P"], -1); r = np.asarray(getattr(b, name), np.float64).reshape(c["P"], -1)
    r32 = np.asarray(getattr(b32, name), np.float64).reshape(c["P"], -1)
    scale = np.abs(r).max()
    e = np.abs(a - r).max(1) / scale
    bad = np.argsort(-e)[:8]
    ex = np.asarray(getattr(bex, name), np.float64).reshape(c["P"], -1)
    print(f"{name}: scale {scale:.3e}, max err {e.max():.2e}, splats over 1e-5: {(e > 1e-5).sum()}, oracle float-vs-double {np.abs(r32 - r).max() / scale:.1e}; "
          f"against the exactly evaluated formulas: HIP {np.abs(a - ex).max() / scale:.2e}, fp32 oracle {np.abs(r - ex).max() / scale:.2e}")
    for i in bad[:4]:
        print(f"   splat {i}: err {e[i]:.2e} gpu {a[i]} ref {r[i]} radius {f.radii[i]} depth {f.stages['depths'][i]:.4f} opac {float(np.ravel(sc.opacities)[i]):.3f} "
              f"conic {f.stages['conic_opacity'][i][:3]} mean2D {f.stages['means2D'][i]} tiles {f.stages['tiles_touched'][i]}")

# ---- which pixels carry the deviation of one splat? The backward is linear in the upstream gradient: mask it by tile column, then by
# pixel column, then by row, and compare the splat's dL_dconic on both sides each time.
if len(sys.argv) > 2:
    sp = int(sys.argv[2])
    W, H = c["W"], c["H"]
    def both(mask):
        g = (g_in * mask[None]).astype(np.float32)
        bb = o.backward(g)
        gg = gsr.backward(st, g)
        return gg.dL_dconic.cpu().numpy().reshape(c["P"], -1)[sp].astype(np.float64), np.asarray(bb.dL_dconic, np.float64).reshape(c["P"], -1)[sp]
    full = both(np.ones((H, W), bool))
    print("splat", sp, "full: gpu", full[0], "ref", full[1], "diff", full[0] - full[1])
    worst = []
    for tx in range((W + 15) // 16):
        m = np.zeros((H, W), bool); m[:, tx * 16:(tx + 1) * 16] = True
        a_, r_ = both(m)
        dd = np.abs(a_ - r_).max()
        worst.append((dd, tx))
        if dd > 1e-6: print("  tile column", tx, "gpu", a_, "ref", r_, "diff %.2e" % dd)
    worst.sort(reverse=True)
    tx = worst[0][1]
    for x in range(tx * 16, min(W, tx * 16 + 16)):
        for y0 in range(0, H, 8):
            m = np.zeros((H, W), bool); m[y0:y0 + 8, x] = True
            a_, r_ = both(m)
            if np.abs(a_ - r_).max() > 1e-6:
                for y in range(y0, min(H, y0 + 8)):
                    m = np.zeros((H, W), bool); m[y, x] = True
                    a_, r_ = both(m)
                    if np.abs(a_ - r_).max() > 1e-7:
                        pl = f.stages["point_list"]; rg = f.stages["ranges"].reshape(-1, 2)
                        t = (y // 16) * ((W + 15) // 16) + x // 16
                        lst = pl[rg[t][0]:rg[t][1]]
                        pos = int(np.where(lst == sp)[0][0]) if (lst == sp).any() else -1
                        print(f"    pixel ({x},{y}): gpu {a_} ref {r_}  n_contrib gpu {d['n_contrib'].reshape(H, W)[y, x]} ref {f.stages['n_contrib'].reshape(H, W)[y, x]} "
                              f"list len {len(lst)} splat at list pos {pos} margin {mc[y, x]:.2e} final_T gpu {d['final_T'].reshape(H, W)[y, x]:.3e} ref {f.stages['final_T'].reshape(H, W)[y, x]:.3e}")
