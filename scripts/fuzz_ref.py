"""Randomised sweep of the HIP library against the REFERENCE's own kernels on this GPU (oracle/_ref/libgsr_ref.so: oracle/build_ref.sh) — no CPU oracle in
the loop, so thousands of frames per minute. Frames drawn like tests/test_gpu_fuzz.py's (17..700 x 17..500 pixels, 1..150 000 splats, scale x0.5..x16,
RGB / depth / SH colours, random poses / backgrounds, culled splats).
  * index stages (radii, tiles_touched, sorted keys, point_list, ranges, num_rendered) and the projected geometry of visible splats: BIT-EXACT, every frame;
  * images: pixels where the two renders took the same branches (n_contrib equal, final T equal to 1e-3 relative) within 1e-4; the others (exp() differs in the last bit
    between the two builds: a blend branch within rounding of its threshold) are counted and stay rare;
  * the nine gradient tensors, upstream gradient zeroed on those pixels for both sides: tensor-scale relative error, worst case reported.
    python scripts/fuzz_ref.py [frames] [seed] [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
from conftest import load_package
from oracle import ref
from util import pose, rel_err

GRADS = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")


def configs(n, seed):
    rng = np.random.default_rng(seed)
    for it in range(n):
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
        yield dict(it=it, W=W, H=H, fx=fx, fy=fy, P=P, mult=mult, mode=mode, kw=kw, Tcw=Tcw, bg=bg)


def run_case(gsr, syn, c, seed):
    cam = syn.make_camera(c["W"], c["H"], c["fx"], c["fy"], Tcw=c["Tcw"], bg=c["bg"])
    sc = syn.make_scene(c["P"], cam, seed=seed * 1000 + c["it"], scale_mult=c["mult"], color_mode=c["mode"], **c["kw"])
    r, fr = ref.forward_scene(sc)
    s = gsr.capi.Settings.from_camera(sc.cam)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
    d = gsr.debug_export(st)
    H, W = c["H"], c["W"]
    assert st.num_rendered == fr.num_rendered, ("num_rendered", c["it"])
    np.testing.assert_array_equal(st.radii.cpu().numpy(), fr.radii)
    for a, b in (("tiles_touched", "tiles_touched"), ("point_list_keys", "keys_sorted"), ("point_list", "point_list"), ("ranges", "ranges")):
        np.testing.assert_array_equal(d[a], fr.stages[b], err_msg="%s frame %d" % (a, c["it"]))
    vis = fr.radii > 0
    for k in ("means2D", "depths", "conic_opacity"):
        x, y = d[k].reshape(len(vis), -1)[vis], fr.stages[k].reshape(len(vis), -1)[vis]
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (k, c["it"])
    col = st.color.cpu().numpy()
    # (a skipped / blended flip anywhere in a pixel's list moves its final T by a factor 1 - alpha, alpha >= 1/255: 4e-3 RELATIVE, however small T has become)
    Th, Tr = d["final_T"].reshape(H, W), fr.stages["final_T"].reshape(H, W)
    same = (d["n_contrib"].reshape(H, W) == fr.stages["n_contrib"].reshape(H, W)) & (np.abs(Th - Tr) <= 1e-3 * np.maximum(Tr, 1e-30))
    scale = max(1.0, float(np.abs(fr.color).max()))
    e_img = float(np.abs(col - fr.color)[:, same].max() / scale) if same.any() else 0.0
    dep_diff = int((st.depth.cpu().numpy()[0][same] != fr.depth[0][same]).sum())   # (the median depth's own branch, T > 0.5, is not visible in the final state)
    g_in = sc.dL_dpix * same[None]
    br = r.backward(g_in)
    gr = gsr.backward(st, g_in)
    torch.cuda.synchronize()
    errs = {n: rel_err(getattr(gr, n).cpu().numpy(), getattr(br, n)) for n in GRADS if getattr(br, n).size}
    cond = None
    if errs and max(errs.values()) > 1e-4:
        # how far is each side from the EXACT value of the reference's formulas here (the CPU oracle with the per-pixel state of backward.cu:470-530 in double:
        # tests/test_gpu_fuzz.py's protocol), and how far is the reference from itself (its float atomics land in any order)?
        from oracle import oracle
        o, fo = oracle.forward_scene(sc, omp=True)
        ex = o.backward(g_in, accum_double=3)
        br2 = r.backward(g_in)
        n = max(errs, key=errs.get)
        cond = {"tensor": n, "hip_vs_exact": rel_err(getattr(gr, n).cpu().numpy(), getattr(ex, n)), "reference_vs_exact": rel_err(getattr(br, n), getattr(ex, n)),
                "reference_vs_its_own_second_run": rel_err(getattr(br2, n), getattr(br, n))}
    return dict(ill_conditioned=cond, it=c["it"], W=W, H=H, P=c["P"], mult=c["mult"], mode=c["mode"], R=int(fr.num_rendered), image=e_img, depth_pixels_differ=dep_diff,
                branch_pixels=float((~same).mean()), worst_grad=max(errs.values()) if errs else 0.0, worst_name=max(errs, key=errs.get) if errs else "")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    gsr = load_package(); syn = gsr.synthetic
    t0 = time.time()
    rows = [run_case(gsr, syn, c, seed) for c in configs(n, seed)]
    bad = [x for x in rows if x["image"] > 1e-4 or x["worst_grad"] > 1e-4 or x["depth_pixels_differ"] > 2]
    out = {"what": "HIP library against the reference's own kernels (oracle/_ref/libgsr_ref.so) on random frames: index stages and projected geometry bit-exact in "
                   "every frame (asserted); below, images / gradients on the pixels where both renders took the same branches",
           "frames": n, "seed": seed, "seconds": round(time.time() - t0, 1), "tile_instances_total": int(sum(x["R"] for x in rows)),
           "image_worst": max(x["image"] for x in rows), "gradient_worst": max(x["worst_grad"] for x in rows),
           "branch_pixel_fraction_mean_max": [float(np.mean([x["branch_pixels"] for x in rows])), max(x["branch_pixels"] for x in rows)],
           "frames_beyond_1e-4": bad, "ten_worst_gradients": sorted(rows, key=lambda x: -x["worst_grad"])[:10]}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
