"""What is "bit-exact tile / sort indices" worth against a reference binary built with floating-point contraction?

The reference's libCudaRasterizer is built by CMake's CUDA language with default flags (DGR/CMakeLists.txt:22-39: no --fmad=false), so nvcc fuses
a*b+c where it chooses; the oracle, and the HIP build the parity tests hold bit-exact against it, use -ffp-contract=off. This script runs the SAME oracle
source twice — contraction off (libgsr_oracle_omp.so) and on (libgsr_oracle_fma.so: -ffp-contract=fast -mfma) — on the bench scenes and counts what moves:
radii, tiles_touched, num_rendered, the sorted point_list (positions whose splat id differs; list length differences), and pixels of n_contrib / the images.
gcc's contractions are not nvcc's: the size of the effect, not a prediction of which entries move. CPU only (test infrastructure).

    python scripts/fma_census.py [out.json]      # headline 1 M / 1200x680 and the 2 M / 640x480 ScanNet shape
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
from oracle import oracle  # noqa: E402


def census(scene):
    out = {}
    t0 = time.time()
    _, a = oracle.forward_scene(scene, omp=True)
    _, b = oracle.forward_scene(scene, omp="fma")
    P = scene.P
    out["P"], out["width"], out["height"] = int(P), int(scene.cam.width), int(scene.cam.height)
    out["num_rendered_off"], out["num_rendered_fma"] = int(a.num_rendered), int(b.num_rendered)
    out["radii_differ"] = int((a.radii != b.radii).sum())
    out["radii_differ_by_more_than_1"] = int((np.abs(a.radii.astype(np.int64) - b.radii) > 1).sum())
    out["visibility_differs"] = int(((a.radii > 0) != (b.radii > 0)).sum())
    out["tiles_touched_differ"] = int((a.stages["tiles_touched"] != b.stages["tiles_touched"]).sum())
    # (that the contracted build really computes something else: last-bit differences of the per-splat floats)
    out["conic_values_differ"] = int((a.stages["conic_opacity"][:, :3] != b.stages["conic_opacity"][:, :3]).any(1).sum())
    out["means2D_differ"] = int((a.stages["means2D"] != b.stages["means2D"]).any(1).sum())
    out["depths_differ"] = int((a.stages["depths"] != b.stages["depths"]).sum())
    ra, rb = a.stages["ranges"], b.stages["ranges"]
    la, lb = (ra[:, 1] - ra[:, 0]).astype(np.int64), (rb[:, 1] - rb[:, 0]).astype(np.int64)
    out["tiles"] = int(len(la))
    out["tile_lists_of_different_length"] = int((la != lb).sum())
    # per tile: positions of the common prefix length whose splat id differs (an inserted / dropped entry shifts the rest of ITS tile only)
    pa, pb = a.stages["point_list"], b.stages["point_list"]
    moved = tiles_differ = 0
    for t in range(len(la)):
        n = int(min(la[t], lb[t]))
        d = int((pa[ra[t, 0]:ra[t, 0] + n] != pb[rb[t, 0]:rb[t, 0] + n]).sum()) + int(abs(la[t] - lb[t]))
        moved += d
        tiles_differ += d != 0
    out["point_list_positions_that_differ"] = int(moved)
    out["tiles_whose_list_differs"] = int(tiles_differ)
    out["point_list_fraction"] = moved / max(int(a.num_rendered), 1)
    out["n_contrib_pixels_differ"] = int((a.stages["n_contrib"] != b.stages["n_contrib"]).sum())
    out["pixels"] = int(scene.cam.width * scene.cam.height)
    out["color_max_abs_diff"] = float(np.abs(a.color - b.color).max())
    out["color_pixels_beyond_1e-4"] = int((np.abs(a.color - b.color).max(0) > 1e-4).sum())
    out["depth_pixels_differ"] = int((a.depth != b.depth).sum())
    out["seconds"] = round(time.time() - t0, 1)
    return out


def main():
    gsr = load_package()
    syn = gsr.synthetic
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import pose
    res = {"what": "oracle source with -ffp-contract=off vs -ffp-contract=fast -mfma (gcc), same inputs: entries that differ. 'camera-frame' = GSORB-SLAM's C++ call "
                   "pattern (means moved into the camera frame by the caller, identity view matrix: src/Render.cc:750-752 — the products with the matrices' zeros and ones "
                   "are exact, so contraction can only touch the covariance chain); 'world-frame' = the Python replay's pattern (a posed view matrix, scripts/replay.py:91-120)",
           "scenes": {}}
    for name, camera, P in (("replica-1M-1200x680", syn.REPLICA, 1_000_000), ("scannet-2M-640x480", syn.SCANNET, 2_000_000)):
        for frame, Tcw in (("camera-frame", None), ("world-frame", pose(0.3, (0.1, -0.2, 0.3)))):
            cam = syn.make_camera(**camera, Tcw=Tcw)
            r = census(syn.make_scene(P, cam, seed=0))
            res["scenes"][name + " " + frame] = r
            print(name, frame, json.dumps(r))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
