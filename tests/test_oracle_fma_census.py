"""CPU: what "bit-exact tile / sort indices" is worth against a reference binary built WITH floating-point contraction ("missing" 1).

The reference's libCudaRasterizer.so is built by CMake's CUDA language with default flags (DGR/CMakeLists.txt:22-39: no --fmad=false): nvcc fuses a*b+c where it
chooses. The oracle and the HIP build the parity tests hold bit-exact against it are -ffp-contract=off. oracle/Makefile's third target builds the SAME oracle source
with -ffp-contract=fast -mfma; this test counts what moves between the two on a 300 k-splat Replica frame (scripts/fma_census.py does the 1 M / 2 M bench scenes:
profiles/r06_fma_census.json). It pins the finding DESIGN.md section 2 quotes: under GSORB-SLAM's C++ call pattern (camera-frame means, IDENTITY view matrix,
src/Render.cc:750-752) every product with the matrices' zeros and ones is exact, contraction only reaches the covariance chain, and no radius, tile count or list
entry moves; with a posed view matrix (the Python replay's pattern) a few per million do. Test infrastructure only: the fma build is not a checker."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from util import pose


def test_contraction_census(syn):
    import fma_census
    cam = syn.make_camera(**syn.REPLICA)
    a = fma_census.census(syn.make_scene(300_000, cam, seed=1))
    print("\n  camera-frame (GSORB C++ pattern), 300 k splats: %s" % {k: a[k] for k in (
        "conic_values_differ", "radii_differ", "tiles_touched_differ", "point_list_positions_that_differ", "n_contrib_pixels_differ", "color_pixels_beyond_1e-4")})
    assert a["conic_values_differ"] > 100_000            # the contracted build does compute something else (last bits of most conics)
    assert a["means2D_differ"] == 0 and a["depths_differ"] == 0   # products with the identity view's 0 / 1 entries are exact either way
    assert a["radii_differ"] == 0 and a["tiles_touched_differ"] == 0 and a["num_rendered_off"] == a["num_rendered_fma"]
    assert a["point_list_positions_that_differ"] == 0 and a["n_contrib_pixels_differ"] == 0
    cam = syn.make_camera(**syn.REPLICA, Tcw=pose(0.3, (0.1, -0.2, 0.3)))
    b = fma_census.census(syn.make_scene(300_000, cam, seed=1))
    print("  world-frame (posed view matrix), 300 k splats: %s" % {k: b[k] for k in (
        "means2D_differ", "depths_differ", "radii_differ", "tiles_touched_differ", "point_list_positions_that_differ", "point_list_fraction", "n_contrib_pixels_differ")})
    assert b["depths_differ"] > 10_000                   # here the sort keys themselves move in their last bit
    assert b["radii_differ_by_more_than_1"] == 0 and b["visibility_differs"] <= 2
    assert b["tiles_touched_differ"] <= 30 and b["point_list_fraction"] < 1e-2   # a few per million splats, list entries that swap with a depth neighbour
