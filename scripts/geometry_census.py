"""Patch-geometry census of the backward blend, on the CPU: for the bench workloads, what the kernel's round
structure (64 records per round, one list per patch, max-over-rows iterations, a reduce phase every 64/rows iterations) costs
with patches of 4x4 pixels (4 rows of 16 lanes: the shipped kernel), 4x2 / 2x4 (8 rows of 8 lanes), 2x2 (16 rows of 4 lanes)
and 8x2 / 8x4 / 8x8 for reference. Counting is oracle/gsr_oracle.c:gsro_geometry_census; the VALU model next to it:
    instructions per launch = 33 x wave iterations + R(geometry) x reduce phases + 240 x rounds
  33  = the compiled blend-loop body per wave iteration
  240 = per-round fixed work (gather, list building, staging, flush) — fitted: (107.4 M - 33 x 1.96 M - 185 x 155 k) / 58 126 rounds
  R   = reduce phase: ring reads, moments of the pw x ph patch through its column / row sums, the colour sums (3 per pixel),
        staging, and the collect step in which lane e gathers entry e's sums from every row (9 FMAs + 1 select per row):
        4x4: 71 + 48 + 26 + 40 = 185 (measured build);  4x2 / 2x4: 40 + 24 + 26 + 80 = 170;  2x2: 22 + 12 + 26 + 160 = 220;
        8x2 (4 rows): 185;  8x4 (2 rows): 71*2 + 96 + 26 + 20 = 284;  8x8 (1 row): 600
    python scripts/geometry_census.py [out.json]     (about ten minutes on 8 cores)"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
from oracle import oracle
gsr = load_package(); syn = gsr.synthetic
GEOMS = [(4, 4), (4, 2), (2, 4), (2, 2), (8, 2), (8, 4), (8, 8)]
RCOST = {(4, 4): 185, (4, 2): 170, (2, 4): 170, (2, 2): 220, (8, 2): 185, (8, 4): 284, (8, 8): 600}
out = {}
for name, P, cam, mult in (("headline 1M replica", 1_000_000, syn.REPLICA, 1.0), ("fat x4 1M replica", 1_000_000, syn.REPLICA, 4.0),
                           ("scannet 2M", 2_000_000, syn.CAMERAS["scannet"], 1.0)):
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    t0 = time.time()
    c = syn.make_camera(**cam); sc = syn.make_scene(P, c, seed=0, scale_mult=mult)
    o = oracle.Oracle(omp=True)
    o.forward(copy_stages=False, means3D=sc.means3D, opacities=sc.opacities, cam=sc.cam, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    rows = o.geometry_census(GEOMS)
    base = None
    out[name] = {}
    for gm, d in zip(GEOMS, rows):
        d["rows"] = 64 // (gm[0] * gm[1])
        d["row_fill"] = d["patch_hits"] / max(d["wave_iterations"] * d["rows"], 1)
        d["useful_lane_frac_of_loop"] = d["blended_pairs"] / max(d["lane_slots"], 1)
        d["valu_model"] = 33 * d["wave_iterations"] + RCOST[gm] * d["reduce_phases"] + 240 * d["rounds"]
        base = base or d["valu_model"]
        d["valu_model_vs_4x4"] = d["valu_model"] / base
        out[name]["%dx%d" % gm] = d
        print("%-20s %dx%d rows %2d: quad hits %9d patch hits %10d wave its %9d (fill %.2f) reduce phases %8d rounds %7d useful %.3f  VALU model %.1f M (%.2f)"
              % (name, gm[0], gm[1], d["rows"], d["quad_hits"], d["patch_hits"], d["wave_iterations"], d["row_fill"], d["reduce_phases"], d["rounds"],
                 d["useful_lane_frac_of_loop"], d["valu_model"] / 1e6, d["valu_model_vs_4x4"]), flush=True)
    print("  (%.0f s)" % (time.time() - t0), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
