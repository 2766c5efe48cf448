"""Row-fill of the backward blend's padded per-patch lists and of its 16-slot reduce blocks, from an
instrumented build (-DGSR_EXP_ROWFILL: five counters in the geometry header; GSR_LIB_OVERRIDE points at it):
    hipcc ... -DGSR_EXP_ROWFILL -o build/libgsr_rowfill.so gsr_api.hip ; GSR_LIB_OVERRIDE=build/libgsr_rowfill.so python scripts/rowfill.py"""
import json, os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
gsr = load_package(); syn = gsr.synthetic
out = {}
for name, P, cam, mult in (("headline 1M replica", 1_000_000, syn.REPLICA, 1.0), ("fat x4 1M replica", 1_000_000, syn.REPLICA, 4.0),
                           ("scannet 2M", 2_000_000, syn.CAMERAS["scannet"], 1.0)):
    c = syn.make_camera(**cam); sc = syn.make_scene(P, c, seed=0, scale_mult=mult)
    s = gsr.capi.Settings.from_camera(c)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    gsr.backward(st, sc.dL_dpix); torch.cuda.synchronize()
    h = st.geom[:256].view(torch.int32).cpu().numpy().astype("int64") & 0xFFFFFFFF
    work, slots, rounds, reduces, parked = (int(x) for x in h[3:8])
    out[name] = {"tile_instances": st.num_rendered, "quad_hits_walked": parked, "patch_hits_walked": work, "row_iterations_run": slots,
                 "row_fill": work / max(slots, 1), "rounds": rounds, "reduce_phases": reduces,
                 "reduce_block_fill": slots / 4 / max(16 * reduces, 1), "iterations_per_round": slots / 4 / max(rounds, 1)}
print(json.dumps(out, indent=1))
