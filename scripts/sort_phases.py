"""Phase times of K_tile_sort_cut's workgroups (instrumented build -DGSR_EXP_SORT_PHASES; GSR_LIB_OVERRIDE points at it): per tile, wall-clock
stamps at workgroup start (0), after the partition (1), at the start of the last chunk (3), after the last chunk's sort (4), at the end (5).
    python scripts/sort_phases.py <splats> <camera> <scale_mult>"""
import ctypes as C, json, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
gsr = load_package(); syn = gsr.synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
camn = sys.argv[2] if len(sys.argv) > 2 else "scannet"
mult = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
c = syn.make_camera(**syn.CAMERAS[camn]); sc = syn.make_scene(P, c, seed=0, scale_mult=mult)
s = gsr.capi.Settings.from_camera(c)
for _ in range(10):
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
torch.cuda.synchronize()
T = ((c.width + 15) // 16) * ((c.height + 15) // 16)
L = gsr.capi.lib(); buf = (C.c_ulonglong * (16 * T))(); L.gsr_debug_sort_phases.argtypes = [C.c_void_p, C.c_int]; assert L.gsr_debug_sort_phases(buf, 16 * T) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(T, 16).astype(np.int64)
t = (a[:, :6] - a[:, 0].min()) / 100.0
tp = (a[:, 7:11] - a[:, 0].min()) / 100.0   # inside the partition: after min/max, after the count, after the scan, before the scatter
n = a[:, 6]
long_ = n > 1024
out = {"tiles": T, "long_lists": int(long_.sum()), "mean_list": float(n.mean()), "kernel_us": float(t[:, 5].max()),
       "start_us_p50_p99": [float(np.percentile(t[:, 0], q)) for q in (50, 99)]}
if long_.any():
    l = t[long_]
    out["long"] = {"workgroup_us": float((l[:, 5] - l[:, 0]).mean()), "partition_us": float((l[:, 1] - l[:, 0]).mean()),
                   "partition_parts_us": {"load_minmax": float((tp[long_][:, 0] - l[:, 0]).mean()), "count": float((tp[long_][:, 1] - tp[long_][:, 0]).mean()),
                                          "scan": float((tp[long_][:, 2] - tp[long_][:, 1]).mean()), "cursors_or_equalise": float((tp[long_][:, 3] - tp[long_][:, 2]).mean()),
                                          "scatter": float((l[:, 1] - tp[long_][:, 3]).mean())},
                   "chunks_before_last_us": float((l[:, 3] - l[:, 1]).mean()), "last_chunk_sort_us": float((l[:, 4] - l[:, 3]).mean()),
                   "last_chunk_emit_us": float((l[:, 5] - l[:, 4]).mean())}
if (~long_ & (n > 0)).any():
    sh_ = t[~long_ & (n > 0)]
    out["short"] = {"workgroup_us": float((sh_[:, 5] - sh_[:, 0]).mean()), "sort_us": float((sh_[:, 4] - sh_[:, 0]).mean()), "emit_us": float((sh_[:, 5] - sh_[:, 4]).mean())}
print(json.dumps(out, indent=1))
