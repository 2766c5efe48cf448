"""Timeline of the backward blend's workgroups on ONE RANK'S CELL of a sharded map (instrumented build -DGSR_EXP_TIMELINE; GSR_LIB_OVERRIDE points at it): where the
active jobs (quads with records) land — XCD, CU, SIMD —, how many run side by side, how long they take.
    GSR_LIB_OVERRIDE=build/libgsr_timeline.so python scripts/timeline_rank.py replica 1000000 4 0"""
import ctypes as C, json, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
gsr = load_package(); syn = gsr.synthetic
sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
camera, total, cells, cell = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
c = syn.make_camera(**syn.CAMERAS[camera]); sc = syn.make_scene(total, c, seed=1234)
t = lambda x: torch.tensor(x, dtype=torch.float32)
part = sharded.KdPartition.build(t(sc.means3D), cells)
idx = np.nonzero(part.assign(t(sc.means3D)).numpy() == cell)[0]
s = gsr.capi.Settings.from_camera(c)
for _ in range(60):
    st = gsr.forward(s, sc.means3D[idx], sc.opacities[idx], colors=sc.colors[idx], scales=sc.scales[idx], rotations=sc.rotations[idx])
    gsr.backward(st, sc.dL_dpix)
torch.cuda.synchronize()
L = gsr.capi.lib(); T = ((c.width + 15) // 16) * ((c.height + 15) // 16); n = 4 * 4 * T
buf = (C.c_ulonglong * n)(); L.gsr_debug_timeline.argtypes = [C.c_void_p, C.c_int]; assert L.gsr_debug_timeline(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)[:4 * T]
t0 = (a[:, 0] - a[:, 0].min()) / 100.0; t1 = (a[:, 1] - a[:, 0].min()) / 100.0
hw = a[:, 2]; xcc = a[:, 3] & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
d = t1 - t0
act = d > 5.0                       # (an empty job leaves within a few us)
key = xcc * 1000 + se * 100 + sh * 50 + cu
out = {"workgroups": int(len(d)), "active_jobs": int(act.sum()), "kernel_us": float(t1.max()), "active_job_us_p10_p50_p90_max": [float(np.percentile(d[act], q)) for q in (10, 50, 90, 100)],
       "empty_job_us_p50_p99": [float(np.percentile(d[~act], q)) for q in (50, 99)],
       "active_jobs_per_xcd": [int((act & (xcc == x)).sum()) for x in range(8)], "xcd_finish_us": [float(t1[xcc == x].max()) for x in range(8)],
       "cus_with_active_jobs": int(len(np.unique(key[act]))), "active_jobs_per_cu_p50_max": [float(np.percentile(np.unique(key[act], return_counts=True)[1], q)) for q in (50, 100)],
       "first_active_start_us_p50_p90_max": [float(np.percentile(t0[act], q)) for q in (50, 90, 100)]}
# concurrency on the busiest SIMD
ks = key[act] * 4 + simd[act]
u, cnt = np.unique(ks, return_counts=True)
out["active_jobs_per_simd_p50_p90_max"] = [float(np.percentile(cnt, q)) for q in (50, 90, 100)]
out["simds_with_active_jobs"] = int(len(u))
edges = np.arange(0, t1.max() + 10, 10.0)
out["active_resident_waves_per_10us"] = [int(round(float((np.minimum(t1[act], lo + 10) - np.maximum(t0[act], lo)).clip(0).sum() / 10.0))) for lo in edges[:-1]]
print(json.dumps(out, indent=1))
