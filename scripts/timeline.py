"""Timeline of the backward blend's workgroups (instrumented build -DGSR_EXP_TIMELINE; GSR_LIB_OVERRIDE points at it):
start / end time, XCD / CU / SIMD of every workgroup of the last launch -> job length distribution, residency over time, tail.
    hipcc ... -DGSR_EXP_TIMELINE -o build/libgsr_timeline.so gsr_api.hip ; GSR_LIB_OVERRIDE=build/libgsr_timeline.so python scripts/timeline.py"""
import ctypes as C, json, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
gsr = load_package(); syn = gsr.synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
c = syn.make_camera(**syn.REPLICA); sc = syn.make_scene(P, c, seed=0)
s = gsr.capi.Settings.from_camera(c)
for _ in range(30):
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    gsr.backward(st, sc.dL_dpix)
torch.cuda.synchronize()
L = gsr.capi.lib(); n = 4 * 4 * ((c.width + 15) // 16) * ((c.height + 15) // 16)
buf = (C.c_ulonglong * n)(); L.gsr_debug_timeline.argtypes = [C.c_void_p, C.c_int]; assert L.gsr_debug_timeline(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
t0 = (a[:, 0] - a[:, 0].min()) / 100.0; t1 = (a[:, 1] - a[:, 0].min()) / 100.0   # us (100 MHz clock)
hw = a[:, 2]; xcc = a[:, 3] & 0xF; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
d = t1 - t0; total = t1.max()
out = {"workgroups": int(len(d)), "kernel_us": float(total), "job_us": {k: float(np.percentile(d, q)) for k, q in (("p1", 1), ("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))},
       "job_us_mean": float(d.mean()), "slot_time_sum_us": float(d.sum()), "mean_residency_waves": float(d.sum() / total)}
edges = np.arange(0, total + 10, 10.0); res = []
for lo in edges[:-1]:
    hi = lo + 10; res.append(float((np.minimum(t1, hi) - np.maximum(t0, lo)).clip(0).sum() / 10.0))
out["resident_waves_per_10us"] = [round(x) for x in res]
order = np.argsort(t0); out["start_us_of_block_quantiles"] = [float(np.percentile(t0, q)) for q in (0, 25, 50, 75, 90, 99, 100)]
out["last_50_finishers_job_us_mean"] = float(d[np.argsort(t1)[-50:]].mean())
key = xcc * 1000 + se * 100 + sh * 50 + cu; out["distinct_cus_seen"] = int(len(np.unique(key)))
per_xcd = [float(t1[xcc == x].max()) for x in range(8) if (xcc == x).any()]; out["xcd_finish_us"] = per_xcd
first = d[order[:3000]]; out["first_round_job_us_mean"] = float(first.mean())
# raw per-job data for offline scheduling simulations (scripts/lpt_sim.py): block id -> (start, end, xcd), and the quad's record counts
def al(x): return (x + 255) & ~255
N = c.width * c.height; T = ((c.width + 15) // 16) * ((c.height + 15) // 16)
off = al(N * 4); off = al(off + N * 4); off = al(off + T * 8); off = al(off + T * 4); off = al(off + T * 4); off = al(off + T * 512 * 4)
qcount = st.image[off:off + T * 16].view(torch.int32).cpu().numpy(); off = al(off + T * 16)
qdone = st.image[off:off + T * 16].view(torch.int32).cpu().numpy()
os.makedirs(R + "/gpurun_out", exist_ok=True)
np.savez_compressed(R + "/gpurun_out/timeline_%s.npz" % (sys.argv[2] if len(sys.argv) > 2 else "bwd"), t0=t0[:4 * T], t1=t1[:4 * T], xcc=xcc[:4 * T], simd=simd[:4 * T], cu=key[:4 * T], qcount=qcount, qdone=qdone)
print(json.dumps(out, indent=1))
