"""Where the time of one (unsharded) tracking iteration of the harness goes at 1 M Gaussians, 1200x680."""
import sys, os, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from conftest import load_package
gsr = load_package(); syn = gsr.synthetic
hz = __import__("gsorb_slam_amd.harness", fromlist=["x"])
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
camd = syn.REPLICA
cam = syn.make_camera(**camd); sc = syn.make_scene(P, cam, seed=1)
g = hz.GaussianMap(hz.Config(), camd["fx"], camd["fy"], device="cuda")
g.add_points(torch.tensor(sc.means3D), torch.tensor(sc.colors))
op = torch.tensor(sc.opacities)
with torch.no_grad():
    g.log_scales.copy_(torch.log(torch.tensor(sc.scales))); g.unnorm_quat.copy_(torch.tensor(sc.rotations)); g.logit_opacities.copy_(torch.log(op / (1 - op)))
r = hz.SlamRenderer(g, cam.width, cam.height)
T = torch.eye(4, device="cuda")
with torch.no_grad():
    rgb, sur, _ = r.render_pair(T, tracking=True)
frame = hz.Frame((rgb * 0.9 + 0.05).clone(), sur[0].clone(), T.clone())
r.track(frame, T, iters=5)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = len(r.track(frame, T, iters=20)[1])
torch.cuda.synchronize(); print("tracking iteration %.2f ms (%d ran)" % ((time.perf_counter() - t0) * 1e3 / n, n))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    n = len(r.track(frame, T, iters=5)[1])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=60))
