#!/usr/bin/env python3
"""The sharded C++ loop at one rank against the SAME loop unsharded on the SAME scene (bench.py's shard_step scene): ms per MapFrame /
Track iteration, three repeats each. usage: r05_shard_ab.py [splats]   (GPU box)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gsorb-slam_amd"))
import importlib
gsr = importlib.import_module("gsorb-slam_amd")
from diff_gaussian_rasterization import _C
syn = gsr.synthetic
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
camd = syn.CAMERAS["replica"]; cam = syn.make_camera(**camd); W, H = cam.width, cam.height
sc = syn.make_scene(P, cam, seed=1234)
t = lambda x: torch.tensor(x, dtype=torch.float32)
op = t(sc.opacities).reshape(-1, 1)
raw = [t(sc.means3D), t(sc.colors), t(sc.rotations), torch.log(op / (1 - op)), torch.log(t(sc.scales))]
T = torch.eye(4, device=dev)
T0 = T.clone(); T0[:3, 3] = torch.tensor([0.004, -0.003, 0.005], device=dev)
import socket
import torch.distributed as td
sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
td.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
def make(shard):
    loop = _C.SlamLoop(W, H, camd["fx"], camd["fy"], dev)
    loop.set_map(*raw)
    if shard: loop.set_shard(td.group.WORLD if shard == 2 else None, 0, 1, torch.zeros(0, 4))
    return loop
ref = make(1)
rgb, sur, _ = ref.render_composite(T)
rgb, depth = (rgb * 0.9 + 0.05).contiguous(), sur[0].contiguous()
host = []
def timed(fn, k=20):
    fn(3); torch.cuda.synchronize(); t0 = time.perf_counter(); n = fn(k); t1 = time.perf_counter(); torch.cuda.synchronize()
    host.append(round((t1 - t0) / n * 1e3, 4))
    return (time.perf_counter() - t0) / n * 1e3
for name, shard in (("unsharded", 0), ("sharded x1", 1), ("sharded x1 rccl", 2), ("unsharded", 0), ("sharded x1", 1), ("sharded x1 rccl", 2)):
    loop = make(shard)
    m = [timed(lambda k: len(loop.map_frame(rgb, depth, T, k))) for _ in range(3)]
    tr = [timed(lambda k: len(loop.track(rgb, depth, T0, k)[0])) for _ in range(3)]
    print(f"{name:12s} mapping {min(m):.4f} ms  tracking {min(tr):.4f} ms   (all: {[round(x, 4) for x in m]} {[round(x, 4) for x in tr]}) host-side call {host[-6:]}", flush=True)
