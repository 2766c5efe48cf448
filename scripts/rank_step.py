"""One rank's share of a sharded map: `total` Gaussians cut into `cells` k-d cells, cell `c` over the whole frame — the plain fwd+bwd step,
or the C++ loop's sharded mapping / tracking iterations at one rank (one-rank RCCL group). Meant to run under rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rank -- python scripts/rank_step.py replica 1000000 4 0 step|map|track [iters]"""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + '/tests'); sys.path.insert(0, R + '/gsorb-slam_amd')
from conftest import load_package
gsr = load_package(); syn = gsr.synthetic
sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
camera, total, cells, cell, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 200
camd = syn.CAMERAS[camera]; cam = syn.make_camera(**camd); W, H = cam.width, cam.height
sc = syn.make_scene(total, cam, seed=1234)
t = lambda x: torch.tensor(x, dtype=torch.float32)
part = sharded.KdPartition.build(t(sc.means3D), cells)
idx = np.nonzero(part.assign(t(sc.means3D)).numpy() == cell)[0]
dev = torch.device("cuda", 0)
if mode == "step":
    s = gsr.capi.Settings.from_camera(cam, device=dev)
    c = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
    ins = dict(means3D=c(sc.means3D[idx]), opacities=c(sc.opacities[idx]), colors=c(sc.colors[idx]), shs=None, scales=c(sc.scales[idx]), rotations=c(sc.rotations[idx]), cov3D=None)
    st0 = gsr.forward(s, ins["means3D"], ins["opacities"], colors=ins["colors"], scales=ins["scales"], rotations=ins["rotations"])
    ws = gsr.capi.Workspace(len(idx), W, H, max_rendered=int(st0.num_rendered * 1.25) + 1024, device=dev)
    grads = gsr.capi.alloc_grads(len(idx), 0, dev, intermediates=False); g = c(sc.dL_dpix)
    def step():
        st = gsr.forward_ws(s, ws, ins, None); gsr.backward(st, g, grads=grads, once=True)
    for _ in range(100): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): step()
    torch.cuda.synchronize(); print("step ms %.4f  P=%d R=%d" % ((time.perf_counter() - t0) / iters * 1e3, len(idx), st0.num_rendered))
else:
    import socket, torch.distributed as td
    from diff_gaussian_rasterization import _C
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    td.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    op = t(sc.opacities[idx]).reshape(-1, 1)
    raw = [t(sc.means3D[idx]), t(sc.colors[idx]), t(sc.rotations[idx]), torch.log(op / (1 - op)), torch.log(t(sc.scales[idx]))]
    loop = _C.SlamLoop(W, H, camd["fx"], camd["fy"], dev); loop.set_map(*raw)
    if os.environ.get("GSR_RANK_UNSHARDED") != "1": loop.set_shard(td.group.WORLD, 0, 1, torch.empty(0))
    T = torch.eye(4, device=dev); rgb, sur, _ = loop.render_composite(T); rgb, depth = (rgb * 0.9 + 0.05).contiguous(), sur[0].contiguous()
    T0 = T.clone(); T0[:3, 3] = torch.tensor([0.004, -0.003, 0.005], device=dev)
    fn = (lambda k: len(loop.map_frame(rgb, depth, T, k))) if mode == "map" else (lambda k: len(loop.track(rgb, depth, T0, k)[0]))
    fn(40); torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    for _ in range(max(iters // 20, 1)): n += fn(20)
    torch.cuda.synchronize(); print("%s ms per iteration %.4f over %d  P=%d transport=%s" % (mode, (time.perf_counter() - t0) / n * 1e3, n, len(idx), loop.shard_transport()))
    td.destroy_process_group()
