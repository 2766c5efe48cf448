"""The multi-process exchange paths of gsorb-slam_amd/sharded.py on the HIP operator.

Two processes, one rank each, driving the PRODUCT backends (HipBandBackend over the C ABI, the Python operator
inside ShardedMapper) with real collectives:
  * backend "gloo", both ranks on cuda:0 — runs on the one-GPU test box (device tensors are staged through the
    host for the gloo all-gather, sharded._all_gather);
  * backend "nccl" (= RCCL over xGMI), rank r on cuda:r — skipped unless two GPUs are visible.
Scheme A (tile bands): the gathered frame must be bit-identical to the one-process C-ABI frame, the gradients
equal up to float summation order. Scheme B (scene shards): the all-reduced pose gradient must equal the
gradient of the same two-layer composite differentiated in one process, and one mapping-loss backward on the
shards must match the rows of the in-process composite's gradients.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, FX, FY = 320, 240, 260.0, 258.0
NAMES = ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales", "dL_drotations")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    gsr = load_package()
    hz = __import__("gsorb_slam_amd.harness", fromlist=["x"])
    sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
    return gsr, hz, sharded


def _scene(gsr):
    syn = gsr.synthetic
    cam = syn.make_camera(W, H, FX, FY)
    return cam, syn.make_scene(6000, cam, seed=17, scale_mult=1.5)


def _fill(hz, sc, idx, device):
    g = hz.GaussianMap(hz.Config(), FX, FY, device=device)
    g.add_points(torch.tensor(sc.means3D[idx]), torch.tensor(sc.colors[idx]))
    op = torch.tensor(sc.opacities[idx])
    with torch.no_grad():
        g.log_scales.copy_(torch.log(torch.tensor(sc.scales[idx])))
        g.unnorm_quat.copy_(torch.tensor(sc.rotations[idx]))
        g.logit_opacities.copy_(torch.log(op / (1 - op)))
    return g


def _pose():
    from util import pose
    return torch.tensor(pose(0.02, (0.01, -0.01, 0.015)), dtype=torch.float32)


def _target(hz, sc, dev):
    g = _fill(hz, sc, np.arange(sc.P), dev)
    with torch.no_grad():
        g.rgb.mul_(0.8).add_(0.1)
        r = hz.SlamRenderer(g, W, H)
        T = _pose().to(dev)
        rgb, sur, _ = r.render_rgb(T, tracking=True)
    return hz.Frame(rgb.clone(), sur[0].clone(), T)


def _smooth_loss(r, frame, Tcw):
    rimage, _, rdepth = r.render_pair(Tcw, tracking=True)
    return ((rimage - frame.rgb) ** 2).sum() + ((rdepth[0] - frame.depth * rdepth[1].detach()) ** 2).sum()


def _worker(rank, world, port, backend, q):
    gsr, hz, sharded = _setup()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        capi = gsr.capi
        cam, sc = _scene(gsr)
        t = lambda a: torch.tensor(a, device=dev)
        res = {}
        # ---- scheme A: tile bands through the C ABI
        r = sharded.TileBandRenderer(sharded.HipBandBackend(capi))
        s = capi.Settings.from_camera(cam, dev)
        color, depth, st = r.forward(s, H, W, dev, means3D=t(sc.means3D), opacities=t(sc.opacities), colors=t(sc.colors),
                                     scales=t(sc.scales), rotations=t(sc.rotations))
        grads = r.backward(st, t(sc.dL_dpix))
        res.update(color=color.cpu().numpy(), depth=depth.cpu().numpy(), bands=r.bands(H), R=st.num_rendered,
                   grads={n: getattr(grads, n).cpu().numpy() for n in NAMES})
        # ---- scheme B: scene shards through the Python operator
        T = _pose().to(dev)
        zc = t(sc.means3D) @ T[2, :3] + T[2, 3]
        slabs = [x.cpu().numpy() for x in sharded.shard_by_depth_slabs(zc, world)]
        frame = _target(hz, sc, dev)
        Mapper = sharded.make_sharded_mapper(hz)
        g = _fill(hz, sc, slabs[rank], dev)
        m = Mapper(g, W, H)
        Tp = T.clone().requires_grad_(True)
        loss = _smooth_loss(m, frame, Tp)
        loss.backward()
        own = Tp.grad.clone()
        total = m.comp.all_reduce_pose_grad(Tp.grad.clone())
        ml = m.mapping_loss(frame)
        ml.backward()
        res.update(pose_own=own.cpu().numpy(), pose_sum=total.cpu().numpy(), loss=float(loss.detach()), map_loss=float(ml.detach()),
                   map_grads={n: getattr(g, n).grad.cpu().numpy() for n in g.NAMES}, idx=slabs[rank],
                   world=dist.get_world_size(), backend=dist.get_backend())
        torch.cuda.synchronize()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _reference(gsr, hz, sharded, world):
    """One process, one GPU: the C-ABI frame + gradients, and the two-layer composite with every layer differentiable."""
    capi = gsr.capi
    dev = torch.device("cuda:0")
    cam, sc = _scene(gsr)
    t = lambda a: torch.tensor(a, device=dev)
    s = capi.Settings.from_camera(cam, dev)
    full = capi.forward(s, means3D=t(sc.means3D), opacities=t(sc.opacities), colors=t(sc.colors), scales=t(sc.scales),
                        rotations=t(sc.rotations))
    gfull = capi.backward(full, t(sc.dL_dpix))
    ref = dict(color=full.color.cpu().numpy(), depth=full.depth.cpu().numpy(), R=full.num_rendered,
               grads={n: getattr(gfull, n).cpu().numpy() for n in NAMES})
    T = _pose().to(dev)
    zc = t(sc.means3D) @ T[2, :3] + T[2, 3]
    slabs = [x.cpu().numpy() for x in sharded.shard_by_depth_slabs(zc, world)]
    frame = _target(hz, sc, dev)

    def composite(Tp, maps, tracking):
        layers = []
        for g in maps:
            r = hz.SlamRenderer(g, W, H)
            rimage, _, rdepth = r.render_pair(Tp, tracking=tracking)
            layers.append(torch.cat([rimage, rdepth[0:2]], 0))
        Tr = torch.ones_like(layers[0][0:1])
        out = torch.zeros_like(layers[0][0:4])
        for L in layers:
            out = out + Tr * L[0:4]
            Tr = Tr * (1.0 - L[4:5])
        return out, 1.0 - Tr

    maps = [_fill(hz, sc, idx, dev) for idx in slabs]
    Tp = T.clone().requires_grad_(True)
    out, sil = composite(Tp, maps, True)
    loss = ((out[0:3] - frame.rgb) ** 2).sum() + ((out[3] - frame.depth * sil[0].detach()) ** 2).sum()
    loss.backward()
    ref.update(pose=Tp.grad.cpu().numpy(), loss=float(loss.detach()))
    return ref


def _run(backend, world=2):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    got, t0 = {}, time.time()
    while len(got) < world:                                   # (never wait for a rank that died)
        try:
            r, res = q.get(timeout=2)
            got[r] = res
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs) or time.time() - t0 > 300:
                for p in procs:
                    p.kill()
                raise AssertionError("ranks died or hung: exit codes %s after %.0f s" % ([p.exitcode for p in procs], time.time() - t0))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    gsr, hz, sharded = _setup()
    ref = _reference(gsr, hz, sharded, world)
    scale = lambda a: np.abs(a).max() + 1e-30
    assert got[0]["world"] == world and got[0]["backend"] == backend
    # ---- scheme A
    assert got[0]["bands"] == [(0, 7), (7, 15)] or sum(b - a for a, b in got[0]["bands"]) == 15
    assert sum(got[r]["R"] for r in range(world)) == ref["R"]          # the bands partition the (splat, tile) pairs
    for r in range(world):
        np.testing.assert_array_equal(got[r]["color"], ref["color"])     # bit-exact frame on every rank
        np.testing.assert_array_equal(got[r]["depth"], ref["depth"])
        for n in NAMES:
            e = np.abs(got[r]["grads"][n] - ref["grads"][n]).max() / scale(ref["grads"][n])
            assert e < 1e-5, (n, e)
    for n in NAMES:
        np.testing.assert_array_equal(got[0]["grads"][n], got[world - 1]["grads"][n])   # identical replicas after the all-reduce
    # ---- scheme B
    own0, own1 = got[0]["pose_own"], got[world - 1]["pose_own"]
    assert world == 1 or np.abs(own0 - own1).max() > 1e-3 * scale(own0)
    np.testing.assert_array_equal(got[0]["pose_sum"], got[world - 1]["pose_sum"])
    e = np.abs(got[0]["pose_sum"] - ref["pose"]).max() / scale(ref["pose"])
    assert e < 1e-4, e                      # float atomics: summation order differs between runs
    assert abs(got[0]["loss"] - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    assert abs(got[0]["map_loss"] - got[world - 1]["map_loss"]) <= 1e-6 * abs(got[0]["map_loss"])
    for r in range(world):
        for n, g in got[r]["map_grads"].items():
            assert np.isfinite(g).all() and np.abs(g).max() > 0, n
    print("\n%s world %d: frame bit-exact, pose-gradient sum vs one-process composite %.1e" % (backend, world, e))


@pytest.mark.gpu
def test_two_processes_on_one_gpu_gloo_drive_the_hip_backends():
    _run("gloo")


@pytest.mark.gpu
def test_two_gpus_rccl_drive_the_hip_backends():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL); the same path runs over gloo on one GPU in the test above")
    _run("nccl")


@pytest.mark.gpu
def test_bench_runs_its_multi_rank_branch_with_two_ranks():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process), here with both ranks on
    whatever GPUs exist and gloo instead of RCCL (GSR_BENCH_BACKEND): rendezvous, barriers, the max-over-ranks timing, the
    strong-scaling `shard_render` headline (k-d cells, the band exchange, pose-gradient all-reduce), the
    collective-free replica figure beside it, and `shard_step` all execute; one JSON line from rank 0."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GSR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--splats", "40000", "--camera", "tum",
           "--steps", "5", "--warmup", "2", "--shard-steps", "3"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["value"] > 0 and d["scaling"] == "strong"
    assert "x2" in d["config"]["parallelism"] and "cpu_baseline" not in d and "parity" not in d
    sr = d["shard_render"]
    assert sr["splats_per_rank"] == 20000 and sr["total_splats"] == 40000 and d["value"] == sr["value"] and d["ms_per_step"] == sr["ms_per_step"]
    # (round 6, the band exchange: rank 0 sends the peer's band — 240 rows — of its six layer planes, and the peer's five gradient planes on its own band)
    assert sr["collective_bytes_per_rank_per_step"]["p2p_forward_sent"] == 6 * 240 * 640 * 4 and sr["collective_bytes_per_rank_per_step"]["p2p_backward_sent"] == 5 * 240 * 640 * 4
    assert d["replica_rasterize"]["scaling"] == "weak" and d["replica_rasterize"]["value"] > 0
    wk = d["shard_render_weak"]   # the weak-scaling scheme-B figure beside the strong headline: --splats per rank, the map grows with the ranks
    assert wk["scaling"] == "weak" and wk["splats_per_rank"] == 40000 and wk["total_splats"] == 80000 and wk["ranks"] == 2 and wk["value"] > 0
    assert sr["ranks"] == 2 and "gloo" in d["config"]["parallelism"]
    ss = d["shard_step"]
    assert ss["rccl_ranks"] == 2 and ss["splats_per_rank"] == 20000 and ss["mapping_ms_per_iter"] > 0 and ss["tracking_ms_per_iter"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world,has_sur", [(2, True), (3, True), (5, False), (8, True)])
def test_fused_compositing_kernels_match_the_dense_composite(world, has_sur):
    """csrc/gsr_shard.h (gsr_composite_forward / _backward_local / _backward_occlusion) for every rank of a simulated world, against
    float64 autograd through the dense "over" composite of all layers in one process: the sum of the ranks' contributions is the
    composite, every rank's (dL/dlayer, dL/dS) is its slice of the dense gradient, the surface-depth pick follows the rule."""
    gsr, hz, sharded = _setup()
    g = torch.Generator().manual_seed(world * 7 + has_sur)
    Hh, Ww = 37, 53
    L = torch.rand((world, 4, Hh, Ww), generator=g)
    S = torch.rand((world, 1, Hh, Ww), generator=g) * 0.95
    S[:, :, :5] = 0.0                                            # rows nothing covers
    S[0, :, 5:9] = 1.0                                           # an opaque layer
    SU = torch.where(torch.rand((world, 1, Hh, Ww), generator=g) < 0.7, 0.5 + 3 * torch.rand((world, 1, Hh, Ww), generator=g), torch.zeros(1))
    keys = torch.rand((world,), generator=g) * 5
    keys[-1] = keys[0]                                           # a tie: broken by rank (stable sort)
    order = torch.argsort(keys.double(), stable=True)
    G4 = torch.randn((4, Hh, Ww), generator=g); Gs = torch.randn((1, Hh, Ww), generator=g)
    # dense reference in float64
    Ld, Sd = L.double().requires_grad_(True), S.double().requires_grad_(True)
    T = torch.ones((1, Hh, Ww), dtype=torch.float64)
    out = torch.zeros((4, Hh, Ww), dtype=torch.float64)
    surf_ref = torch.zeros((1, Hh, Ww)); found = torch.zeros((1, Hh, Ww), dtype=torch.bool)
    for k in order.tolist():
        out = out + T * Ld[k]
        T = T * (1 - Sd[k])
        has = SU[k] > 0
        surf_ref = torch.where(~found & has, SU[k], surf_ref)
        found = found | (has & (T.detach() <= 0.5))
    sil = 1 - T
    ((out * G4.double()).sum() + (sil * Gs.double()).sum()).backward()
    # the kernels, rank by rank
    pad = torch.zeros((world, 1, Hh, Ww)); pad[:, 0, 0, 0] = keys
    gathered = torch.cat([S, SU if has_sur else torch.zeros_like(S), pad], 1).cuda().contiguous()
    order_d = order.cuda()
    contribs, locals_ = [], []
    for r in range(world):
        contrib, sil_tot, surf = gsr.capi.composite_forward(world, r, order_d, gathered, L[r].cuda().contiguous(), has_sur)
        contribs.append(contrib)
        assert (sil_tot.cpu().double() - sil.detach()).abs().max() < 2e-6
        if has_sur:
            assert torch.equal(surf.cpu(), surf_ref)
        else:
            assert float(surf.abs().max()) == 0.0
        locals_.append(gsr.capi.composite_backward_local(world, r, order_d, gathered, L[r].cuda().contiguous(), G4.cuda()))
    assert (torch.stack(contribs).sum(0).cpu().double() - out.detach()).abs().max() < 5e-6
    c_all = torch.stack([c for _, c in locals_]).contiguous()
    for r in range(world):
        assert (locals_[r][0].cpu().double() - Ld.grad[r]).abs().max() < 5e-6
        dS = gsr.capi.composite_backward_occlusion(world, r, order_d, gathered, c_all, Gs.cuda())
        assert (dS.cpu().double() - Sd.grad[r]).abs().max() < 2e-5 * max(1.0, float(Sd.grad[r].abs().max()))
    # no upstream gradient on the colours / on the silhouette
    d0, c0 = gsr.capi.composite_backward_local(world, 0, order_d, gathered, L[0].cuda().contiguous(), None)
    assert float(d0.abs().max()) == 0.0 and float(c0.abs().max()) == 0.0
    dS0 = gsr.capi.composite_backward_occlusion(world, 0, order_d, gathered, c_all, None)
    Ld.grad = None; Sd.grad = None


@pytest.mark.gpu
def test_one_rank_rccl_runs_every_collective_of_both_schemes():
    """RCCL itself on the one-GPU box: a process group of ONE rank with backend "nccl". The collectives are trivial, but they are
    RCCL's — device buffers handed to its all-gather / all-reduce on the rank's HIP stream, inside autograd's backward too — i.e.
    the branches of sharded.py (_all_gather, all_reduce_vector, the compositor's exchange) that the gloo legs never take. The frame
    and the gradients must be the single-process ones."""
    _run("nccl", world=1)
