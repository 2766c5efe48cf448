"""The sharded C++ loop (torch_ext/DirectLoop.cpp: SlamLoop::SetShard; multi-GPU scheme B) on the HIP kernels, driven through the `_C.SlamLoop`
binding with real process groups:
  * two processes on the one GPU of the test box, backend "gloo" (the loop stages its three collectives through the host there);
  * one rank with backend "nccl": RCCL itself executes the all-gathers / all-reduces on the loop's stream.
Every rank owns one cell of a k-d partition of the map (sharded.KdPartition). Compared with the UNSHARDED C++ loop on the whole map:
the mapping loss curve, the tracking loss curve and the tracked pose, the twelve pose sums of the first tracking iteration (summed over
the ranks by the loop's all-reduce), the composite render (PSNR), and map growth under the owner rule.
Reference: src/Render.cc:402-483, :1054-1126 (the loops), :557-594 (growth); the reference itself is single-GPU.
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
P, MAP_ITERS, TRACK_ITERS = 30_000, 12, 10


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
    sys.path.insert(0, os.path.join(ROOT, "gsorb-slam_amd"))
    from diff_gaussian_rasterization import _C
    sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
    return gsr, _C, sharded


def _scene(gsr):
    syn = gsr.synthetic
    sc = syn.make_scene(P, syn.make_camera(W, H, FX, FY), seed=17, scale_mult=1.5)
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    op = t(sc.opacities).reshape(-1, 1)
    raw = [t(sc.means3D), t(sc.colors), t(sc.rotations), torch.log(op / (1 - op)), torch.log(t(sc.scales))]
    return sc, raw


def _poses():
    from util import pose
    t = lambda a, tr: torch.tensor(pose(a, tr), dtype=torch.float32)
    return t(0.02, (0.01, -0.01, 0.015)), t(0.026, (0.02, -0.004, 0.03))


def _loop(_C, raw, dev=0, **cfg):
    loop = _C.SlamLoop(W, H, FX, FY, torch.device("cuda", dev), **cfg)
    loop.set_map(*raw)
    return loop


def _frame(_C, raw, dev=0):
    """the observation: the whole map with perturbed colours under the true pose"""
    T, _ = _poses()
    obs = [x.clone() for x in raw]
    obs[1] = obs[1] * 0.8 + 0.1
    rgb, sur, _ = _loop(_C, obs, dev).render_composite(T.cuda(dev))
    return rgb.contiguous(), sur[0].contiguous(), T.cuda(dev)


_RANK = {}
def rank_of(loop):
    return _RANK.get(id(loop), 0)


def _schedule(loop, frame, T0, cfg_note, rebalance=None):
    rgb, depth, T = frame
    res = {"note": cfg_note}
    comp = loop.render_composite(T)
    res["render"] = comp[0].cpu().numpy()
    res["map"] = loop.map_frame(rgb, depth, T, MAP_ITERS)
    # (with the feature matches' reprojection term: every rank of a sharded run holds it in full, its gradient enters each rank's rows with 1 / world)
    obs, Xw, s2, cx, cy = _matches(_poses()[0])
    m = (obs.to(T.device), Xw.to(T.device), s2.to(T.device), cx, cy)
    hist, best = loop.track(rgb, depth, T0, 1, *m)
    res["pose_sums"] = loop.last_pose_sums().cpu().numpy()
    res["track1"] = hist
    hist, best = loop.track(rgb, depth, T0, TRACK_ITERS, *m)
    res["track"], res["pose"] = hist, best.cpu().numpy()
    res["added"] = loop.add_gaussians(rgb * 0.0 + 0.9, depth, T)     # a bright frame nobody explains: the dark-pixel rule adds nothing, the silhouette rule might
    res["size"] = loop.size()
    if rebalance is not None:                                         # growth left the cells out of balance: re-split, rows travel with their moments
        before = loop.render_composite(T)[0].cpu().numpy()
        rows0 = loop.export_rows().cpu()
        part = rebalance(loop)
        rows1 = loop.export_rows().cpu()
        after = loop.render_composite(T)[0].cpu().numpy()
        res.update(rebalanced=part is not None, size_rebalanced=loop.size(), psnr_rebalance=_psnr(after, before),
                   rows_checksum=(float(rows0.double().sum()), float(rows1.double().sum())),
                   misplaced=int((part.assign(rows1[:, 0:3]) != rank_of(loop)).sum()) if part is not None else 0,
                   loss_after=loop.map_frame(rgb, depth, T, 2))
    return res


def _worker(rank, world, port, backend, q):
    gsr, _C, sharded = _setup()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = rank if backend == "nccl" and world > 1 else 0               # (gloo: both ranks share the test box's one GPU)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        sc, raw = _scene(gsr)
        part = sharded.KdPartition.build(raw[0], world)
        idx = torch.nonzero(part.assign(raw[0]) == rank).squeeze(-1)
        frame = _frame(_C, raw, dev)
        loop = _loop(_C, [x[idx] for x in raw], dev, fused_update=True)
        loop.set_shard(dist.group.WORLD, rank, world, part.nodes)
        assert loop.shard_transport() == ("rccl" if backend == "nccl" else "staged"), loop.shard_transport()   # (never the silent fallback)
        _RANK[id(loop)] = rank
        res = _schedule(loop, frame, _poses()[1].cuda(dev), "sharded %s world %d" % (backend, world),
                        rebalance=(lambda l: sharded.rebalance_loop(l, rank, world, dist.group.WORLD, tolerance=1.02)) if world > 1 else None)
        res.update(count=int(idx.numel()), order_dev=gsr.capi.shard_order(part.nodes.cuda(), frame[2]).cpu().tolist() if world > 1 else [0],
                   order_cpu=part.order(frame[2]))
        torch.cuda.synchronize()
        q.put((rank, res))
    except Exception:                                                      # (the parent must not wait for a rank that died)
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise
    finally:
        dist.destroy_process_group()


def _reference():
    gsr, _C, sharded = _setup()
    sc, raw = _scene(gsr)
    frame = _frame(_C, raw)
    # (fused_update = False: the unsharded tracking iteration then goes through gsr_pose_step's rows, which LastPoseSums reads)
    return _schedule(_loop(_C, raw, fused_update=False), frame, _poses()[1].cuda(), "unsharded")


def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return 99.0 if mse == 0 else float(10 * np.log10(1.0 / mse))


def _run(backend, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    got, t0 = {}, time.time()
    while len(got) < world:
        try:
            r, res = q.get(timeout=2)
            assert "error" not in res, "rank %d failed:\n%s" % (r, res.get("error"))
            got[r] = res
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > 300:
                for p in procs:
                    p.kill()
                raise AssertionError("ranks died or hung: exit codes %s after %.0f s" % ([p.exitcode for p in procs], time.time() - t0))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _reference()
    scale = lambda a: np.abs(np.asarray(a)).max() + 1e-30
    for r in range(1, world):                                             # replicas: every rank evaluates the same loss on the same composite
        np.testing.assert_allclose(got[r]["map"], got[0]["map"], rtol=1e-6)
        np.testing.assert_allclose(got[r]["track"], got[0]["track"], rtol=1e-6)
        np.testing.assert_allclose(got[r]["pose"], got[0]["pose"], atol=1e-7)
    assert got[0]["order_dev"] == got[0]["order_cpu"] and sorted(got[0]["order_cpu"]) == list(range(world))
    assert sum(got[r]["count"] for r in range(world)) == P
    psnr = _psnr(got[0]["render"], ref["render"])
    e_map = float(np.abs(np.array(got[0]["map"]) - np.array(ref["map"])).max() / scale(ref["map"]))
    e_sum = float(np.abs(got[0]["pose_sums"] - ref["pose_sums"]).max() / scale(ref["pose_sums"]))
    n = min(len(got[0]["track"]), len(ref["track"]))
    e_track = float(np.abs(np.array(got[0]["track"][:n]) - np.array(ref["track"][:n])).max() / scale(ref["track"]))
    e_pose = float(np.abs(got[0]["pose"] - ref["pose"]).max())
    print("\n%s world %d: composite vs one-GPU render %.1f dB; mapping loss curve %.1e (first %.5f / %.5f, last %.5f / %.5f); pose sums %.1e; "
          "tracking loss curve %.1e over %d iterations; tracked pose %.1e; growth %s vs %d" %
          (backend, world, psnr, e_map, got[0]["map"][0], ref["map"][0], got[0]["map"][-1], ref["map"][-1], e_sum, e_track, n, e_pose,
           [got[r]["added"] for r in range(world)], ref["added"]))
    assert len(got[0]["map"]) == MAP_ITERS and got[0]["map"][-1] < got[0]["map"][0]
    if world == 1:                                                         # one layer: the composite IS the render, the exchange changes nothing
        assert psnr >= 90.0 and e_map < 1e-4 and e_sum < 2e-3 and e_track < 1e-3 and e_pose < 1e-4   # (the pose sums: float atomics in the backward, sums that cancel)
    else:                                                                  # cells side by side: exact order, splats that straddle a boundary are what differs
        assert psnr >= 38.0 and e_map < 1e-2 and e_sum < 5e-2 and e_track < 2e-2 and e_pose < 2e-3
        assert n >= 3
    added = sum(got[r]["added"] for r in range(world))
    assert abs(added - ref["added"]) <= 0.02 * ref["added"] + 5
    assert sum(got[r]["size"] for r in range(world)) == P + added
    if world > 1:       # the re-balance: rows move with their moments, none is lost, every Gaussian ends in its owner's cell, the composite is the same map
        sizes = [got[r]["size_rebalanced"] for r in range(world)]
        print("  re-balance: %s -> %s, composite after vs before %.1f dB, two more mapping iterations %s" %
              ([got[r]["size"] for r in range(world)], sizes, got[0]["psnr_rebalance"], got[0]["loss_after"]))
        assert all(got[r]["rebalanced"] for r in range(world)) and sum(sizes) == P + added and max(sizes) - min(sizes) <= world
        assert all(got[r]["misplaced"] == 0 for r in range(world)) and got[0]["psnr_rebalance"] >= 35.0
        tot0, tot1 = sum(got[r]["rows_checksum"][0] for r in range(world)), sum(got[r]["rows_checksum"][1] for r in range(world))
        assert abs(tot0 - tot1) <= 1e-6 * abs(tot0)                      # parameters AND moments: the same multiset of rows
        assert got[0]["loss_after"][-1] < 1.05 * got[0]["map"][-1]


@pytest.mark.gpu
def test_two_processes_on_one_gpu_gloo_drive_the_sharded_cpp_loop():
    _run("gloo", world=2)


@pytest.mark.gpu
def test_one_rank_rccl_drives_the_sharded_cpp_loop():
    _run("nccl", world=1)


@pytest.mark.gpu
def test_two_gpus_rccl_drive_the_sharded_cpp_loop():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run("nccl", world=2)


def _empty_cell_worker(rank, world, port, q):
    """rank 1 owns a cell that holds no Gaussian at all (a split plane outside the scene): its SlamLoop has n = 0 and never launches a rasterizer kernel"""
    gsr, _C, sharded = _setup()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, raw = _scene(gsr)
        nodes = torch.tensor([[0.0, 1.0e3, -1.0, -2.0]])                   # x < 1000 -> leaf (rank) 0, else rank 1: everything is rank 0's
        mine = [x if rank == 0 else x[:0] for x in raw]
        frame = _frame(_C, raw, 0)
        # (recycled allocator memory under the empty rank's workspace: what an uninitialised header would be read from)
        junk = torch.full((1 << 22,), 0x7F7F7F7F, dtype=torch.int32, device="cuda"); del junk
        loop = _loop(_C, mine, 0, fused_update=True)
        loop.set_shard(dist.group.WORLD, rank, world, nodes)
        rgb, depth, T = frame
        res = {"size": loop.size()}
        res["render"] = loop.render_composite(T)[0].cpu().numpy()
        res["map"] = loop.map_frame(rgb, depth, T, 6)
        hist, best = loop.track(rgb, depth, _poses()[1].cuda(), 6)
        res["track"], res["pose"] = hist, best.cpu().numpy()
        rows = loop.shard_render_step(T, torch.ones((5, H, W), device="cuda"))
        res["rows"] = rows.sum(0).cpu().numpy()
        torch.cuda.synchronize()
        q.put((rank, res))
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_an_empty_cell_takes_part_in_the_sharded_loop():
    """a rank whose cell is empty (d.n == 0) launches no rasterizer kernel, so nothing ever writes its geometry header — the loss / pose kernels and
    the overflow bookkeeping must not read an uninitialised overflow flag there (a spurious skip on one rank desynchronises the replicated pose and the
    collectives). World 2 over gloo on one GPU, every Gaussian in rank 0's cell: both ranks must reproduce the UNSHARDED loop (one layer: the composite is the render)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_empty_cell_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    got, t0 = {}, time.time()
    while len(got) < 2:
        try:
            r, res = q.get(timeout=2)
            assert "error" not in res, "rank %d failed:\n%s" % (r, res.get("error"))
            got[r] = res
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() - t0 > 300:
                for p in procs:
                    p.kill()
                raise AssertionError("ranks died or hung: exit codes %s" % [p.exitcode for p in procs])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    gsr, _C, sharded = _setup()
    sc, raw = _scene(gsr)
    frame = _frame(_C, raw)
    ref = _loop(_C, raw, fused_update=True)
    ref_render = ref.render_composite(frame[2])[0].cpu().numpy()
    ref_map = ref.map_frame(frame[0], frame[1], frame[2], 6)
    ref_track, ref_pose = ref.track(frame[0], frame[1], _poses()[1].cuda(), 6)
    assert got[0]["size"] == P and got[1]["size"] == 0
    for r in (0, 1):
        assert np.all(np.isfinite(got[r]["map"])) and np.all(np.isfinite(got[r]["track"])), (r, got[r]["map"], got[r]["track"])
        assert _psnr(got[r]["render"], ref_render) >= 90.0
        np.testing.assert_allclose(got[r]["map"], ref_map, rtol=2e-4)
        n = min(len(got[r]["track"]), len(ref_track))
        assert n >= 3
        np.testing.assert_allclose(got[r]["track"][:n], ref_track[:n], rtol=2e-3)
        assert np.abs(got[r]["pose"] - ref_pose.cpu().numpy()).max() < 1e-4
    np.testing.assert_allclose(got[1]["map"], got[0]["map"], rtol=1e-6)
    np.testing.assert_allclose(got[1]["track"], got[0]["track"], rtol=1e-6)
    np.testing.assert_allclose(got[1]["rows"], got[0]["rows"], rtol=1e-6)       # (the all-reduced pose rows: the empty rank added zeros)


def _band_worker(rank, world, port, backend, q, extra=None):
    """the same cell through the sharded loop twice: the band exchange of round 6 and round 5's replicated composite"""
    gsr, _C, sharded = _setup()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = rank if backend == "nccl" and world > 1 else 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        sc, raw = _scene(gsr)
        part = sharded.KdPartition.build(raw[0], world)
        idx = torch.nonzero(part.assign(raw[0]) == rank).squeeze(-1)
        frame = _frame(_C, raw, dev)
        rgb, depth, T = frame
        obs, Xw, s2, cx, cy = _matches(_poses()[0])
        m = (obs.to(T.device), Xw.to(T.device), s2.to(T.device), cx, cy)
        res = {}
        for name, band in (("band", True), ("replicated", False)):
            loop = _loop(_C, [x[idx] for x in raw], dev, fused_update=True, band_exchange=band, **(extra or {}))
            loop.set_shard(dist.group.WORLD if world > 1 or backend == "nccl" else None, rank, world, part.nodes)
            r = {"map": loop.map_frame(rgb, depth, T, 8)}
            hist, best = loop.track(rgb, depth, _poses()[1].cuda(dev), 1, *m)
            r["pose_sums"], r["track1"] = loop.last_pose_sums().cpu().numpy(), hist[0]
            hist, best = loop.track(rgb, depth, _poses()[1].cuda(dev), 8, *m)
            r["track"], r["pose"] = hist, best.cpu().numpy()
            r["xyz"] = loop.params()[0].double().sum().item()
            r["transport"] = loop.shard_transport()
            res[name] = r
            del loop
        torch.cuda.synchronize()
        q.put((rank, res))
    except Exception:
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise
    finally:
        dist.destroy_process_group()


def _run_band(backend, world, extra=None):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, backend, q, extra)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    got, t0 = {}, time.time()
    while len(got) < world:
        try:
            r, res = q.get(timeout=2)
            assert "error" not in res, "rank %d failed:\n%s" % (r, res.get("error"))
            got[r] = res
        except queue.Empty:
            if [p.exitcode for p in procs if p.exitcode not in (None, 0)] or time.time() - t0 > 400:
                for p in procs:
                    p.kill()
                raise AssertionError("ranks died or hung: exit codes %s" % [p.exitcode for p in procs])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-30))
    for r in range(world):
        a, b = got[r]["band"], got[r]["replicated"]
        n = min(len(a["track"]), len(b["track"]))
        e = (rel(a["map"], b["map"]), rel(a["pose_sums"], b["pose_sums"]), rel(a["track"][:n], b["track"][:n]), float(np.abs(a["pose"] - b["pose"]).max()))
        if r == 0:
            print("\n  %s world %d, band exchange vs replicated composite: mapping loss curve %.1e, pose sums %.1e, tracking loss curve %.1e over %d iterations, tracked pose %.1e"
                  % (backend, world, *e[:3], n, e[3]))
        # the same render, loss and gradients — summed front to back instead of by an all-reduce, float atomics in both backward passes
        assert len(a["map"]) == 8 and n >= 3
        assert e[0] < 1e-5 and e[3] < 1e-5, e
        assert abs(a["track1"] - b["track1"]) <= 1e-5 * abs(b["track1"])   # the tracking loss of the same pose
        # (over the iterations the two runs' poses drift apart by ~1e-6 — float atomics in both — and the tracking loss is a masked SUM: a pixel whose
        # silhouette crosses 0.99 enters or leaves it whole)
        assert e[2] < 3e-4, e
        assert e[1] < 1e-3, e                                              # (sums of ~1e4 signed terms that cancel: 1e-5 of the terms' own scale)
        assert abs(a["xyz"] - b["xyz"]) <= 1e-7 * abs(b["xyz"])            # the maps after eight Adam steps
        for rr in range(1, world):                                         # replicas: every rank records the same losses
            np.testing.assert_allclose(got[rr]["band"]["map"], got[0]["band"]["map"], rtol=1e-6)
            np.testing.assert_allclose(got[rr]["band"]["track"], got[0]["band"]["track"], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 7])
def test_band_exchange_equals_the_replicated_composite_gloo(world):
    """LoopConfig::band_exchange (round 6: every rank composites, evaluates the loss and differentiates the composite on its band of pixel rows only, two grouped
    point-to-point exchanges per mapping iteration) against round 5's replicated composite (all-gather, all-reduce, all-gather; every rank the whole frame), same cells,
    same frames: 2, 4 and 7 processes on the test box's one GPU over gloo (the exchange staged through alltoall_base; seven ranks cut the 240 rows into bands of
    35 and a last one of 30: messages of unequal sizes between the pairs)."""
    _run_band("gloo", world)


@pytest.mark.gpu
def test_band_exchange_with_tracking_on_the_blended_depth_gloo():
    """Tracking.use_sur_depth = false: the tracking exchange then ships every plane (on the surface depth the blended-depth plane stays at home, both ways)
    and the rank's backward takes the depth and silhouette planes of its layer's gradient; three ranks."""
    _run_band("gloo", 3, dict(use_sur_depth=False))


@pytest.mark.gpu
def test_band_exchange_one_rank_rccl():
    _run_band("nccl", 1)


@pytest.mark.gpu
def test_band_exchange_two_gpus_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_band("nccl", 2)


def _matches(T, n=60, seed=9):
    """feature matches: world points in front of the camera of T and their (noisy) pixel observations"""
    g = torch.Generator().manual_seed(seed)
    Xc = torch.stack([torch.rand(n, generator=g) * 1.2 - 0.6, torch.rand(n, generator=g) * 0.9 - 0.45, 1.0 + 2.0 * torch.rand(n, generator=g)], 1)
    Xw = (Xc - T[:3, 3]) @ T[:3, :3]
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    obs = torch.stack([FX * Xc[:, 0] / Xc[:, 2] + cx, FY * Xc[:, 1] / Xc[:, 2] + cy], 1) + 0.5 * torch.randn(n, 2, generator=g)
    obs[::7] += 8.0                                                       # a few outliers: frozen out halfway through (chi-square 5.991)
    return obs, Xw, torch.full((n,), 1.0), cx, cy


@pytest.mark.gpu
@pytest.mark.parametrize("sur", [True, False])
def test_cpp_track_with_feature_matches_follows_the_python_harness(sur):
    """(sur: Tracking.use_sur_depth — the depth term on the surface (median) depth, the reference's default and the C++ loop's plain three-channel forward with
    the mask from the final transmittance, or on the alpha-blended depth through the fused pair.) SlamLoop::Track with the ORB matches' reprojection term (Render.cc:1031-1096; gsr_reproj_loss inside the direct loop, fused pose step and
    two-launch form; the tensor expressions on the autograd path) against the Python harness's track() with the same matches on the same map:
    loss curves, the pose, and the term really pulls (without it the curve differs)."""
    gsr, _C, sharded = _setup()
    hz = __import__("gsorb_slam_amd.harness", fromlist=["x"])
    sc, raw = _scene(gsr)
    frame = _frame(_C, raw)
    T, T0 = _poses()
    obs, Xw, s2, cx, cy = _matches(T)
    FW, iters = 5.0, 12
    K = torch.tensor([[FX, 0.0, cx], [0.0, FY, cy], [0.0, 0.0, 1.0]], device="cuda")
    g = hz.GaussianMap(hz.Config(), FX, FY, device="cuda")
    g.add_points(raw[0], raw[1])
    with torch.no_grad():
        g.unnorm_quat.copy_(raw[2].cuda()); g.logit_opacities.copy_(raw[3].cuda()); g.log_scales.copy_(raw[4].cuda())
    g.cfg.feature_weight_tracking = FW
    g.cfg.use_sur_depth = sur
    r = hz.SlamRenderer(g, W, H)
    mt = (torch.cat([obs, torch.ones(len(obs), 1)], 1).reshape(-1, 3, 1).cuda(), torch.cat([Xw, torch.ones(len(Xw), 1)], 1).reshape(-1, 4, 1).cuda(), s2.reshape(-1, 1).cuda())
    T_ref, h_ref = r.track(hz.Frame(frame[0], frame[1], frame[2]), T0.cuda(), iters=iters, matches=mt, K=K)
    out = {}
    for name, cfg in (("fused", dict()), ("two-launch", dict(fused_update=False)), ("autograd", dict(direct=False)), ("no-matches", dict())):
        loop = _loop(_C, raw, feature_weight_tracking=FW, use_sur_depth=sur, **cfg)
        args = () if name == "no-matches" else (obs.cuda(), Xw.cuda(), s2.cuda(), cx, cy)
        h, best = loop.track(frame[0], frame[1], T0.cuda(), iters, *args)
        out[name] = (np.array(h), best.cpu().numpy())
    n = min(len(h_ref), *(len(out[k][0]) for k in ("fused", "two-launch", "autograd")))
    assert n >= 6
    scale = np.abs(np.array(h_ref)).max()
    for k in ("fused", "two-launch", "autograd"):
        e = np.abs(out[k][0][:n] - np.array(h_ref)[:n]).max() / scale
        print("  track with matches, C++ %-10s vs Python harness: loss curve %.1e over %d iterations, pose %.1e" % (k, e, n, np.abs(out[k][1] - T_ref.cpu().numpy()).max()))
        assert e < 5e-3 and np.abs(out[k][1] - T_ref.cpu().numpy()).max() < 1e-3
    gap = np.abs(out["no-matches"][0][:n] - out["fused"][0][:n]).max() / scale
    assert gap > 0.05, gap                                                # the term is a real share of the objective


@pytest.mark.gpu
def test_shard_render_step_pose_sums_equal_autograd_through_the_python_operator():
    """SlamLoop::ShardRenderStep (what bench.py's N > 1 headline times) with one rank and no group: the composite is the render, so the pose sums it returns
    must be the gradient of sum(G . (rgb, depth, silhouette)) w.r.t. the pose that autograd finds through the Python operator's fused pair."""
    gsr, _C, sharded = _setup()
    import diff_gaussian_rasterization as dgr
    sc, raw = _scene(gsr)
    T = _poses()[0].cuda()
    loop = _loop(_C, raw)
    loop.set_shard(None, 0, 1, torch.empty(0))
    G = torch.randn((5, H, W), generator=torch.Generator().manual_seed(3)).cuda().contiguous()
    rows = loop.shard_render_step(T, G).clone()
    rows2 = loop.shard_render_step(T, G)                                   # (a second call on the same pose: same result up to the atomics' order)
    sums = rows.sum(0).cpu().double()
    assert (rows2.sum(0).cpu().double() - sums).abs().max() <= 1e-3 * sums.abs().max()
    cam = gsr.synthetic.make_camera(W, H, FX, FY)
    s = gsr.capi.Settings.from_camera(cam, device="cuda")
    rs = dgr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=s.tanfovx, tanfovy=s.tanfovy, bg=torch.zeros(3, device="cuda"), scale_modifier=1.0,
                                           viewmatrix=s.viewmatrix, projmatrix=s.projmatrix, sh_degree=0, campos=s.campos, prefiltered=False)
    rast = dgr.GaussianRasterizer(raster_settings=rs)
    Tp = T.clone().requires_grad_(True)
    xyz = raw[0].cuda()
    mc = gsr.capi.to_camera(Tp, xyz)
    img, ds, _, _ = rast.forward_pair(means3D=mc, means2D=torch.zeros_like(mc, requires_grad=True), opacities=torch.sigmoid(raw[3].cuda()), colors_precomp=raw[1].cuda(),
                                      scales=torch.exp(raw[4].cuda()), rotations=torch.nn.functional.normalize(raw[2].cuda()))
    ((img * G[0:3]).sum() + (ds * G[3:5]).sum()).backward()
    ref = torch.cat([Tp.grad[:3, :3].reshape(-1), Tp.grad[:3, 3]]).cpu().double()
    e = float((sums - ref).abs().max() / ref.abs().max())
    print("\n  ShardRenderStep pose sums vs autograd through the Python operator: %.1e" % e)
    assert e < 2e-3                                                          # (sums of ~1e5 signed terms that cancel: float atomics in the backward)
