"""CPU, world_size 2, gloo: the sharded MAPPING / TRACKING step (gsorb-slam_amd/sharded.py:ShardedMapper,
SURVEY.md §8e scheme B, BASELINE.json configs 4-5) against the unsharded harness.

Every rank owns a depth slab of the map in its own GaussianMap (own Adam state) and rasterizes it with the
CPU oracle wrapped as an autograd op (tests/oracle_op.py; the GPU twin of this test runs the HIP operator).
Checked:
  * pose gradient: each rank back-propagates the tracking loss through composite() to the pose it fed its own
    shard with; the all-reduced sum equals the single-process gradient d loss / d pose of the SAME composited
    loss (all layers differentiable in one process), and is close to the gradient of the unsharded render;
  * one mapping iteration: loss and the per-Gaussian gradients of every shard agree with the rows of the
    unsharded iteration; the Adam step of the shards equals the unsharded step where the gradient is not ~0;
  * tracking: a sharded track() follows the unsharded one.
"""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, FX, FY = 160, 120, 130.0, 129.0


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
    return syn.make_scene(2500, cam, seed=21, scale_mult=1.2)


def _fill(hz, sc, idx, device="cpu"):
    g = hz.GaussianMap(hz.Config(), FX, FY, device=device)
    g.add_points(torch.tensor(sc.means3D[idx]), torch.tensor(sc.colors[idx]))
    op = torch.tensor(sc.opacities[idx])
    with torch.no_grad():
        g.log_scales.copy_(torch.log(torch.tensor(sc.scales[idx])))
        g.unnorm_quat.copy_(torch.tensor(sc.rotations[idx]))
        g.logit_opacities.copy_(torch.log(op / (1 - op)))
        g.log_scales[torch.tensor(np.asarray(idx) % 7 == 0)] += 1.6   # some oversized splats (chosen by GLOBAL index): the scale regularisers are active
    return g


def _target(hz, sc, OracleRasterizer, Tcw):
    """observation: the whole map with perturbed colours, seen from Tcw."""
    g = _fill(hz, sc, np.arange(sc.P))
    with torch.no_grad():
        g.rgb.mul_(0.8).add_(0.1)
    r = hz.SlamRenderer(g, W, H, rasterizer_cls=OracleRasterizer)
    with torch.no_grad():
        rgb, sur, _ = r.render_rgb(Tcw, tracking=True)
    return hz.Frame(rgb.clone(), sur[0].clone(), Tcw.clone())


def _matches(Tcw, n=40, seed=9):
    """feature matches for track(): world points in front of the camera and their pixel observations under Tcw (+ noise)."""
    g = torch.Generator().manual_seed(seed)
    Xc = torch.stack([torch.rand(n, generator=g) * 1.2 - 0.6, torch.rand(n, generator=g) * 0.9 - 0.45, 1.0 + 2.0 * torch.rand(n, generator=g)], 1)
    Twc = torch.inverse(Tcw)
    Xw = Xc @ Twc[:3, :3].t() + Twc[:3, 3]
    K = torch.tensor([[FX, 0.0, (W - 1) / 2.0], [0.0, FY, (H - 1) / 2.0], [0.0, 0.0, 1.0]])
    uv = (K @ (Xc / Xc[:, 2:3]).t()).t() + torch.cat([0.3 * torch.randn(n, 2, generator=g), torch.zeros(n, 1)], 1)
    return (uv.reshape(n, 3, 1), torch.cat([Xw, torch.ones(n, 1)], 1).reshape(n, 4, 1), torch.full((n, 1), 1.0)), K


FEATURE_WEIGHT = 20.0   # the reprojection term and the render terms pull on the pose with comparable strength


def _record_pose_grads(g):
    """the pose gradient the optimiser sees (after the ranks' gradients are summed), one entry per iteration"""
    rec, init = [], g.init_camera_pose

    def init_and_wrap(Tcw):                              # track() creates the pose optimiser: wrap its step() right after
        out = init(Tcw)
        step = g.opt_pose.step

        def wrapped(*a, **k):
            rec.append(torch.cat([g.cam_quat.grad.reshape(-1), g.cam_trans.grad.reshape(-1)]).detach().clone())
            return step(*a, **k)
        g.opt_pose.step = wrapped
        return out
    g.init_camera_pose = init_and_wrap
    return rec


def _tracking_loss(hz, r, frame, Tcw):
    c = r.map.cfg
    rimage, rsur, rdepth = r.render_pair(Tcw, tracking=True)
    certain = (rdepth[1] > 0.99) & ~torch.isnan(frame.depth)
    return (c.im_weight_tracking * hz.l1_tracking(rimage, frame.rgb, certain.unsqueeze(0).repeat(3, 1, 1).detach())
            + c.depth_weight_tracking * hz.l1_tracking(rdepth[0], frame.depth, certain.detach()))


def _smooth_loss(r, frame, Tcw):
    """un-masked squared error: no pixel enters or leaves the loss when the silhouette moves by 1e-3"""
    rimage, _, rdepth = r.render_pair(Tcw, tracking=True)
    return ((rimage - frame.rgb) ** 2).sum() + ((rdepth[0] - frame.depth * rdepth[1].detach()) ** 2).sum()


def _worker(rank, world, port, q):
    gsr, hz, sharded = _setup()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_op import OracleRasterizer
        from util import pose
        OracleRasterizer.omp = False                       # deterministic summation order in the checker
        torch.manual_seed(0)
        sc = _scene(gsr)
        Tcw = torch.tensor(pose(0.02, (0.01, -0.01, 0.015)), dtype=torch.float32)
        zc = torch.tensor(sc.means3D) @ Tcw[2, :3] + Tcw[2, 3]
        slabs = [s.numpy() for s in sharded.shard_by_depth_slabs(zc, world)]
        frame = _target(hz, sc, OracleRasterizer, Tcw)
        Mapper = sharded.make_sharded_mapper(hz)
        res = {}

        # ---- (1) pose gradient through composite(), all-reduced
        g = _fill(hz, sc, slabs[rank])
        m = Mapper(g, W, H, rasterizer_cls=OracleRasterizer)
        Tp = Tcw.clone().requires_grad_(True)
        loss = _tracking_loss(hz, m, frame, Tp)
        loss.backward()
        own = Tp.grad.clone()
        total = m.comp.all_reduce_pose_grad(Tp.grad.clone())
        res.update(pose_own=own.numpy(), pose_sum=total.numpy(), track_loss=float(loss.detach()))
        Tp = Tcw.clone().requires_grad_(True)
        _smooth_loss(m, frame, Tp).backward()
        res.update(pose_sum_smooth=m.comp.all_reduce_pose_grad(Tp.grad.clone()).numpy())

        # ---- (2) one mapping iteration on the shard
        g = _fill(hz, sc, slabs[rank])
        m = Mapper(g, W, H, rasterizer_cls=OracleRasterizer)
        before = {n: getattr(g, n).detach().clone() for n in g.NAMES}
        ml = m.mapping_loss(frame)
        ml.backward()
        grads = {n: getattr(g, n).grad.detach().clone().numpy() for n in g.NAMES}
        with torch.no_grad():
            g.opt.step()
        steps = {n: (getattr(g, n).detach() - before[n]).numpy() for n in g.NAMES}
        res.update(map_loss=float(ml.detach()), grads=grads, steps=steps, idx=slabs[rank])

        # ---- (3) a short sharded track()
        g = _fill(hz, sc, slabs[rank])
        m = Mapper(g, W, H, rasterizer_cls=OracleRasterizer)
        T0 = torch.tensor(pose(0.025, (0.02, -0.005, 0.03)), dtype=torch.float32)
        T_est, hist = m.track(frame, T0, iters=12)
        res.update(track_T=T_est.numpy(), track_hist=hist)

        # ---- (4) the same with feature matches: the reprojection term depends on the pose only, every rank evaluates it in
        #      full, and the summed pose gradient must contain it ONCE
        g = _fill(hz, sc, slabs[rank])
        m = Mapper(g, W, H, rasterizer_cls=OracleRasterizer)
        g.cfg.feature_weight_tracking = FEATURE_WEIGHT
        mt, K = _matches(Tcw)
        rec = _record_pose_grads(g)
        T_est, hist = m.track(frame, T0, iters=8, matches=mt, K=K)
        res.update(trackm_T=T_est.numpy(), trackm_hist=hist, trackm_grad0=rec[0].numpy())
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _single_process_reference():
    """Same quantities without any sharding (and the two-layer composite differentiated in ONE process)."""
    gsr, hz, sharded = _setup()
    from oracle_op import OracleRasterizer
    from util import pose
    OracleRasterizer.omp = False
    sc = _scene(gsr)
    Tcw = torch.tensor(pose(0.02, (0.01, -0.01, 0.015)), dtype=torch.float32)
    zc = torch.tensor(sc.means3D) @ Tcw[2, :3] + Tcw[2, 3]
    slabs = [s.numpy() for s in sharded.shard_by_depth_slabs(zc, 2)]
    frame = _target(hz, sc, OracleRasterizer, Tcw)
    ref = {}
    # (1a) composite of the two layers in one process: every layer keeps its autograd history
    Tp = Tcw.clone().requires_grad_(True)
    layers = []
    for idx in slabs:
        r = hz.SlamRenderer(_fill(hz, sc, idx), W, H, rasterizer_cls=OracleRasterizer)
        rimage, _, rdepth = r.render_pair(Tp, tracking=True)
        layers.append(torch.cat([rimage, rdepth[0:2]], 0))
    T = torch.ones_like(layers[0][0:1])
    out = torch.zeros_like(layers[0][0:4])
    for L in layers:                                      # slabs come front to back
        out = out + T * L[0:4]
        T = T * (1.0 - L[4:5])
    sil = 1.0 - T
    c = hz.Config()
    certain = (sil[0] > 0.99) & ~torch.isnan(frame.depth)
    loss = (c.im_weight_tracking * hz.l1_tracking(out[0:3], frame.rgb, certain.unsqueeze(0).repeat(3, 1, 1).detach())
            + c.depth_weight_tracking * hz.l1_tracking(out[3], frame.depth, certain.detach()))
    loss.backward()
    ref.update(pose_composite=Tp.grad.numpy().copy(), track_loss_composite=float(loss.detach()))
    # (1b) unsharded render
    g = _fill(hz, sc, np.arange(sc.P))
    r = hz.SlamRenderer(g, W, H, rasterizer_cls=OracleRasterizer)
    Tp = Tcw.clone().requires_grad_(True)
    loss = _tracking_loss(hz, r, frame, Tp)
    loss.backward()
    ref.update(pose_full=Tp.grad.numpy().copy(), track_loss_full=float(loss.detach()))
    Tp = Tcw.clone().requires_grad_(True)
    _smooth_loss(r, frame, Tp).backward()
    ref.update(pose_full_smooth=Tp.grad.numpy().copy())
    # (2) unsharded mapping iteration
    g = _fill(hz, sc, np.arange(sc.P))
    r = hz.SlamRenderer(g, W, H, rasterizer_cls=OracleRasterizer)
    before = {n: getattr(g, n).detach().clone() for n in g.NAMES}
    ml = r.mapping_loss(frame)
    ml.backward()
    ref["grads"] = {n: getattr(g, n).grad.detach().clone().numpy() for n in g.NAMES}
    with torch.no_grad():
        g.opt.step()
    ref["steps"] = {n: (getattr(g, n).detach() - before[n]).numpy() for n in g.NAMES}
    ref["map_loss"] = float(ml.detach())
    # (3) unsharded track
    g = _fill(hz, sc, np.arange(sc.P))
    r = hz.SlamRenderer(g, W, H, rasterizer_cls=OracleRasterizer)
    T0 = torch.tensor(pose(0.025, (0.02, -0.005, 0.03)), dtype=torch.float32)
    T_est, hist = r.track(frame, T0, iters=12)
    ref.update(track_T=T_est.numpy(), track_hist=hist)
    # (4) unsharded track with feature matches
    g = _fill(hz, sc, np.arange(sc.P))
    g.cfg.feature_weight_tracking = FEATURE_WEIGHT
    r = hz.SlamRenderer(g, W, H, rasterizer_cls=OracleRasterizer)
    mt, K = _matches(Tcw)
    rec = _record_pose_grads(g)
    T_est, hist = r.track(frame, T0, iters=8, matches=mt, K=K)
    ref.update(trackm_T=T_est.numpy(), trackm_hist=hist, trackm_grad0=rec[0].numpy())
    # the feature term alone (render terms switched off): its share of the pose gradient
    g = _fill(hz, sc, np.arange(sc.P))
    g.cfg.feature_weight_tracking, g.cfg.im_weight_tracking, g.cfg.depth_weight_tracking = FEATURE_WEIGHT, 0.0, 0.0
    r = hz.SlamRenderer(g, W, H, rasterizer_cls=OracleRasterizer)
    rec = _record_pose_grads(g)
    r.track(frame, T0, iters=1, matches=mt, K=K)
    ref.update(trackm_grad0_feature=rec[0].numpy())
    return ref


def test_two_rank_sharded_mapping_and_tracking_match_the_unsharded_harness():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _single_process_reference()
    scale = lambda a: np.abs(a).max() + 1e-30

    # (1) pose gradient: per-rank parts differ, their all-reduced sum is the single-process gradient
    own0, own1 = got[0]["pose_own"], got[1]["pose_own"]
    assert np.abs(own0 - own1).max() > 1e-3 * scale(own0)
    for r in range(world):
        np.testing.assert_allclose(got[r]["pose_sum"], own0 + own1, rtol=0, atol=1e-6 * scale(own0 + own1))
        assert abs(got[r]["track_loss"] - ref["track_loss_composite"]) <= 1e-6 * abs(ref["track_loss_composite"])
    e_comp = np.abs(got[0]["pose_sum"] - ref["pose_composite"]).max() / scale(ref["pose_composite"])
    # against the UNSHARDED render scheme B is exact only up to the residual T of early-stopped pixels; compared on a
    # smooth loss (the tracking loss is a masked L1 sum: a pixel whose silhouette crosses 0.99 enters or leaves it whole)
    e_full = np.abs(got[0]["pose_sum_smooth"] - ref["pose_full_smooth"]).max() / scale(ref["pose_full_smooth"])
    print("\npose gradient: sum over ranks vs one-process composite %.2e, vs unsharded render (smooth loss) %.2e" % (e_comp, e_full))
    assert e_comp < 1e-5, e_comp
    assert e_full < 2e-2, e_full
    assert (got[0]["pose_sum"][3] == 0).all()      # the last row of Tcw feeds nothing

    # (2) mapping iteration
    assert abs(got[0]["map_loss"] - got[1]["map_loss"]) <= 1e-6 * abs(got[0]["map_loss"])
    assert abs(got[0]["map_loss"] - ref["map_loss"]) <= 2e-3 * abs(ref["map_loss"]), (got[0]["map_loss"], ref["map_loss"])
    worst = {}
    for r in range(world):
        idx = got[r]["idx"]
        for n, gsh in got[r]["grads"].items():
            gref = ref["grads"][n][idx]
            worst[n] = max(worst.get(n, 0.0), float(np.abs(gsh - gref).max() / scale(ref["grads"][n])))
            st, sref = got[r]["steps"][n], ref["steps"][n][idx]
            big = np.abs(gref) > 1e-2 * scale(gref)       # Adam's first step is lr * sign(g): compare where g is not ~0
            assert big.any(), n
            same = np.sign(gsh) == np.sign(gref)        # the composite is approximate: a few gradients change sign
            assert (~same & big).sum() <= 0.03 * big.sum(), (n, int((~same & big).sum()), int(big.sum()))
            np.testing.assert_allclose(st[big & same], sref[big & same], rtol=1e-3, atol=1e-9, err_msg=n)
    print("mapping gradients, sharded vs unsharded, max |diff| / max |ref|:", {k: "%.1e" % v for k, v in worst.items()})
    assert max(worst.values()) < 2e-2, worst

    # (3) tracking: identical pose copies on the ranks, close to the unsharded track
    np.testing.assert_array_equal(got[0]["track_T"], got[1]["track_T"])
    assert len(got[0]["track_hist"]) == len(ref["track_hist"])
    np.testing.assert_allclose(got[0]["track_hist"], ref["track_hist"], rtol=2e-2)   # the composited surface depth is approximate
    assert np.abs(got[0]["track_T"] - ref["track_T"]).max() < 2e-3

    # (4) with feature matches (the term every rank holds in full enters the summed gradient once: with it counted `world`
    #     times the second iteration's loss is already off by several percent)
    np.testing.assert_array_equal(got[0]["trackm_T"], got[1]["trackm_T"])
    assert len(got[0]["trackm_hist"]) == len(ref["trackm_hist"])
    gsum, gref, gfeat = got[0]["trackm_grad0"], ref["trackm_grad0"], ref["trackm_grad0_feature"]
    share = np.abs(gfeat).max() / scale(gref)
    err = np.abs(gsum - gref).max() / scale(gref)
    print("track() with matches: feature term's share of the pose gradient %.2f, summed gradient vs unsharded %.1e" % (share, err))
    assert share > 0.2                                   # counted twice, the gradient would be off by about this much
    assert err < 0.1 * share, (err, share)
    np.testing.assert_allclose(got[0]["trackm_hist"], ref["trackm_hist"], rtol=2e-2)
    assert np.abs(got[0]["trackm_T"] - ref["trackm_T"]).max() < 2e-3
