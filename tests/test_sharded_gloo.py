"""CPU, world_size 2, gloo: the multi-GPU exchange of the path (gsorb-slam_amd/sharded.py).
Each rank renders its depth slab of the scene with the CPU oracle (test stand-in for the HIP
op), the layers are all-gathered and composited, and the result is compared with the
single-process render of the whole scene; the pose-gradient all-reduce is checked too."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_package
        from oracle import oracle
        gsr = load_package()
        syn = gsr.synthetic
        sharded = __import__("gsorb_slam_amd.sharded", fromlist=["x"])
        cam = syn.make_camera(128, 96, 100.0, 100.0)
        sc = syn.make_scene(1500, cam, seed=11, scale_mult=2.5)
        z = torch.tensor(sc.means3D[:, 2])
        mine = sharded.shard_by_depth_slabs(z, world)[rank].numpy()

        def render(idx, colors):
            o = oracle.Oracle()
            f = o.forward(copy_stages=False, means3D=sc.means3D[idx], opacities=sc.opacities[idx], cam=cam,
                          colors=colors, scales=sc.scales[idx], rotations=sc.rotations[idx])
            return f.color

        zc = lambda idx: np.stack([sc.means3D[idx, 2], np.ones(len(idx)), np.zeros(len(idx))], 1).astype(np.float32)
        rgb = torch.tensor(render(mine, sc.colors[mine]), requires_grad=True)
        ds = torch.tensor(render(mine, zc(mine))[:2], requires_grad=True)
        comp = sharded.LayerCompositor()
        out_rgb, out_depth, out_sil = comp.composite(rgb, ds, float(z[mine].min()))
        # the same loss on every rank; autograd must reach only this rank's layer
        w = torch.linspace(0.5, 1.5, 128)[None, None, :]
        loss = (out_rgb * w).sum() + out_depth.sum()
        loss.backward()
        pose_g = comp.all_reduce_pose_grad(torch.full((4, 4), float(rank + 1)))
        if rank == 0:
            allidx = np.arange(sc.P)
            full_rgb = render(allidx, sc.colors)
            full_ds = render(allidx, zc(allidx))
            q.put(dict(rgb=out_rgb.detach().numpy(), depth=out_depth.detach().numpy(), sil=out_sil.detach().numpy(),
                       full_rgb=full_rgb, full_ds=full_ds, pose=pose_g.numpy(), grad_rgb=rgb.grad.numpy(),
                       grad_ds=ds.grad.numpy(), front=bool(z[mine].min() <= z.min())))
    finally:
        dist.destroy_process_group()


def test_two_rank_layer_compositing_matches_single_process_render():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # depth slabs are depth-separable: compositing equals the whole-scene render except where the
    # reference's stop rule fires (forward.cu:360-364: a pixel that stops keeps its residual T, which
    # the single render discards but the composite still passes to the slab behind); residual T is
    # small by construction, so the two agree to a few 1e-3 at worst and ~1e-7 almost everywhere
    d = np.abs(res["rgb"] - res["full_rgb"])
    assert d.max() < 5e-3 and np.median(d) < 1e-6 and (d > 1e-4).mean() < 0.02
    assert np.abs(res["depth"][0] - res["full_ds"][0]).max() < 5e-2
    assert np.abs(res["sil"][0] - res["full_ds"][1]).max() < 5e-3
    np.testing.assert_array_equal(res["pose"], np.full((4, 4), 3.0))      # 1 + 2
    # rank 0 owns the front slab: its layer enters with weight 1 (w for the loss above)
    assert res["front"]
    np.testing.assert_allclose(res["grad_rgb"], np.broadcast_to(np.linspace(0.5, 1.5, 128, dtype=np.float32), (3, 96, 128)), rtol=1e-6)
    assert np.isfinite(res["grad_ds"]).all() and np.abs(res["grad_ds"][1]).max() > 0   # silhouette shades what is behind
