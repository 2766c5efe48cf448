"""Map I/O and the replay harness."""
import os

import numpy as np
import pytest

from oracle import oracle
from util import pose


def _model(syn, P=4000, seed=3):
    cam = syn.make_camera(320, 240, 260.0, 258.0)
    sc = syn.make_scene(P, cam, seed=seed, scale_mult=2.0)
    rng = np.random.default_rng(seed)
    unq = (sc.rotations * rng.uniform(0.5, 2.0, (P, 1))).astype(np.float32)
    return sc, unq


def test_ply_round_trip_and_header(gsr, syn, tmp_path):
    rp = __import__("gsorb_slam_amd.replay", fromlist=["x"])
    sc, unq = _model(syn, 500)
    m = rp.GaussianModel(sc.means3D, sc.colors, np.log(sc.opacities / (1 - sc.opacities)), np.log(sc.scales), unq)
    p = str(tmp_path / "GaussianModel.ply")
    rp.write_ply(p, m)
    head = open(p, "rb").read(400).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 500\nproperty float x\n")
    for name in ("rgb_0", "opacity", "scale_2", "rot_3"):
        assert "property float %s\n" % name in head
    m2 = rp.read_ply(p)
    for a in ("xyz", "rgb", "opacity", "scaling", "rotation"):
        np.testing.assert_array_equal(getattr(m, a).astype(np.float32).reshape(getattr(m2, a).shape), getattr(m2, a))
    assert os.path.getsize(p) == head.index("end_header\n") + len("end_header\n") + 500 * 14 * 4


def test_trajectory_formats(gsr, tmp_path):
    rp = __import__("gsorb_slam_amd.replay", fromlist=["x"])
    T = pose(0.2, (0.1, 0.2, 0.3)).astype(np.float32)
    (tmp_path / "rep.txt").write_text(" ".join("%.8f" % v for v in T.ravel()) + "\n")
    (tmp_path / "scan.txt").write_text("# comment\n7 " + " ".join("%.8f" % v for v in T.ravel()) + "\n")
    # quaternion (x,y,z,w) of a rotation about y by 0.2 rad
    (tmp_path / "tum.txt").write_text("1305031102.17 0.1 0.2 0.3 0 %.9f 0 %.9f\n" % (np.sin(0.1), np.cos(0.1)))
    for f, kind in (("rep.txt", "replica"), ("scan.txt", "scannet"), ("tum.txt", "tum")):
        poses, _ = rp.read_trajectory(str(tmp_path / f), kind)
        assert len(poses) == 1
        np.testing.assert_allclose(poses[0], T, atol=1e-6)


@pytest.mark.gpu
def test_replay_renders_match_oracle(gsr, syn, tmp_path):
    import torch
    rp = __import__("gsorb_slam_amd.replay", fromlist=["x"])
    sc, unq = _model(syn)
    W, H, fx, fy = 320, 240, 260.0, 258.0
    m = rp.GaussianModel(sc.means3D, sc.colors, np.log(sc.opacities / (1 - sc.opacities)), np.log(sc.scales), unq)
    p = str(tmp_path / "GaussianModel.ply")
    rp.write_ply(p, m)
    K = [[fx, 0, W / 2], [0, fy, H / 2], [0, 0, 1]]
    r = rp.Replayer(rp.read_ply(p), K, W, H)
    # the replay camera equals the reference construction used by the synthetic generator
    cam = syn.make_camera(W, H, fx, fy)
    np.testing.assert_allclose(r.cam.projmatrix.cpu().numpy(), cam.projmatrix, rtol=1e-6, atol=1e-7)
    poses = [np.eye(4, dtype=np.float32), pose(0.05, (0.02, 0.0, -0.05)).astype(np.float32)]
    gts, gds = [], []
    q = unq / np.linalg.norm(unq, axis=1, keepdims=True)
    for Tp in poses:   # ground truth = oracle render of the same map; saved poses are camera-to-world,
        T = np.linalg.inv(np.linalg.inv(poses[0]) @ Tp).astype(np.float32)   # w2c relative to frame 0 (replay.py:308-312)
        mc = (sc.means3D.astype(np.float64) @ T[:3, :3].T.astype(np.float64) + T[:3, 3]).astype(np.float32)
        o = oracle.Oracle()
        f = o.forward(copy_stages=False, means3D=mc, opacities=sc.opacities, cam=cam, colors=sc.colors,
                      scales=sc.scales, rotations=q.astype(np.float32))
        gts.append(f.color); gds.append(f.depth)
    res = r.evaluate(poses, colors=gts, depths=gds)
    assert res["frames"] == 2 and res["mean_psnr"] > 60.0          # same image up to rounding
    assert res["mean_depth_l1"] < 1e-3
    rgb, surf, dep = r.render(np.linalg.inv(poses[1]))
    assert rgb.shape == (3, H, W) and surf.shape == (1, H, W) and dep.shape == (1, H, W)
