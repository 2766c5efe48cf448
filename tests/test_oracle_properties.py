"""CPU: size-independent properties of the rasterizer, checked on the oracle. The same
properties are re-checked on the HIP path at BASELINE sizes in tests/test_gpu_parity.py."""
import numpy as np

from oracle import oracle


def _scene(syn, P=4000, **kw):
    cam = syn.make_camera(320, 208, 250.0, 250.0, **{k: v for k, v in kw.items() if k in ("bg", "Tcw")})
    return syn.make_scene(P, cam, seed=5, scale_mult=kw.get("mult", 2.0))


def test_stage_invariants(syn):
    sc = _scene(syn)
    o, f = oracle.forward_scene(sc)
    st = f.stages
    assert f.num_rendered == int(st["tiles_touched"].sum()) == st["point_list"].size
    assert np.all(np.diff(st["keys_sorted"].astype(np.uint64)) >= 0)          # sorted by (tile, depth)
    r = st["ranges"]
    nz = r[:, 1] > r[:, 0]
    assert int((r[nz, 1] - r[nz, 0]).sum()) == f.num_rendered
    tiles = (st["keys_sorted"] >> np.uint64(32)).astype(np.int64)
    for t in np.nonzero(nz)[0][:50]:
        assert np.all(tiles[r[t, 0]:r[t, 1]] == t)
    # stable: equal keys keep splat-index order
    k = st["keys_sorted"]
    same = k[1:] == k[:-1]
    assert np.all(st["point_list"][1:][same] > st["point_list"][:-1][same])
    assert np.all((f.radii > 0) == (st["tiles_touched"] > 0))


def test_background_enters_linearly(syn):
    sc0 = _scene(syn, bg=(0, 0, 0))
    sc1 = _scene(syn, bg=(0.2, 0.4, 0.9))
    _, f0 = oracle.forward_scene(sc0)
    _, f1 = oracle.forward_scene(sc1)
    T = f0.stages["final_T"].reshape(208, 320)
    exp = f0.color + T[None] * np.array([0.2, 0.4, 0.9], np.float32)[:, None, None]
    np.testing.assert_allclose(f1.color, exp, atol=2e-6)


def test_transparent_splats_are_noops(syn):
    sc = _scene(syn)
    _, f0 = oracle.forward_scene(sc)
    keep = sc.opacities.ravel() >= 0.6
    op = sc.opacities.copy()
    op[~keep] = 1.0 / 512.0           # below 1/255: can never be blended (forward.cu:357-359)
    sc2 = type(sc)(sc.cam, sc.means3D, sc.scales, sc.rotations, op, sc.colors, None, sc.dL_dpix)
    o2, f2 = oracle.forward_scene(sc2)
    sc3 = type(sc)(sc.cam, sc.means3D[keep], sc.scales[keep], sc.rotations[keep], sc.opacities[keep],
                   sc.colors[keep], None, sc.dL_dpix)
    _, f3 = oracle.forward_scene(sc3)
    np.testing.assert_array_equal(f2.color, f3.color)
    np.testing.assert_array_equal(f2.depth, f3.depth)
    b2 = o2.backward(sc.dL_dpix)
    assert np.all(b2.dL_dmeans3D[~keep] == 0) and np.all(b2.dL_dopacity[~keep] == 0)


def test_splat_order_in_memory_is_irrelevant_when_depths_differ(syn):
    sc = _scene(syn, P=1500)
    perm = np.random.default_rng(0).permutation(sc.P)
    sc2 = type(sc)(sc.cam, sc.means3D[perm], sc.scales[perm], sc.rotations[perm], sc.opacities[perm],
                   sc.colors[perm], None, sc.dL_dpix)
    _, f0 = oracle.forward_scene(sc)
    _, f1 = oracle.forward_scene(sc2)
    assert len(np.unique(sc.means3D[:, 2])) == sc.P
    np.testing.assert_array_equal(f0.color, f1.color)
    np.testing.assert_array_equal(f0.radii[perm], f1.radii)


def test_empty_and_all_culled(syn):
    cam = syn.make_camera(64, 48, 50.0, 50.0, bg=(0.1, 0.2, 0.3))
    o = oracle.Oracle()
    z = lambda *s: np.zeros(s, np.float32)
    f = o.forward(means3D=z(0, 3), opacities=z(0, 1), cam=cam, colors=z(0, 3), scales=z(0, 3), rotations=z(0, 4))
    assert f.num_rendered == 0 and np.all(f.color == 0)      # src/Rasterizer.cu:183: P==0 -> zero image, no bg
    P = 10
    m = np.tile(np.array([0, 0, -1.0], np.float32), (P, 1))   # behind the camera
    f = o.forward(means3D=m, opacities=np.full((P, 1), .5, np.float32), cam=cam, colors=z(P, 3),
                  scales=np.full((P, 3), .1, np.float32), rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)))
    assert f.num_rendered == 0 and np.all(f.radii == 0)
    np.testing.assert_allclose(f.color, np.broadcast_to(cam.bg[:, None, None], f.color.shape))
    assert not oracle.mark_visible(m, cam).any()


def test_filter_radii_equals_forward_radii(syn):
    sc = _scene(syn, P=2000)
    _, f = oracle.forward_scene(sc)
    r = oracle.filter_radii(sc.means3D, sc.scales, sc.rotations, sc.cam)
    np.testing.assert_array_equal(r, f.radii)
    assert oracle.lib().gsro_higher_msb(75 * 43) == 12 and oracle.lib().gsro_higher_msb(1200) == 11
