"""the fused SSIM and Adam kernels (csrc/gsr_train.h) against plain PyTorch fp32 references of the
same operations — the torch-convolution SSIM of harness.ssim_torch (Utils.cc:77-100, the reference's asymmetric window)
and torch.optim.Adam as Gaussian.cc:144-175 configures it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hz(gsr):
    return __import__("gsorb_slam_amd.harness", fromlist=["x"])


@pytest.mark.parametrize("shape", [(3, 37, 53), (1, 16, 16), (3, 5, 200), (3, 680, 1200)])
def test_fused_ssim_matches_the_torch_convolutions(gsr, hz, shape):
    g = torch.Generator().manual_seed(3)
    a = torch.rand(shape, generator=g).cuda()
    b = (a + 0.2 * torch.randn(shape, generator=g).cuda()).clamp(0, 1)
    a1, a2 = a.clone().requires_grad_(True), a.clone().requires_grad_(True)
    ref = hz.ssim_torch(a1, b)
    got = hz.ssim(a2, b)
    assert got.is_cuda and abs(float(got.detach()) - float(ref.detach())) <= 2e-6 * max(1.0, abs(float(ref.detach())))
    w = 0.37
    (w * (1.0 - ref)).backward()
    (w * (1.0 - got)).backward()
    scale = float(a1.grad.abs().max())
    assert scale > 0
    assert float((a2.grad - a1.grad).abs().max()) <= 2e-5 * scale       # tolerance: fp32 summation order (121 taps x 5 sums)
    # no gradient requested: no derivative maps, same value
    with torch.no_grad():
        assert abs(float(hz.ssim(a, b)) - float(ref.detach())) <= 2e-6


@pytest.mark.parametrize("name", ["a", "b"])
def test_fused_ssim_matches_the_reference_ssim_golden(gsr, hz, name):
    """tests/golden/ref_loss.npz: loss_utils._ssim of the REFERENCE (imported in the build container) fed GSORB-SLAM's window."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_loss.npz"))
    x = torch.tensor(G[f"{name}_img1"]).cuda().requires_grad_(True)
    v = hz.ssim(x, torch.tensor(G[f"{name}_img2"]).cuda())
    v.backward()
    assert abs(float(v.detach()) - float(G[f"{name}_ssim_gsorb_window"])) <= 3e-6
    gref = G[f"{name}_dssim_dimg1_gsorb_window"]
    assert np.abs(x.grad.cpu().numpy() - gref).max() <= 3e-5 * np.abs(gref).max()


def test_fused_ssim_gradient_reaches_a_non_contiguous_render(gsr, hz):
    """A cropped / permuted render is copied by .contiguous() inside forward(), where grad mode is off: the copy has
    requires_grad False, and the need for a gradient must come from the autograd context, not from that copy."""
    g = torch.Generator().manual_seed(4)
    big = torch.rand((3, 50, 70), generator=g).cuda()
    b = torch.rand((3, 40, 60), generator=g).cuda()
    a1, a2 = big.clone().requires_grad_(True), big.clone().requires_grad_(True)
    crop1, crop2 = a1[:, 5:45, 3:63], a2[:, 5:45, 3:63]
    assert not crop2.is_contiguous()
    (1.0 - hz.ssim_torch(crop1, b)).backward()
    (1.0 - hz.ssim(crop2, b)).backward()
    assert a2.grad is not None and float(a2.grad.abs().max()) > 0
    assert float((a2.grad - a1.grad).abs().max()) <= 2e-5 * float(a1.grad.abs().max())
    hwc = torch.rand((40, 60, 3), generator=g).cuda().requires_grad_(True)     # channels-last storage, permuted view
    (1.0 - hz.ssim(hwc.permute(2, 0, 1), b)).backward()
    assert hwc.grad is not None and float(hwc.grad.abs().max()) > 0


def test_fused_ssim_uses_the_window_it_is_given(gsr, hz):
    """A symmetric and the reference's asymmetric 11-tap window give different losses; each matches its own convolution."""
    g = torch.Generator().manual_seed(5)
    a, b = torch.rand((3, 40, 44), generator=g).cuda(), torch.rand((3, 40, 44), generator=g).cuda()
    taps_ref = [float(x) for x in hz._ssim_taps(11, 1.5, "cpu")]
    sym = torch.tensor([np.exp(-((x - 5) ** 2) / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    sym = [float(x) for x in sym / sym.sum()]
    assert taps_ref != taps_ref[::-1]
    v_ref, v_sym = float(gsr.capi.ssim_mean(a, b, taps_ref)), float(gsr.capi.ssim_mean(a, b, sym))
    assert abs(v_ref - float(hz.ssim_torch(a, b))) <= 2e-6 and abs(v_ref - v_sym) > 1e-5


def test_fused_adam_matches_torch_adam_over_several_steps(gsr):
    g = torch.Generator().manual_seed(11)
    shapes = [(1000, 3), (1001, 1), (7,), (4, 1), (12345, 4)]
    lrs = [1e-4, 2.5e-3, 1e-3, 4e-4, 5e-2]
    init = [torch.randn(s, generator=g).cuda() for s in shapes]
    pa = [t.clone().requires_grad_(True) for t in init]
    pb = [t.clone().requires_grad_(True) for t in init]
    oa = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(pa, lrs)], lr=0.0, eps=1e-15)
    ob = gsr.capi.FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(pb, lrs)], lr=0.0, eps=1e-15)
    for it in range(6):
        grads = [torch.randn(s, generator=g).cuda() * (10.0 ** (it % 3 - 1)) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if it == 3:
            pa[2].grad = pb[2].grad = None            # a parameter without a gradient keeps its step count
        oa.step(); ob.step()
    for p, q in zip(pa, pb):
        d = (p - q).abs().max()
        assert float(d) <= 2e-6 * float(p.abs().max()), float(d)
        sa, sb = oa.state[p], ob.state[q]
        assert float(sa["step"]) == float(sb["step"])
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-12)
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-12)


def test_harness_map_uses_the_fused_optimiser_and_keeps_state_through_surgery(gsr, hz):
    m = hz.GaussianMap(hz.Config(), 300.0, 300.0, device="cuda")
    pts = torch.rand(500, 3) + torch.tensor([0.0, 0.0, 1.0])
    m.add_points(pts, torch.rand(500, 3))
    assert isinstance(m.opt, gsr.capi.FusedAdam)
    for n in m.NAMES:
        getattr(m, n).grad = torch.randn_like(getattr(m, n))
    m.opt.step()
    before = m.opt.state[m.xyz]["exp_avg"].clone()
    m.add_points(pts[:50] + 0.01, torch.rand(50, 3))                     # CatTensorToOptimizer
    st = m.opt.state[m.xyz]
    assert st["exp_avg"].shape[0] == 550 and torch.equal(st["exp_avg"][:500], before) and float(st["exp_avg"][500:].abs().max()) == 0
    keep_mask = torch.zeros(550, dtype=torch.bool, device="cuda"); keep_mask[::2] = True
    m.prune(keep_mask)                                                   # PruneOptimizer
    assert m.opt.state[m.xyz]["exp_avg"].shape[0] == 275
    for n in m.NAMES:
        getattr(m, n).grad = torch.randn_like(getattr(m, n))
    m.opt.step()
    assert float(m.opt.state[m.xyz]["step"]) == 2


def test_to_camera_pose_gradient_matches_autograd_through_the_product(gsr):
    """N1 of the verdict's table: dL/dTcw through mc = X R^T + t. The fused reduction against autograd through the plain
    product (what the reference does, Render.cc:750-752), in float64 for the comparison's reference."""
    g = torch.Generator().manual_seed(2)
    n = 300_001
    X = (torch.randn((n, 3), generator=g) * 2).cuda()
    T = torch.eye(4); T[:3, :3] = torch.linalg.qr(torch.randn((3, 3), generator=g))[0]; T[:3, 3] = torch.randn(3, generator=g)
    T = T.cuda()
    w = torch.randn((n, 3), generator=g).cuda()
    Ta, Xa = T.clone().requires_grad_(True), X.clone().requires_grad_(True)
    mc = gsr.capi.to_camera(Ta, Xa)
    (mc * w).sum().backward()
    Tb, Xb = T.double().clone().requires_grad_(True), X.double().clone().requires_grad_(True)
    ref = Xb @ Tb[:3, :3].t() + Tb[:3, 3]
    (ref * w.double()).sum().backward()
    assert float((mc.double() - ref).abs().max()) <= 1e-5
    assert float((Xa.grad.double() - Xb.grad).abs().max()) <= 1e-5
    scale = float(Tb.grad.abs().max())
    assert float((Ta.grad.double() - Tb.grad).abs().max()) <= 2e-5 * scale   # fp32 sums of 3e5 terms
    assert float(Ta.grad[3].abs().max()) == 0.0


def test_rt2T_kernel_matches_the_tensor_formulas_and_their_autograd(gsr, hz):
    g = torch.Generator().manual_seed(4)
    for _ in range(5):
        q = (torch.randn((4, 1), generator=g) * 1.7)
        t = torch.randn((3, 1), generator=g)
        w = torch.randn((4, 4), generator=g)
        qa, ta = q.double().clone().requires_grad_(True), t.double().clone().requires_grad_(True)
        Ta = hz.rt2T(qa, ta)                                   # CPU double: the tensor formulas (include/Utils.h:56-77)
        (Ta * w.double()).sum().backward()
        qb, tb = q.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
        Tb = hz.rt2T(qb, tb)
        (Tb * w.cuda()).sum().backward()
        assert float((Tb.cpu().double() - Ta).abs().max()) <= 1e-6
        assert float((qb.grad.cpu().double() - qa.grad).abs().max()) <= 1e-5 * max(1.0, float(qa.grad.abs().max()))
        assert float((tb.grad.cpu().double() - ta.grad).abs().max()) <= 1e-6


# ---- the fused loss terms of the loops (include/gsr.h: gsr_pixel_loss*, gsr_scale_reg*) against the tensor expressions of
# ---- SlamLoop.cpp / harness.py (= src/Render.cc:1088-1105, :436-471, :449-462) in float64
def _loss_inputs(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand((3, H, W), generator=g)
    dep = 1.0 + 2.0 * torch.rand((H, W), generator=g)
    sur = dep + 0.05 * torch.randn((H, W), generator=g)
    sil = 0.9 + 0.1 * torch.rand((H, W), generator=g) + 0.02     # ~ a fifth of the pixels under the 0.99 threshold
    frgb = torch.rand((3, H, W), generator=g)
    fd = 1.0 + 2.0 * torch.rand((H, W), generator=g)
    img[:, 3, 5] = frgb[:, 3, 5]                                 # |0|: gradient 0, like torch.abs
    return img, dep, sur, sil, frgb, fd


@pytest.mark.parametrize("shape,surface", [((37, 53), False), ((480, 640), False), ((120, 200), True)])
def test_fused_tracking_loss_matches_the_masked_l1_sums(gsr, shape, surface):
    H, W = shape
    img, dep, sur, sil, frgb, fd = _loss_inputs(H, W, 3)
    fd[::7, ::5] = float("nan")                                  # invalid sensor depth (Render.cc:1088: ~isnan)
    wi, wd = 0.5, 1.0
    a = img.double().requires_grad_(True); d = (sur if surface else dep).double().requires_grad_(True)
    certain = (sil.double() > 0.99) & ~torch.isnan(fd)
    ref = wi * torch.where(certain.unsqueeze(0).expand(3, H, W), (a - frgb.double()).abs(), torch.zeros_like(a)).sum() \
        + wd * torch.where(certain, (d - fd.double()).abs(), torch.zeros_like(d)).sum()
    ref.backward()
    ai = img.cuda().requires_grad_(True); di = (sur if surface else dep).cuda().requires_grad_(True)
    out = gsr.capi.tracking_pixel_loss(ai, di, sil.cuda(), frgb.cuda(), fd.cuda(), wi, wd, depth_is_surface=surface)
    (out * 2.0).backward()                                       # an upstream gradient other than 1
    assert abs(float(out) - float(ref)) <= 2e-5 * abs(float(ref))
    assert torch.equal(ai.grad.cpu(), (2.0 * a.grad).float())
    if surface:
        assert di.grad is None                                   # the median-depth plane has no gradient
    else:
        assert torch.equal(di.grad.cpu(), (2.0 * d.grad).float())


@pytest.mark.parametrize("shape", [(680, 1200), (37, 53), (5, 7), (48, 53)])   # 1024 / 8 / 1 / 10 workgroups: every shape of the arrival counters
@pytest.mark.parametrize("surface", [False, True])
def test_one_pass_tracking_loss_equals_the_two_kernels(gsr, surface, shape):
    """gsr_track_loss (the direct tracking loop's: sums and gradient planes from one pass over the render) against gsr_pixel_loss +
    gsr_pixel_loss_backward_add, and against the float64 tensor expression."""
    import ctypes as C
    H, W = shape
    img, dep, sur, sil, frgb, fd = (t.cuda().contiguous() for t in _loss_inputs(H, W, 5))
    fd[::7, ::5] = float("nan")
    L = gsr.lib(); p = gsr.capi._p; st = gsr.capi._stream
    w3 = (C.c_float * 3)(0.5, 1.25, 0.0)
    z = lambda *sh: torch.full(sh, 7.0, device="cuda")               # (garbage in every output)
    part_a, sums_a, gi_a, gd_a = z(1024 * 5), z(8), z(3, H, W), z(H, W)
    part_b, sums_b, gi_b, gd_b = z(1024 * 5), z(8), z(3, H, W), z(H, W)
    d_ptr, s_ptr = (None, p(sur)) if surface else (p(dep), None)
    gsr.capi._check(L.gsr_pixel_loss(p(img), d_ptr, s_ptr, p(sil), p(frgb), p(fd), H, W, 0, 0.99, w3, p(part_a), p(sums_a), st()))
    gsr.capi._check(L.gsr_pixel_loss_backward_add(p(img), d_ptr, p(sil), p(frgb), p(fd), H, W, 0, 0.99, w3, p(sums_a), None, None, p(gi_a), p(gd_a), st()))
    gsr.capi._check(L.gsr_track_loss(p(img), d_ptr, s_ptr, p(sil), p(frgb), p(fd), H, W, 0.99, w3, p(part_b), p(sums_b), p(gi_b), p(gd_b), None, st()))
    torch.cuda.synchronize()
    assert torch.equal(sums_a, sums_b) and torch.equal(gi_a, gi_b) and torch.equal(gd_a, gd_b)
    # the sums finished inside the same launch (a ticket word that is zero between calls): three calls in a row, the word is put back every time
    ticket = torch.zeros(144, dtype=torch.int32, device="cuda")   # GSR_TICKET_WORDS
    for _ in range(3):
        sums_c, part_c = z(8), z(1024 * 5)
        gsr.capi._check(L.gsr_track_loss(p(img), d_ptr, s_ptr, p(sil), p(frgb), p(fd), H, W, 0.99, w3, p(part_c), p(sums_c), p(gi_b), p(gd_b), p(ticket), st()))
        torch.cuda.synchronize()
        assert torch.equal(sums_a, sums_c) and int(ticket.abs().sum()) == 0
    certain = (sil.double() > 0.99) & ~torch.isnan(fd)
    dd = (sur if surface else dep).double()
    ref = 0.5 * torch.where(certain.unsqueeze(0).expand(3, H, W), (img.double() - frgb.double()).abs(), torch.zeros(3, H, W, device="cuda", dtype=torch.float64)).sum() \
        + 1.25 * torch.where(certain, (dd - fd.double()).abs(), torch.zeros_like(dd)).sum()
    assert abs(float(sums_b[5]) - float(ref)) <= 2e-5 * abs(float(ref))
    assert float(gd_b.abs().max()) in ((0.0,) if surface else (0.0, 1.25))


@pytest.mark.parametrize("shape", [(37, 53), (680, 1200)])
def test_fused_mapping_pixel_loss_matches_the_masked_means(gsr, shape):
    H, W = shape
    img, dep, sur, sil, frgb, fd = _loss_inputs(H, W, 4)
    fd[::3, ::4] = 0.0                                           # pixels without a depth measurement
    w1, w2, w3 = 0.45, 1.0, 0.3
    a = img.double().requires_grad_(True); d = dep.double().requires_grad_(True)
    valid = fd.double() > 0
    valid_sur = valid & (sil.double() > 0.99)
    ref = w1 * (a - frgb.double()).abs().mean() + w2 * torch.where(valid, (d - fd.double()).abs(), torch.zeros_like(d)).sum() / valid.sum() \
        + w3 * torch.where(valid_sur, (sur.double() - fd.double()).abs(), torch.zeros_like(d)).sum() / valid_sur.sum().clamp_min(1)
    ref.backward()
    ai = img.cuda().requires_grad_(True); di = dep.cuda().requires_grad_(True)
    out, sums = gsr.capi.mapping_pixel_loss(ai, di, sur.cuda(), sil.cuda(), frgb.cuda(), fd.cuda(), w1, w2, w3)
    out.backward()
    assert abs(float(out) - float(ref)) <= 2e-5 * abs(float(ref))
    assert int(sums[2]) == int(valid.sum()) and int(sums[4]) == int(valid_sur.sum())
    assert (ai.grad.cpu().double() - a.grad).abs().max() <= 1e-6 * a.grad.abs().max()
    assert (di.grad.cpu().double() - d.grad).abs().max() <= 1e-6 * d.grad.abs().max()
    # no surface pixel at all: the term is an exact zero (the deviation from the reference's NaN that harness.py documents)
    out0, sums0 = gsr.capi.mapping_pixel_loss(ai.detach(), di.detach(), sur.cuda(), torch.zeros_like(sil).cuda(), frgb.cuda(), fd.cuda(), 0.0, 0.0, 1.0)
    assert float(out0) == 0.0 and float(sums0[4]) == 0.0


@pytest.mark.parametrize("n,frac", [(1000, 0.1), (300_000, 0.01), (5000, 0.0)])
def test_fused_scale_regularisers_match_the_tensor_expressions(gsr, n, frac):
    g = torch.Generator().manual_seed(n)
    ls = torch.log(0.01 + 0.02 * torch.rand((n, 3), generator=g))
    big = torch.rand((n,), generator=g) < frac
    ls[big, 0] = torch.log(torch.tensor(0.5)) + 0.3 * torch.rand((int(big.sum()),), generator=g)   # one oversized axis ...
    ls[big & (torch.arange(n) % 2 == 0), 2] = torch.log(torch.tensor(0.4))                        # ... or two
    limit, wl, ws = 0.3, 0.7, 0.2
    x = ls.double().requires_grad_(True)
    sc = torch.exp(x)
    w = (sc > limit).sum(1).to(sc.dtype)
    mx, mn = sc.max(1)[0], sc.min(1)[0]
    cnt = w.sum()
    reg_scalar = (w * (mx - limit)).sum()
    spread = (w * (mx - mn)).sum()
    reg_long = torch.where(cnt > 0, spread / cnt.clamp_min(1), torch.zeros_like(spread))
    ref = wl * reg_long + ws * reg_scalar
    ref.backward()
    xi = ls.cuda().requires_grad_(True)
    val, out = gsr.capi.scale_regularisers(xi, limit, wl, ws)
    (val * 3.0).backward()
    assert float(out[0]) == float(cnt)
    assert abs(float(val) - float(ref)) <= 2e-5 * max(abs(float(ref)), 1e-12)
    assert (xi.grad.cpu().double() - 3.0 * x.grad).abs().max() <= 2e-6 * max(float(x.grad.abs().max()) * 3.0, 1e-12)


# ---- the direct loops' kernels (round 4: gsr_map_prepare / gsr_map_update / gsr_pose_update / gsr_map_loss_total) ----------------
def _pose_T(seed=0):
    from util import pose
    return torch.tensor(pose(0.05 + 0.01 * seed, (0.03, -0.02, 0.04)), dtype=torch.float32)


@pytest.mark.parametrize("n", [1, 257, 5000])
def test_map_prepare_matches_the_tensor_expressions(gsr, n):
    """gsr_map_prepare against what the reference forms with libtorch (Render.cc:750-758: camera transform, sigmoid, exp, normalize)
    and against gsr_scale_reg's regulariser sums of the same log-scales."""
    g = torch.Generator().manual_seed(n)
    xyz = torch.randn((n, 3), generator=g); logit = torch.randn((n, 1), generator=g) * 2
    ls = torch.log(0.01 + 0.3 * torch.rand((n, 3), generator=g)); q = torch.randn((n, 4), generator=g)
    T = _pose_T()
    limit, wl, ws = 0.2, 0.7, 0.3
    mc, op, sc, rot, out = gsr.capi.map_prepare(xyz.cuda(), logit.cuda(), ls.cuda(), q.cuda(), T.cuda(), reg=(limit, wl, ws))
    ref_mc = xyz.double() @ T[:3, :3].double().T + T[:3, 3].double()
    assert (mc.cpu().double() - ref_mc).abs().max() < 2e-6 * max(float(ref_mc.abs().max()), 1.0)
    assert (op.cpu() - torch.sigmoid(logit[:, 0])).abs().max() < 2e-7
    assert ((sc.cpu() - torch.exp(ls)).abs() / torch.exp(ls)).max() < 1e-6
    assert (rot.cpu() - torch.nn.functional.normalize(q)).abs().max() < 2e-7
    val, out2 = gsr.capi.scale_regularisers(ls.cuda(), limit, wl, ws)
    assert float(out[0]) == float(out2[0])
    assert (out.cpu() - out2.cpu()).abs().max() <= 2e-5 * max(float(out2.abs().max()), 1e-12)


@pytest.mark.parametrize("with_reg,n", [(False, 3003), (True, 3003), (True, (1 << 18) + 3)])
def test_map_update_is_autograd_through_the_activations_plus_adam(gsr, with_reg, n):
    """gsr_map_update against float64 autograd through the camera transform / sigmoid / exp / normalize (and the regularisers) followed
    by torch.optim.Adam (eps 1e-15) on the five raw tensors, over three steps with changing upstream gradients."""
    # (n < 2^18: one Gaussian per thread; above: four per thread as whole float4s, and the tail of a count that is not a multiple of four)
    g = torch.Generator().manual_seed(7)
    raw = [torch.randn((n, 3), generator=g), torch.rand((n, 3), generator=g), torch.randn((n, 4), generator=g),
           torch.randn((n, 1), generator=g), torch.log(0.01 + 0.3 * torch.rand((n, 3), generator=g))]
    lrs = [1e-4, 2.5e-3, 1e-3, 5e-2, 1e-3]
    limit, wl, ws = 0.2, 5.0, 10.0
    T = _pose_T(1)
    ref = [t.double().clone().requires_grad_(True) for t in raw]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], eps=1e-15)
    dev = [t.cuda().contiguous() for t in raw]
    m = [torch.zeros_like(t) for t in dev]; v = [torch.zeros_like(t) for t in dev]
    for step in range(1, 4):
        ups = [torch.randn((n, 3), generator=g), torch.randn((n, 3), generator=g), torch.randn((n, 4), generator=g),
               torch.randn((n,), generator=g), torch.randn((n, 3), generator=g)]   # dL/d(mc, colour, rot, opac, scale)
        xyz, rgb, q, logit, ls = ref
        mc = xyz @ T[:3, :3].double().T + T[:3, 3].double()
        loss = (mc * ups[0].double()).sum() + (rgb * ups[1].double()).sum() + (torch.nn.functional.normalize(q) * ups[2].double()).sum() \
            + (torch.sigmoid(logit[:, 0]) * ups[3].double()).sum() + (torch.exp(ls) * ups[4].double()).sum()
        if with_reg:
            sc = torch.exp(ls)
            w = (sc > limit).sum(1).to(sc.dtype); mx, mn = sc.max(1)[0], sc.min(1)[0]; cnt = w.sum()
            loss = loss + wl * torch.where(cnt > 0, (w * (mx - mn)).sum() / cnt.clamp_min(1), torch.zeros_like(cnt)) + ws * (w * (mx - limit)).sum()
        opt.zero_grad(); loss.backward(); opt.step()
        outs = gsr.capi.map_prepare(dev[0], dev[3], dev[4], dev[2], T.cuda(), reg=(limit, wl, ws) if with_reg else None)
        gsr.capi.map_update(dev, (m, v), [u.cuda().contiguous() for u in ups], (outs[1], outs[2]), T.cuda(), lrs, [step] * 5,
                            reg=(outs[4], limit, wl, ws) if with_reg else None)
    for name, a, b, lr in zip(("xyz", "rgb", "quat", "logit", "log_scales"), dev, ref, lrs):
        moved = (b.detach() - raw[("xyz", "rgb", "quat", "logit", "log_scales").index(name)].double()).abs().max()
        assert moved > lr                                                               # three Adam steps of ~lr each
        assert (a.cpu().double() - b.detach()).abs().max() < 2e-3 * float(moved), name  # (sign flips of tiny gradients aside: Adam's m / sqrt(v) is +-1 at step 1)


def test_map_update_skips_the_step_of_an_overflowed_forward(gsr, syn):
    """The predicate of the direct loops: a forward that ran out of workspace leaves its overflow flag in the geometry blob and
    gsr_map_update must then leave the parameters and the moments alone; gsr_map_loss_total reports NaN."""
    cam = syn.make_camera(160, 120, 130.0, 130.0)
    sc = syn.make_scene(3000, cam, seed=1)
    s = gsr.capi.Settings.from_camera(cam)
    ws = gsr.capi.Workspace(3000, 160, 120, 64)        # far too small
    gsr.forward_ws(s, ws, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    torch.cuda.synchronize()
    assert ws.status()[1]
    n = 3000
    dev = [torch.randn((n, k), device="cuda") for k in (3, 3, 4, 1, 3)]
    before = [x.clone() for x in dev]
    m = [torch.zeros_like(x) for x in dev]; v = [torch.zeros_like(x) for x in dev]
    T = torch.eye(4, device="cuda")
    outs = gsr.capi.map_prepare(dev[0], dev[3], dev[4], dev[2], T)
    ups = [torch.randn((n, 3), device="cuda"), torch.randn((n, 3), device="cuda"), torch.randn((n, 4), device="cuda"), torch.randn((n,), device="cuda"),
           torch.randn((n, 3), device="cuda")]
    gsr.capi.map_update(dev, (m, v), ups, (outs[1], outs[2]), T, [1e-2] * 5, [1] * 5, geom=ws.geom)
    assert all(torch.equal(a, b) for a, b in zip(dev, before)) and all(float(x.abs().max()) == 0.0 for x in m + v)
    gsr.capi.map_update(dev, (m, v), ups, (outs[1], outs[2]), T, [1e-2] * 5, [1] * 5, geom=None)
    assert not torch.equal(dev[0], before[0])
    loss = torch.zeros((1,), device="cuda"); sums = torch.ones((8,), device="cuda")
    gsr.capi._check(gsr.lib().gsr_map_loss_total(sums.data_ptr(), None, 0, 0, 0.0, None, ws.geom.data_ptr(), loss.data_ptr(), None))
    assert bool(torch.isnan(loss).all())
    gsr.capi._check(gsr.lib().gsr_map_loss_total(sums.data_ptr(), None, 0, 0, 0.0, None, None, loss.data_ptr(), None))
    assert float(loss) == 1.0


def test_pose_update_is_rt2T_backward_plus_adam_and_keeps_the_best_pose(gsr, hz):
    """gsr_pose_update against float64 autograd through harness.rt2T + torch.optim.Adam on (quat, trans), the best-pose bookkeeping of
    Render.cc:1107-1112 (NaN never wins) and the pose matrix it leaves for the next iteration."""
    g = torch.Generator().manual_seed(3)
    q0 = torch.tensor([0.9, 0.1, -0.2, 0.05]); t0 = torch.tensor([0.1, -0.2, 0.3])
    qr = q0.double().reshape(4, 1).clone().requires_grad_(True); tr = t0.double().reshape(3, 1).clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [qr], "lr": 4e-4}, {"params": [tr], "lr": 4e-4}], eps=1e-15)
    pose = torch.cat([q0, t0]).cuda(); mom = torch.zeros(14, device="cuda"); best = torch.zeros(8, device="cuda"); best[0] = float("inf")
    hist = torch.zeros(4, device="cuda"); Tcw = torch.zeros(16, device="cuda")
    losses = [3.0, 2.0, float("nan"), 2.5]
    poses_before = []
    for it, lv in enumerate(losses):
        G = torch.randn((512, 12), generator=g) * 0.01                       # gsr_pose_grad's rows: dL/dR row-major, dL/dt
        s = G.double().sum(0)
        T = hz.rt2T(qr, tr)
        opt.zero_grad()
        ((T[:3, :3] * s[:9].reshape(3, 3)).sum() + (T[:3, 3] * s[9:]).sum()).backward()
        poses_before.append(torch.cat([qr.detach().reshape(4), tr.detach().reshape(3)]).clone())
        opt.step()
        gsr.capi.pose_update(pose, mom, best, hist[it:], Tcw, G.cuda().contiguous(), torch.tensor([lv], device="cuda"), 4e-4, it + 1)
        ref = torch.cat([qr.detach().reshape(4), tr.detach().reshape(3)])
        assert (pose.cpu().double() - ref).abs().max() < 2e-6, it
        assert (Tcw.cpu().double().reshape(4, 4) - hz.rt2T(qr.detach(), tr.detach())).abs().max() < 2e-6
    h = hist.cpu()
    assert float(h[0]) == 3.0 and float(h[1]) == 2.0 and bool(torch.isnan(h[2])) and float(h[3]) == 2.5
    assert float(best[0]) == 2.0 and (best[1:].cpu().double() - poses_before[1]).abs().max() < 2e-6


def test_pose_step_equals_pose_grad_plus_pose_update(gsr):
    """gsr_pose_step (the pose sums and the step in one launch: the last workgroup of the sums takes the step) against gsr_pose_grad followed by
    gsr_pose_update on the same state, four iterations (a NaN loss among them): bit-identical pose, moments, best pose, history and matrix."""
    import ctypes as C
    L = gsr.lib(); p = gsr.capi._p; st = gsr.capi._stream
    g = torch.Generator().manual_seed(11)
    n = 300_001
    X = torch.randn((n, 3), generator=g).cuda()
    q0 = torch.tensor([0.9, 0.1, -0.2, 0.05]); t0 = torch.tensor([0.1, -0.2, 0.3])
    mk = lambda: dict(pose=torch.cat([q0, t0]).cuda(), mom=torch.zeros(14, device="cuda"), best=torch.tensor([float("inf")] + [0.0] * 7, device="cuda"),
                      hist=torch.zeros(4, device="cuda"), Tcw=torch.eye(4, device="cuda").reshape(16).clone(), part=torch.full((512, 12), 9.0, device="cuda"))
    A, B = mk(), mk()
    ticket = torch.zeros(144, dtype=torch.int32, device="cuda")   # GSR_TICKET_WORDS
    for it, lv in enumerate([3.0, 2.0, float("nan"), 2.5]):
        dmc = (torch.randn((n, 3), generator=g) * 1e-3).cuda()
        loss = torch.tensor([lv], device="cuda")
        gsr.capi._check(L.gsr_pose_grad(p(X), p(dmc), n, p(A["Tcw"]), p(A["part"]), None, st()))
        gsr.capi.pose_update(A["pose"], A["mom"], A["best"], A["hist"][it:], A["Tcw"], A["part"], loss, 4e-4, it + 1)
        a = gsr.capi.PoseUpdateArgs(p(B["pose"]), p(B["mom"]), p(B["best"]), p(B["hist"][it:]), p(B["Tcw"]), p(B["part"]), p(loss), None, 4e-4, 0.9, 0.999, 1e-15, it + 1)
        gsr.capi._check(L.gsr_pose_step(p(X), p(dmc), n, C.byref(a), p(ticket), st()))
        torch.cuda.synchronize()
        assert int(ticket.abs().sum()) == 0
        for k in ("pose", "mom", "best", "Tcw"):
            assert torch.equal(A[k], B[k]), (it, k)
    assert torch.equal(A["hist"].isnan(), B["hist"].isnan()) and torch.equal(A["hist"].nan_to_num(0.0), B["hist"].nan_to_num(0.0))


def test_backward_with_the_pose_step_inside_equals_backward_plus_pose_step(gsr, syn):
    """gsr_backward_args.fused_pose_step (a tracking iteration: the per-splat stage adds the pose sums to accumulator rows, a one-wave kernel behind it takes the pose step) against
    gsr_backward writing dL_dmean3D followed by gsr_pose_step: the same pose, moments, best pose, history and matrix up to the order of the sums,
    over three iterations on one workspace (the arrival counters go back to zero every time); with culled splats and a map size that is not a multiple of 256."""
    import ctypes as C
    cam = syn.make_camera(width=203, height=149, fx=150.0, fy=152.0)
    sc = syn.make_scene(30_001, cam, seed=5, scale_mult=2.0, frac_behind=0.1, frac_offscreen=0.2)
    s = gsr.capi.Settings.from_camera(sc.cam)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda").contiguous()
    X = t(sc.means3D) + 0.01                                                   # (the "world-frame" means: any [P,3] tensor)
    p = gsr.capi._p
    q0 = torch.tensor([0.9, 0.1, -0.2, 0.05]); t0 = torch.tensor([0.1, -0.2, 0.3])
    mk = lambda: dict(pose=torch.cat([q0, t0]).cuda(), mom=torch.zeros(14, device="cuda"), best=torch.tensor([float("inf")] + [0.0] * 7, device="cuda"),
                      hist=torch.zeros(4, device="cuda"), Tcw=torch.eye(4, device="cuda").reshape(16).clone(),
                      part=torch.full((512, 12), 9.0, device="cuda"))
    A, B, D = mk(), mk(), mk()
    B["part"].zero_()                                                          # (the fused step's accumulator rows: zero between calls)
    D["part"].zero_()                                                          # (third leg: the sums alone out of the backward, the step by gsr_pose_finish — a sharded run's form)
    sums = torch.zeros(12, device="cuda")
    tickets = torch.zeros(144, dtype=torch.int32, device="cuda")
    for it, lv in enumerate([3.0, float("nan"), 2.5]):
        g = torch.Generator().manual_seed(it)
        dpix = t(torch.randn((3, sc.cam.height, sc.cam.width), generator=g).numpy())
        loss = torch.tensor([lv], device="cuda")
        st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
        gr = gsr.backward(st, dpix)
        a = gsr.capi.PoseUpdateArgs(p(A["pose"]), p(A["mom"]), p(A["best"]), p(A["hist"][it:]), p(A["Tcw"]), p(A["part"]), p(loss), None, 4e-4, 0.9, 0.999, 1e-15, it + 1)
        gsr.capi._check(gsr.lib().gsr_pose_step(p(X), p(gr.dL_dmeans3D), sc.P, C.byref(a), p(tickets), gsr.capi._stream()))
        b = gsr.capi.PoseUpdateArgs(p(B["pose"]), p(B["mom"]), p(B["best"]), p(B["hist"][it:]), p(B["Tcw"]), p(B["part"]), p(loss), None, 4e-4, 0.9, 0.999, 1e-15, it + 1)
        ps = gsr.capi.PoseStepArgs(p(X), C.cast(C.pointer(b), C.c_void_p), 0)
        st2 = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
        gr2 = gsr.backward(st2, dpix, fused_pose_step=ps)
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0 and float(B["part"][:64].abs().sum()) == 0.0
        assert (gr2.dL_dmeans3D - gr.dL_dmeans3D).abs().max() <= 1e-5 * gr.dL_dmeans3D.abs().max()      # (still written when given; atomics' order only)
        for k in ("pose", "mom", "Tcw", "best"):
            assert (A[k] - B[k]).abs().max() <= 1e-5 * max(1e-6, float(A[k].abs().max())), (it, k)
        d = gsr.capi.PoseUpdateArgs(p(D["pose"]), p(D["mom"]), p(D["best"]), p(D["hist"][it:]), p(D["Tcw"]), p(D["part"]), p(loss), None, 4e-4, 0.9, 0.999, 1e-15, it + 1)
        pd = gsr.capi.PoseStepArgs(p(X), C.cast(C.pointer(d), C.c_void_p), 1)
        pose_before = D["pose"].clone()
        st3 = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
        gsr.backward(st3, dpix, fused_pose_step=pd)
        torch.cuda.synchronize()
        assert torch.equal(D["pose"], pose_before) and float(D["part"][:64].abs().sum()) > 0.0        # (no step yet: the rows hold the sums)
        rows = D["part"][:64].sum(0)
        gsr.capi._check(gsr.lib().gsr_pose_finish(C.byref(d), p(D["part"]), p(sums), gsr.capi._stream()))
        torch.cuda.synchronize()
        assert float(D["part"][:64].abs().sum()) == 0.0 and (sums - rows).abs().max() <= 1e-5 * max(1e-6, float(rows.abs().max()))
        for k in ("pose", "mom", "Tcw", "best"):
            assert (A[k] - D[k]).abs().max() <= 1e-5 * max(1e-6, float(A[k].abs().max())), (it, k, "sums_only + gsr_pose_finish")
    assert torch.equal(A["hist"].isnan(), B["hist"].isnan()) and torch.equal(A["hist"].nan_to_num(0.0), B["hist"].nan_to_num(0.0))
    assert torch.equal(A["hist"].isnan(), D["hist"].isnan()) and torch.equal(A["hist"].nan_to_num(0.0), D["hist"].nan_to_num(0.0))


def _fused_update_setup(gsr, syn, n=20_000):
    """A fused-pair forward and a raw parameter set whose activations are the scene's (so that the rasterizer inputs and the update's
    parameters belong together): returns a function that runs one backward with fused_map_update and hands back the stepped raw tensors."""
    cam = syn.make_camera(203, 149, 150.0, 152.0)
    sc = syn.make_scene(n, cam, seed=3, scale_mult=2.0)
    s = gsr.capi.Settings.from_camera(cam)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda").contiguous()
    g = torch.Generator().manual_seed(5)
    gA = t(torch.randn((3, 149, 203), generator=g)); gB = t(torch.randn((2, 149, 203), generator=g)); gB[1] = 0.0
    T = torch.eye(4, device="cuda")
    opac = t(sc.opacities).reshape(-1).clamp(1e-4, 1 - 1e-4)
    raw0 = [t(sc.means3D), t(sc.colors), t(sc.rotations), torch.log(opac / (1 - opac)).reshape(-1, 1).contiguous(), torch.log(t(sc.scales))]

    def run(detach, grads, n_override=None, also_pose=False, lrs=(1e-4, 2.5e-3, 1e-3, 5e-2, 1e-3)):
        raw = [x.clone() for x in raw0]
        m = [torch.zeros_like(x) for x in raw]; v = [torch.zeros_like(x) for x in raw]
        mc, op, scl, rot = gsr.capi.map_prepare(raw[0], raw[3], raw[4], raw[2], T)
        st = gsr.forward(s, mc, op, colors=raw[1], scales=scl, rotations=rot, dual=True)
        a = gsr.capi.map_update_args(raw, (m, v), None, (op, scl), T, lrs, [1] * 5)
        if n_override is not None:
            a.n = n_override
        kw = {}
        if also_pose:
            import ctypes as C
            z = lambda k: torch.zeros(k, device="cuda")
            pu = gsr.capi.PoseUpdateArgs(*(gsr.capi._p(x) for x in (z(7), z(14), z(8), z(4), z(16), z(64 * 12), z(1))), None, 4e-4, 0.9, 0.999, 1e-15, 1)
            kw["fused_pose_step"] = gsr.capi.PoseStepArgs(gsr.capi._p(raw[0]), C.cast(C.pointer(pu), C.c_void_p))
        gsr.backward(st, gA, grads=grads, dL_dds=gB, detach_depth_color=detach, once=True, fused_map_update=a, **kw)
        torch.cuda.synchronize()
        return raw, st
    return run, raw0


def test_fused_map_update_steps_the_colours_when_the_depth_colour_is_detached(gsr, syn):
    """with fused_map_update the per-splat stage steps Adam on rgb from the blend stage's colour sums although no dL_dcolor buffer
    is handed over. With ds_detach_depth = 1 and no gradient buffers at all the call used to pick the blend kernel WITHOUT those sums
    (the tracking variant) and step the colours with a zero gradient. The colour update must not depend on the detach flag."""
    run, raw0 = _fused_update_setup(gsr, syn)
    none = gsr.capi.Grads(*([None] * 9))
    a, _ = run(False, none)
    b, _ = run(True, none)
    moved = float((a[1] - raw0[1]).abs().max())
    assert moved > 1e-3                                              # one Adam step of lr 2.5e-3 on the colours
    assert float((b[1] - raw0[1]).abs().max()) > 1e-3                # ... also with the depth channel's colour detached
    assert float((a[1] - b[1]).abs().max()) <= 1e-3 * moved          # (sign flips of ~0 gradients aside: the first Adam step is +-lr)
    for k in (2, 3, 4):                                              # rotations, opacity, scales see no depth-colour term either
        assert float((a[k] - b[k]).abs().max()) <= 2e-2 * float((a[k] - raw0[k]).abs().max())
    assert float((a[0] - b[0]).abs().max()) > 0.0                    # the means do: that is what the flag detaches


def test_backward_rejects_bad_fused_arguments_before_it_launches_anything(gsr, syn):
    """fused_map_update / fused_pose_step are validated at the top of gsr_backward. A rejected call leaves the accumulators as
    the forward left them: the same state then takes a correct backward (no double count of the blend stage); both fused steps at once
    are an error."""
    run, raw0 = _fused_update_setup(gsr, syn, n=5000)
    none = gsr.capi.Grads(*([None] * 9))
    good, _ = run(False, none)
    with pytest.raises(Exception):
        run(False, none, also_pose=True)
    cam = syn.make_camera(203, 149, 150.0, 152.0)
    # a wrong n: EINVAL, then the SAME forward state is taken back by a plain backward and must equal a clean one
    sc = syn.make_scene(5000, cam, seed=3, scale_mult=2.0)
    s = gsr.capi.Settings.from_camera(cam)
    g = torch.Generator().manual_seed(5)
    gA = torch.randn((3, 149, 203), generator=g).cuda()
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    ref = gsr.backward(st, gA)
    st2 = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda").contiguous()
    raw = [t(sc.means3D), t(sc.colors), t(sc.rotations), torch.zeros((5000, 1), device="cuda"), torch.log(t(sc.scales))]
    m = [torch.zeros_like(x) for x in raw]; v = [torch.zeros_like(x) for x in raw]
    a = gsr.capi.map_update_args(raw, (m, v), None, (t(sc.opacities).reshape(-1), t(sc.scales)), torch.eye(4, device="cuda"), [1e-3] * 5, [1] * 5)
    a.n = 4999
    before = [x.clone() for x in raw]
    with pytest.raises(Exception):
        gsr.backward(st2, gA, fused_map_update=a)
    st2.dirty = False                                                # (the rejected call launched nothing: the accumulators are still clean)
    out = gsr.backward(st2, gA)
    assert all(torch.equal(x, y) for x, y in zip(raw, before))
    for nme in ("dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales"):
        x, y = getattr(out, nme), getattr(ref, nme)
        assert float((x - y).abs().max()) <= 2e-6 * float(y.abs().max()), nme


@pytest.mark.parametrize("shape", [(37, 53), (680, 1200)])
def test_fused_mapping_loss_equals_its_separate_kernels(gsr, hz, shape):
    """gsr_map_loss_forward / _finish / _backward (SSIM and the pixel terms of the mapping loss in the same two passes, one finish kernel that
    also closes the regularisers and forms the iteration's loss) against the kernels they replace: capi.mapping_pixel_loss, capi.ssim_mean (both
    through autograd) and capi.scale_regularisers."""
    import ctypes as C
    H, W = shape
    g = torch.Generator().manual_seed(H)
    image = torch.rand((3, H, W), generator=g).cuda().requires_grad_(True)
    frgb = torch.rand((3, H, W), generator=g).cuda()
    fd = (0.5 + 3 * torch.rand((H, W), generator=g)); fd[::5, ::3] = 0.0; fd = fd.cuda()
    depth = (fd + 0.1 * torch.randn((H, W), generator=g).cuda()).requires_grad_(True)
    sur = (fd + 0.1 * torch.randn((H, W), generator=g).cuda())
    sil = torch.rand((H, W), generator=g).cuda() * 0.2 + 0.85
    taps = hz._ssim_taps().tolist() if hasattr(hz, "_ssim_taps") else None
    assert taps is not None
    n = 5000
    ls = torch.log(0.01 + 0.3 * torch.rand((n, 3), generator=g)).cuda()
    w = (0.8, 0.7, 0.35); c_ssim = 0.2; limit, wl, ws = 0.2, 5.0, 10.0
    # separate kernels
    pix, sums_ref = gsr.capi.mapping_pixel_loss(image, depth, sur, sil, frgb, fd, *w)
    ssim = gsr.capi.ssim_mean(image, frgb, taps)
    reg, reg_ref = gsr.capi.scale_regularisers(ls, limit, wl, ws)
    total = pix + c_ssim * (1.0 - ssim) + reg
    total.backward()
    # fused
    L = gsr.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    np6 = int(L.gsr_ssim_partials(3, H, W))
    partial6 = torch.empty((np6 * 6,), device="cuda"); dmaps = torch.empty((3, 3, H, W), device="cuda")
    t11 = (C.c_float * 11)(*taps); w3 = (C.c_float * 3)(*w)
    img = image.detach().contiguous(); dep = depth.detach().contiguous()
    gsr.capi._check(L.gsr_map_loss_forward(p(img), p(dep), p(sur), p(sil), p(frgb), p(fd), H, W, t11, 0.99, p(partial6), p(dmaps), None))
    xyz = torch.zeros((n, 3), device="cuda"); q = torch.ones((n, 4), device="cuda"); lg = torch.zeros((n, 1), device="cuda")
    reg_partial = torch.empty((3 * ((n + 255) // 256),), device="cuda")
    gsr.capi._check(L.gsr_map_prepare(n, None, None, p(ls), None, None, None, None, None, None, limit, wl, ws, p(reg_partial), None, None))
    sums = torch.empty((8,), device="cuda"); reg_out = torch.empty((4,), device="cuda"); loss = torch.empty((1,), device="cuda")
    gsr.capi._check(L.gsr_map_loss_finish(p(partial6), p(reg_partial), n, H, W, w3, c_ssim, wl, ws, None, p(sums), p(reg_out), p(loss), None))
    neg_c = torch.tensor([-c_ssim], device="cuda"); gi = torch.empty_like(img); gd = torch.empty_like(dep)
    gsr.capi._check(L.gsr_map_loss_backward(p(img), p(dep), p(frgb), p(fd), p(dmaps), H, W, t11, w3, p(neg_c), p(sums), p(gi), p(gd), None))
    assert abs(float(loss) - float(total)) <= 2e-6 * abs(float(total))
    assert (sums[:6].cpu() - sums_ref[:6].cpu()).abs().max() <= 2e-5 * float(sums_ref[:5].abs().max())
    assert (reg_out.cpu() - reg_ref.cpu()).abs().max() <= 2e-5 * max(float(reg_ref.abs().max()), 1e-12)
    assert (gi - image.grad).abs().max() <= 2e-6 * float(image.grad.abs().max())
    assert (gd - depth.grad).abs().max() <= 2e-6 * float(depth.grad.abs().max())


def test_backward_without_colour_outputs_takes_the_lean_plain_kernel(gsr):
    """gsr_backward on a plain render with no colour / SH gradient buffer and no fused channels (round 6: an unsharded tracking iteration on the surface depth) runs the
    backward blend WITHOUT the colour sums (K_blend_bwd<64, false, false, false>: the lean body at four waves per SIMD). Every other gradient must equal the full call's
    (to the order of the float atomics), and the accumulators are left clean."""
    syn = gsr.synthetic
    cam = syn.make_camera(320, 240, 260.0, 258.0, bg=(0.2, 0.1, 0.3))
    sc = syn.make_scene(30000, cam, seed=8, scale_mult=2.0)
    s = gsr.capi.Settings.from_camera(cam)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    g = t(sc.dL_dpix)
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations)
    full = gsr.backward(st, g)
    lean = gsr.capi.alloc_grads(sc.P, 0, "cuda", intermediates=False)
    lean.dL_dcolors = None
    lean.dL_dsh = None
    out = gsr.backward(st, g, grads=lean)
    assert float(gsr.capi.acc_view(st).abs().max()) == 0.0
    for n in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
        a, b = getattr(out, n), getattr(full, n)
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), n
    again = gsr.backward(st, g)                                       # the state is reusable: a full backward afterwards gives the full gradients
    assert float((again.dL_dcolors - full.dL_dcolors).abs().max()) <= 2e-6 * float(full.dL_dcolors.abs().max())


def test_silhouette_only_backward_equals_the_fused_pair_with_a_zero_depth_gradient(gsr):
    """gsr_backward_args.dds_depth_only = 2 (round 6: a sharded tracking iteration on the surface depth — the layer receives a colour gradient and what it occludes,
    nothing through the blended depth): the plain no-colour kernel with the silhouette's plane folded into the background factor must give what the fused pair's
    no-colour kernel gives for dL_dds = [0, g_sil]; the combination with colour outputs is refused before anything is launched."""
    syn = gsr.synthetic
    cam = syn.make_camera(320, 240, 260.0, 258.0, bg=(0.0, 0.0, 0.0))
    sc = syn.make_scene(30000, cam, seed=9, scale_mult=2.0)
    s = gsr.capi.Settings.from_camera(cam)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device="cuda")
    g = t(sc.dL_dpix)
    gs = torch.randn((1, cam.height, cam.width), generator=torch.Generator().manual_seed(4)).cuda()
    st = gsr.forward(s, sc.means3D, sc.opacities, colors=sc.colors, scales=sc.scales, rotations=sc.rotations, dual=True)
    def nocol():
        gr = gsr.capi.alloc_grads(sc.P, 0, "cuda", intermediates=False)
        gr.dL_dcolors = None; gr.dL_dsh = None
        return gr
    ref = gsr.backward(st, g, grads=nocol(), dL_dds=torch.cat([torch.zeros_like(gs), gs]).contiguous(), detach_depth_color=True)
    out = gsr.backward(st, g, grads=nocol(), dL_dds=gs.contiguous(), detach_depth_color=True, dds_depth_only=2)
    for n in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
        a, b = getattr(out, n), getattr(ref, n)
        assert float((a - b).abs().max()) <= 3e-6 * float(b.abs().max()), n
    assert float(gsr.capi.acc_view(st).abs().max()) == 0.0
    with pytest.raises(gsr.capi.GsrError):                            # colour outputs asked for: the silhouette-only form does not exist
        gsr.backward(st, g, dL_dds=gs.contiguous(), detach_depth_color=True, dds_depth_only=2)
    assert float(gsr.capi.acc_view(st).abs().max()) == 0.0            # (refused before the first launch: the accumulators are as the forward left them)


@pytest.mark.parametrize("M", [1, 37, 700])
def test_reprojection_term_adds_its_pose_sums_and_its_value(gsr, hz, M):
    """gsr_reproj_loss (the ORB matches' term of the tracking loss, src/Render.cc:1031-1096) against float64 autograd through the reference's
    expressions: value added to the loss, the twelve sums dL/dR (row-major), dL/dt ADDED to the pose row; inliers recomputed (chi-square 5.991),
    stored and re-used; the gradient scale of a sharded run."""
    g = torch.Generator().manual_seed(M)
    fx, fy, cx, cy, w = 260.0, 258.0, 159.5, 119.5, 0.1
    T = torch.tensor(__import__("util").pose(0.03, (0.02, -0.01, 0.04)), dtype=torch.float64)
    Xc = torch.stack([torch.rand(M, generator=g) * 1.2 - 0.6, torch.rand(M, generator=g) * 0.9 - 0.45, 1.0 + 2.0 * torch.rand(M, generator=g)], 1).double()
    Xw = (Xc - T[:3, 3]) @ T[:3, :3]                                          # Xc = R Xw + t
    noise = torch.randn(M, 2, generator=g).double() * torch.where(torch.arange(M) % 5 == 0, 6.0, 0.7).unsqueeze(1)   # every fifth match an outlier
    obs = torch.stack([fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy], 1) + noise
    s2 = (0.5 + torch.rand(M, generator=g)).double()

    def ref(mask):
        Tr = T.clone().requires_grad_(True)
        X = Xw @ Tr[:3, :3].t() + Tr[:3, 3]
        e = torch.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1) - obs
        werr = s2 * (e * e).sum(1)
        m = mask if mask is not None else torch.ones(M, dtype=torch.bool)
        val = werr[m].sum()
        val.backward()
        return float(val), torch.cat([Tr.grad[:3, :3].reshape(-1), Tr.grad[:3, 3]]), werr.detach() < 5.991
    c = lambda t: t.to(torch.float32).cuda().contiguous()
    Td = c(T.reshape(-1))
    val_all, grad_all, inl_ref = ref(None)
    row = torch.full((12,), 3.0, device="cuda"); loss = torch.full((1,), 7.0, device="cuda")
    gsr.capi.reproj_loss(c(obs), c(Xw), c(s2), Td, fx, fy, cx, cy, w, row, loss, refresh=2)
    assert abs(float(loss) - 7.0 - w * val_all) <= 2e-5 * (1 + w * val_all)
    assert ((row.cpu().double() - 3.0) - w * grad_all).abs().max() <= 2e-4 * max(1.0, float((w * grad_all).abs().max()))
    # recompute the inliers, store them, use them again (with a 1/world gradient scale)
    inl = torch.ones(M, dtype=torch.uint8, device="cuda")
    row.zero_(); loss.zero_()
    gsr.capi.reproj_loss(c(obs), c(Xw), c(s2), Td, fx, fy, cx, cy, w, row, loss, inliers=inl, refresh=1)
    assert torch.equal(inl.cpu().bool(), inl_ref) and (M < 5 or not bool(inl_ref.all()))
    val_in, grad_in, _ = ref(inl_ref)
    assert abs(float(loss) - w * val_in) <= 2e-5 * (1 + w * val_in)
    row2 = torch.zeros(12, device="cuda"); loss2 = torch.zeros(1, device="cuda")
    gsr.capi.reproj_loss(c(obs), c(Xw), c(s2), Td, fx, fy, cx, cy, w, row2, loss2, inliers=inl, refresh=0, grad_scale=0.25)
    assert float(loss2) == float(loss) and (row2 * 4 - row).abs().max() <= 1e-6 * max(1.0, float(row.abs().max()))
    assert (row.cpu().double() - w * grad_in).abs().max() <= 2e-4 * max(1.0, float((w * grad_in).abs().max()))


def test_camera_transform_inside_the_projection_kernel_is_gsr_to_camera_bit_for_bit(gsr, syn):
    """gsr_forward_args.pre_Tcw: world means + the pose on the device -> the same render, the same radii / tile lists and the same camera-frame means
    as gsr_to_camera followed by the plain forward (what a tracking iteration did before: src/Render.cc:750-752 + the render)."""
    import util
    cam = syn.make_camera(**syn.TUM1)
    sc = syn.make_scene(20000, cam, seed=11, scale_mult=2.0, frac_behind=0.1, frac_offscreen=0.2)
    s = gsr.capi.Settings.from_camera(cam)
    dev = s.viewmatrix.device
    t = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
    T = t(util.pose(0.05, (0.02, -0.03, 0.04)))
    Xc = t(sc.means3D)
    Xw = ((Xc - T[:3, 3]) @ T[:3, :3]).contiguous()                      # Xc ~ R Xw + t
    mc = torch.empty_like(Xw)
    gsr.capi._check(gsr.lib().gsr_to_camera(gsr.capi._p(Xw), Xw.shape[0], gsr.capi._p(T), gsr.capi._p(mc), gsr.capi._stream()))
    ws = gsr.capi.Workspace(Xw.shape[0], cam.width, cam.height, max_rendered=4_000_000, device=dev)
    kw = dict(colors=t(sc.colors), scales=t(sc.scales), rotations=t(sc.rotations))
    a = gsr.forward_ws(s, ws, mc, t(sc.opacities), dual=True, **kw)
    ref = [x.clone() for x in (a.color, a.depth, a.radii, a.ds)]
    da = gsr.debug_export(a)
    out = torch.zeros_like(Xw)
    b = gsr.forward_ws(s, ws, Xw, t(sc.opacities), dual=True, pre_Tcw=T, means_cam_out=out, **kw)
    db = gsr.debug_export(b)
    assert torch.equal(out, mc)
    for x, y in zip(ref, (b.color, b.depth, b.radii, b.ds)):
        assert torch.equal(x, y)
    import numpy as np
    np.testing.assert_array_equal(da["ranges"], db["ranges"]); np.testing.assert_array_equal(da["point_list"], db["point_list"])
    # rejected before any launch: a transform without a place for the camera-frame means
    with pytest.raises(Exception):
        gsr.forward_ws(s, ws, Xw, t(sc.opacities), dual=True, pre_Tcw=T, means_cam_out=None, **kw)


def test_activations_inside_the_projection_kernel_are_gsr_map_prepare_bit_for_bit(gsr, syn):
    """gsr_forward_args.raw (+ pre_Tcw): raw parameters in -> the render, the radii / tile lists, the activated tensors the backward takes and the scale
    regularisers' partial sums that gsr_map_prepare followed by the plain forward produce (what a mapping iteration did before)."""
    import util
    cam = syn.make_camera(**syn.TUM1)
    sc = syn.make_scene(30000, cam, seed=12, scale_mult=3.0, frac_behind=0.1, frac_offscreen=0.2)
    s = gsr.capi.Settings.from_camera(cam)
    dev = s.viewmatrix.device
    t = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous()
    n = sc.means3D.shape[0]
    T = t(util.pose(0.04, (0.01, 0.02, -0.03)))
    Xw = ((t(sc.means3D) - T[:3, 3]) @ T[:3, :3]).contiguous()
    op = t(sc.opacities).reshape(-1)
    logit = torch.log(op / (1 - op)).contiguous(); quat = (t(sc.rotations) * 1.7).contiguous()
    ls = torch.log(t(sc.scales).abs().clamp_min(1e-6)).contiguous()      # (the scene's splats behind the camera carry negative scales)
    p = gsr.capi._p
    L = gsr.lib()
    mc = torch.empty((n, 3), device=dev); o1 = torch.empty((n,), device=dev); s1 = torch.empty((n, 3), device=dev); r1 = torch.empty((n, 4), device=dev)
    rows = (n + 255) // 256
    reg1 = torch.empty((3 * rows,), device=dev)
    limit = float(torch.quantile(torch.exp(ls).reshape(-1), 0.8))      # some scales beyond the limit
    gsr.capi._check(L.gsr_map_prepare(n, p(Xw), p(logit), p(ls), p(quat), p(T), p(mc), p(o1), p(s1), p(r1), limit, 5.0, 10.0, p(reg1), None, None))
    ws = gsr.capi.Workspace(n, cam.width, cam.height, max_rendered=6_000_000, device=dev)
    a = gsr.forward_ws(s, ws, mc, o1.reshape(-1, 1), colors=t(sc.colors), scales=s1, rotations=r1, dual=True)
    ref = [x.clone() for x in (a.color, a.depth, a.radii, a.ds)]
    da = gsr.debug_export(a)
    mc2 = torch.zeros_like(mc); o2 = torch.zeros_like(o1); s2 = torch.zeros_like(s1); r2 = torch.zeros_like(r1); reg2 = torch.zeros_like(reg1)
    b = gsr.forward_ws(s, ws, Xw, logit.reshape(-1, 1), colors=t(sc.colors), scales=ls, rotations=quat, dual=True, pre_Tcw=T, means_cam_out=mc2,
                       raw=(o2, s2, r2, limit, reg2))
    db = gsr.debug_export(b)
    for x, y in ((mc, mc2), (o1, o2), (s1, s2), (r1, r2), (reg1, reg2)):
        assert torch.equal(x, y)
    assert float(reg1.reshape(-1, 3)[:, 0].sum()) > 0                   # (the regularisers are not vacuous here)
    for x, y in zip(ref, (b.color, b.depth, b.radii, b.ds)):
        assert torch.equal(x, y)
    np.testing.assert_array_equal(da["ranges"], db["ranges"]); np.testing.assert_array_equal(da["point_list"], db["point_list"])
