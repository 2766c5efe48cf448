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
