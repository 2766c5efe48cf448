"""GPU (-m gpu): the kernels of the sharded loop's BAND exchange (round 6; csrc/gsr_shard.h: gsr_band_composite_forward / _backward, gsr_shard_map_totals;
csrc/gsr_train.h: the loss kernels on a band of rows). A rank composites, evaluates the loss and differentiates the composite on its band of pixel rows
only — for every rank's layer. Checked here, kernel by kernel, on a simulated world:
  * the bands of the composite, tiled over the ranks, are the dense float64 composite; every rank's (dL/dlayer, dL/dS) rows are its slice of the dense
    gradient (float64 autograd), and equal what the three round-4 kernels (gsr_composite_forward / _backward_local / _backward_occlusion) give;
  * the mapping loss evaluated band by band: the bands' sums add up to the whole image's, the gradient planes and derivative maps are BIT-identical
    to the whole-image launch on the band's rows, gsr_shard_map_totals reproduces the whole-image finish; the same for the tracking loss.
The reference is single-GPU (src/Render.cc:420-483, :1054-1126): the exchange has no counterpart there; what is pinned is that banding changes nothing."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hz(gsr):
    return __import__("gsorb_slam_amd.harness", fromlist=["x"])


def _bands(H, world):
    hb = -(-H // world)
    return [(r * hb, min(H, (r + 1) * hb)) for r in range(world)]


@pytest.mark.parametrize("world", [1, 2, 4, 8, 11])
def test_band_compositor_equals_the_dense_composite_and_the_round4_kernels(gsr, world):
    g = torch.Generator().manual_seed(world)
    H, W, halo = 97, 53, 10
    L = torch.rand((world, 6, H, W), generator=g)
    L[:, 4] *= 0.95
    L[:, 4, :5] = 0.0                        # rows nothing covers
    L[0, 4, 5:9] = 1.0                       # an opaque layer
    L[:, 5] = torch.where(torch.rand((world, H, W), generator=g) < 0.7, 0.5 + 3 * L[:, 5], torch.zeros(1))
    keys = torch.rand((world,), generator=g)
    order = torch.argsort(keys.double(), stable=True)
    G4 = torch.randn((4, H, W), generator=g); Gs = torch.randn((H, W), generator=g)
    # dense float64 reference
    Ld = L[:, :4].double().requires_grad_(True); Sd = L[:, 4:5].double().requires_grad_(True)
    T = torch.ones((1, H, W), dtype=torch.float64); out = torch.zeros((4, H, W), dtype=torch.float64)
    surf_ref = torch.zeros((1, H, W)); found = torch.zeros((1, H, W), dtype=torch.bool)
    for k in order.tolist():
        out = out + T * Ld[k]
        T = T * (1 - Sd[k])
        has = L[k, 5:6] > 0
        surf_ref = torch.where(~found & has, L[k, 5:6], surf_ref)
        found = found | (has & (T.detach() <= 0.5))
    ((out * G4.double()).sum() + ((1 - T[0]) * Gs.double()).sum()).backward()
    Ld_c, order_d, G4d, Gsd = L.cuda().contiguous(), order.cuda(), G4.cuda().contiguous(), Gs.cuda().contiguous()
    comp = torch.full((4, H, W), float("nan"), device="cuda"); sil = torch.full((H, W), float("nan"), device="cuda"); sur = torch.full((H, W), float("nan"), device="cuda")
    d_from = [torch.zeros((world, 5, H, W), device="cuda") for _ in range(world)]     # d_from[r][k]: what rank r computed for rank k's layer (its band's rows)
    for r, (b0, b1) in enumerate(_bands(H, world)):
        if b0 >= b1:
            continue
        own = Ld_c[r].contiguous()
        c_r = torch.full((4, H, W), float("nan"), device="cuda"); s_r = torch.full((H, W), float("nan"), device="cuda"); u_r = torch.full((H, W), float("nan"), device="cuda")
        gsr.capi.band_composite_forward(world, r, order_d, Ld_c if world > 1 else None, own, b0, b1, halo, c_r, s_r, u_r)
        e0, e1 = max(0, b0 - halo), min(H, b1 + halo)
        assert (c_r[:3, e0:e1].cpu().double() - out.detach()[:3, e0:e1]).abs().max() < 5e-6       # rgb on the band and its halo
        assert torch.isnan(c_r[:3, :e0]).all() and torch.isnan(c_r[:3, e1:]).all()                # nothing else is written
        assert torch.isnan(c_r[3, :b0]).all() and torch.isnan(c_r[3, b1:]).all()
        comp[:, b0:b1] = c_r[:, b0:b1]; sil[b0:b1] = s_r[b0:b1]; sur[b0:b1] = u_r[b0:b1]
        d_own = torch.zeros((5, H, W), device="cuda")
        gsr.capi.band_composite_backward(world, r, order_d, Ld_c if world > 1 else None, own, G4d, b0, b1, d_from[r] if world > 1 else None, d_own, g_sil=Gsd)
        d_from[r][r] = d_own
    assert (comp.cpu().double() - out.detach()).abs().max() < 5e-6
    assert (sil.cpu().double() - (1 - T.detach()[0])).abs().max() < 2e-6
    assert torch.equal(sur.cpu(), surf_ref[0])
    # every rank's layer gradient: the rows it gets back from the ranks' bands
    d_layer = torch.stack(d_from).sum(0)                                 # [k][5][H][W]: each row was written by exactly one rank
    for k in range(world):
        assert (d_layer[k, :4].cpu().double() - Ld.grad[k]).abs().max() < 5e-6
        assert (d_layer[k, 4].cpu().double() - Sd.grad[k, 0]).abs().max() < 2e-5 * max(1.0, float(Sd.grad[k].abs().max()))
    # ... and the same numbers as the replicated path's kernels
    gathered = torch.stack([L[:, 4], L[:, 5]], 1).cuda().contiguous()
    locs = [gsr.capi.composite_backward_local(world, r, order_d, gathered, Ld_c[r, :4].contiguous(), G4d) for r in range(world)]
    c_all = torch.stack([c for _, c in locs]).contiguous()
    for k in range(world):
        assert (d_layer[k, :4] - locs[k][0]).abs().max() <= 1e-6
        dS = gsr.capi.composite_backward_occlusion(world, k, order_d, gathered, c_all, Gsd.reshape(1, H, W))
        assert (d_layer[k, 4] - dS[0]).abs().max() <= 1e-5 * max(1.0, float(dS.abs().max()))


@pytest.mark.parametrize("shape,world", [((97, 53), 1), ((97, 53), 4), ((96, 130), 3), ((680, 1200), 8)])
def test_losses_band_by_band_equal_the_whole_image(gsr, hz, shape, world):
    H, W = shape
    g = torch.Generator().manual_seed(H + world)
    image = torch.rand((3, H, W), generator=g).cuda(); frgb = torch.rand((3, H, W), generator=g).cuda()
    fd = (0.5 + 3 * torch.rand((H, W), generator=g)); fd[::5, ::3] = 0.0; fd = fd.cuda()
    depth = fd + 0.1 * torch.randn((H, W), generator=g).cuda(); sur = fd + 0.1 * torch.randn((H, W), generator=g).cuda()
    sil = torch.rand((H, W), generator=g).cuda() * 0.2 + 0.85
    taps = hz._ssim_taps().tolist()
    w = (0.8, 0.7, 0.35); c_ssim, wl, ws = 0.2, 5.0, 10.0
    L = gsr.lib(); p = lambda t: C.c_void_p(t.data_ptr()); chk = gsr.capi._check
    t11 = (C.c_float * 11)(*taps); w3 = (C.c_float * 3)(*w)
    neg_c = torch.tensor([-c_ssim], device="cuda")
    # whole image
    n6 = int(L.gsr_ssim_partials(3, H, W))
    part = torch.empty((6 * n6,), device="cuda"); dm = torch.zeros((3, 3, H, W), device="cuda")
    sums = torch.empty((8,), device="cuda"); reg3 = torch.tensor([3.0, 0.5, 0.25], device="cuda"); reg_out = torch.empty((4,), device="cuda"); loss = torch.empty((1,), device="cuda")
    gi = torch.empty_like(image); gd = torch.empty_like(depth)
    chk(L.gsr_map_loss_forward(p(image), p(depth), p(sur), p(sil), p(frgb), p(fd), H, W, t11, 0.99, p(part), p(dm), None))
    chk(L.gsr_map_loss_finish(p(part), p(reg3), 1, H, W, w3, c_ssim, wl, ws, None, p(sums), p(reg_out), p(loss), None))
    chk(L.gsr_map_loss_backward(p(image), p(depth), p(frgb), p(fd), p(dm), H, W, t11, w3, p(neg_c), p(sums), p(gi), p(gd), None))
    # band by band: every rank holds its band of the composite plus ten rows either side (the rest is poison), and its share of the regulariser sums
    rows = torch.zeros((world, 16), device="cuda")
    gi_b = torch.full_like(image, float("nan")); gd_b = torch.full_like(depth, float("nan"))
    for r, (b0, b1) in enumerate(_bands(H, world)):
        if b0 >= b1:
            rows[r, 12] = 0.0
            continue
        e0, e1 = max(0, b0 - 10), min(H, b1 + 10)
        img_r = torch.full_like(image, float("nan")); img_r[:, e0:e1] = image[:, e0:e1]
        dep_r = torch.full_like(depth, float("nan")); dep_r[b0:b1] = depth[b0:b1]
        n6r = int(L.gsr_map_loss_partials_rows(H, W, b0, b1))
        part_r = torch.empty((6 * n6r,), device="cuda"); dm_r = torch.full((3, 3, H, W), float("nan"), device="cuda")
        chk(L.gsr_map_loss_forward_rows(p(img_r), p(dep_r), p(sur), p(sil), p(frgb), p(fd), H, W, t11, 0.99, p(part_r), p(dm_r), b0, b1, None))
        m0, m1 = max(0, b0 - 5), min(H, b1 + 5)
        assert torch.equal(dm_r[:, :, m0:m1], dm[:, :, m0:m1])                       # the derivative maps the band's gradient needs: bit-identical
        reg_r = reg3 / world
        chk(L.gsr_map_loss_finish_rows(p(part_r), p(reg_r), 1, H, W, w3, c_ssim, wl, ws, None, p(rows[r]), C.c_void_p(rows[r].data_ptr() + 32),
                                       C.c_void_p(rows[r].data_ptr() + 48), b0, b1, None))
        chk(L.gsr_map_loss_backward_rows(p(img_r), p(dep_r), p(frgb), p(fd), p(dm_r), H, W, t11, w3, p(neg_c), p(sums), p(gi_b), p(gd_b), b0, b1, None))
    assert torch.equal(gi_b, gi) and torch.equal(gd_b, gd)                             # same arithmetic, row by row
    sums_t = torch.empty((8,), device="cuda"); reg_t = torch.empty((4,), device="cuda"); loss_t = torch.empty((1,), device="cuda")
    chk(L.gsr_shard_map_totals(world, p(rows), H, W, w3, c_ssim, wl, ws, p(sums_t), p(reg_t), p(loss_t), None))
    assert (sums_t - sums).abs().max() <= 2e-6 * float(sums.abs().max())
    assert (reg_t - reg_out).abs().max() <= 2e-6 * float(reg_out.abs().max())
    assert abs(float(loss_t) - float(loss)) <= 3e-6 * abs(float(loss))
    rows[world - 1, 12] = float("nan")                                                 # one rank's forward overflowed: the iteration's loss is NaN on every rank
    chk(L.gsr_shard_map_totals(world, p(rows), H, W, w3, c_ssim, wl, ws, p(sums_t), p(reg_t), p(loss_t), None))
    assert torch.isnan(loss_t).all()
    # the tracking loss (masked L1 sums, gradient planes): bands tile the whole image's
    part_t = torch.empty((1024 * 5,), device="cuda"); s_full = torch.empty((8,), device="cuda"); di = torch.empty_like(image); dd = torch.empty_like(depth)
    wt = (C.c_float * 3)(0.7, 1.0, 0.0)
    chk(L.gsr_track_loss(p(image), p(depth), None, p(sil), p(frgb), p(fd), H, W, 0.99, wt, p(part_t), p(s_full), p(di), p(dd), None, None))
    di_b = torch.full_like(image, float("nan")); dd_b = torch.full_like(depth, float("nan")); acc = torch.zeros((8,), device="cuda")
    tick = torch.zeros((144,), dtype=torch.int32, device="cuda")
    for r, (b0, b1) in enumerate(_bands(H, world)):
        if b0 >= b1:
            continue
        s_r = torch.empty((8,), device="cuda")
        chk(L.gsr_track_loss_rows(p(image), p(depth), None, p(sil), p(frgb), p(fd), H, W, 0.99, wt, p(part_t), p(s_r), p(di_b), p(dd_b), p(tick) if r % 2 else None, b0, b1, 0, None))
        acc += s_r
    assert torch.equal(di_b, di) and torch.equal(dd_b, dd)
    assert (acc[:6] - s_full[:6]).abs().max() <= 3e-6 * float(s_full[:6].abs().max())
