"""Independent fp64 torch-autograd statement of what the reference rasterizer computes.

This is NOT a restatement of the reference's code: it is the textbook math
(EWA projection, per-tile depth-ordered alpha compositing) written densely over
(pixel, splat) pairs and differentiated by autograd, used to cross-check the C
oracle (oracle/gsr_oracle.c) on small scenes. Only the discrete decisions that
depend on fp32 rounding are taken from the caller (tile membership from the fp32
radii / pixel centres), exactly as SURVEY.md Appendix C step 6 describes.

Reference semantics reproduced (file:line under /root/reference/Thirdparty/
diff_gaussian_rasterization/cuda_rasterizer):
  * near cull z<=0.2 (auxiliary.h:154); 1.3*tanfov clamp on t.x/t.z with the clamped
    value treated as a constant in the backward (forward.cu:82-87, backward.cu:168-176,
    262-264); +0.3 low-pass (forward.cu:110-111)
  * alpha = min(0.99, o*exp(power)) with a straight-through clamp (backward.cu:499);
    skip power>0 and alpha<1/255; stop when T(1-alpha)<1e-4 (forward.cu:346-362)
  * colour = sum c*alpha*T + T_final*bg (forward.cu:398); median depth = depth of the
    last blended splat seen while T>0.5 (forward.cu:374-379), no gradient
"""
from __future__ import annotations

import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
         0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def tile_rects(means2D_f32: np.ndarray, radii: np.ndarray, W: int, H: int):
    """auxiliary.h:46-56 in fp32."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    p = means2D_f32.astype(np.float32)
    r = radii.astype(np.float32)
    f16 = np.float32(16.0)
    def lo(a, g):
        return np.minimum(g, np.maximum(0, np.trunc((a - r) / f16).astype(np.int64)))
    def hi(a, g):
        return np.minimum(g, np.maximum(0, np.trunc((((a + r) + f16) - np.float32(1.0)) / f16).astype(np.int64)))
    return lo(p[:, 0], gx), lo(p[:, 1], gy), hi(p[:, 0], gx), hi(p[:, 1], gy)


def eval_sh(deg, sh, d):
    """sh [P,M,3], d [P,3] unit."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
               + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
               + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
               + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res + 0.5


def render(scene, radii: np.ndarray, means2D_f32: np.ndarray, dL_dpix: np.ndarray | None = None,
           cov3D_precomp: np.ndarray | None = None, chunk: int = 4096):
    """Returns dict(color [3,H,W], depth [H,W], grads{...}) in fp64. With `cov3D_precomp` [P,6] (upper triangle
    xx xy xz yy yz zz, forward.cu:94-97) the 3D covariance is an input and scales / rotations are ignored.
    The dense (pixel, splat) part runs in chunks of `chunk` pixels; gradients accumulate over the chunks."""
    cam = scene.cam
    W, H = cam.width, cam.height
    dd = torch.float64
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=dd)
    means = t64(scene.means3D).requires_grad_(True)
    scales = t64(scene.scales).requires_grad_(True)
    rots = t64(scene.rotations).requires_grad_(True)
    opac = t64(scene.opacities).reshape(-1).requires_grad_(True)
    colors_in = shs_in = None
    if scene.colors is not None:
        colors_in = t64(scene.colors).requires_grad_(True)
    else:
        shs_in = t64(scene.shs).requires_grad_(True)
    V = t64(cam.viewmatrix).T          # Tcw (row-major math)
    PV = t64(cam.projmatrix).T         # P @ Tcw
    bg = t64(cam.bg)
    P = means.shape[0]
    fx = W / (2.0 * cam.tanfovx)
    fy = H / (2.0 * cam.tanfovy)

    hom = torch.cat([means, torch.ones(P, 1, dtype=dd)], 1)
    t = (hom @ V.T)[:, :3]
    ph = hom @ PV.T
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    # 3D covariance: R S S R^T with the quaternion used as given
    r, x, y, z = rots[:, 0], rots[:, 1], rots[:, 2], rots[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    S = torch.diag_embed(scales * cam.scale_modifier)
    Mm = R @ S
    Sigma = Mm @ Mm.transpose(1, 2)
    cov_in = None
    if cov3D_precomp is not None:
        cov_in = t64(cov3D_precomp).requires_grad_(True)
        c6 = cov_in
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4], c6[:, 5]], 1).reshape(P, 3, 3)

    limx, limy = 1.3 * cam.tanfovx, 1.3 * cam.tanfovy
    tz = t[:, 2]
    rx, ry = t[:, 0] / tz, t[:, 1] / tz
    cx_ = (rx < -limx) | (rx > limx)
    cy_ = (ry < -limy) | (ry > limy)
    tx = torch.where(cx_, (rx.clamp(-limx, limx) * tz).detach(), t[:, 0])
    ty = torch.where(cy_, (ry.clamp(-limy, limy) * tz).detach(), t[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz),
                     zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(P, 2, 3)
    Wr = V[:3, :3]
    JW = J @ Wr
    cov = JW @ Sigma @ JW.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    ca, cb, cc = c / det, -b / det, a / det

    if colors_in is not None:
        col = colors_in
    else:
        dirv = means - t64(cam.campos)
        dirv = dirv / dirv.norm(dim=1, keepdim=True)
        col = torch.clamp_min(eval_sh(cam.sh_degree, shs_in, dirv), 0.0)

    vis = torch.tensor(radii > 0)
    x0, y0, x1, y1 = [torch.tensor(v) for v in tile_rects(means2D_f32, radii, W, H)]
    depth = tz.detach()
    # stable order by fp32 depth bits (what the sort key holds), ties by index
    d32 = torch.tensor(depth.numpy().astype(np.float32).astype(np.float64))
    order = torch.argsort(d32, stable=True)
    order = order[vis[order]]

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pxs_all, pys_all = xs.reshape(-1), ys.reshape(-1)
    o = order
    N = W * H
    leaves = [means, opac] + ([colors_in] if colors_in is not None else [shs_in]) + \
             ([cov_in] if cov_in is not None else [scales, rots])
    names = ["means3D", "opacities", "colors" if colors_in is not None else "shs"] + \
            (["cov3D"] if cov_in is not None else ["scales", "rotations"])
    acc = [torch.zeros_like(l) for l in leaves]
    g_all = None if dL_dpix is None else t64(dL_dpix).reshape(3, -1).T
    color_out = np.zeros((N, 3)); depth_out = np.zeros(N); T_out = np.zeros(N)
    for c0 in range(0, N, chunk):
        pxs, pys = pxs_all[c0:c0 + chunk], pys_all[c0:c0 + chunk]
        tix, tiy = pxs // 16, pys // 16
        member = (tix[:, None] >= x0[o][None]) & (tix[:, None] < x1[o][None]) & \
                 (tiy[:, None] >= y0[o][None]) & (tiy[:, None] < y1[o][None])
        dx = px[o][None, :] - pxs[:, None].to(dd)
        dy = py[o][None, :] - pys[:, None].to(dd)
        power = -0.5 * (ca[o][None] * dx * dx + cc[o][None] * dy * dy) - cb[o][None] * dx * dy
        araw = opac[o][None] * torch.exp(power)
        alpha = araw - (araw - 0.99).clamp(min=0).detach()
        valid = member & (power <= 0) & (alpha >= 1.0 / 255.0)
        am = torch.where(valid, alpha, torch.zeros_like(alpha))
        Tafter = torch.cumprod(1 - am, 1)
        Tbefore = torch.cat([torch.ones(am.shape[0], 1, dtype=dd), Tafter[:, :-1]], 1)
        stop = valid & (Tafter < 1e-4)
        excluded = torch.cumsum(stop.to(torch.int64), 1) >= 1
        inc = valid & ~excluded
        w = torch.where(inc, am * Tbefore, torch.zeros_like(am))
        Tfinal = torch.prod(torch.where(inc, 1 - am, torch.ones_like(am)), 1)
        color = w @ col[o] + Tfinal[:, None] * bg[None]
        med = inc & (Tbefore > 0.5)
        idx = torch.arange(am.shape[1])[None].expand_as(med)
        last = torch.where(med, idx, torch.full_like(idx, -1)).max(1).values if am.shape[1] else torch.full((am.shape[0],), -1)
        dsel = torch.where(last >= 0, depth[o][last.clamp(min=0)] if am.shape[1] else torch.zeros(am.shape[0], dtype=dd),
                           torch.zeros(am.shape[0], dtype=dd))
        color_out[c0:c0 + chunk] = color.detach().numpy()
        depth_out[c0:c0 + chunk] = dsel.numpy()
        T_out[c0:c0 + chunk] = Tfinal.detach().numpy()
        if g_all is not None:
            loss = (color * g_all[c0:c0 + chunk]).sum()
            grads = torch.autograd.grad(loss, leaves, allow_unused=True, retain_graph=True)
            for a, gg in zip(acc, grads):
                if gg is not None:
                    a += gg
    out = dict(color=color_out.T.reshape(3, H, W), depth=depth_out.reshape(H, W), final_T=T_out.reshape(H, W))
    if g_all is not None:
        out["grads"] = {n: a.numpy() for n, a in zip(names, acc)}
    return out
