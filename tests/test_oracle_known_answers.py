"""Known-answer test of the oracle on a two-splat scene small enough to work out BY HAND from the reference's
source text. Every expected value below is a closed form evaluated in float64 next to the reference line it was read
from (DGR = Thirdparty/diff_gaussian_rasterization/cuda_rasterizer). It shares no code with the oracle, with
tests/spec_fp64.py (dense autograd) or with the kernels: a third, scalar statement of the same pipeline. It is still
the builder's reading of the reference, not an output of the compiled reference (which cannot be built here).

Scene: 32x32 image (2x2 tiles), fx = fy = 16 (tanfov = 1), identity view, background (0.1, 0.2, 0.3).
  A: mean (0, 0, 2),    isotropic scale 0.25, identity rotation, opacity 0.8, colour (1.0, 0.5, 0.25)
  B: mean (0.25, 0, 4), isotropic scale 0.5,  identity rotation, opacity 0.6, colour (0.2, 0.4, 0.9)
"""
import math

import numpy as np

from oracle import oracle

W = H = 32
FX = FY = 16.0
BG = np.array([0.1, 0.2, 0.3])
OA, OB = 0.8, 0.6
CA, CB = np.array([1.0, 0.5, 0.25]), np.array([0.2, 0.4, 0.9])


def _scene(syn):
    cam = syn.make_camera(W, H, FX, FY, bg=tuple(BG))
    means = np.array([[0, 0, 2.0], [0.25, 0, 4.0]], np.float32)
    scales = np.array([[0.25] * 3, [0.5] * 3], np.float32)
    rots = np.array([[1, 0, 0, 0], [1, 0, 0, 0]], np.float32)
    opac = np.array([[OA], [OB]], np.float32)
    cols = np.stack([CA, CB]).astype(np.float32)
    return cam, means, scales, rots, opac, cols


def _expected():
    e = {}
    # forward.cu:74-116 computeCov2D: J = [[fx/z, 0, -fx*x/z^2], [0, fy/z, -fy*y/z^2]], W = I, cov3D = s^2 I,
    # cov = J cov3D J^T, then +0.3 on the diagonal (:110-111)
    covA = (FX / 2.0) ** 2 * 0.25 ** 2 + 0.3                                  # 4.3 (both axes)
    JB = np.array([[FX / 4.0, 0, -FX * 0.25 / 16.0], [0, FY / 4.0, 0]])
    covB = 0.5 ** 2 * JB @ JB.T + 0.3 * np.eye(2)                             # [[4.315625, 0], [0, 4.3]]
    e["cov"] = [np.array([covA, 0.0, covA]), np.array([covB[0, 0], covB[0, 1], covB[1, 1]])]
    # forward.cu:216-221: conic = (c, -b, a) / det
    e["conic"] = [np.array([c[2], -c[1], c[0]]) / (c[0] * c[2] - c[1] ** 2) for c in e["cov"]]
    # forward.cu:229-232: lambda1 = mid + sqrt(max(0.1, mid^2 - det)); radius = ceil(3 sqrt(lambda1))
    e["radius"] = []
    for c in e["cov"]:
        mid, det = 0.5 * (c[0] + c[2]), c[0] * c[2] - c[1] ** 2
        e["radius"].append(math.ceil(3.0 * math.sqrt(mid + math.sqrt(max(0.1, mid * mid - det)))))   # 7 and 7
    # forward.cu:197-200 + auxiliary.h:41-44: p_hom.xy = mean.xy / tanfov, p_w = 1/(z + 1e-7), pix = ((ndc + 1) * S - 1) / 2
    e["pix"] = [np.array([15.5, 15.5]), np.array([((0.25 / (4.0 + 1e-7) + 1.0) * W - 1.0) * 0.5, 15.5])]   # B: 16.5
    # auxiliary.h:46-56: rect = [int((p - r)/16), int((p + r + 15)/16)) clamped to the 2x2 grid: both splats reach all 4 tiles
    e["tiles_touched"] = [4, 4]
    return e


def _pixel(e, x, y):
    """forward.cu:339-391 at one pixel, front to back (A is nearer)."""
    T, C, depth, n, parts = 1.0, np.zeros(3), 0.0, 0, []
    for k, (o, col, z) in enumerate(((OA, CA, 2.0), (OB, CB, 4.0))):
        d = e["pix"][k] - np.array([x, y], float)
        con = e["conic"][k]
        power = -0.5 * (con[0] * d[0] ** 2 + con[2] * d[1] ** 2) - con[1] * d[0] * d[1]      # :348
        G = math.exp(power)
        alpha = min(0.99, o * G)                                                               # :356
        assert power <= 0 and T * (1 - alpha) >= 1e-4                                          # never triggered in this scene
        parts.append(dict(d=d, G=G, alpha=alpha, T=T, con=con))
        if alpha < 1.0 / 255.0:                                                                # :357-358 skipped, not counted
            continue
        C = C + col * alpha * T                                                               # :370
        if T > 0.5:
            depth = z                                                                         # :374-379
        T *= 1 - alpha
        n = k + 1                                                                             # :381-383 last_contributor
    return C + T * BG, T, depth, n, parts                                                     # :398


def test_two_splat_scene_worked_out_by_hand(syn):
    cam, means, scales, rots, opac, cols = _scene(syn)
    e = _expected()
    assert e["radius"] == [7, 7] and abs(e["pix"][1][0] - 16.5) < 1e-6
    o = oracle.Oracle()
    f = o.forward(means3D=means, opacities=opac, cam=cam, colors=cols, scales=scales, rotations=rots)
    # ---- per-splat stages
    np.testing.assert_array_equal(f.radii, e["radius"])
    np.testing.assert_array_equal(f.stages["tiles_touched"], e["tiles_touched"])
    assert f.num_rendered == 8
    np.testing.assert_allclose(f.stages["means2D"], np.stack(e["pix"]), atol=2e-6)
    np.testing.assert_allclose(f.stages["conic_opacity"][:, :3], np.stack(e["conic"]), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(f.stages["conic_opacity"][:, 3], [OA, OB], rtol=1e-7)
    np.testing.assert_array_equal(f.stages["depths"], [2.0, 4.0])
    # keys: tile << 32 | depth bits, every tile lists A (z = 2) before B (z = 4)  (rasterizer_impl.cu:103-105)
    bits = lambda z: int(np.float32(z).view(np.uint32))
    exp_keys = [(t << 32) | bits(z) for t in range(4) for z in (2.0, 4.0)]
    np.testing.assert_array_equal(f.stages["keys_sorted"], np.array(exp_keys, np.uint64))
    np.testing.assert_array_equal(f.stages["point_list"], [0, 1] * 4)
    np.testing.assert_array_equal(f.stages["ranges"], [[0, 2], [2, 4], [4, 6], [6, 8]])
    # ---- pixels: one per tile, plus one next to the centre
    for (x, y) in ((15, 15), (16, 15), (12, 18), (19, 13), (22, 15), (23, 15), (30, 30)):
        C, T, depth, n, _ = _pixel(e, x, y)
        np.testing.assert_allclose(f.color[:, y, x], C, rtol=3e-6)
        assert abs(f.stages["final_T"].reshape(H, W)[y, x] - T) < 3e-7
        assert f.depth[0, y, x] == depth and f.stages["n_contrib"].reshape(H, W)[y, x] == n
    assert f.depth[0, 15, 15] == 2.0            # T = 1 - alpha_A ~ 0.245 < 0.5 when B arrives: the median depth stays on A
    assert f.depth[0, 15, 23] == 4.0 and f.stages["n_contrib"].reshape(H, W)[15, 23] == 2   # A skipped (alpha < 1/255), B blended
    assert f.depth[0, 30, 30] == 0.0 and f.stages["n_contrib"].reshape(H, W)[30, 30] == 0   # nothing reaches the corner
    np.testing.assert_allclose(f.color[:, 30, 30], BG, rtol=1e-7)

    # ---- backward of L = sum_ch g_ch * C_ch at ONE pixel (backward.cu:425-557)
    x, y = 16, 14
    g = np.array([0.7, -0.3, 0.5])
    dL = np.zeros((3, H, W), np.float32)
    dL[:, y, x] = g
    b = o.backward(dL)
    _, Tf, _, _, (pa, pb) = _pixel(e, x, y)
    # dL/dcolour = alpha * T * g  (:511-523)
    np.testing.assert_allclose(b.dL_dcolors[0], pa["alpha"] * pa["T"] * g, rtol=3e-6)
    np.testing.assert_allclose(b.dL_dcolors[1], pb["alpha"] * pb["T"] * g, rtol=3e-6)
    # dL/dalpha: ((c - accum_rec) . g) * T, accum_rec = colour accumulated BEHIND the splat (:511-523), plus the
    # background term -T_final / (1 - alpha) * (bg . g)  (:531-534)
    dA_B = ((CB - 0.0) @ g) * pb["T"] - Tf / (1 - pb["alpha"]) * (BG @ g)
    dA_A = ((CA - pb["alpha"] * CB) @ g) * pa["T"] - Tf / (1 - pa["alpha"]) * (BG @ g)
    # the same numbers are the analytic derivatives of C = cA aA + cB aB (1-aA) + bg (1-aA)(1-aB)
    assert abs(dA_A - (CA - CB * pb["alpha"] - BG * (1 - pb["alpha"])) @ g) < 1e-12
    assert abs(dA_B - ((CB - BG) * (1 - pa["alpha"])) @ g) < 1e-12
    # dL/dopacity = G * dL/dalpha (:555)
    np.testing.assert_allclose(b.dL_dopacity.ravel(), [pa["G"] * dA_A, pb["G"] * dA_B], rtol=5e-6)
    # dL/dconic: (-0.5 gdx dx, -0.5 gdx dy [stored in .y], -0.5 gdy dy) * dL_dG, dL_dG = opacity * dL/dalpha, gdx = G dx (:537-552)
    for k, (p, o_k, dA) in enumerate(((pa, OA, dA_A), (pb, OB, dA_B))):
        dG = o_k * dA
        gdx, gdy = p["G"] * p["d"][0], p["G"] * p["d"][1]
        np.testing.assert_allclose(b.dL_dconic[k].ravel()[[0, 1, 3]],
                                   [-0.5 * gdx * p["d"][0] * dG, -0.5 * gdx * p["d"][1] * dG, -0.5 * gdy * p["d"][1] * dG],
                                   rtol=1e-5, atol=1e-9)
        # dL/dmean2D = dL_dG * (-gdx*con.x - gdy*con.y, -gdy*con.z - gdx*con.y) * (W/2, H/2)   (:460-461, :541-548)
        np.testing.assert_allclose(b.dL_dmeans2D[k][:2],
                                   [dG * (-gdx * p["con"][0] - gdy * p["con"][1]) * 0.5 * W,
                                    dG * (-gdy * p["con"][2] - gdx * p["con"][1]) * 0.5 * H], rtol=1e-5, atol=1e-9)


# ---------------------------------------------------------------------------------------------------------------------
# Second scene: rotated, anisotropic splats. The per-splat backward (computeCov2DCUDA backward.cu:148-334, computeCov3D
# backward :340-396, preprocessCUDA :399-424) is pinned to the FORWARD formulas read off the reference: the loss at one
# pixel is written out in float64 straight from forward.cu, and its central finite differences with respect to every
# input are the expected gradients. No code is shared with the oracle, the fp64 spec or the kernels.
# ---------------------------------------------------------------------------------------------------------------------
def _rot_matrix(q):
    """forward.cu:118-148 computeCov3D: glm fills the matrix column by column, M = S * R, Sigma = transpose(M) * M, which
    is R_std S^2 R_std^T with the standard rotation matrix of (r, x, y, z) — q is used as given, NOT normalised (:128)."""
    r, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                     [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                     [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])


def _pixel_loss(means, scales, quats, opac, cols, px, py, g):
    """L = g . C(px, py): forward.cu:74-116 (cov2D), :190-232 (projection, conic), :339-398 (blend); identity view,
    fx = fy = 16 on a 32x32 image (tan_fov = 1), front-to-back order by depth."""
    order = np.argsort(means[:, 2], kind="stable")
    T, C = 1.0, np.zeros(3)
    for k in order:
        m, s, q = means[k], scales[k], quats[k]
        R = _rot_matrix(q)
        Sigma = R @ np.diag(s * s) @ R.T
        tx, ty, tz = m
        lim = 1.3 * 1.0                                                      # forward.cu:86-91 clamp of t.x/t.z (inactive here)
        assert abs(tx / tz) < lim and abs(ty / tz) < lim
        J = np.array([[FX / tz, 0, -FX * tx / tz ** 2], [0, FY / tz, -FY * ty / tz ** 2]])
        cov = J @ Sigma @ J.T + 0.3 * np.eye(2)
        det = cov[0, 0] * cov[1, 1] - cov[0, 1] ** 2
        con = np.array([cov[1, 1], -cov[0, 1], cov[0, 0]]) / det
        pix = np.array([((tx / (tz + 1e-7) + 1.0) * W - 1.0) * 0.5, ((ty / (tz + 1e-7) + 1.0) * H - 1.0) * 0.5])
        d = pix - np.array([px, py], float)
        power = -0.5 * (con[0] * d[0] ** 2 + con[2] * d[1] ** 2) - con[1] * d[0] * d[1]
        alpha = min(0.99, opac[k] * math.exp(power))
        assert power <= 0 and alpha >= 1.0 / 255.0 and alpha < 0.99 and T * (1 - alpha) >= 1e-4   # smooth regime only
        C = C + cols[k] * alpha * T
        T *= 1 - alpha
    return float(g @ (C + T * BG))


def test_per_splat_backward_against_finite_differences_of_the_forward_formulas(syn):
    cam = syn.make_camera(W, H, FX, FY, bg=tuple(BG))
    means = np.array([[0.05, -0.1, 2.0], [0.3, 0.1, 3.5]])
    scales = np.array([[0.30, 0.20, 0.25], [0.45, 0.6, 0.35]])
    quats = np.array([[0.9, 0.1, -0.2, 0.3], [0.7, -0.4, 0.3, 0.5]])        # deliberately NOT unit length
    opac = np.array([0.8, 0.6])
    cols = np.stack([CA, CB])
    px, py, g = 17, 14, np.array([0.7, -0.3, 0.5])
    f32 = lambda a: np.asarray(a, np.float32)
    o = oracle.Oracle()
    f = o.forward(means3D=f32(means), opacities=f32(opac).reshape(-1, 1), cam=cam, colors=f32(cols), scales=f32(scales), rotations=f32(quats))
    L0 = _pixel_loss(means, scales, quats, opac, cols, px, py, g)
    assert abs(float(g @ f.color[:, py, px].astype(np.float64)) - L0) < 2e-6
    dL = np.zeros((3, H, W), np.float32)
    dL[:, py, px] = g
    b = o.backward(dL)

    def fd(arr, k, j, h=1e-6):
        args = dict(means=means.copy(), scales=scales.copy(), quats=quats.copy(), opac=opac.copy(), cols=cols.copy())
        lo, hi = {n: v.copy() for n, v in args.items()}, {n: v.copy() for n, v in args.items()}
        if arr == "opac":
            lo[arr][k] -= h; hi[arr][k] += h
        else:
            lo[arr][k, j] -= h; hi[arr][k, j] += h
        return (_pixel_loss(px=px, py=py, g=g, **hi) - _pixel_loss(px=px, py=py, g=g, **lo)) / (2 * h)

    for k in range(2):
        exp_mean = np.array([fd("means", k, j) for j in range(3)])
        exp_scale = np.array([fd("scales", k, j) for j in range(3)])
        exp_rot = np.array([fd("quats", k, j) for j in range(4)])
        exp_col = np.array([fd("cols", k, j) for j in range(3)])
        exp_op = fd("opac", k, 0)
        for got, exp, name in ((b.dL_dmeans3D[k], exp_mean, "mean3D"), (b.dL_dscales[k], exp_scale, "scale"),
                               (b.dL_drotations[k], exp_rot, "rotation"), (b.dL_dcolors[k], exp_col, "colour"),
                               (b.dL_dopacity[k].ravel(), np.array([exp_op]), "opacity")):
            scale = max(1e-6, float(np.abs(exp).max()))
            assert float(np.abs(np.asarray(got, np.float64).ravel() - exp).max()) <= 2e-4 * scale, (k, name, got, exp)


# ---------------------------------------------------------------------------------------------------------------------
# Third scene: view-dependent colours. computeColorFromSH (forward.cu:20-71) written out by hand for degree 2 — basis
# constants as printed in auxiliary.h:21-38 — with the +0.5 shift and the clamp at zero; its backward (backward.cu:24-140,
# incl. the direction normalisation dnormvdv, :13-22... and the clamp rule: a clamped channel passes no gradient) and the
# extra mean gradient through the view direction (:399-424) are pinned by finite differences of this forward text.
# ---------------------------------------------------------------------------------------------------------------------
_C0, _C1 = 0.28209479177387814, 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)


def _sh_colour(sh, mean, campos):
    d = mean - campos
    x, y, z = d / np.linalg.norm(d)
    c = _C0 * sh[0] - _C1 * y * sh[1] + _C1 * z * sh[2] - _C1 * x * sh[3]
    c = c + _C2[0] * x * y * sh[4] + _C2[1] * y * z * sh[5] + _C2[2] * (2 * z * z - x * x - y * y) * sh[6] \
          + _C2[3] * x * z * sh[7] + _C2[4] * (x * x - y * y) * sh[8]
    return np.maximum(c + 0.5, 0.0)


def test_sh_colour_backward_against_finite_differences_of_the_forward_formulas(syn):
    cam = syn.make_camera(W, H, FX, FY, bg=tuple(BG))
    cam.sh_degree = 2
    campos = np.asarray(cam.campos, np.float64)
    means = np.array([[0.05, -0.1, 2.0], [0.3, 0.1, 3.5]])
    scales = np.array([[0.30, 0.20, 0.25], [0.45, 0.6, 0.35]])
    quats = np.array([[0.9, 0.1, -0.2, 0.3], [0.7, -0.4, 0.3, 0.5]])
    quats = quats / np.linalg.norm(quats, axis=1, keepdims=True)
    opac = np.array([0.8, 0.6])
    rng = np.random.default_rng(6)
    shs = rng.normal(0, 0.35, (2, 9, 3))
    shs[0, 0] = [1.2, 0.3, -2.5]          # the blue channel of splat 0 clamps at zero: no gradient through it
    px, py, g = 17, 14, np.array([0.7, -0.3, 0.5])
    f32 = lambda a: np.asarray(a, np.float32)

    def loss(means_, shs_):
        cols = np.stack([_sh_colour(shs_[k], means_[k], campos) for k in range(2)])
        return _pixel_loss(means_, scales, quats, opac, cols, px, py, g), cols

    L0, cols0 = loss(means, shs)
    assert cols0[0, 2] == 0.0 and (cols0[0, :2] > 0).all() and (cols0[1] > 0).all()
    o = oracle.Oracle()
    f = o.forward(means3D=f32(means), opacities=f32(opac).reshape(-1, 1), cam=cam, shs=f32(shs), scales=f32(scales), rotations=f32(quats))
    assert abs(float(g @ f.color[:, py, px].astype(np.float64)) - L0) < 2e-6
    np.testing.assert_array_equal(f.stages["clamped"].reshape(2, 3), [[0, 0, 1], [0, 0, 0]])
    dL = np.zeros((3, H, W), np.float32)
    dL[:, py, px] = g
    b = o.backward(dL)
    h = 1e-6
    exp_sh = np.zeros_like(shs)
    for idx in np.ndindex(*shs.shape):
        lo, hi = shs.copy(), shs.copy()
        lo[idx] -= h; hi[idx] += h
        exp_sh[idx] = (loss(means, hi)[0] - loss(means, lo)[0]) / (2 * h)
    exp_mean = np.zeros_like(means)
    for idx in np.ndindex(*means.shape):
        lo, hi = means.copy(), means.copy()
        lo[idx] -= h; hi[idx] += h
        exp_mean[idx] = (loss(hi, shs)[0] - loss(lo, shs)[0]) / (2 * h)
    got_sh = np.asarray(b.dL_dsh, np.float64)[:, :9]
    assert float(np.abs(got_sh - exp_sh).max()) <= 2e-4 * float(np.abs(exp_sh).max())
    assert float(np.abs(got_sh[0, :, 2]).max()) == 0.0                       # the clamped channel
    assert float(np.abs(np.asarray(b.dL_dmeans3D, np.float64) - exp_mean).max()) <= 2e-4 * float(np.abs(exp_mean).max())


# ---------------------------------------------------------------------------------------------------------------------
# Fourth scene: a rotated, translated camera and covariances given directly (cov3D_precomp). Pins the view-matrix
# convention of computeCov2D (forward.cu:74-116: t = view * mean, T = W * J with glm's column-major W, cov = T^T Vrk T —
# i.e. J R_cw Sigma R_cw^T J^T) and the 6-vector layout (xx, xy, xz, yy, yz, zz) of cov3D and of its gradient (an
# off-diagonal entry stands for both symmetric elements, backward.cu:200-215).
# ---------------------------------------------------------------------------------------------------------------------
def test_rotated_camera_and_precomputed_covariances_against_finite_differences(syn):
    a, bq = 0.15, -0.1
    Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(bq), -math.sin(bq)], [0, math.sin(bq), math.cos(bq)]])
    Rcw, tcw = Ry @ Rx, np.array([0.1, -0.05, 0.3])
    Tcw = np.eye(4)
    Tcw[:3, :3], Tcw[:3, 3] = Rcw, tcw
    cam = syn.make_camera(W, H, FX, FY, Tcw=Tcw, bg=tuple(BG))
    means = np.array([[0.1, -0.15, 2.2], [-0.2, 0.2, 3.0]])
    rng = np.random.default_rng(12)
    A = rng.normal(0, 0.25, (2, 3, 3))
    Sig = np.stack([m @ m.T + 0.02 * np.eye(3) for m in A])
    cov6 = np.stack([[S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]] for S in Sig])
    opac, cols = np.array([0.75, 0.65]), np.stack([CA, CB])
    px, py, g = 16, 15, np.array([0.7, -0.3, 0.5])

    def loss(means_, cov6_):
        T, C = 1.0, np.zeros(3)
        cam_pts = means_ @ Rcw.T + tcw
        for k in np.argsort(cam_pts[:, 2], kind="stable"):
            tx, ty, tz = cam_pts[k]
            c = cov6_[k]
            S = np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]])
            J = np.array([[FX / tz, 0, -FX * tx / tz ** 2], [0, FY / tz, -FY * ty / tz ** 2]])
            cov = J @ Rcw @ S @ Rcw.T @ J.T + 0.3 * np.eye(2)
            det = cov[0, 0] * cov[1, 1] - cov[0, 1] ** 2
            con = np.array([cov[1, 1], -cov[0, 1], cov[0, 0]]) / det
            pix = np.array([((tx / (tz + 1e-7) + 1.0) * W - 1.0) * 0.5, ((ty / (tz + 1e-7) + 1.0) * H - 1.0) * 0.5])
            d = pix - np.array([px, py], float)
            power = -0.5 * (con[0] * d[0] ** 2 + con[2] * d[1] ** 2) - con[1] * d[0] * d[1]
            alpha = min(0.99, opac[k] * math.exp(power))
            assert power <= 0 and 1.0 / 255.0 <= alpha < 0.99 and T * (1 - alpha) >= 1e-4
            C = C + cols[k] * alpha * T
            T *= 1 - alpha
        return float(g @ (C + T * BG))

    f32 = lambda x: np.asarray(x, np.float32)
    o = oracle.Oracle()
    f = o.forward(means3D=f32(means), opacities=f32(opac).reshape(-1, 1), cam=cam, colors=f32(cols), cov3D_precomp=f32(cov6))
    assert abs(float(g @ f.color[:, py, px].astype(np.float64)) - loss(means, cov6)) < 2e-6
    dL = np.zeros((3, H, W), np.float32)
    dL[:, py, px] = g
    b = o.backward(dL)
    h = 1e-6
    exp_cov, exp_mean = np.zeros_like(cov6), np.zeros_like(means)
    for idx in np.ndindex(*cov6.shape):
        lo, hi = cov6.copy(), cov6.copy()
        lo[idx] -= h; hi[idx] += h
        exp_cov[idx] = (loss(means, hi) - loss(means, lo)) / (2 * h)
    for idx in np.ndindex(*means.shape):
        lo, hi = means.copy(), means.copy()
        lo[idx] -= h; hi[idx] += h
        exp_mean[idx] = (loss(hi, cov6) - loss(lo, cov6)) / (2 * h)
    assert float(np.abs(np.asarray(b.dL_dcov3D, np.float64) - exp_cov).max()) <= 2e-4 * float(np.abs(exp_cov).max())
    assert float(np.abs(np.asarray(b.dL_dmeans3D, np.float64) - exp_mean).max()) <= 2e-4 * float(np.abs(exp_mean).max())


# ---------------------------------------------------------------------------------------------------------------------
# Fifth scene: the 0.99 clamp and early termination (forward.cu:356-364): a stack of four large co-axial splats. The first
# saturates (alpha = 0.99), the second is half transparent, the third would push T below 1e-4 — the pixel is DONE before
# blending it, it does not count as a contributor, and the fourth is never looked at. Backward (backward.cu:476-557): only
# the contributors receive gradients, and the clamp is transparent to dL/dopacity (straight-through: :555).
# ---------------------------------------------------------------------------------------------------------------------
def test_alpha_clamp_and_early_termination_worked_out_by_hand(syn):
    cam = syn.make_camera(W, H, FX, FY, bg=tuple(BG))
    zs = np.array([2.0, 2.5, 3.0, 3.5])
    means = np.stack([np.zeros(4), np.zeros(4), zs], 1)
    scales = np.full((4, 3), 2.0) * (zs / 2.0)[:, None]            # the same 2D footprint for all four: cov = 256.3
    quats = np.tile([1.0, 0, 0, 0], (4, 1))
    opac = np.array([1.0, 0.5, 1.0, 0.9])
    cols = np.array([[1.0, 0.5, 0.25], [0.2, 0.4, 0.9], [0.6, 0.1, 0.3], [0.9, 0.9, 0.1]])
    px, py, g = 15, 16, np.array([0.7, -0.3, 0.5])
    cov = (FX / 2.0) ** 2 * 2.0 ** 2 + 0.3                         # 256.3 on both axes, no off-diagonal
    G = math.exp(-0.5 * (0.5 ** 2 + 0.5 ** 2) / cov)               # pixel centre 15.5: d = (0.5, -0.5)
    a1, a2, a3 = min(0.99, 1.0 * G), 0.5 * G, min(0.99, 1.0 * G)
    assert a1 == 0.99 and a3 == 0.99
    T1 = 1.0 - a1
    T2 = T1 * (1.0 - a2)
    assert T2 * (1.0 - a3) < 1e-4 * 0.6 and T1 * (1 - a2) > 1e-4 * 10   # far from the knife edge in either direction
    C = cols[0] * a1 + cols[1] * a2 * T1 + BG * T2
    f32 = lambda a: np.asarray(a, np.float32)
    o = oracle.Oracle()
    f = o.forward(means3D=f32(means), opacities=f32(opac).reshape(-1, 1), cam=cam, colors=f32(cols), scales=f32(scales), rotations=f32(quats))
    np.testing.assert_allclose(f.color[:, py, px], C, rtol=3e-6)
    assert abs(f.stages["final_T"].reshape(H, W)[py, px] - T2) < 1e-8
    assert f.stages["n_contrib"].reshape(H, W)[py, px] == 2
    assert f.depth[0, py, px] == 2.0                                # T = 1 > 0.5 only when the first splat arrives
    dL = np.zeros((3, H, W), np.float32)
    dL[:, py, px] = g
    b = o.backward(dL)
    assert float(np.abs(b.dL_dcolors[2:]).max()) == 0.0 and float(np.abs(b.dL_dopacity[2:]).max()) == 0.0
    assert float(np.abs(b.dL_dmeans3D[2:]).max()) == 0.0
    np.testing.assert_allclose(b.dL_dcolors[0], a1 * 1.0 * g, rtol=3e-6)
    np.testing.assert_allclose(b.dL_dcolors[1], a2 * T1 * g, rtol=3e-6)
    dA2 = (cols[1] @ g) * T1 - T2 / (1 - a2) * (BG @ g)
    dA1 = ((cols[0] - a2 * cols[1]) @ g) * 1.0 - T2 / (1 - a1) * (BG @ g)
    np.testing.assert_allclose(b.dL_dopacity.ravel()[:2], [G * dA1, G * dA2], rtol=2e-5)   # the clamp passes the gradient
