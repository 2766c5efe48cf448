"""Optimisation harness around the operator: the build's counterpart of the
reference's `Render` + `Gaussian` classes, i.e. everything that sits between the SLAM front end
and the rasterizer — parameter store with two Adam optimisers, activations, camera-frame
transform, the two colour modes, losses, the tracking and mapping loops, densification mask and
pruning with optimiser-state surgery. Plain torch on top of the drop-in Python operator; nothing
here is accelerated code. The ORB-SLAM2 side (features, covisibility, keyframes) is NOT here:
candidate frames and optional feature matches are inputs.

Reference behaviour reproduced (file:line under /root/reference/src unless noted):
  * Render.cc:750-760   means -> camera frame by bmm (pose gradient via autograd), sigmoid / exp /
                        normalize activations, viewmatrix = I in the rasterizer settings
  * Render.cc:927-981   colour render (raw rgb) and depth render (colours [z_cam, 1, 0])
  * Utils.cc:39-100     L1 losses (mean for mapping, sum for tracking) and SSIM with the 11x11
                        window whose taps are exp(-floor((x-11)/2)^2 / (2*1.5^2)) (:68-75)
  * Render.cc:420-483   mapping iteration: random candidate frame, image/depth/surface-depth
                        losses, scale regularisers, Adam step   (surface-depth has no gradient)
  * Render.cc:1054-1126 tracking iteration: pose-only Adam, best-pose bookkeeping with NaN guard,
                        early stop when |loss - last| < 1e-3, optional reprojection term
  * Gaussian.cc:50-95   insertion (logit opacity 1, identity quaternion, three scale inits)
  * Gaussian.cc:144-175 Adam per group, eps 1e-15; the pose optimiser uses lrCamQuat for BOTH groups
  * Gaussian.cc:180-258 pruning / concatenation with exp_avg / exp_avg_sq surgery
  * Render.cc:557-594   densification mask from the rendered silhouette / depth error
"""
from __future__ import annotations

import math
import os
import sys
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class Config:
    """Examples/RGB-D/replica.yaml:87-117 (Mapping / Tracking blocks)."""
    mapping_iters: int = 60
    im_weight_mapping: float = 1.0
    depth_weight_mapping: float = 0.7
    sur_depth_weight_mapping: float = 0.35
    reg_long_weight: float = 5.0
    reg_scalar_weight: float = 10.0
    lam: float = 0.8
    lr_mean3d: float = 0.0001
    lr_rgb: float = 0.0025
    lr_rotation: float = 0.001
    lr_opacities: float = 0.05
    lr_scales: float = 0.001
    prune_opacities: float = 0.005
    scale_modifier: float = 1.0
    init_scalar_method: int = 2          # 0 Distance, 1 DistanceMean, 2 SinglePixel (Gaussian.cc:59-79)
    radius_depth_ratio: float = 3.0
    median_mul: float = 40.0
    tracking_iters: int = 40
    lr_cam_quat: float = 0.0004
    lr_cam_trans: float = 0.002          # parsed by the reference but unused (Gaussian.cc:149-150)
    im_weight_tracking: float = 0.7
    feature_weight_tracking: float = 0.1
    depth_weight_tracking: float = 1.0
    use_sur_depth: bool = True


def _dgr():
    pkg = os.path.dirname(os.path.abspath(__file__))
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    import diff_gaussian_rasterization as dgr
    return dgr


# ---- losses (Utils.cc:39-100) -----------------------------------------------------------
def l1_mapping(a, b, mask=None):
    """Utils.cc:39-56: mean |a - b|, over the masked elements if a mask is given. The reference selects them
    (masked_select: a device->host sync for the count); the same mean is sum(where(mask, d, 0)) / count(mask) without one.
    An empty mask gives NaN, like the mean of an empty selection."""
    d = torch.abs(a - b)
    return d.mean() if mask is None else torch.where(mask, d, torch.zeros_like(d)).sum() / mask.sum()


def l1_tracking(a, b, mask=None):
    d = torch.abs(a - b)
    return d.sum() if mask is None else torch.where(mask, d, torch.zeros_like(d)).sum()


def ssim_window(window_size=11, sigma=1.5, channel=3, device="cpu"):
    g = torch.tensor([math.exp(-(math.floor((x - window_size) / 2.0) ** 2) / (2.0 * sigma * sigma))
                      for x in range(window_size)], dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    w = g.mm(g.t()).unsqueeze(0).unsqueeze(0)
    return w.expand(channel, 1, window_size, window_size).contiguous().to(device)


def _ssim_taps(window_size=11, sigma=1.5, device="cpu"):
    g = torch.tensor([math.exp(-(math.floor((x - window_size) / 2.0) ** 2) / (2.0 * sigma * sigma))
                      for x in range(window_size)], dtype=torch.float32, device=device)
    return g / g.sum()


def _capi():
    """The C-ABI wrappers (fused SSIM / Adam kernels); using them raises ImportError when the HIP library is not built."""
    from . import capi
    return capi


_TAPS = None


def ssim(img1, img2):
    """Utils.cc:77-100. The reference convolves with the 11x11 window outer(g, g) (ssim_window above); the window is
    separable, so the same sums are formed by an 11x1 and a 1x11 pass. On the GPU the whole loss — five window sums, the
    SSIM map, its mean, and the backward — is two HIP kernels (csrc/gsr_train.h: the convolutions through MIOpen were
    2.2 ms of a 6.0 ms mapping iteration at 1200x680); CPU tensors (the oracle-driven test loops) take the torch path."""
    global _TAPS
    if img1.is_cuda:
        if _TAPS is None:
            _TAPS = [float(x) for x in _ssim_taps(11, 1.5, "cpu")]
        return _capi().ssim_mean(img1, img2.detach(), _TAPS)
    return ssim_torch(img1, img2)


def ssim_torch(img1, img2):
    """The same loss through torch convolutions (CPU loops; the fp32 reference the fused kernels are tested against)."""
    C1, C2 = 0.01 * 0.01, 0.03 * 0.03
    ch = img1.shape[0]
    g = _ssim_taps(11, 1.5, img1.device)
    wv = g.reshape(1, 1, 11, 1).expand(ch, 1, 11, 1).contiguous()
    wh = g.reshape(1, 1, 1, 11).expand(ch, 1, 1, 11).contiguous()
    conv = lambda x: F.conv2d(F.conv2d(x.unsqueeze(0), wv, padding=(5, 0), groups=ch), wh, padding=(0, 5), groups=ch).squeeze(0)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq
    s2 = conv(img2 * img2) - mu2_sq
    s12 = conv(img1 * img2) - mu12
    m = ((2.0 * mu12 + C1) * (2.0 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean()


def _adam(groups, device):
    """Adam as Gaussian.cc:144-175 configures it (eps 1e-15). GPU parameters: the fused one-kernel-per-tensor step
    (capi.FusedAdam, same state layout); CPU parameters (oracle-driven test loops): torch.optim.Adam."""
    if torch.device(device).type == "cuda":
        return _capi().FusedAdam(groups, lr=0.0, eps=1e-15)
    return torch.optim.Adam(groups, lr=0.0, eps=1e-15)


# ---- pose parameterisation (include/Utils.h:56-77, Utils.cc:170-179) ----------------------------
def rt2T(quat, trans):
    """quat [4,1] un-normalised (r,x,y,z), trans [3,1] -> Tcw [4,4]. GPU parameters: one kernel forwards, one backwards
    (csrc/gsr_train.h: the same formulas; as scalar-tensor arithmetic it is ~120 launches per tracking iteration)."""
    if quat.is_cuda and quat.dtype == torch.float32 and trans.dtype == torch.float32:
        return _capi().rt2T(quat, trans)
    q = quat.reshape(4)
    q = q / torch.sqrt((q * q).sum())
    r, x, y, z = q[0], q[1], q[2], q[3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]).reshape(3, 3)
    top = torch.cat([R, trans.reshape(3, 1)], 1)
    return torch.cat([top, torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=top.device, dtype=top.dtype)], 0)


def rot_to_quat(R):
    """Rotation matrix -> (w,x,y,z), as cv::Quatd::createFromRotMat gives InitCameraPose (Gaussian.cc:97-128)."""
    R = R.double()
    t = R.trace()
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(torch.argmax(torch.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return torch.tensor([float(v) for v in q], dtype=torch.float32)


@dataclass
class Frame:
    rgb: torch.Tensor      # [3,H,W] 0..1
    depth: torch.Tensor    # [H,W] metres, 0 = invalid
    Tcw: torch.Tensor      # [4,4]


class GaussianMap:
    """Parameter store + optimisers (reference class Gaussian)."""
    NAMES = ("xyz", "rgb", "unnorm_quat", "logit_opacities", "log_scales")

    def __init__(self, cfg: Config, fx: float, fy: float, device="cuda"):
        self.cfg, self.fx, self.fy, self.device = cfg, fx, fy, torch.device(device)
        self.xyz = self.rgb = self.unnorm_quat = self.logit_opacities = self.log_scales = None
        self.opt = None
        self.scene_radius = 1.0
        self.cam_quat = self.cam_trans = self.opt_pose = None

    def __len__(self):
        return 0 if self.xyz is None else int(self.xyz.shape[0])

    def _new_params(self, pts, cols):
        n = pts.shape[0]
        dev = self.device
        quat = torch.zeros(n, 4, device=dev)
        quat[:, 0] = 1.0
        logit = torch.ones(n, 1, device=dev)
        m = self.cfg.init_scalar_method
        if m in (0, 1):
            from . import capi
            dis = torch.clamp_min(capi.dist2(pts), 1e-7)
            s = torch.sqrt(dis)
            if m == 1:
                s = torch.clamp_max(s, 8 * s.mean())
            logs = torch.log(s).unsqueeze(-1).repeat(1, 3)
        elif m == 2:   # SinglePixel: one pixel wide at its depth
            logs = torch.log(torch.sqrt((pts[:, 2] / ((self.fx + self.fy) * 0.5)) ** 2)).unsqueeze(-1).repeat(1, 3)
        else:
            raise ValueError("Unknown Init Scalar Method")
        return [pts.clone(), cols.clone(), quat, logit, logs.contiguous()]

    def _lrs(self):
        c = self.cfg
        return (c.lr_mean3d, c.lr_rgb, c.lr_rotation, c.lr_opacities, c.lr_scales)

    def add_points(self, pts, cols):
        """Gaussian.cc:50-95: first call creates the optimiser, later calls concatenate."""
        pts, cols = pts.to(self.device, torch.float32), cols.to(self.device, torch.float32)
        new = self._new_params(pts, cols)
        if self.xyz is None:
            for n, t in zip(self.NAMES, new):
                setattr(self, n, t.requires_grad_(True))
            groups = [{"params": [getattr(self, n)], "lr": lr} for n, lr in zip(self.NAMES, self._lrs())]
            self.opt = _adam(groups, self.device)
            return
        for gi, (n, t) in enumerate(zip(self.NAMES, new)):       # CatTensorToOptimizer, Gaussian.cc:236-258
            old = getattr(self, n)
            st = self.opt.state.pop(old, None)
            merged = torch.cat([old.detach(), t], 0).requires_grad_(True)
            if st is not None and "exp_avg" in st:
                st["exp_avg"] = torch.cat([st["exp_avg"], torch.zeros_like(t)], 0)
                st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"], torch.zeros_like(t)], 0)
                self.opt.state[merged] = st
            self.opt.param_groups[gi]["params"][0] = merged
            setattr(self, n, merged)

    def prune(self, remove_mask):
        """Gaussian.cc:196-234 (RemovePoints + PruneOptimizer)."""
        keep = torch.nonzero(~remove_mask).squeeze(-1)
        for gi, n in enumerate(self.NAMES):
            old = getattr(self, n)
            st = self.opt.state.pop(old, None)
            new = old.detach().index_select(0, keep).requires_grad_(True)
            if st is not None and "exp_avg" in st:
                st["exp_avg"] = st["exp_avg"].index_select(0, keep)
                st["exp_avg_sq"] = st["exp_avg_sq"].index_select(0, keep)
                self.opt.state[new] = st
            self.opt.param_groups[gi]["params"][0] = new
            setattr(self, n, new)

    def append_rows(self, params, exp_avg, exp_avg_sq):
        """Rows that arrive with their history (a sharded map re-balancing its cells): params / moments [k, 14] in NAMES order."""
        if params.shape[0] == 0:
            return
        widths = [int(getattr(self, n)[0].numel()) for n in self.NAMES] if len(self) else [3, 3, 4, 1, 3]
        off = 0
        for gi, (n, w) in enumerate(zip(self.NAMES, widths)):
            old = getattr(self, n)
            shape = (params.shape[0],) + tuple(old.shape[1:])
            st = self.opt.state.pop(old, None)
            merged = torch.cat([old.detach(), params[:, off:off + w].reshape(shape)], 0).requires_grad_(True)
            if st is not None and "exp_avg" in st:
                st["exp_avg"] = torch.cat([st["exp_avg"], exp_avg[:, off:off + w].reshape(shape)], 0)
                st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"], exp_avg_sq[:, off:off + w].reshape(shape)], 0)
                self.opt.state[merged] = st
            self.opt.param_groups[gi]["params"][0] = merged
            setattr(self, n, merged)
            off += w

    def low_opacity_mask(self):
        return (torch.sigmoid(self.logit_opacities) < self.cfg.prune_opacities).squeeze(-1)   # Gaussian.cc:180-185

    def init_camera_pose(self, Tcw):
        Tcw = Tcw.to(self.device, torch.float32)
        self.cam_quat = rot_to_quat(Tcw[:3, :3].cpu()).reshape(4, 1).to(self.device).requires_grad_(True)
        self.cam_trans = Tcw[:3, 3].clone().reshape(3, 1).requires_grad_(True)
        # both groups use lrCamQuat, like the reference (Gaussian.cc:149-150)
        self.opt_pose = _adam([{"params": [self.cam_quat], "lr": self.cfg.lr_cam_quat},
                               {"params": [self.cam_trans], "lr": self.cfg.lr_cam_quat}], self.device)
        return Tcw


class SlamRenderer:
    """Reference class Render reduced to its rasterizer-facing part."""

    def __init__(self, gmap: GaussianMap, width: int, height: int, near=0.01, far=100.0, seed=0, rasterizer_cls=None):
        """rasterizer_cls: class with the GaussianRasterizer(raster_settings) / forward(**kw) interface; None = the
        drop-in operator. (The loop-level parity test passes an oracle-backed class to run the SAME loop on the
        CPU checker; the product never does.)"""
        self.map, self.W, self.H = gmap, width, height
        dgr = _dgr()
        dev = gmap.device
        tanfovx, tanfovy = width / (2 * gmap.fx), height / (2 * gmap.fy)
        P = torch.tensor([[1 / tanfovx, 0, 0, 0], [0, 1 / tanfovy, 0, 0],
                          [0, 0, far / (far - near), -(far * near) / (far - near)], [0, 0, 1, 0]], dtype=torch.float32, device=dev)
        self.settings = dgr.GaussianRasterizationSettings(
            image_height=height, image_width=width, tanfovx=tanfovx, tanfovy=tanfovy,
            bg=torch.zeros(3, device=dev), scale_modifier=gmap.cfg.scale_modifier, viewmatrix=torch.eye(4, device=dev),
            projmatrix=P.t().contiguous(), sh_degree=1, campos=torch.zeros(3, device=dev), prefiltered=False)
        self.rasterizer = (rasterizer_cls or dgr.GaussianRasterizer)(self.settings)
        self.rng = torch.Generator().manual_seed(seed)
        self.tracking_counts = self.mapping_counts = 0
        self.fused_pair = True   # render_pair: one fused pass when the rasterizer has forward_pair (False: two passes, like the reference)
        # the pixel terms and scale regularisers of the losses as fused kernels (capi.tracking_pixel_loss / mapping_pixel_loss /
        # scale_regularisers) when the renders are on the GPU; False: the tensor expressions below (what CPU-side oracle loops run)
        self.fused_losses = True
        self._track_act = None   # track(): the (fixed) map's activations, formed once per call

    @staticmethod
    def to_camera(Tcw, mean3D):
        """Render.cc:750-752 moves the means into the camera frame with Tcw.repeat(n,1,1).bmm([x;1]) — n tiny matrix products
        (0.6 ms per call at 1 M splats on MI355X through rocBLAS batched GEMM, forward and backward). The same numbers come
        from one [n,3] x [3,3] product plus the translation; the pose still receives its gradient through autograd."""
        if mean3D.is_cuda and mean3D.dtype == torch.float32 and Tcw.dtype == torch.float32:
            return _capi().to_camera(Tcw, mean3D)   # same product; dL/dTcw from one reduction kernel (a 3 x N x 3 GEMM: 1.7 ms)
        return mean3D @ Tcw[:3, :3].t() + Tcw[:3, 3]

    # Render.cc:711-781 with useRadiusFilter = false
    def splat(self, Tcw, mean3D, rgb, unnorm_quat, logit_opacities, log_scales, mc=None, act=None):
        mc = self.to_camera(Tcw, mean3D) if mc is None else mc
        mean2D = torch.zeros_like(mc, requires_grad=True)
        opac, scales, rots = act if act is not None else self.activations(unnorm_quat, logit_opacities, log_scales)
        image, radii, depth = self.rasterizer(
            means3D=mc, means2D=mean2D, opacities=opac, colors_precomp=rgb, scales=scales, rotations=rots)
        return image, depth, radii

    @staticmethod
    def activations(unnorm_quat, logit_opacities, log_scales):
        """Render.cc:754-758: sigmoid / exp / normalize of the raw parameters."""
        return torch.sigmoid(logit_opacities), torch.exp(log_scales), F.normalize(unnorm_quat)

    def _params(self, tracking):
        g = self.map
        p = [g.xyz, g.rgb, g.unnorm_quat, g.logit_opacities, g.log_scales]
        return [t.detach() for t in p] if tracking else p

    def render_rgb(self, Tcw, tracking=False):               # GSParamRGBUpdata, Render.cc:927-946
        xyz, rgb, q, o, s = self._params(tracking)
        return self.splat(Tcw, xyz, rgb, q, o, s)

    def render_depth(self, Tcw, tracking=False, mc=None, act=None):   # GSParamDepthUpdata, Render.cc:949-981
        xyz, _, q, o, s = self._params(tracking)
        mc = self.to_camera(Tcw, xyz) if mc is None else mc
        z = mc[:, 2:3]
        col = torch.cat([z, torch.ones_like(z), torch.zeros_like(z)], 1)
        if tracking:
            col = col.detach()
        return self.splat(Tcw, xyz, col, q, o, s, mc=mc, act=act)

    # ---- the three hooks a sharded mapper overrides (gsorb-slam_amd/sharded.py:ShardedMapper) ----------------
    def render_pair(self, Tcw, tracking=False):
        """Both renders of one iteration: (colour image [3,H,W], surface (median) depth [1,H,W] — no gradient,
        depth/silhouette render [>=2,H,W]: [0] alpha-blended depth, [1] accumulated opacity)."""
        # the reference forms the camera-frame means and the activations once per render; both renders of an iteration see
        # the same parameters and pose, so they are formed once here (autograd adds the two gradients: same numbers)
        xyz, rgb, q, o, s = self._params(tracking)
        mc = self.to_camera(Tcw, xyz)
        act = self._track_act if tracking and self._track_act is not None else self.activations(q, o, s)
        if self.fused_pair and hasattr(self.rasterizer, "forward_pair"):
            # ONE pass of the rasterizer for both renders (the view matrix is the identity and the means are camera-frame:
            # the depth channel's colour IS mc[:, 2]; tracking detaches it, like render_depth)
            mean2D = torch.zeros_like(mc, requires_grad=True)
            rimage, rdepth, _, rsur = self.rasterizer.forward_pair(
                means3D=mc, means2D=mean2D, opacities=act[0], colors_precomp=rgb, scales=act[1], rotations=act[2],
                detach_depth_color=tracking)
            return rimage, rsur, rdepth
        rdepth, _, _ = self.render_depth(Tcw, tracking, mc=mc, act=act)
        rimage, rsur, _ = self.splat(Tcw, xyz, rgb, q, o, s, mc=mc, act=act)
        return rimage, rsur, rdepth

    def _reduce_regularisers(self, sum_over, sum_spread, count):
        """(sum of max-scale excess, sum of max-min spread, number of oversized splats) over the whole map."""
        return sum_over, sum_spread, count

    def _sync_pose_grads(self):
        """Single process: the pose gradient is already complete."""

    def _replicated_term_grad_scale(self):
        """Weight of the gradient of a loss term that every rank of a sharded run evaluates in full (the feature
        reprojection term of track(): it depends on the pose only, not on a rank's shard). Single process: 1."""
        return 1.0

    # Render.cc:420-483
    def mapping_loss(self, fr: Frame):
        g, c = self.map, self.map.cfg
        Tcw = fr.Tcw.to(g.device)
        rimage, rsur, rdepth = self.render_pair(Tcw)
        strict = getattr(self, "strict_empty_terms", False)
        fused = self.fused_losses and rimage.is_cuda and not strict
        max_scalar = 0.1 * g.scene_radius
        if fused:
            # the same terms as below: the pixel terms in two launches, the regularisers in two (own hook: a sharded mapper
            # sums the regulariser sums over its ranks first and keeps the tensor expressions)
            pix, _ = _capi().mapping_pixel_loss(rimage, rdepth[0], rsur[0], rdepth[1], fr.rgb, fr.depth, c.im_weight_mapping * c.lam,
                                                c.depth_weight_mapping, c.sur_depth_weight_mapping)
            loss = pix + (c.im_weight_mapping * (1 - c.lam)) * (1.0 - ssim(rimage, fr.rgb))
            if type(self)._reduce_regularisers is SlamRenderer._reduce_regularisers:
                return loss + _capi().scale_regularisers(g.log_scales, max_scalar, c.reg_long_weight, c.reg_scalar_weight)[0]
            return loss + self._regularisers(max_scalar, strict)
        valid = fr.depth > 0
        valid_sur = (fr.depth > 0) & (rdepth[1] > 0.99)
        image_loss = c.lam * l1_mapping(rimage, fr.rgb) + (1 - c.lam) * (1.0 - ssim(rimage, fr.rgb))
        depth_loss = l1_mapping(rdepth[0], fr.depth, valid.detach())
        # DELIBERATE DEVIATION: with no surface pixel the reference's masked_select(mask).mean() over an empty selection is NaN
        # (Utils.cc:39-44, Render.cc:455) and so is its reg_long over an empty set of oversized splats; here both terms are
        # an exact zero with a zero gradient (and nothing asks the host), so a step that would poison Adam's moments with
        # NaN in the reference is an ordinary step here. (strict_empty_terms=True reproduces the reference's NaN.)
        n_sur = valid_sur.sum()
        sur_loss = torch.where(valid_sur.detach(), torch.abs(rsur[0] - fr.depth), torch.zeros_like(fr.depth)).sum() / n_sur.clamp_min(1)
        if strict:
            sur_loss = torch.where(n_sur > 0, sur_loss, torch.full_like(sur_loss, float("nan")))
        return (c.im_weight_mapping * image_loss + c.depth_weight_mapping * depth_loss + c.sur_depth_weight_mapping * sur_loss
                + self._regularisers(max_scalar, strict))

    def _regularisers(self, max_scalar, strict=False):
        """reg_long_weight * reg_long + reg_scalar_weight * reg_scalar (Render.cc:449-462) as tensor expressions."""
        g, c = self.map, self.map.cfg
        sc = torch.exp(g.log_scales)
        # Render.cc:449-462 gathers the rows of every scale COMPONENT above the limit (torch::where(...)[0]: a row with two
        # oversized axes counts twice) and takes max / min of those rows. The same sums with the multiplicity as a weight,
        # without the index list (nonzero: another host sync)
        w = (sc > max_scalar).sum(1).to(sc.dtype)
        mx, mn = sc.max(1)[0], sc.min(1)[0]
        over, spread, cnt = self._reduce_regularisers((w * (mx - max_scalar)).sum(), (w * (mx - mn)).sum(), w.sum())
        cnt = torch.as_tensor(cnt, dtype=sc.dtype, device=sc.device)
        reg_scalar = over
        reg_long = torch.where(cnt > 0, spread / cnt.clamp_min(1), torch.zeros_like(spread))   # mean over the oversized splats
        if strict:
            reg_long = torch.where(cnt > 0, reg_long, torch.full_like(reg_long, float("nan")))
        return c.reg_long_weight * reg_long + c.reg_scalar_weight * reg_scalar

    def mapping_iteration(self, frames):
        g = self.map
        k = int(torch.randint(0, len(frames), (1,), generator=self.rng))
        loss = self.mapping_loss(frames[k])
        loss.backward()
        with torch.no_grad():
            g.opt.step()
            g.opt.zero_grad()
            self.mapping_counts += 1
        return float(loss.detach())

    def map_frames(self, frames, iters=None):
        return [self.mapping_iteration(frames) for _ in range(iters or self.map.cfg.mapping_iters)]

    # Render.cc:985-1141
    def track(self, frame: Frame, Tcw_init, iters=None, matches=None, K=None):
        """matches = (obs [M,3,1] pixels (u,v,1), Xw4 [M,4,1], inv_sigma2 [M,1]) from the feature front end, or None."""
        g, c = self.map, self.map.cfg
        g.init_camera_pose(Tcw_init)
        best_q, best_t = g.cam_quat.detach().clone(), g.cam_trans.detach().clone()
        min_loss, last_loss = float("inf"), 0.0
        iters = iters or c.tracking_iters
        inline = None
        history = []
        if self.fused_losses and g.xyz.is_cuda:   # the map does not move while the pose is tracked
            with torch.no_grad():
                self._track_act = self.activations(g.unnorm_quat, g.logit_opacities, g.log_scales)
        try:
            return self._track_loop(frame, iters, matches, K, best_q, best_t, min_loss, last_loss, inline, history)
        finally:
            self._track_act = None

    def _track_loop(self, frame, iters, matches, K, best_q, best_t, min_loss, last_loss, inline, history):
        g, c = self.map, self.map.cfg
        for it in range(iters):
            Tcw = rt2T(g.cam_quat.clone(), g.cam_trans.clone())
            lrpj = None
            if matches is not None:
                obs, Xw4, inv_s2 = matches
                M = obs.shape[0]
                Xc = Tcw.unsqueeze(0).repeat(M, 1, 1).bmm(Xw4).transpose(1, 2)[..., :3]
                Xc = Xc / Xc[..., 0, 2].reshape(M, 1, 1)
                uv = K.unsqueeze(0).repeat(M, 1, 1).bmm(Xc.transpose(1, 2))
                e = (uv - obs)[:, 0:2, 0]
                werr = (e * e * inv_s2).sum(1, keepdim=True)
                if inline is None:
                    inline = torch.ones_like(werr, dtype=torch.bool)
                if it == int(iters / 2.0):
                    inline = werr < 5.991
                lrpj = werr.masked_select(inline).sum()
                # a sharded run sums the pose gradients of the ranks: a term every rank holds in full must enter each rank's
                # gradient with weight 1/world (its VALUE is unchanged)
                gs = self._replicated_term_grad_scale()
                if gs != 1.0:
                    lrpj = lrpj * gs + lrpj.detach() * (1.0 - gs)
            rimage, rsur, rdepth = self.render_pair(Tcw, tracking=True)
            if self.fused_losses and rimage.is_cuda:   # Render.cc:1088-1105 in two launches
                loss = _capi().tracking_pixel_loss(rimage, rsur[0] if c.use_sur_depth else rdepth[0], rdepth[1], frame.rgb, frame.depth,
                                                   c.im_weight_tracking, c.depth_weight_tracking, depth_is_surface=c.use_sur_depth)
                if matches is not None:
                    loss = loss + c.feature_weight_tracking * lrpj
            else:
                certain = (rdepth[1] > 0.99) & ~torch.isnan(frame.depth)
                image_l1 = l1_tracking(rimage, frame.rgb, certain.unsqueeze(0).repeat(3, 1, 1).detach())
                depth_l1 = l1_tracking(rsur[0] if c.use_sur_depth else rdepth[0], frame.depth, certain.detach())
                loss = c.im_weight_tracking * image_l1 + c.depth_weight_tracking * depth_l1 + c.feature_weight_tracking * (lrpj if lrpj is not None else Tcw.sum() * 0)
            loss.backward()
            with torch.no_grad():
                lv = float(loss.detach())
                history.append(lv)
                if not math.isnan(lv) and lv < min_loss:
                    best_q, best_t, min_loss = g.cam_quat.detach().clone(), g.cam_trans.detach().clone(), lv
                if abs(last_loss - lv) < 10e-4:
                    break
                last_loss = lv
                self._sync_pose_grads()
                g.opt_pose.step()
                g.opt_pose.zero_grad()
                self.tracking_counts += 1
        return rt2T(best_q, best_t).detach(), history

    # Render.cc:557-594 + ProjectPixel :618-700 (back-projection of the masked pixels)
    def densify(self, frame: Frame):
        g, c = self.map, self.map.cfg
        with torch.no_grad():
            Tcw = frame.Tcw.to(g.device)
            rim, rdep = self._densify_renders(Tcw)
            gray = (rim[0] * 299 + rim[1] * 587 + rim[2] * 114) / 1000
            black = gray < 50 / 255.0
            diff = torch.abs(frame.depth - rdep[0])
            dmask = (diff < 0.05) & (frame.depth > 0) & (rdep[0] > 0)
            if bool(dmask.any()):
                vals = diff.masked_select(dmask)
                th = float(vals.sum() / dmask.sum()) + c.median_mul * float(vals.median())
            else:
                th = 0.0
            th = max(th, 0.01)
            add = (~(rdep[1] > 0.99) & black & (diff > th)) | (rdep[1] < 0.8)
            add = add & (frame.depth > 0)
            v, u = torch.nonzero(add, as_tuple=True)
            if v.numel() == 0:
                return 0
            z = frame.depth[v, u]
            cx, cy = (self.W - 1) / 2.0, (self.H - 1) / 2.0
            pc = torch.stack([(u.float() - cx) * z / g.fx, (v.float() - cy) * z / g.fy, z], 1)
            Twc = torch.inverse(Tcw)
            pw = pc @ Twc[:3, :3].t() + Twc[:3, 3]
            cols = frame.rgb[:, v, u].t().contiguous()
            own = self._owned(pw)                                 # (a sharded map: only the points of this rank's cell)
            if own is not None:
                pw, cols = pw[own], cols[own]
            if pw.shape[0]:
                g.add_points(pw, cols)
            return int(pw.shape[0])

    def _densify_renders(self, Tcw):
        rim, _, _ = self.render_rgb(Tcw, tracking=True)
        rdep, _, _ = self.render_depth(Tcw, tracking=True)
        return rim, rdep

    def _owned(self, pw):
        return None

    def remove_low_opacity(self):                              # Render.cc:598-616
        m = self.map.low_opacity_mask()
        n = int(m.sum())
        if n > 0:
            self.map.prune(m)
        return n
