"""TEST INFRASTRUCTURE: the CPU oracle wrapped as a torch.autograd.Function with the Python operator's
interface (GaussianRasterizer(raster_settings)(means3D=..., colors_precomp=..., ...) -> (color, radii, depth)).

Lets the tests run the SAME optimisation loops (gsorb-slam_amd/harness.py, sharded.py) once on the HIP
operator and once on the checker, on CPU tensors. Never imported by the product package.
Interface mirrored: Thirdparty/diff_gaussian_rasterization/diff_gaussian_rasterization/__init__.py:17-196.
"""
import numpy as np
import torch

from oracle import oracle


class _Cam:
    """What oracle.Oracle reads from a camera (gsorb-slam_amd/synthetic.py:Camera)."""

    def __init__(self, rs):
        n = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        self.width, self.height = int(rs.image_width), int(rs.image_height)
        self.tanfovx, self.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
        self.viewmatrix, self.projmatrix = n(rs.viewmatrix), n(rs.projmatrix)
        self.campos, self.bg = n(rs.campos), n(rs.bg)
        self.scale_modifier, self.sh_degree = float(rs.scale_modifier), int(rs.sh_degree)


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, omp):
        n = lambda t: None if t is None or t.numel() == 0 else np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
        o = oracle.Oracle(omp)
        f = o.forward(copy_stages=False, means3D=n(means3D), opacities=n(opacities), cam=_Cam(rs), colors=n(colors_precomp),
                      shs=n(sh), scales=n(scales), rotations=n(rotations), cov3D_precomp=n(cov3Ds_precomp))
        ctx.o = o
        ctx.shapes = [None if t is None else tuple(t.shape) for t in (sh, colors_precomp, scales, rotations, cov3Ds_precomp)]
        ctx.opac_shape = tuple(opacities.shape)
        color, radii, depth = torch.from_numpy(f.color), torch.from_numpy(f.radii.copy()), torch.from_numpy(f.depth)
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, g_color, _r, _d):
        b = ctx.o.backward(np.ascontiguousarray(g_color.detach().cpu().numpy(), dtype=np.float32))
        t = torch.from_numpy
        like = lambda g, shp: None if shp is None or int(np.prod(shp)) == 0 else t(g).reshape(shp)
        sh_s, col_s, sc_s, rot_s, cov_s = ctx.shapes
        return (t(b.dL_dmeans3D), t(b.dL_dmeans2D), like(b.dL_dsh, sh_s), like(b.dL_dcolors, col_s),
                t(b.dL_dopacity).reshape(ctx.opac_shape), like(b.dL_dscales, sc_s), like(b.dL_drotations, rot_s),
                like(b.dL_dcov3D, cov_s), None, None)


class OracleRasterizer(torch.nn.Module):
    """Drop-in for diff_gaussian_rasterization.GaussianRasterizer on CPU tensors, backed by the oracle."""
    omp = True   # the OpenMP build: same arithmetic, float atomics in the backward (summation order varies)

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = torch.zeros(0)
        return _OracleRasterize.apply(means3D, means2D, e if shs is None else shs, e if colors_precomp is None else colors_precomp,
                                      opacities, e if scales is None else scales, e if rotations is None else rotations,
                                      e if cov3D_precomp is None else cov3D_precomp, self.raster_settings, self.omp)
