"""diff_gaussian_rasterization — MI355X implementation of the Python operator.

Same public surface as the reference package
(Thirdparty/diff_gaussian_rasterization/diff_gaussian_rasterization/__init__.py:17-196):
`GaussianRasterizationSettings`, `GaussianRasterizer(nn.Module)` with `forward` / `markVisible`,
`rasterize_gaussians`, and an extension module `_C` exporting `rasterize_gaussians`,
`rasterize_gaussians_backward`, `mark_visible` with the reference's argument lists. Put the
directory that contains this package (gsorb-slam_amd/) on sys.path and
`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
works as in scripts/replay.py:324-325.

`_C` is the libtorch host layer (gsorb-slam_amd/torch_ext) over the HIP kernels; importing
this package without it fails loudly — there is no fallback.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C  # noqa: F401  (ImportError here means the extension has not been built)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


def _on(t, dev):
    """Absent inputs arrive as empty CPU tensors (torch.Tensor([])), like in the reference."""
    return t if t.device == dev else t.to(dev)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        dev = means3D.device
        sh, colors_precomp, scales, rotations, cov3Ds_precomp = (_on(t, dev) for t in (
            sh, colors_precomp, scales, rotations, cov3Ds_precomp))
        args = (raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy,
                raster_settings.image_height, raster_settings.image_width, sh, raster_settings.sh_degree,
                raster_settings.campos, raster_settings.prefiltered)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth = _C.rasterize_gaussians(*args)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, opacities)
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _radii, _depth):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer, opacities) = ctx.saved_tensors
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
                geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer)
        # the graph node knows how often it ran: the first backward skips the re-zero of the per-splat
        # accumulators (stages 2|4), a repeated one (retain_graph) starts with a clear (1|2|4)
        stages = 7 if getattr(ctx, "backward_ran", False) else 6
        ctx.backward_ran = True
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _C.rasterize_gaussians_backward_staged(*args, stages)

        def like(g, x):
            return None if x.numel() == 0 else g.reshape(x.shape)

        return (grad_means3D, grad_means2D, like(grad_sh, sh), like(grad_colors_precomp, colors_precomp),
                grad_opacities.reshape(opacities.shape), like(grad_scales, scales), like(grad_rotations, rotations),
                like(grad_cov3Ds_precomp, cov3Ds_precomp), None)


class _RasterizeGaussiansPair(torch.autograd.Function):
    """The fused pair (new capability): colours and [view depth, 1] blended in ONE pass — what GSORB-SLAM renders as two passes
    with the same geometry (src/Render.cc:927-981, scripts/replay.py:324-325). Outputs: color [3,H,W], ds [2,H,W] (alpha-blended
    depth, accumulated opacity; background 0), radii, depth (median). color and ds are differentiable."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                detach_depth_color):
        dev = means3D.device
        sh, colors_precomp, scales, rotations, cov3Ds_precomp = (_on(t, dev) for t in (
            sh, colors_precomp, scales, rotations, cov3Ds_precomp))
        rs = raster_settings
        ctx.detach_depth_color = bool(detach_depth_color)
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered)
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth, ds = _C.rasterize_gaussians_pair(*args)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, opacities)
        ctx.mark_non_differentiable(radii, depth)
        return color, ds, radii, depth

    @staticmethod
    def backward(ctx, grad_color, grad_ds, _radii, _depth):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer, opacities) = ctx.saved_tensors
        if grad_color is None:
            grad_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        if grad_ds is None:
            grad_ds = torch.zeros((2, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        stages = 7 if getattr(ctx, "backward_ran", False) else 6
        ctx.backward_ran = True
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _C.rasterize_gaussians_pair_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_color, grad_ds, sh, rs.sh_degree, rs.campos, geomBuffer,
            ctx.num_rendered, binningBuffer, imgBuffer, stages, ctx.detach_depth_color)

        def like(g, x):
            return None if x.numel() == 0 else g.reshape(x.shape)

        return (grad_means3D, grad_means2D, like(grad_sh, sh), like(grad_colors_precomp, colors_precomp),
                grad_opacities.reshape(opacities.shape), like(grad_scales, scales), like(grad_rotations, rotations),
                like(grad_cov3Ds_precomp, cov3Ds_precomp), None, None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)

    def forward_pair(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                     cov3D_precomp=None, detach_depth_color=False):
        """The fused pair: (color [3,H,W], ds [2,H,W], radii, depth) — see _RasterizeGaussiansPair. detach_depth_color: the depth
        channel's colours are constants in the backward (GSORB-SLAM's tracking iterations detach their [z, 1, 0] colours)."""
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        return _RasterizeGaussiansPair.apply(means3D, means2D, empty if shs is None else shs,
                                             empty if colors_precomp is None else colors_precomp, opacities,
                                             empty if scales is None else scales, empty if rotations is None else rotations,
                                             empty if cov3D_precomp is None else cov3D_precomp, self.raster_settings,
                                             detach_depth_color)
