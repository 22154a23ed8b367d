"""diff_gaussian_rasterization -- MI355X-native drop-in for the reference's rasterizer package.

Public surface kept identical to DGR/diff_gaussian_rasterization/__init__.py so gaussian_renderer / train.py /
render.py call it unchanged:
  * GaussianRasterizationSettings: same 15 fields in the same order (:176-191)
  * GaussianRasterizer(raster_settings)(means3D, means2D, opacities, tongue_class, shs, colors_precomp, scales,
    rotations, cov3D_precomp) -> (color, radii, coord, mcoord, depth, mdepth, alpha, tongue, normal)  (:209-243, :105)
  * GaussianRasterizer.markVisible(positions)  (:198-207)
  * rasterize_gaussians(...) / _RasterizeGaussians autograd.Function with the reference's gradient slots (:161-172):
    grads for means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp; None for
    tongue_class (Q2) and raster_settings.
The native code behind it is the C-ABI HIP library (include/ed3dgs.h) reached through `_C`.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    kernel_size: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    require_depth: bool
    require_coord: bool
    debug: bool


def _snapshot(args):
    return tuple(a.detach().cpu().clone() if torch.is_tensor(a) else a for a in args)


_ZEROS = {}


def _zeros(shape, like):
    """Read-only zero plane standing in for the upstream gradient of an output the loss does not use (autograd would
    otherwise allocate and fill one per call; the C ABI only reads it)."""
    key = (tuple(shape), like.device, like.dtype)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(shape, dtype=like.dtype, device=like.device)
    return z


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, tongue_class, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        ctx.set_materialize_grads(False)   # unused outputs arrive as None in backward, not as freshly filled zeros
        call = (rs.bg, means3D, colors_precomp, opacities, tongue_class, scales, rotations, rs.scale_modifier,
                cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, rs.image_height,
                rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.require_coord, rs.require_depth,
                rs.debug)
        if rs.debug:
            # debug aid of the reference (:90-97): keep a CPU copy of the arguments and dump it if the call throws
            saved = _snapshot(call)
            try:
                out = _C.rasterize_gaussians(*call)
            except Exception:
                torch.save(saved, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            out = _C.rasterize_gaussians(*call)
        (num_rendered, color, coord, mcoord, alpha, tongue, normal, depth, mdepth, radii, geom_buf, binning_buf,
         img_buf) = out
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, normal, radii, sh, geom_buf,
                              binning_buf, img_buf, alpha)
        ctx.mark_non_differentiable(radii)
        return color, radii, coord, mcoord, depth, mdepth, alpha, tongue, normal

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_coord, grad_mcoord, grad_depth, grad_mdepth, grad_alpha,
                 grad_tongue, grad_normal):
        # grad_radii and grad_tongue are ignored, as in the reference (:108)
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, normal, radii, sh, geom_buf, binning_buf, img_buf,
         alpha) = ctx.saved_tensors
        H, W = alpha.shape[-2], alpha.shape[-1]
        z3, z1 = (3, H, W), (1, H, W)
        grad_color = grad_color if grad_color is not None else _zeros(z3, alpha)
        grad_coord = grad_coord if grad_coord is not None else _zeros(z3, alpha)
        grad_mcoord = grad_mcoord if grad_mcoord is not None else _zeros(z3, alpha)
        grad_depth = grad_depth if grad_depth is not None else _zeros(z1, alpha)
        grad_mdepth = grad_mdepth if grad_mdepth is not None else _zeros(z1, alpha)
        grad_alpha = grad_alpha if grad_alpha is not None else _zeros(z1, alpha)
        grad_normal = grad_normal if grad_normal is not None else _zeros(z3, alpha)
        call = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.kernel_size, grad_color, grad_coord,
                grad_mcoord, grad_depth, grad_mdepth, grad_alpha, grad_normal, normal, sh, rs.sh_degree, rs.campos,
                geom_buf, ctx.num_rendered, binning_buf, img_buf, alpha, rs.require_coord, rs.require_depth, rs.debug)
        if rs.debug:
            saved = _snapshot(call)
            try:
                res = _C.rasterize_gaussians_backward(*call)
            except Exception:
                torch.save(saved, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            res = _C.rasterize_gaussians_backward(*call)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = res
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, None, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, tongue_class, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, tongue_class, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, tongue_class, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, tongue_class, scales, rotations,
                                   cov3D_precomp, self.raster_settings)

    def integrate(self, points3D, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                  cov3D_precomp=None, view2gaussian_precomp=None):
        """GaussianRasterizer.integrate (DGR/diff_gaussian_rasterization/__init__.py:245-312): the mesh-extraction probe.
        Returns (color [9,H,W], alpha_integrated, color_integrated, point_coordinate, point_sdf, radii)."""
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        view2gaussian_precomp = empty if view2gaussian_precomp is None else view2gaussian_precomp
        # the reference passes kernel_size 0.0 and an all-zero subpixel_offset here (:271, :289); the offset is unused
        args = (rs.bg, points3D, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3D_precomp,
                view2gaussian_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, 0.0, None, rs.image_height,
                rs.image_width, shs, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        with torch.no_grad():
            (num_rendered, color, alpha_integrated, color_integrated, point_coordinate, point_sdf, radii, geomBuffer,
             binningBuffer, imgBuffer) = _C.integrate_gaussians_to_points(*args)
        return color, alpha_integrated, color_integrated, point_coordinate, point_sdf, radii
