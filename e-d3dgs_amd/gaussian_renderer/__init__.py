"""gaussian_renderer -- MI355X-native drop-in for the reference's render glue (gaussian_renderer/__init__.py).

`render`, `render_tongue`, `render_without_tongue` keep the reference's signatures (:8, :145, :289) and result-dict
keys (:128-142), so train.py / render.py call them unchanged.  What differs is how it runs:
  * scalars of the raster settings are plain Python numbers (the reference wraps six of them in 0-d CUDA tensors
    that pybind turns back into host scalars -> six device syncs per call, :29-37);
  * the camera time is passed to the deformation network as a float (no (P,1) tensor, :45);
  * deformation = one fused HIP launch per direction, activations fused, rasterizer = HIP tile kernels.
`integrate` (mesh probing) is below; `render_old` (DGR-old) is outside the scope.
"""
import math
import os

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from ed3dgs_amd.activations import fused_activations


# False: the activations as a launch of their own behind the deformation (round 3's form; A/B runs: ED3DGS_SEPARATE_ACTIVATIONS=1)
FUSE_ACTIVATIONS = os.environ.get("ED3DGS_SEPARATE_ACTIVATIONS", "0") in ("", "0")


def _standard_activations(pc):
    """True when the model's activation callables are the ones the reference installs
    (scene/gaussian_model.py:37-45): only then may the fused HIP activation kernel stand in for them."""
    return (getattr(pc, "scaling_activation", None) is torch.exp and getattr(pc, "opacity_activation", None) is torch.sigmoid
            and getattr(pc, "rotation_activation", None) is torch.nn.functional.normalize)


def _settings(viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, require_coord, require_depth):
    dev = pc.get_xyz.device
    return GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        kernel_size=float(kernel_size),
        bg=bg_color.to(dev),
        scale_modifier=float(scaling_modifier),
        viewmatrix=viewpoint_camera.world_view_transform.to(dev),
        projmatrix=viewpoint_camera.full_proj_transform.to(dev),
        sh_degree=int(pc.active_sh_degree),
        campos=viewpoint_camera.camera_center.to(dev),
        prefiltered=False,
        require_depth=bool(require_depth),
        require_coord=bool(require_coord),
        debug=bool(pipe.debug),
    )


def _eval_sh():
    """The caller's utils.sh_utils.eval_sh when render() runs inside the reference's source tree (:5), else the same
    polynomials from ed3dgs_amd.sh."""
    try:
        from utils.sh_utils import eval_sh
    except ImportError:
        from ed3dgs_amd.sh import eval_sh
    return eval_sh


_ZERO_SCALARS = {}


def _zero_scalar(device, dtype):
    key = (str(device), dtype)
    z = _ZERO_SCALARS.get(key)
    if z is None:
        z = _ZERO_SCALARS[key] = torch.zeros(1, device=device, dtype=dtype)
    return z


def _render_impl(viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, require_coord, require_depth,
                 override_color, cam_no, iter, num_down_emb_c, num_down_emb_f, disable_filter3D, select):
    """select: None = all Gaussians; +1 = rows with round(tongue_class) != 0 (render_tongue :155,246-253);
    -1 = the complement (render_without_tongue :299,390-397).  Deformation always runs on all P."""
    means3D = pc.get_xyz
    # zero tensor whose .grad receives the screen-space mean gradients (train.py:346-348 reads it).  Nothing reads its
    # values, so it is a fresh leaf over ONE cached zero (expanded view: no fill launch per call; an in-place write raises)
    screenspace_points = _zero_scalar(means3D.device, means3D.dtype).expand(means3D.shape).requires_grad_()
    raster_settings = _settings(viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, require_coord,
                                require_depth)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    opacity = pc._opacity
    # the model's split SH storage goes to the deformation as it is (scene/gaussian_model.py:57-58): get_features (:128-131)
    # would copy both tensors into one per call, and autograd would copy the gradient back out of it slice by slice
    shs_rest = None
    if (torch.is_tensor(getattr(pc, "_features_dc", None)) and torch.is_tensor(getattr(pc, "_features_rest", None))
            and pc._features_dc.dim() == 3 and pc._features_rest.dim() == 3 and pc._features_rest.shape[1] > 0):
        shs, shs_rest = pc._features_dc, pc._features_rest
    else:
        shs = pc.get_features
    # pipe.compute_cov3D_python (:68-72): the reference hands scales = rotations = None to the deformation network, whose
    # first line subscripts them (scene/deformation.py:109) -- the branch raises TypeError there.  What the branch is written
    # to do is kept here instead: the covariance comes from the model's UNDEFORMED scaling / rotation through
    # pc.get_covariance (scene/gaussian_model.py:143-144), the deformation still moves means, opacity and SH (it is given
    # the base scaling / rotation it needs; their deformed values are then unused), and the rasterizer gets cov3D_precomp.
    cov3D_precomp = pc.get_covariance(scaling_modifier) if pipe.compute_cov3D_python else None
    scales = pc._scaling
    rotations = pc._rotation

    # north_star "the deformation MLP fused with the preprocess": when the model's activations are the reference's own, the MLP
    # kernel's epilogue writes the rasterizer's inputs (exp / normalize / sigmoid of the final values) next to the raw ones and
    # the activation backward runs inside the deformation backward's first pass -- no activation launch in either direction
    fuse = (FUSE_ACTIVATIONS and cov3D_precomp is None and _standard_activations(pc) and getattr(pc._deformation, "supports_activated", False)
            and (disable_filter3D or getattr(pc, "fused_filter3D", False)))
    kw = dict(activated=(None if disable_filter3D else pc.filter_3D,)) if fuse else {}
    # (viewpoint_camera.time goes in as a Python number; the reference builds a (P,1) tensor of it (:45) of which the network
    # reads element [0,0] (scene/deformation.py:58) -- a tensor is accepted just the same)
    (means3D_final, scales_final, rotations_final, opacity_final, shs_final, extras) = pc._deformation(
        means3D, scales, rotations, opacity, float(viewpoint_camera.time), cam_no, pc, None, shs, iter=iter,
        num_down_emb_c=num_down_emb_c, num_down_emb_f=num_down_emb_f, sh_coefs_rest=shs_rest, **kw)

    if fuse:
        opacity = opacity_final
    elif cov3D_precomp is not None:
        scales_final = rotations_final = None
        if disable_filter3D:
            opacity = pc.opacity_activation(opacity_final)
        else:
            _, opacity = pc.apply_scaling_n_opacity_with_3D_filter(opacity=opacity_final, scales=pc._scaling)
    elif _standard_activations(pc) and (disable_filter3D or getattr(pc, "fused_filter3D", False)):
        # one fused launch per direction (csrc/activations.hip) instead of normalize / exp / sigmoid (/ 3D filter)
        scales_final, rotations_final, opacity = fused_activations(
            scales_final, rotations_final, opacity_final, None if disable_filter3D else pc.filter_3D)
    else:
        rotations_final = pc.rotation_activation(rotations_final)
        if disable_filter3D:
            scales_final = pc.scaling_activation(scales_final)
            opacity = pc.opacity_activation(opacity_final)
        else:
            scales_final, opacity = pc.apply_scaling_n_opacity_with_3D_filter(opacity=opacity_final, scales=scales_final)

    colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            eval_sh = _eval_sh()
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.to(means3D.device).repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized) + 0.5, 0.0)
    else:
        colors_precomp = override_color

    tongue_class = pc.tongue_class
    m3, m2, sh_r, op_r, sc_r, ro_r, cp_r, cov_r, tg_r = (means3D_final, screenspace_points, shs_final, opacity,
                                                       scales_final, rotations_final, colors_precomp, cov3D_precomp,
                                                       tongue_class)
    mask = None
    if select is not None:
        is_tongue = torch.round(tongue_class).bool().reshape(-1)  # filter_mask of the reference (:155, :299)
        mask = is_tongue if select > 0 else ~is_tongue
        pick = lambda t: None if t is None else t[mask]
        m3, m2, sh_r, op_r, sc_r, ro_r, cp_r, cov_r, tg_r = [pick(t) for t in (m3, m2, sh_r, op_r, sc_r, ro_r, cp_r,
                                                                              cov_r, tg_r)]
    if colors_precomp is not None:
        sh_r = None

    outputs = rasterizer(means3D=m3, means2D=m2, shs=sh_r, colors_precomp=cp_r, opacities=op_r, tongue_class=tg_r,
                         scales=sc_r, rotations=ro_r, cov3D_precomp=cov_r)
    assert len(outputs) == 9, "only (depth-)diff-gaussian-rasterization from RaDe-GS supported!"
    (rendered_image, radii, expected_coord, median_coord, expected_depth, median_depth, rendered_alpha,
     rendered_tongue, rendered_normal) = outputs
    # note: for the masked variants `radii` / `visibility_filter` have the subset's length, as in the reference
    return {"render": rendered_image,
            "mask": rendered_alpha,
            "expected_coord": expected_coord,
            "median_coord": median_coord,
            "expected_depth": expected_depth,
            "median_depth": median_depth,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii,
            "normal": rendered_normal,
            "sh_coefs_final": shs_final,
            "extras": extras,
            "deformed_gaussian_positions": means3D_final,
            "tongue_mask": rendered_tongue}


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, kernel_size, scaling_modifier=1.0,
           require_coord: bool = True, require_depth: bool = True, override_color=None, cam_no=None, iter=None,
           train_coarse=False, num_down_emb_c=5, num_down_emb_f=5, disable_filter3D=True):
    """Render the scene.  Background tensor (bg_color) must be on the GPU."""
    return _render_impl(viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, require_coord,
                        require_depth, override_color, cam_no, iter, num_down_emb_c, num_down_emb_f, disable_filter3D,
                        None)


def render_tongue(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, kernel_size, scaling_modifier=1.0,
                  require_coord: bool = True, require_depth: bool = True, override_color=None, cam_no=None, iter=None,
                  train_coarse=False, num_down_emb_c=5, num_down_emb_f=5, disable_filter3D=True):
    return _render_impl(viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, require_coord,
                        require_depth, override_color, cam_no, iter, num_down_emb_c, num_down_emb_f, disable_filter3D,
                        +1)


def render_without_tongue(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, kernel_size, scaling_modifier=1.0,
                          require_coord: bool = True, require_depth: bool = True, override_color=None, cam_no=None,
                          iter=None, train_coarse=False, num_down_emb_c=5, num_down_emb_f=5, disable_filter3D=True):
    return _render_impl(viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, require_coord,
                        require_depth, override_color, cam_no, iter, num_down_emb_c, num_down_emb_f, disable_filter3D,
                        -1)


def integrate(points3D, viewpoint_camera, pc, pipe, bg_color: torch.Tensor, kernel_size: float, loaded_iter,
              scaling_modifier=1.0, override_color=None, num_down_emb_c=5, num_down_emb_f=5):
    """gaussian_renderer.integrate (gaussian_renderer/__init__.py:551-661): deform the Gaussians at the camera's time
    (cam_no = None), apply the 3D-filter activations, and integrate them at `points3D` (mesh_extract_tetrahedra.py:90-124).
    Inference only: nothing here is differentiated (the reference's rasterizer.integrate has no backward either)."""
    with torch.no_grad():
        raster_settings = _settings(viewpoint_camera, pc, pipe, bg_color, kernel_size, scaling_modifier, True, True)
        rasterizer = GaussianRasterizer(raster_settings=raster_settings)
        means3D = pc.get_xyz
        screenspace_points = torch.zeros_like(means3D)
        (means3D_final, scales_deformed, rotations_deformed, opacity_deformed, shs_final, extras) = pc._deformation(
            means3D, pc._scaling, pc._rotation, pc._opacity, float(viewpoint_camera.time), None, pc, None, pc.get_features,
            iter=loaded_iter, num_down_emb_c=num_down_emb_c, num_down_emb_f=num_down_emb_f)
        scales_final = rotations_final = cov3D_precomp = None
        if pipe.compute_cov3D_python:
            # the reference leaves opacity_final undefined on this branch (:606-607 vs :649); the filtered opacity is used
            cov3D_precomp = pc.get_covariance(scaling_modifier)
            _, opacity_final = pc.apply_scaling_n_opacity_with_3D_filter(opacity=opacity_deformed, scales=scales_deformed)
        else:
            scales_final, opacity_final = pc.apply_scaling_n_opacity_with_3D_filter(opacity=opacity_deformed,
                                                                                    scales=scales_deformed)
            rotations_final = pc.rotation_activation(rotations_deformed)
        colors_precomp = None
        shs = shs_final
        if override_color is not None:
            colors_precomp, shs = override_color, None
        elif pipe.convert_SHs_python:
            eval_sh = _eval_sh()
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.to(means3D.device).repeat(pc.get_features.shape[0], 1)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp / dir_pp.norm(dim=1, keepdim=True)) + 0.5, 0.0)
            shs = None   # the reference passes both here and its rasterizer raises (:651-652)
        rendered_image, alpha_integrated, color_integrated, point_coordinate, point_sdf, radii = rasterizer.integrate(
            points3D=points3D, means3D=means3D_final, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
            opacities=opacity_final, scales=scales_final, rotations=rotations_final, cov3D_precomp=cov3D_precomp,
            view2gaussian_precomp=None)
    return {"render": rendered_image,
            "alpha_integrated": alpha_integrated,
            "color_integrated": color_integrated,
            "point_coordinate": point_coordinate,
            "point_sdf": point_sdf,
            "visibility_filter": radii > 0,
            "radii": radii}
