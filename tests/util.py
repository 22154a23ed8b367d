"""Shared helpers of the parity tests: run the same synthetic inputs through the CPU oracle and the HIP path."""
import math

import numpy as np
import torch

from ed3dgs_amd import synthetic as S

VARIANTS = {"FFF": (False, False), "FTT": (False, True), "TFT": (True, False), "TTT": (True, True)}


def scene_inputs(P, W, H, scene_seed=0, cam_seed=1, cam_index=0, n_cams=1, kernel_size=0.0, tongue=False, sh_degree=3,
                 scale_modifier=1.0, bg=(1.0, 1.0, 1.0)):
    """sh_degree < 3 keeps the M = 16 coefficient rows, as the reference does while active_sh_degree grows
    (scene/gaussian_model.py:49,146-148, train.py:129-130); the rows above (deg+1)^2 are then present but unused."""
    sc = S.make_scene(P, seed=scene_seed)
    cam = S.make_cameras(n_cams, W, H, seed=cam_seed)[cam_index]
    a = S.activated(sc)
    if tongue:
        g = torch.Generator().manual_seed(7)
        sc.tongue_class = (torch.rand(P, 1, generator=g) > 0.5).float()
    return dict(
        P=P, W=W, H=H, bg=torch.tensor(bg, dtype=torch.float32), means3D=sc.xyz, opacities=a["opacities"], tongue_class=sc.tongue_class,
        scales=a["scales"], rotations=a["rotations"], shs=a["shs"], viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, campos=cam.camera_center, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), kernel_size=kernel_size, scale_modifier=float(scale_modifier), sh_degree=int(sh_degree))


# Exclusion margin of the parity tests (relative distance of a blend decision to its threshold in the ORACLE's forward):
# MARGIN on alpha >= 1/255 (and power > 0), MARGIN x 5 on T (1 - alpha) < 1e-4 and on T > 0.5 -- T is a product of up to
# hundreds of (1 - alpha) factors, so a 2.4e-7 disagreement in single alphas is a larger one in T.
MARGIN = 1e-6
MARGIN_WEIGHTS = (1.0, 5.0, 5.0)


def oracle_forward(inp, variant, with_margin=True, colors_precomp=None, cov3D_precomp=None):
    from oracle import raster_oracle as O
    O.set_margin_weights(*MARGIN_WEIGHTS)
    rc, rd = VARIANTS[variant]
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    use_cov = cov3D_precomp is not None
    return O.forward(n(inp["bg"]), n(inp["means3D"]), n(colors_precomp), n(inp["opacities"]), n(inp["tongue_class"]),
                     None if use_cov else n(inp["scales"]), None if use_cov else n(inp["rotations"]),
                     inp["scale_modifier"], n(cov3D_precomp), n(inp["viewmatrix"]), n(inp["projmatrix"]),
                     inp["tanfovx"], inp["tanfovy"], inp["kernel_size"], inp["H"], inp["W"],
                     None if colors_precomp is not None else n(inp["shs"]), inp["sh_degree"], n(inp["campos"]), rc, rd,
                     with_margin=with_margin)


def oracle_backward(inp, fw, grads, variant, colors_precomp=None, cov3D_precomp=None, reference_q1=True):
    from oracle import raster_oracle as O
    rc, rd = VARIANTS[variant]
    n = lambda t: None if t is None else t.detach().cpu().numpy()
    use_cov = cov3D_precomp is not None
    return O.backward(fw, n(inp["bg"]), n(inp["means3D"]), n(colors_precomp), None if use_cov else n(inp["scales"]),
                      None if use_cov else n(inp["rotations"]), inp["scale_modifier"], n(cov3D_precomp),
                      n(inp["viewmatrix"]), n(inp["projmatrix"]), inp["tanfovx"], inp["tanfovy"], inp["kernel_size"],
                      n(grads["color"]), n(grads["coord"]), n(grads["mcoord"]), n(grads["depth"]), n(grads["mdepth"]),
                      n(grads["alpha"]), n(grads["normal"]), None if colors_precomp is not None else n(inp["shs"]),
                      inp["sh_degree"], n(inp["campos"]), rc, rd, reference_q1=reference_q1)


def hip_settings(inp, variant, device="cuda", debug=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    rc, rd = VARIANTS[variant]
    return GaussianRasterizationSettings(
        image_height=inp["H"], image_width=inp["W"], tanfovx=inp["tanfovx"], tanfovy=inp["tanfovy"],
        kernel_size=inp["kernel_size"], bg=inp["bg"].to(device), scale_modifier=inp["scale_modifier"],
        viewmatrix=inp["viewmatrix"].to(device), projmatrix=inp["projmatrix"].to(device), sh_degree=inp["sh_degree"],
        campos=inp["campos"].to(device), prefiltered=False, require_depth=rd, require_coord=rc, debug=debug)


def hip_forward_raw(inp, variant, device="cuda", colors_precomp=None, cov3D_precomp=None):
    """Calls _C.rasterize_gaussians directly; returns the 13-tuple and a state view."""
    from diff_gaussian_rasterization import _C
    rc, rd = VARIANTS[variant]
    d = lambda t: t.to(device).contiguous()
    e = torch.Tensor([])
    use_cov = cov3D_precomp is not None
    out = _C.rasterize_gaussians(
        d(inp["bg"]), d(inp["means3D"]), e if colors_precomp is None else d(colors_precomp), d(inp["opacities"]),
        d(inp["tongue_class"]), e if use_cov else d(inp["scales"]), e if use_cov else d(inp["rotations"]),
        inp["scale_modifier"], d(cov3D_precomp) if use_cov else e, d(inp["viewmatrix"]), d(inp["projmatrix"]),
        inp["tanfovx"], inp["tanfovy"], inp["kernel_size"], inp["H"], inp["W"],
        e if colors_precomp is not None else d(inp["shs"]), inp["sh_degree"], d(inp["campos"]), False, rc, rd, False)
    sv = _C.state_view(inp["P"], inp["H"], inp["W"], out[0], out[10], out[11], out[12])
    return out, sv


def rel_linf(a, b, mask=None):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    if mask is not None:
        d = d[..., mask] if d.ndim > mask.ndim else d[mask]
    scale = max(np.abs(b).max(), 1e-30)
    return d.max() / scale if d.size else 0.0


GRAD_NAMES = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]


def zero_unused_grads(grads, variant):
    rc, rd = VARIANTS[variant]
    if not rc:
        grads["coord"].zero_(); grads["mcoord"].zero_()
    if not rd:
        grads["depth"].zero_(); grads["mdepth"].zero_()
    if not (rc or rd):
        grads["normal"].zero_()
    return grads


def mask_marginal(grads, fw, margin):
    """Pixels whose blend decision sits within `margin` of a threshold get no upstream gradient on either side; returns
    the masked gradients and the excluded pixel fraction."""
    good = torch.from_numpy((fw["margin"] >= margin).astype(np.float32))
    return {k: v * good for k, v in grads.items()}, float(1.0 - good.mean())


def oracle_state_from_hip(fw, out, sv):
    """The oracle forward's dict with the HIP forward's saved per-pixel state substituted (kernel-level backward checks)."""
    fw_hip = dict(fw)
    fw_hip.update(alpha=out[4].cpu().numpy(), normal=out[6].cpu().numpy(), n_contrib=sv["n_contrib"],
                  accum_coord=sv["accum_coord"], accum_depth=sv["accum_depth"], normal_length=sv["normal_length"])
    return fw_hip


def hip_backward_raw(inp, out, grads, variant, device="cuda", debug=False, colors_precomp=None, cov3D_precomp=None):
    """Calls _C.rasterize_gaussians_backward with the 13-tuple `out` of hip_forward_raw; returns {name: numpy}."""
    from diff_gaussian_rasterization import _C
    rc, rd = VARIANTS[variant]
    d = lambda t: t.to(device).contiguous()
    e = torch.Tensor([])
    use_cov = cov3D_precomp is not None
    res = _C.rasterize_gaussians_backward(
        d(inp["bg"]), d(inp["means3D"]), out[9], e if colors_precomp is None else d(colors_precomp),
        e if use_cov else d(inp["scales"]), e if use_cov else d(inp["rotations"]), inp["scale_modifier"],
        d(cov3D_precomp) if use_cov else e, d(inp["viewmatrix"]), d(inp["projmatrix"]), inp["tanfovx"], inp["tanfovy"],
        inp["kernel_size"], d(grads["color"]), d(grads["coord"]), d(grads["mcoord"]), d(grads["depth"]),
        d(grads["mdepth"]), d(grads["alpha"]), d(grads["normal"]), out[6],
        e if colors_precomp is not None else d(inp["shs"]), inp["sh_degree"], d(inp["campos"]), out[10], out[0], out[11],
        out[12], out[4], rc, rd, debug)
    return {n: t.cpu().numpy() for n, t in zip(GRAD_NAMES, res)}


def grad_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
