"""Drives a `diff_gaussian_rasterization` package's PYTHON surface (GaussianRasterizationSettings, GaussianRasterizer.forward /
.markVisible / .integrate, the autograd Function's forward and backward) against a RECORDING stand-in for its native `_C` module and
returns what crossed the boundary: for every `_C` call the positional arguments as tags (each input tensor is filled with its own
constant, each setting has its own value), the order in which the Function hands `_C`'s results back, and which `_C` gradient lands
in which input's .grad.  Used twice: tools/gen_surface_golden.py runs it on the REFERENCE's package (container only) and writes the
result; tests/test_surface_contract_cpu.py runs it on this repo's package and compares.  No GPU, no native code."""
import types

import torch

P, H, W = 5, 4, 6
IN_TAGS = dict(means3D=1.0, means2D=2.0, sh=3.0, colors_precomp=4.0, opacities=5.0, tongue_class=6.0, scales=7.0, rotations=8.0,
               cov3D_precomp=9.0, points3D=10.0, view2gaussian_precomp=11.0)
IN_SHAPES = dict(means3D=(P, 3), means2D=(P, 3), sh=(P, 16, 3), colors_precomp=(P, 3), opacities=(P, 1), tongue_class=(P, 1),
                 scales=(P, 3), rotations=(P, 4), cov3D_precomp=(P, 6), points3D=(7, 3), view2gaussian_precomp=(P, 10))
FWD_OUT = ["num_rendered", "color", "coord", "mcoord", "alpha", "tongue", "normal", "depth", "mdepth", "radii", "geomBuffer",
           "binningBuffer", "imgBuffer"]      # the tuple `_C.rasterize_gaussians` returns (rasterize_points.cu:37-165)
BWD_OUT = ["grad_means2D", "grad_colors_precomp", "grad_opacities", "grad_means3D", "grad_cov3Ds_precomp", "grad_sh", "grad_scales",
           "grad_rotations"]                  # ... and `_C.rasterize_gaussians_backward` (rasterize_points.cu:167-290)
BWD_SHAPE = dict(grad_means2D="means2D", grad_colors_precomp="colors_precomp", grad_opacities="opacities", grad_means3D="means3D",
                 grad_cov3Ds_precomp="cov3D_precomp", grad_sh="sh", grad_scales="scales", grad_rotations="rotations")


def describe(a):
    if torch.is_tensor(a):
        if a.numel() == 0:
            return ["tensor", "empty", list(a.shape)]
        return ["tensor", round(float(a.reshape(-1)[0]), 3), list(a.shape)]
    if isinstance(a, bool):
        return ["bool", a]
    if isinstance(a, int):
        return ["int", a]
    if isinstance(a, float):
        return ["float", round(a, 6)]
    return [type(a).__name__, repr(a)]


def make_recorder(log, used):
    """A module object with the four entry points of the reference's extension (ext.cpp); every call is logged."""
    rec = types.ModuleType("_C_recorder")

    def rasterize_gaussians(*args):
        log.append(["rasterize_gaussians", [describe(a) for a in args]])
        out = []
        for i, name in enumerate(FWD_OUT):
            if name == "num_rendered":
                out.append(7)
            elif name == "radii":
                out.append(torch.full((P,), 100 + i, dtype=torch.int32))
            elif name.endswith("Buffer"):
                out.append(torch.full((3,), 100.0 + i))
            else:
                c = 3 if name in ("color", "coord", "mcoord", "normal") else 1
                out.append(torch.full((c, H, W), 100.0 + i))
        return tuple(out)

    def rasterize_gaussians_backward(*args):
        log.append(["rasterize_gaussians_backward", [describe(a) for a in args]])
        res = []
        for i, name in enumerate(BWD_OUT):
            src = BWD_SHAPE[name]
            shape = IN_SHAPES[src] if used[src] else (0,)
            res.append(torch.full(shape, 300.0 + i))
        return tuple(res)

    def mark_visible(*args):
        log.append(["mark_visible", [describe(a) for a in args]])
        return torch.ones(P, dtype=torch.bool)

    def integrate_gaussians_to_points(*args):
        log.append(["integrate_gaussians_to_points", [describe(a) for a in args]])
        names = ["num_rendered", "color", "alpha_integrated", "color_integrated", "point_coordinate", "point_sdf", "radii", "geomBuffer",
                 "binningBuffer", "imgBuffer"]
        return tuple(7 if n == "num_rendered" else torch.full((2,), 400.0 + i) for i, n in enumerate(names))

    rec.rasterize_gaussians, rec.rasterize_gaussians_backward = rasterize_gaussians, rasterize_gaussians_backward
    rec.mark_visible, rec.integrate_gaussians_to_points = mark_visible, integrate_gaussians_to_points
    return rec


def settings_of(pkg, debug=False):
    return pkg.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=0.31, tanfovy=0.32, kernel_size=0.33, bg=torch.full((3,), 20.0), scale_modifier=0.34,
        viewmatrix=torch.full((4, 4), 21.0), projmatrix=torch.full((4, 4), 22.0), sh_degree=3, campos=torch.full((3,), 23.0),
        prefiltered=False, require_depth=True, require_coord=True, debug=debug)


def probe(pkg, install):
    """pkg: the loaded package; install(recorder): makes `pkg`'s Function / Module code call the recorder as its `_C`."""
    result = {"settings_fields": list(pkg.GaussianRasterizationSettings._fields)}
    for variant, use in (("sh_scales_rotations", dict(sh=True, colors_precomp=False, scales=True, rotations=True, cov3D_precomp=False)),
                         ("precomputed_colour_and_covariance", dict(sh=False, colors_precomp=True, scales=False, rotations=False, cov3D_precomp=True))):
        used = dict(means3D=True, means2D=True, opacities=True, **use)
        log = []
        install(make_recorder(log, used))
        t = {k: torch.full(IN_SHAPES[k], IN_TAGS[k], requires_grad=(k != "tongue_class")) for k in IN_SHAPES}
        ras = pkg.GaussianRasterizer(raster_settings=settings_of(pkg))
        outs = ras(means3D=t["means3D"], means2D=t["means2D"], opacities=t["opacities"], tongue_class=t["tongue_class"],
                   shs=t["sh"] if use["sh"] else None, colors_precomp=t["colors_precomp"] if use["colors_precomp"] else None,
                   scales=t["scales"] if use["scales"] else None, rotations=t["rotations"] if use["rotations"] else None,
                   cov3D_precomp=t["cov3D_precomp"] if use["cov3D_precomp"] else None)
        returned = [describe(o.detach()) for o in outs]
        # upstream gradients with their own tags for every differentiable output (radii, an int tensor, takes none)
        diff = [(i, o) for i, o in enumerate(outs) if o.dtype.is_floating_point]
        torch.autograd.backward([o for _, o in diff], [torch.full(o.shape, 200.0 + i) for i, o in diff])
        landed = {k: (describe(t[k].grad) if t[k].grad is not None else None) for k in IN_SHAPES if k not in ("points3D", "view2gaussian_precomp")}
        result[variant] = {"calls": log, "returned": returned, "input_grads": landed}
    log = []
    install(make_recorder(log, {}))
    ras = pkg.GaussianRasterizer(raster_settings=settings_of(pkg))
    vis = ras.markVisible(torch.full((P, 3), 30.0))
    result["mark_visible"] = {"calls": log, "returned": describe(vis)}
    # GaussianRasterizer.integrate (:245-312): 23 positional arguments to _C.integrate_gaussians_to_points, ten results back, six returned
    log = []
    install(make_recorder(log, {}))
    ras = pkg.GaussianRasterizer(raster_settings=settings_of(pkg))
    t = {k: torch.full(IN_SHAPES[k], IN_TAGS[k]) for k in IN_SHAPES}
    outs = ras.integrate(points3D=t["points3D"], means3D=t["means3D"], means2D=t["means2D"], opacities=t["opacities"], shs=t["sh"],
                         scales=t["scales"], rotations=t["rotations"])
    result["integrate"] = {"calls": log, "returned": [describe(o) for o in outs]}
    for bad in (dict(), dict(shs=1, colors_precomp=1, scales=1, rotations=1), dict(shs=1), dict(shs=1, scales=1, rotations=1, cov3D_precomp=1)):
        try:
            ras(means3D=None, means2D=None, opacities=None, tongue_class=None, **bad)
            msg = None
        except Exception as ex:   # the reference raises bare Exception with these texts (:213-217)
            msg = str(ex)
        result.setdefault("argument_errors", []).append([sorted(bad), msg])
    return result
