"""Drives a `gaussian_renderer` module's render glue (render, render_tongue, render_without_tongue) with stand-ins on every side --
a fake camera, a fake model whose tensors carry their own constants and whose activations are marked multiplications, a RECORDING
deformation network, a RECORDING rasterizer -- and returns what the glue did: the raster settings it built, what it handed to the
deformation network, what it handed to the rasterizer (after which activation), and its result dictionary.  Run twice:
tools/gen_render_golden.py on the REFERENCE's gaussian_renderer/__init__.py (container only), tests/test_render_glue_cpu.py on this
repo's.  CPU only, no native code."""
import types
from typing import NamedTuple

import torch

P, H, W = 5, 4, 6


def describe(a):
    if a is None:
        return None
    if torch.is_tensor(a):
        if a.dtype == torch.bool:
            return ["bool_tensor", a.reshape(-1).tolist()]
        if a.numel() == 0:
            return ["tensor", "empty", list(a.shape)]
        v = a.detach().reshape(-1).double()
        return ["tensor", round(float(v[0]), 4), list(a.shape)] if bool((v == v[0]).all()) else ["tensor_values", [round(float(x), 4) for x in v[:12]], list(a.shape)]
    if isinstance(a, bool):
        return ["bool", a]
    if isinstance(a, (int, float)):
        # as the native side receives it: a C float (the reference wraps tan(fov / 2) in a float32 0-d tensor, this repo hands the
        # Python double to a ctypes c_float -- the same rounding)
        return ["number", round(float(torch.tensor(float(a), dtype=torch.float32)), 6)]
    if isinstance(a, str):
        return ["str", a]
    return [type(a).__name__]


def number_of(a):
    """a raster setting as the native side sees it: the reference wraps scalars in 0-d tensors, this repo passes plain numbers"""
    if torch.is_tensor(a) and a.dim() == 0:
        return ["number", round(float(a.to(torch.float32)), 6)]
    return describe(a)


class Settings(NamedTuple):
    image_height: object
    image_width: object
    tanfovx: object
    tanfovy: object
    kernel_size: object
    bg: object
    scale_modifier: object
    viewmatrix: object
    projmatrix: object
    sh_degree: object
    campos: object
    prefiltered: object
    require_depth: object
    require_coord: object
    debug: object


def make_rasterizer_package(log):
    class GaussianRasterizer:
        def __init__(self, raster_settings):
            self.raster_settings = raster_settings
            log["settings"] = {k: number_of(getattr(raster_settings, k)) for k in Settings._fields}

        def __call__(self, **kw):
            log["rasterizer_kwargs"] = {k: describe(v) for k, v in kw.items()}
            n = kw["means3D"].shape[0]
            radii = torch.tensor([3, 0, 5, 0, 9][:n], dtype=torch.int32)
            planes = [torch.full((3, H, W), 101.0), radii, torch.full((3, H, W), 102.0), torch.full((3, H, W), 103.0), torch.full((1, H, W), 104.0),
                      torch.full((1, H, W), 105.0), torch.full((1, H, W), 106.0), torch.full((1, H, W), 107.0), torch.full((3, H, W), 108.0)]
            return tuple(planes)   # color, radii, coord, mcoord, depth, mdepth, alpha, tongue, normal

        def integrate(self, **kw):
            log["integrate_kwargs"] = {k: describe(v) for k, v in kw.items()}
            radii = torch.tensor([3, 0, 5, 0, 9], dtype=torch.int32)
            return (torch.full((9, H, W), 201.0), torch.full((7,), 202.0), torch.full((7, 3), 203.0), torch.full((7, 2), 204.0),
                    torch.full((7,), 205.0), radii)   # color, alpha_integrated, color_integrated, point_coordinate, point_sdf, radii
    mod = types.ModuleType("diff_gaussian_rasterization")
    mod.GaussianRasterizationSettings, mod.GaussianRasterizer = Settings, GaussianRasterizer
    return mod


class FakeCamera:
    FoVx, FoVy, image_height, image_width, time = 0.7, 0.5, H, W, 0.37
    world_view_transform = torch.full((4, 4), 31.0)
    full_proj_transform = torch.full((4, 4), 32.0)
    camera_center = torch.full((3,), 33.0)


class FakeModel:
    """Activations are MARKED (x -> k x) and deliberately not torch.exp / sigmoid / normalize, so that a glue that fuses the standard
    activations into a native kernel takes its generic branch."""
    active_sh_degree, max_sh_degree = 2, 3

    def __init__(self, log):
        self.get_xyz = torch.full((P, 3), 1.0)
        self._opacity = torch.full((P, 1), 2.0)
        self.get_features = torch.full((P, 16, 3), 3.0)
        self._scaling = torch.full((P, 3), 4.0)
        self._rotation = torch.full((P, 4), 5.0)
        self.tongue_class = torch.tensor([[0.2], [0.7], [1.0], [0.0], [0.6]])
        self.filter_3D = torch.full((P, 1), 6.0)
        self.scaling_activation = lambda x: x * 2.0
        self.rotation_activation = lambda x: x * 3.0
        self.opacity_activation = lambda x: x * 5.0
        self.log = log
        model = self

        def deformation(*args, **kw):
            pos = []
            for a in args:
                pos.append(["the_model"] if a is model else describe(a))
            # the camera time arrives as a (P, 1) tensor in the reference and as a number here: recorded as its value
            if pos[4] is not None and pos[4][0] in ("tensor", "number"):
                pos[4] = ["time", pos[4][1]]
            log["deformation_args"] = pos
            log["deformation_kwargs"] = {k: describe(v) for k, v in kw.items() if v is not None}
            return (torch.full((P, 3), 21.0), torch.full((P, 3), 22.0), torch.full((P, 4), 23.0), torch.full((P, 1), 24.0),
                    torch.full((P, 16, 3), 25.0), "EXTRAS")
        self._deformation = deformation

    def apply_scaling_n_opacity_with_3D_filter(self, opacity, scales):
        return scales * 7.0, opacity * 11.0

    def get_covariance(self, scaling_modifier=1):
        return torch.full((P, 6), 12.0 * scaling_modifier)


class Pipe:
    debug, compute_cov3D_python, convert_SHs_python = False, False, False


CASES = [("render", dict()), ("render", dict(disable_filter3D=False, scaling_modifier=0.7, require_coord=False)),
         ("render", dict(override_color=torch.full((P, 3), 40.0))), ("render_tongue", dict()), ("render_without_tongue", dict()),
         ("render_tongue", dict(disable_filter3D=False)), ("integrate", dict()), ("integrate", dict(scaling_modifier=0.7))]


def probe(module, install):
    """install(rasterizer_package): makes `module`'s glue build its settings / rasterizer from the recording package."""
    out = []
    for fn, kw in CASES:
        log = {}
        install(make_rasterizer_package(log))
        model = FakeModel(log)
        if fn == "integrate":   # gaussian_renderer.integrate (:551-661): the mesh-extraction probe's glue
            res = module.integrate(torch.full((7, 3), 60.0), FakeCamera(), model, Pipe(), torch.full((3,), 50.0), 0.3, 4321,
                                   num_down_emb_c=7, num_down_emb_f=8, **kw)
        else:
            res = getattr(module, fn)(FakeCamera(), model, Pipe(), torch.full((3,), 50.0), 0.3, cam_no=2, iter=1234, num_down_emb_c=7,
                                      num_down_emb_f=8, **kw)
        log["result"] = {k: describe(v) for k, v in res.items()}
        out.append({"function": fn, "kwargs": sorted(kw), **log})
    return out
