"""tools/gen_raster_golden.py -- generates tests/golden/raster_k1_*.npz from the REFERENCE's own importable Python.

The reference holds no fixtures for its CUDA rasterizer and the rasterizer cannot run here, so the
raster oracle (oracle/raster_ref.c) is "parity unpinned by the reference".  Three of K1's stages, though, exist a second
time in the reference as plain torch code -- the `convert_SHs_python` / `compute_cov3D_python` paths of render()
(gaussian_renderer/__init__.py:68-72,85-95) and the camera matrices every caller builds:

  * utils/sh_utils.py:57          eval_sh                      -> SH -> RGB at degrees 0..3 (CR/forward.cu:23-74)
  * utils/general_utils.py:78-112 strip_symmetric, build_rotation, build_scaling_rotation, composed as
    scene/gaussian_model.py:31-35 does                         -> cov3D (CR/forward.cu:270-304)
  * utils/graphics_utils.py:106-141 getWorld2View2, getProjectionMatrix, composed as scene/cameras.py:84-92 does
                                                               -> viewmatrix, projmatrix, campos
  * scene/gaussian_model.py:594-603, :538-592 apply_scaling_n_opacity_with_3D_filter, compute_3D_filter
                                                               -> the 3D-filter activations (a7) and the filter (f3)

This script imports those (container only: needs /root/reference) and writes inputs + the reference's outputs as .npz.
Harness-side shims, no edits to the reference: the modules get a `torch` proxy whose zeros()/ones()/tensor() drop the
hard-coded device="cuda" (general_utils.py:85,101); gaussian_model.py is loaded by path behind import-only stubs for
`plyfile`, `simple_knn._C`, `tkinter` and a namespace stub for the `scene` package (its __init__ pulls the dataset
readers).  Only data is written -- no reference text.
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "e-d3dgs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


class _TorchCpu:
    """`torch` as the reference's modules see it: factory calls lose a hard-coded device (this host has no GPU)."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def _strip(fn):
        def call(*a, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k.pop("device")
            return fn(*a, **k)
        return call


for _n in ("zeros", "ones", "tensor", "empty", "zeros_like"):
    setattr(_TorchCpu, _n, staticmethod(_TorchCpu._strip(getattr(torch, _n))))


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    if hasattr(mod, "torch"):
        mod.torch = _TorchCpu()
    return mod


def load_reference_utils():
    pkg = types.ModuleType("utils")
    pkg.__path__ = [os.path.join(REF, "utils")]
    sys.modules["utils"] = pkg
    sh = _load("utils.sh_utils", "utils/sh_utils.py")
    gen = _load("utils.general_utils", "utils/general_utils.py")
    gfx = _load("utils.graphics_utils", "utils/graphics_utils.py")
    return sh, gen, gfx


def load_reference_gaussian_model():
    """scene/gaussian_model.py by path; returns the module or raises the ordinary ImportError it ran into."""
    for name, attrs in (("plyfile", ("PlyData", "PlyElement")), ("simple_knn", ()), ("simple_knn._C", ("distCUDA2",)),
                        ("tkinter", ("W",))):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
    torch.Tensor.cuda = lambda self, *a, **k: self
    scene = types.ModuleType("scene")
    scene.__path__ = [os.path.join(REF, "scene")]      # namespace stub: scene/__init__.py (dataset readers) is not run
    sys.modules["scene"] = scene
    _load("utils.system_utils", "utils/system_utils.py")
    _load("scene.deformation", "scene/deformation.py")
    return _load("scene.gaussian_model", "scene/gaussian_model.py")


def main():
    from ed3dgs_amd import synthetic as S
    sh_utils, gen, gfx = load_reference_utils()
    os.makedirs(OUT, exist_ok=True)

    # ---- SH -> RGB, degrees 0..3, M = 16 rows always present (render(): gaussian_renderer/__init__.py:88-92)
    P = 512
    sc = S.make_scene(P, seed=41)
    g = torch.Generator().manual_seed(42)
    shs = torch.cat((sc.f_dc, sc.f_rest * 4.0), 1).contiguous()       # strong higher bands: some colours clamp at 0
    campos = S.make_cameras(1, 400, 400, seed=1)[0].camera_center.clone()   # the C1 camera: every Gaussian of the cube is visible
    shs_view = shs.transpose(1, 2).view(-1, 3, 16)
    dir_pp = sc.xyz - campos.repeat(P, 1)
    dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    out = dict(means3D=sc.xyz.numpy(), campos=campos.numpy(), shs=shs.numpy())
    for deg in range(4):
        raw = sh_utils.eval_sh(deg, shs_view, dirn) + 0.5
        out["rgb_deg%d" % deg] = torch.clamp_min(raw, 0.0).numpy()
        out["clamped_deg%d" % deg] = (raw < 0).numpy()
    assert out["clamped_deg3"].any() and not out["clamped_deg3"].all()
    np.savez_compressed(os.path.join(OUT, "raster_k1_sh.npz"), **out)

    # ---- cov3D = strip_symmetric(L L^T), L = R(q / |q|) diag(mod s)  (scene/gaussian_model.py:31-35)
    a = S.activated(sc)
    out = dict(scales=a["scales"].numpy(), rotations_raw=sc.rot.numpy(), rotations=a["rotations"].numpy())
    for mod in (1.0, 0.7):
        L = gen.build_scaling_rotation(mod * a["scales"], sc.rot)      # build_rotation normalises the quaternion itself
        out["cov3D_mod%02d" % round(mod * 10)] = gen.strip_symmetric(L @ L.transpose(1, 2)).numpy()
    np.savez_compressed(os.path.join(OUT, "raster_k1_cov3d.npz"), **out)

    # ---- camera matrices of synthetic.make_cameras' poses, composed as scene/cameras.py:84-92
    out = {}
    for tag, (n, W, H) in (("c1", (1, 400, 400)), ("c3", (8, 1920, 1080)), ("c4", (16, 1100, 1604))):
        cams = S.make_cameras(n, W, H, seed=1)
        wvt, proj, full, center, Rs, Ts = [], [], [], [], [], []
        for c in cams:
            w = torch.tensor(gfx.getWorld2View2(c.R, c.T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
            pm = gfx.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=c.FoVx, fovY=c.FoVy).transpose(0, 1)
            f = (w.unsqueeze(0).bmm(pm.unsqueeze(0))).squeeze(0)
            wvt.append(w.numpy()); proj.append(pm.numpy()); full.append(f.numpy()); center.append(w.inverse()[3, :3].numpy())
            Rs.append(c.R); Ts.append(c.T)
        out.update({tag + "_R": np.stack(Rs), tag + "_T": np.stack(Ts), tag + "_fov": np.array([cams[0].FoVx, cams[0].FoVy]),
                    tag + "_size": np.array([W, H]), tag + "_world_view_transform": np.stack(wvt),
                    tag + "_projection_matrix": np.stack(proj), tag + "_full_proj_transform": np.stack(full),
                    tag + "_camera_center": np.stack(center)})
    np.savez_compressed(os.path.join(OUT, "raster_k1_cameras.npz"), **out)

    # ---- optional: the 3D-filter activations and compute_3D_filter from scene/gaussian_model.py
    try:
        gm = load_reference_gaussian_model()
    except Exception as e:   # an ordinary ImportError of a missing dependency: recorded, nothing else to do
        print("scene/gaussian_model.py not importable here: %r -- a7 / f3 fixtures skipped" % (e,))
        return
    gm.torch = _TorchCpu()
    model = gm.GaussianModel.__new__(gm.GaussianModel)
    model.setup_functions()
    P = 384
    sc = S.make_scene(P, seed=43)
    cams = S.make_cameras(5, 320, 200, seed=44)
    for c in cams:
        c.original_image = 0          # not None: compute_3D_filter would call load_image()
    model._xyz = sc.xyz
    model.compute_3D_filter(cams)
    filt = model.filter_3D.clone()
    scales_f, opac_f = model.apply_scaling_n_opacity_with_3D_filter(opacity=sc.opacity, scales=sc.log_scale)
    np.savez_compressed(os.path.join(OUT, "raster_filter3d.npz"), xyz=sc.xyz.numpy(), log_scale=sc.log_scale.numpy(),
                        opacity_logit=sc.opacity.numpy(), filter_3D=filt.numpy(), scales_filtered=scales_f.numpy(),
                        opacity_filtered=opac_f.numpy(), cam_R=np.stack([c.R for c in cams]), cam_T=np.stack([c.T for c in cams]),
                        cam_fov=np.array([cams[0].FoVx, cams[0].FoVy]), cam_size=np.array([320, 200]))
    print("wrote raster_k1_sh / raster_k1_cov3d / raster_k1_cameras / raster_filter3d .npz to", OUT)


if __name__ == "__main__":
    main()
