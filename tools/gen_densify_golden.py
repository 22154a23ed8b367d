"""tools/gen_densify_golden.py -- tests/golden/densify_reference.npz from the REFERENCE's own GaussianModel.

`ed3dgs_amd.densify_stats` restates what the reference's training loop does with the screen-space gradients of the path:
`add_densification_stats` (scene/gaussian_model.py:516-518, called at train.py:404-407 with the running max of the radii) and
`densify` = `densify_and_clone` + `densify_and_split` (:452-514, with `densification_postfix` / `prune_points` and their optimizer
surgery).  Until round 4 those restatements were pinned by hand-written formulas only.  This script (container only: needs
/root/reference) runs the reference's methods themselves on a small synthetic model on the CPU and writes inputs + the
reference's outputs; tests/test_densify_golden_cpu.py compares `DensificationStats.add` / `densify_tensors` with them bit for bit.

Harness-side shims, no edits to the reference (the loaders are tools/gen_raster_golden.py's): the module's `torch` is a proxy
whose factory calls drop a hard-coded device="cuda" -- and whose `normal(mean=, std=)` draws the split's samples from a CPU
generator seeded with the iteration number, which is densify_tensors' documented convention (every rank must draw the same
numbers; the reference draws from the device's global RNG, single-GPU).  The model object is made with `__new__` +
`setup_functions()` and given the seven parameter groups `cat_tensors_to_optimizer` / `_prune_optimizer` walk (an Adam optimizer
with state, so the exp_avg surgery runs as well).  Only data is written -- no reference text."""
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_raster_golden as G  # noqa: E402  (also puts ROOT and e-d3dgs_amd on sys.path)

OUT = os.path.join(ROOT, "tests", "golden")
ITERATION = 7300
P, VIEWS = 1200, 6
MAX_GRAD, EXTENT, PERCENT_DENSE = 2.2e-4, 6.0, 0.01


def model_tensors():
    g = torch.Generator().manual_seed(5)
    return dict(xyz=torch.randn(P, 3, generator=g), features_dc=torch.randn(P, 1, 3, generator=g),
                features_rest=torch.randn(P, 15, 3, generator=g), opacity=torch.randn(P, 1, generator=g) * 2,
                scaling=torch.randn(P, 3, generator=g) * 0.7 - 3.0, rotation=torch.randn(P, 4, generator=g),
                embedding=torch.randn(P, 32, generator=g), tongue_class=(torch.rand(P, 1, generator=g) > 0.5).float())


def view(i):
    g = torch.Generator().manual_seed(300 + i)
    grad = torch.randn(P, 3, generator=g) * 2e-4
    grad[:, 2] = grad[:, :2].abs().sum(1)
    radii = (torch.rand(P, generator=g) * 40 - 8).clamp(min=0).floor().int()
    radii[:64] = 0                                      # rows no view ever sees: denom 0, accum / denom = NaN -> 0 (densify :510-511)
    return grad, radii > 0, radii


def main():
    G.load_reference_utils()
    gm = G.load_reference_gaussian_model()
    proxy = G._TorchCpu()

    def seeded_normal(mean=None, std=None, **kw):      # densify_and_split: torch.normal(mean=means, std=stds)
        gen = torch.Generator().manual_seed(ITERATION)
        return torch.randn(std.shape, generator=gen, dtype=torch.float32) * std + mean
    proxy.normal = seeded_normal
    gm.torch = proxy
    sys.modules["utils.general_utils"].torch = proxy    # build_rotation's device="cuda" (densify_and_split calls it)

    t = model_tensors()
    m = gm.GaussianModel.__new__(gm.GaussianModel)
    m.setup_functions()
    m.percent_dense = PERCENT_DENSE
    names = dict(xyz="xyz", features_dc="f_dc", features_rest="f_rest", opacity="opacity", scaling="scaling", rotation="rotation",
                 embedding="embedding")
    params = {k: nn.Parameter(t[k].clone().requires_grad_(True)) for k in names}
    m._xyz, m._features_dc, m._features_rest = params["xyz"], params["features_dc"], params["features_rest"]
    m._opacity, m._scaling, m._rotation, m._embedding = params["opacity"], params["scaling"], params["rotation"], params["embedding"]
    m.tongue_class = t["tongue_class"].clone()
    m.optimizer = torch.optim.Adam([{"params": [params[k]], "lr": 1e-3, "name": names[k]} for k in names], lr=0.0, eps=1e-15)
    for p in params.values():                           # one step so that every group has exp_avg / exp_avg_sq to be cut and extended
        p.grad = torch.zeros_like(p)
    m.optimizer.step()
    m.xyz_gradient_accum = torch.zeros((P, 1))
    m.denom = torch.zeros((P, 1))
    m.max_radii2D = torch.zeros((P,))
    # train.py:404-407, VIEWS iterations of batch size 1
    for i in range(VIEWS):
        grad, vis, radii = view(i)
        m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis].float())
        m.add_densification_stats(grad, vis)
    out = {"in_" + k: v.numpy() for k, v in t.items()}
    out.update(accum=m.xyz_gradient_accum.numpy().copy(), denom=m.denom.numpy().copy(), max_radii2D=m.max_radii2D.numpy().copy(),
               params=np.array([P, VIEWS, ITERATION], dtype=np.int64), thresholds=np.array([MAX_GRAD, EXTENT, PERCENT_DENSE], dtype=np.float64))
    for i in range(VIEWS):
        grad, vis, radii = view(i)
        out["view%d_grad" % i], out["view%d_radii" % i] = grad.numpy(), radii.numpy()
    m.densify(MAX_GRAD, 0.005, EXTENT, None)            # train.py:414 (min_opacity / size threshold are unused by densify itself)
    got = dict(xyz=m._xyz, features_dc=m._features_dc, features_rest=m._features_rest, opacity=m._opacity, scaling=m._scaling,
               rotation=m._rotation, embedding=m._embedding, tongue_class=m.tongue_class)
    for k, v in got.items():
        out["out_" + k] = v.detach().numpy()
    n0, n1 = P, got["xyz"].shape[0]
    assert n1 != n0 and got["tongue_class"].shape[0] == n1
    for g_ in m.optimizer.param_groups:                 # the optimizer surgery ran: state rows follow the parameter rows
        st = m.optimizer.state[g_["params"][0]]
        assert st["exp_avg"].shape[0] == n1
    np.savez_compressed(os.path.join(OUT, "densify_reference.npz"), **out)
    print("wrote densify_reference.npz: %d -> %d Gaussians" % (n0, n1))


def ply_golden(gm):
    """tests/golden/ply_reference.npz: the reference's own `save_ply` (scene/gaussian_model.py:261-283) run on a 37-Gaussian model with
    stand-ins for plyfile's two classes that CAPTURE the structured array it hands them (plyfile itself is not installed; for an
    all-`f4` vertex element `binary_little_endian` plyfile writes exactly that array's bytes behind the header): attribute names in
    `construct_list_of_attributes` order (:231-248) + the vertex bytes.  ed3dgs_amd.model.save_ply must write the same names and bytes."""
    import tempfile
    from ed3dgs_amd import synthetic as S
    captured = {}

    class _Element:
        @staticmethod
        def describe(elements, name):
            captured["elements"], captured["name"] = elements, name
            return "element"

    class _Data:
        def __init__(self, els):
            pass

        def write(self, path):
            captured["path"] = path
    gm.PlyElement, gm.PlyData = _Element, _Data
    Pn = 37
    sc = S.make_scene(Pn, seed=9)
    g = torch.Generator().manual_seed(10)
    m = gm.GaussianModel.__new__(gm.GaussianModel)
    m._xyz, m._features_dc, m._features_rest = sc.xyz, sc.f_dc, sc.f_rest
    m._opacity, m._scaling, m._rotation, m._embedding = sc.opacity, sc.log_scale, sc.rot, sc.embedding
    m.tongue_class = (torch.rand(Pn, 1, generator=g) > 0.5).float()
    m.filter_3D = torch.rand(Pn, 1, generator=g) * 0.01
    with tempfile.TemporaryDirectory() as td:
        m.save_ply(os.path.join(td, "point_cloud", "iteration_7", "point_cloud.ply"))
    el = captured["elements"]
    assert captured["name"] == "vertex" and el.shape == (Pn,)
    names = list(el.dtype.names)
    assert names == m.construct_list_of_attributes() and all(el.dtype[n] == np.dtype("f4") for n in names)
    np.savez_compressed(os.path.join(OUT, "ply_reference.npz"), names=np.array(names), vertex_bytes=np.frombuffer(el.tobytes(), dtype=np.uint8),
                        xyz=sc.xyz.numpy(), f_dc=sc.f_dc.numpy(), f_rest=sc.f_rest.numpy(), opacity=sc.opacity.numpy(),
                        log_scale=sc.log_scale.numpy(), rot=sc.rot.numpy(), embedding=sc.embedding.numpy(),
                        tongue_class=m.tongue_class.numpy(), filter_3D=m.filter_3D.numpy())
    print("wrote ply_reference.npz: %d properties, %d vertex bytes" % (len(names), el.nbytes))


if __name__ == "__main__":
    main()
    ply_golden(sys.modules["scene.gaussian_model"])
