"""SynthGaussianModel -- the slice of scene/gaussian_model.py:GaussianModel that render() touches, built from a
synthetic scene (ed3dgs_amd.synthetic) or from a reference checkpoint (`load_checkpoint`: point_cloud.ply +
deformation.pth, scene/gaussian_model.py:250-347).  Attribute names, activation callables and the 3D-filter formula
follow scene/gaussian_model.py:37-45, 112-141, 538-603; training bookkeeping (optimizer, densification) is out of scope."""
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from scene.deformation import deform_network


def default_hyper(**over):
    """ModelHiddenParams defaults (arguments/__init__.py:85-108) with the nersemble overrides
    (arguments/nersemble/default.py:5-16)."""
    d = dict(net_width=128, defor_depth=1, min_embeddings=30, max_embeddings=150, no_ds=False, no_dr=False,
             no_do=False, no_dc=False, temporal_embedding_dim=256, gaussian_embedding_dim=32,
             use_coarse_temporal_embedding=True, no_c2f_temporal_embedding=False, no_coarse_deform=False,
             no_fine_deform=False, total_num_frames=300, c2f_temporal_iter=10000, deform_from_iter=5000,
             use_anneal=True, zero_temporal=False)
    d.update(over)
    return SimpleNamespace(**d)


class SynthGaussianModel:
    def __init__(self, scene, args=None, deform_seed=2, device="cuda", table_scale=100.0):
        args = args or default_hyper()
        self.args = args
        self.active_sh_degree = scene.sh_degree
        self.max_sh_degree = scene.sh_degree
        p = lambda t: nn.Parameter(t.to(device).contiguous().float())
        self._xyz = p(scene.xyz)
        self._scaling = p(scene.log_scale)
        self._rotation = p(scene.rot)
        self._opacity = p(scene.opacity)
        self._features_dc = p(scene.f_dc)
        self._features_rest = p(scene.f_rest)
        self._embedding = p(scene.embedding)
        self.tongue_class = scene.tongue_class.to(device).contiguous().float()
        self.filter_3D = scene.filter_3D.to(device).contiguous().float()
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(deform_seed)
        self._deformation = deform_network(W=args.net_width, D=args.defor_depth, min_embeddings=args.min_embeddings,
                                           max_embeddings=args.max_embeddings, num_frames=args.total_num_frames,
                                           args=args)
        torch.random.set_rng_state(gen_state)
        with torch.no_grad():
            self._deformation.weight.mul_(table_scale)  # make the synthetic deformation visible
        self._deformation = self._deformation.to(device)
        self.fused_filter3D = True  # apply_scaling_n_opacity_with_3D_filter below is the reference formula -> fusable
        self.scaling_activation = torch.exp
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # properties render() reads
    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_embedding(self):
        return self._embedding

    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        """scene/gaussian_model.py:31-35,143-144: strip_symmetric(L L^T), L = R(q / |q|) diag(modifier * exp(_scaling)) of the
        UNDEFORMED parameters, as six values per Gaussian in the rasterizer's order (xx, xy, xz, yy, yz, zz)."""
        s = scaling_modifier * self.get_scaling
        q = torch.nn.functional.normalize(self._rotation)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), 1).reshape(-1, 3, 3)
        L = R * s[:, None, :]
        c = L @ L.transpose(1, 2)
        return torch.stack((c[:, 0, 0], c[:, 0, 1], c[:, 0, 2], c[:, 1, 1], c[:, 1, 2], c[:, 2, 2]), 1)

    def apply_scaling_n_opacity_with_3D_filter(self, opacity, scales):
        """scene/gaussian_model.py:594-603"""
        opacity = self.opacity_activation(opacity)
        scales = self.scaling_activation(scales)
        scales_square = torch.square(scales)
        det1 = scales_square.prod(dim=1)
        scales_after_square = scales_square + torch.square(self.filter_3D)
        det2 = scales_after_square.prod(dim=1)
        coef = torch.sqrt(det1 / det2)
        return torch.sqrt(scales_after_square), opacity * coef[..., None]

    def parameters(self):
        return [self._xyz, self._scaling, self._rotation, self._opacity, self._features_dc, self._features_rest,
                self._embedding] + list(self._deformation.parameters())

    # ---- the reference's on-disk formats (scene/gaussian_model.py:250-347) ----
    def save_ply(self, path):
        """save_ply (:261-283): one `vertex` element, all-float properties in construct_list_of_attributes order."""
        from . import ply
        d = lambda t: t.detach().cpu().numpy().astype(np.float32)
        xyz = d(self._xyz)
        f_dc = d(self._features_dc.transpose(1, 2).flatten(start_dim=1))
        f_rest = d(self._features_rest.transpose(1, 2).flatten(start_dim=1))
        cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, d(self._opacity), d(self._scaling), d(self._rotation),
                d(self._embedding), d(self.tongue_class), d(self.filter_3D)]
        names = ply.attribute_names(f_dc.shape[1], f_rest.shape[1], self._scaling.shape[1], self._rotation.shape[1],
                                    self._embedding.shape[1])
        flat = np.concatenate(cols, axis=1)
        assert flat.shape[1] == len(names)
        elements = np.empty(xyz.shape[0], dtype=[(n, "f4") for n in names])
        for i, n in enumerate(names):
            elements[n] = flat[:, i]
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        ply.write_vertices(path, elements)

    def load_ply(self, path, device="cuda"):
        """load_ply (:295-347): same field handling (missing tongue_class -> zeros; f_rest_* / scale_* / rot_* /
        embedding_* sorted by their numeric suffix)."""
        from . import ply
        v = ply.read_vertices(path)
        names = v.dtype.names
        col = lambda n: np.asarray(v[n], np.float32)
        by_suffix = lambda pre: sorted([n for n in names if n.startswith(pre)], key=lambda x: int(x.split("_")[-1]))
        xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
        P = xyz.shape[0]
        opac = col("opacity")[:, None]
        tongue = col("tongue_class")[:, None] if "tongue_class" in names else np.zeros_like(opac)
        filt = col("filter_3D")[:, None]
        f_dc = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], axis=1)[:, :, None]      # (P, 3, 1)
        rest = by_suffix("f_rest_")
        if len(rest) != 3 * (self.max_sh_degree + 1) ** 2 - 3:
            raise ValueError("load_ply: %d f_rest_* properties do not match max_sh_degree=%d" % (len(rest), self.max_sh_degree))
        f_rest = np.stack([col(n) for n in rest], axis=1).reshape(P, 3, (self.max_sh_degree + 1) ** 2 - 1)
        scales = np.stack([col(n) for n in by_suffix("scale_")], axis=1)
        rots = np.stack([col(n) for n in by_suffix("rot")], axis=1)
        emb = np.stack([col(n) for n in by_suffix("embedding")], axis=1)
        p = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=device).contiguous().requires_grad_(True))
        self._xyz, self._opacity, self._scaling, self._rotation, self._embedding = p(xyz), p(opac), p(scales), p(rots), p(emb)
        self._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float, device=device).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(f_rest, dtype=torch.float, device=device).transpose(1, 2).contiguous().requires_grad_(True))
        self.filter_3D = torch.tensor(filt, dtype=torch.float, device=device)
        self.tongue_class = torch.tensor(tongue, dtype=torch.float, device=device)
        self.active_sh_degree = self.max_sh_degree

    def save_deformation(self, path):
        """save_deformation (:258-259)"""
        os.makedirs(path, exist_ok=True)
        torch.save(self._deformation.state_dict(), os.path.join(path, "deformation.pth"))

    def load_model(self, path, device="cuda"):
        """load_model (:250-256); the file is read with weights_only=True (tensors only, nothing is executed)."""
        sd = torch.load(os.path.join(path, "deformation.pth"), map_location="cpu", weights_only=True)
        self._deformation.load_state_dict(sd)
        self._deformation = self._deformation.to(device)

    @torch.no_grad()
    def compute_3D_filter(self, cameras):
        """compute_3D_filter (:538-592) on the GPU (csrc/filter3d.hip)."""
        from .filter3d import compute_3D_filter
        self.filter_3D = compute_3D_filter(self.get_xyz, cameras)

    @torch.no_grad()
    def reset_3D_filter(self):
        """reset_3D_filter (:533-536)"""
        self.filter_3D = torch.zeros([self.get_xyz.shape[0], 1], device=self.get_xyz.device)


def read_cfg_args(model_path):
    """`<model_path>/cfg_args` (train.py:483-484 writes `str(Namespace(**vars(args)))`; arguments/__init__.py:174-194 reads
    it back with eval()).  Parsed here with `ast` -- literals only, nothing in the file is executed.  Returns a
    SimpleNamespace; a missing file gives an empty one (as the reference's `Namespace()` default)."""
    import ast
    path = os.path.join(model_path, "cfg_args")
    if not os.path.exists(path):
        return SimpleNamespace()
    tree = ast.parse(open(path).read().strip(), mode="eval")
    call = tree.body
    if not (isinstance(call, ast.Call) and isinstance(call.func, ast.Name) and call.func.id == "Namespace" and not call.args):
        raise ValueError("cfg_args: expected `Namespace(key=value, ...)`")
    out = {}
    for kw in call.keywords:
        if kw.arg is None:
            raise ValueError("cfg_args: **kwargs are not allowed")
        try:
            out[kw.arg] = ast.literal_eval(kw.value)
        except ValueError:
            raise ValueError("cfg_args: value of %r is not a literal" % kw.arg)
    return SimpleNamespace(**out)


def write_cfg_args(model_path, args):
    """train.py:483-484: the run's arguments as `Namespace(...)` text."""
    from argparse import Namespace
    os.makedirs(model_path, exist_ok=True)
    with open(os.path.join(model_path, "cfg_args"), "w") as f:
        f.write(str(Namespace(**vars(args))))


def load_checkpoint(model_path, iteration, args=None, sh_degree=None, device="cuda"):
    """A render-ready model from `<model_path>/point_cloud/iteration_<n>/{point_cloud.ply, deformation.pth}` (the layout
    scene/__init__.py writes: Scene.save -> save_ply + save_deformation).  Hyper-parameters not given in `args` and the
    SH degree come from `<model_path>/cfg_args` when that file exists (read_cfg_args), else the defaults."""
    from .synthetic import make_scene
    cfg = read_cfg_args(model_path)
    if args is None:
        known = vars(default_hyper())
        args = default_hyper(**{k: v for k, v in vars(cfg).items() if k in known})
    if sh_degree is None:
        sh_degree = int(getattr(cfg, "sh_degree", 3))
    d = os.path.join(model_path, "point_cloud", "iteration_%d" % iteration)
    m = SynthGaussianModel(make_scene(1, sh_degree=sh_degree), args=args, device=device, table_scale=1.0)
    m.load_ply(os.path.join(d, "point_cloud.ply"), device=device)
    m.load_model(d, device=device)
    return m


PIPE = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
