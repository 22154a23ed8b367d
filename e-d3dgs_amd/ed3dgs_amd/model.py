"""SynthGaussianModel -- the slice of scene/gaussian_model.py:GaussianModel that render() touches, built from a
synthetic scene (ed3dgs_amd.synthetic).  Attribute names, activation callables and the 3D-filter formula follow
scene/gaussian_model.py:37-45, 112-141, 594-603; training bookkeeping (optimizer, densification, PLY I/O) is out of
this round's hot-path scope."""
from types import SimpleNamespace

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


PIPE = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
