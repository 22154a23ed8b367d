"""Synthetic scenes / cameras / upstream gradients for tests and bench.

Everything is generated on the CPU with fixed seeds (scene S=0, cameras S=1, MLP S=2, upstream grads S=3) and
moved to the requested device afterwards, so the GPU path and the CPU oracle see bit-identical inputs.
Camera matrices are built the way the reference builds them (utils/graphics_utils.py:106-141 getWorld2View2 /
getProjectionMatrix, scene/cameras.py:84-92): row-major tensors that hold the *transposed* 4x4.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch


def make_scene(P, seed=0, device="cpu", sh_degree=3, embedding_dim=32):
    """Raw (pre-activation) Gaussian parameters, as GaussianModel stores them (scene/gaussian_model.py:48-71)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(P, 3, generator=g) * 2 - 1
    mean_ls = math.log(0.6 * max(P, 1) ** (-1.0 / 3.0))
    log_scale = torch.randn(P, 3, generator=g) * 0.4 + mean_ls
    rot = torch.randn(P, 4, generator=g)
    opacity = torch.randn(P, 1, generator=g) * 1.5
    f_dc = torch.randn(P, 1, 3, generator=g)
    f_rest = torch.randn(P, (sh_degree + 1) ** 2 - 1, 3, generator=g) * 0.1
    emb = torch.randn(P, embedding_dim, generator=g) * 0.1
    s = SimpleNamespace(
        xyz=xyz, log_scale=log_scale, rot=rot, opacity=opacity, f_dc=f_dc, f_rest=f_rest, embedding=emb,
        tongue_class=torch.zeros(P, 1), filter_3D=torch.zeros(P, 1), sh_degree=sh_degree)
    for k, v in list(vars(s).items()):
        if torch.is_tensor(v):
            setattr(s, k, v.to(device))
    return s


def _look_at_w2c(eye, target=np.zeros(3), up=np.array([0.0, -1.0, 0.0])):
    """World-to-camera with +z forward, +x right, +y down (COLMAP / 3DGS convention). Returns R (c2w), T (w2c)."""
    f = target - eye
    f = f / np.linalg.norm(f)
    r = np.cross(f, -up)
    r = r / np.linalg.norm(r)
    d = np.cross(f, r)
    R_w2c = np.stack([r, d, f], axis=0)
    T = -R_w2c @ eye
    return R_w2c.T, T  # the reference's Camera takes R = c2w rotation (transposed inside getWorld2View2)


def _world2view2(R, t):
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return np.float32(Rt)


def _projection(znear, zfar, fovX, fovY):
    tanY, tanX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tanY * znear, tanX * znear
    bottom, left = -top, -right
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = (zfar + znear) / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


class SynthCamera:
    """The subset of scene/cameras.py:Camera that render() reads."""

    def __init__(self, R, T, FoVx, FoVy, width, height, time=0.0, device="cpu"):
        self.R, self.T = np.asarray(R, np.float64), np.asarray(T, np.float64)   # as scene/cameras.py:Camera keeps them
        self.FoVx, self.FoVy = FoVx, FoVy
        self.image_width, self.image_height = int(width), int(height)
        self.time = float(time)
        self.znear, self.zfar = 0.01, 100.0
        self.world_view_transform = torch.tensor(_world2view2(R, T)).transpose(0, 1).contiguous()
        self.projection_matrix = _projection(self.znear, self.zfar, FoVx, FoVy).transpose(0, 1).contiguous()
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0))).squeeze(0)
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()
        self.to(device)

    def to(self, device):
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
            setattr(self, k, getattr(self, k).to(device))
        return self

    def with_time(self, t):
        c = SynthCamera.__new__(SynthCamera)
        c.__dict__.update(self.__dict__)
        c.time = float(t)
        return c


def make_cameras(n, W, H, seed=1, device="cpu", radius=4.0):
    """n cameras on a circle of radius 4 in the xz-plane, height U(-0.5,0.5), looking at the origin, fx=fy=1.2*W."""
    rng = np.random.RandomState(seed)
    fx = 1.2 * W
    FoVx = 2 * math.atan(W / (2 * fx))
    FoVy = 2 * math.atan(H / (2 * fx))
    cams = []
    for i in range(n):
        ang = 2 * math.pi * i / max(n, 1) + 0.3
        eye = np.array([radius * math.cos(ang), rng.uniform(-0.5, 0.5), radius * math.sin(ang)])
        R, T = _look_at_w2c(eye)
        cams.append(SynthCamera(R, T, FoVx, FoVy, W, H, 0.0, device))
    return cams


def make_upstream_grads(H, W, seed=3, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    HW = H * W
    d = dict(
        color=torch.randn(3, H, W, generator=g) / (3 * HW),
        depth=torch.randn(1, H, W, generator=g) / HW,
        mdepth=torch.randn(1, H, W, generator=g) / HW,
        normal=torch.randn(3, H, W, generator=g) / HW,
        coord=torch.randn(3, H, W, generator=g) / HW,
        mcoord=torch.randn(3, H, W, generator=g) / HW,
        alpha=torch.randn(1, H, W, generator=g) / HW,
    )
    return {k: v.to(device) for k, v in d.items()}


def activated(scene):
    """The activations render() applies when disable_filter3D=True (gaussian_renderer/__init__.py:77-81)."""
    return dict(
        scales=torch.exp(scene.log_scale),
        rotations=torch.nn.functional.normalize(scene.rot),
        opacities=torch.sigmoid(scene.opacity),
        shs=torch.cat((scene.f_dc, scene.f_rest), dim=1).contiguous(),
    )
