"""Times gaussian_renderer.integrate / distCUDA2 at the C3 size (200k Gaussians, 1080p) with 1M query points.
usage (GPU box): python tools/time_integrate.py [n_points]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ed3dgs_amd import synthetic as S  # noqa: E402
from ed3dgs_amd.model import PIPE, SynthGaussianModel, default_hyper  # noqa: E402
from gaussian_renderer import integrate  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
model = SynthGaussianModel(S.make_scene(200_000, seed=0), args=default_hyper(), device="cuda")
cam = S.make_cameras(8, 1920, 1080, seed=1, device="cuda")[3].with_time(0.3)
g = torch.Generator(device="cuda").manual_seed(0)
xyz = model.get_xyz.detach()
pts = xyz[torch.randint(0, xyz.shape[0], (n,), device="cuda", generator=g)] + 0.02 * torch.randn(n, 3, device="cuda", generator=g)
bg = torch.ones(3, device="cuda")
for _ in range(2):
    r = integrate(pts, cam, model, PIPE, bg, 0.0, 20000, num_down_emb_c=30, num_down_emb_f=30)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    r = integrate(pts, cam, model, PIPE, bg, 0.0, 20000, num_down_emb_c=30, num_down_emb_f=30)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
inside = int((r["point_coordinate"].abs().sum(1) > 0).sum())
print("integrate: %.2f ms per call (%d points, %d inside the image, max points per pixel %d)" %
      (dt * 1e3, n, inside, int(r["render"][8].max())))
