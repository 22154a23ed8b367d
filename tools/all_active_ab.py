"""Cost of the active-row walk when EVERY row is active: a loss term on sh_coefs_final gives every Gaussian a non-zero upstream
gradient; the deformation backward then gathers all P rows through the list.  Compare with ED3DGS_DEFORM_DENSE_BWD=1.
usage: python tools/all_active_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def run(n=30):
    from ed3dgs_amd import dist as D
    from ed3dgs_amd.model import PIPE
    from gaussian_renderer import render
    dev = torch.device("cuda:0")
    wl, model, cams, grads = bench.build("C3", dev)
    ups = [grads["color"], grads["depth"], grads["mdepth"], grads["normal"]]
    w = torch.full((wl["P"], 16, 3), 1e-9, device=dev)
    bg = torch.ones(3, device=dev)

    def step(k):
        ci, fi = D.item_of(k, wl["cams"], wl["frames"])
        pkg = render(cams[ci].with_time(fi / wl["frames"]), model, PIPE, bg, kernel_size=0.0, require_coord=False, require_depth=True,
                     cam_no=None, iter=20000, num_down_emb_c=30, num_down_emb_f=30, disable_filter3D=True)
        outs = [pkg["render"], pkg["expected_depth"], pkg["median_depth"], pkg["normal"], pkg["sh_coefs_final"]]
        torch.autograd.backward(outs, ups + [w])
        act = int((model._embedding.grad.abs().amax(dim=1) > 0).sum())
        for p in model.parameters():
            p.grad = None
        return act
    for k in range(5):
        act = step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        step(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, act


if __name__ == "__main__":
    ms, act = run()
    print("DENSE_BWD=%s: %.3f ms/step, active rows %d" % (os.environ.get("ED3DGS_DEFORM_DENSE_BWD", "0"), ms, act))
