"""How many Gaussians of a bench workload receive an exactly-zero upstream gradient in one training step?

A Gaussian that no pixel blends (outside the frustum, radius 0, or behind the last contributor of every tile it touches)
gets all-zero rows in every gradient render()'s backward hands to the deformation backward, and the deformation
backward's work for such a row (data gradient, its share of every weight gradient) is exactly zero.  This prints the
fractions for a few frames: the number that says whether compacting the deformation backward over the active rows pays.

    python tools/zero_grad_fraction.py [--workload C3] [--frames 4]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--frames", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    wl, model, cams, grads = bench.build(a.workload, dev)
    step = bench.make_step(model, cams, grads, wl, dev)
    P = wl["P"]
    for item in range(a.frames):
        it = item * 37 % (wl["cams"] * wl["frames"])
        # make_step clears the gradients at its end: repeat its body here without the clearing
        from ed3dgs_amd import dist as D
        from ed3dgs_amd.model import PIPE
        from gaussian_renderer import render
        ci, fi = D.item_of(it, wl["cams"], wl["frames"])
        cam = cams[ci].with_time(fi / wl["frames"])
        pkg = render(cam, model, PIPE, torch.ones(3, device=dev), kernel_size=0.0, require_coord=False, require_depth=True,
                     cam_no=None, iter=20000, num_down_emb_c=30, num_down_emb_f=30, disable_filter3D=True)
        outs = [pkg["render"], pkg["expected_depth"], pkg["median_depth"], pkg["normal"]]
        torch.autograd.backward(outs, [grads["color"], grads["depth"], grads["mdepth"], grads["normal"]])
        z = lambda t: (t.reshape(P, -1) == 0).all(dim=1)
        zero_all = z(model._xyz.grad) & z(model._scaling.grad) & z(model._rotation.grad) & z(model._opacity.grad) \
            & z(model._features_dc.grad) & z(model._features_rest.grad)
        vis = pkg["visibility_filter"]
        zero_emb = z(model._embedding.grad)
        print(f"item {it}: visible {int(vis.sum())}/{P} = {float(vis.float().mean()):.4f}; all-zero upstream rows "
              f"{int(zero_all.sum())} = {float(zero_all.float().mean()):.4f}; of the visible ones "
              f"{float((zero_all & vis).sum()) / max(int(vis.sum()), 1):.4f}; zero g_embedding rows {float(zero_emb.float().mean()):.4f}",
              flush=True)
        # how clustered are the active rows? strips of 32 consecutive Gaussians that are entirely inactive
        strips = zero_all[: P // 32 * 32].reshape(-1, 32).all(dim=1)
        print(f"         strips of 32 entirely inactive: {float(strips.float().mean()):.4f}", flush=True)
        for p in model.parameters():
            p.grad = None


if __name__ == "__main__":
    main()
