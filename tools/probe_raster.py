"""Timing probe of the rasterizer alone (static scene): python tools/probe_raster.py P W H variant iters"""
import os, sys, math, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import util
from diff_gaussian_rasterization import GaussianRasterizer
from ed3dgs_amd import synthetic as S

P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variant = sys.argv[4] if len(sys.argv) > 4 else "FTT"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
inp = util.scene_inputs(P, W, H)
rs = util.hip_settings(inp, variant)
rast = GaussianRasterizer(rs)
leaf = lambda t: t.cuda().clone().requires_grad_(True)
means3D, opac, scales, rots, shs = leaf(inp["means3D"]), leaf(inp["opacities"]), leaf(inp["scales"]), leaf(inp["rotations"]), leaf(inp["shs"])
means2D = torch.zeros_like(means3D, requires_grad=True)
tongue = inp["tongue_class"].cuda()
g = {k: v.cuda() for k, v in S.make_upstream_grads(H, W).items()}
def step(bwd=True):
    outs = rast(means3D=means3D, means2D=means2D, opacities=opac, tongue_class=tongue, shs=shs, scales=scales, rotations=rots)
    color, radii, coord, mcoord, depth, mdepth, alpha, tng, normal = outs
    if bwd:
        torch.autograd.backward([color, depth, mdepth, normal, alpha], [g["color"], g["depth"], g["mdepth"], g["normal"], g["alpha"]])
    return outs
for _ in range(3): step()
torch.cuda.synchronize()
for mode in (False, True):
    t0 = time.perf_counter()
    for _ in range(iters): step(mode)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{'fwd+bwd' if mode else 'fwd'}: {dt*1e3:.3f} ms/iter  ({1/dt:.1f}/s)")
outs = step(False)
print("num_rendered", rast and int((outs[1] > 0).sum()), "visible;")
