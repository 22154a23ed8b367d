"""How many of a C3 item's Gaussians pass K1's culling (radii > 0), how they sit in waves of 64, and how many receive a gradient."""
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "e-d3dgs_amd")]
import torch, bench
wl, model, cams, grads = bench.build(sys.argv[1] if len(sys.argv) > 1 else "C3", "cuda")
step = bench.make_step(model, cams, grads, wl, "cuda")
from ed3dgs_amd import dist as D
mine = D.shard_items(wl["cams"] * wl["frames"], 0, 1)
for k in range(8):
    with torch.no_grad():
        pkg, _ = step(D.visit_item(mine, k, wl["cams"], wl["frames"]), backward=False)
    r = pkg["radii"]
    vis = r > 0
    P = vis.numel()
    pad = (-P) % 64
    v64 = torch.cat([vis, vis.new_zeros(pad)]).view(-1, 64).sum(1)
    print("item %d: visible %.3f  waves with any visible lane %.3f  mean visible lanes in such a wave %.1f  waves needed if compacted %.3f"
          % (k, vis.float().mean().item(), (v64 > 0).float().mean().item(), v64[v64 > 0].float().mean().item(),
             ((v64.sum() + 63) // 64).item() / v64.numel()))
