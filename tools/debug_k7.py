"""Debug helper: which Gaussians get a non-finite gradient from K7 and what their records look like."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "e-d3dgs_amd"))
import numpy as np, torch
import util
from diff_gaussian_rasterization import _C
from ed3dgs_amd import synthetic as S
variant = sys.argv[1] if len(sys.argv) > 1 else "FTT"
inp = util.scene_inputs(10000, 400, 400, kernel_size=0.0)
H, W = inp["H"], inp["W"]
grads = S.make_upstream_grads(H, W)
rc, rd = util.VARIANTS[variant]
for z in os.environ.get("ZERO", "").split(","):
    if z: grads[z].zero_()
out, sv = util.hip_forward_raw(inp, variant)
d = lambda t: t.cuda().contiguous()
e = torch.Tensor([])
res = _C.rasterize_gaussians_backward(
    d(inp["bg"]), d(inp["means3D"]), out[9], e, d(inp["scales"]), d(inp["rotations"]), inp["scale_modifier"], e,
    d(inp["viewmatrix"]), d(inp["projmatrix"]), inp["tanfovx"], inp["tanfovy"], inp["kernel_size"],
    d(grads["color"]), d(grads["coord"]), d(grads["mcoord"]), d(grads["depth"]), d(grads["mdepth"]),
    d(grads["alpha"]), d(grads["normal"]), out[6], d(inp["shs"]), inp["sh_degree"], d(inp["campos"]), out[10],
    out[0], out[11], out[12], out[4], rc, rd, False)
rec = sv["rec"]
print("non-finite records:", np.argwhere(~np.isfinite(rec))[:20], "count", (~np.isfinite(rec)).sum())
radii = out[9].cpu().numpy()
vis = radii > 0
print("non-finite among visible:", (~np.isfinite(rec[vis])).sum(), "absmax visible per slot", np.nanmax(np.abs(rec[vis]), axis=0))
for n, t in zip(["m2d", "col", "op", "m3d", "cov", "sh", "sc", "rot"], res):
    a = t.cpu().numpy().reshape(inp["P"], -1)
    bad = np.argwhere(~np.isfinite(a).all(axis=1)).ravel()
    print(n, "bad rows", len(bad), bad[:10])
alpha = out[4].cpu().numpy()
print("alpha min", alpha.min(), "zeros", (alpha == 0).sum(), "n_contrib zeros", (sv["n_contrib"][0] == 0).sum(), "nl min", sv["normal_length"].min())
