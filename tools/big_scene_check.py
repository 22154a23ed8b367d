import os, sys, time
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
import torch, bench
bench.WORKLOADS["big"] = dict(P=2_000_000, W=1920, H=1080, cams=2, frames=4, deform=True, name="big: 2M Gaussians 1080p")
dev = torch.device("cuda:0")
wl, model, cams, grads = bench.build("big", dev)
step = bench.make_step(model, cams, grads, wl, dev)
chk = {}
def probe(m):
    chk["emb"] = float(m._embedding.grad.abs().sum()); chk["act"] = int((m._embedding.grad.abs().amax(dim=1) > 0).sum())
    chk["w"] = float(sum(p.grad.abs().sum() for p in m._deformation.parameters() if p.grad is not None))
step.probe = probe
for k in range(3):
    t0 = time.perf_counter(); pkg, st = step(k); torch.cuda.synchronize(); print("step", k, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), chk, "R", int(pkg["radii"].gt(0).sum()), flush=True)
import math
assert all(math.isfinite(v) for v in chk.values() if isinstance(v, float))
print("ok, peak GB", torch.cuda.max_memory_allocated() / 2**30)
