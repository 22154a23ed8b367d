"""How long does the HOST take to enqueue one training step / one render (no device wait inside the loop)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
import torch
import bench
wl, model, cams, grads = bench.build("C3", "cuda")
step = bench.make_step(model, cams, grads, wl, "cuda")
for k in range(5):
    step(k)
torch.cuda.synchronize()
for mode in ("train", "render"):
    t0 = time.perf_counter()
    for k in range(40):
        if mode == "train":
            step(k)
        else:
            with torch.no_grad():
                step(k, backward=False, coord=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(mode, "host enqueue ms/iter %.3f" % ((t1 - t0) / 40 * 1e3), "total ms/iter %.3f" % ((t2 - t0) / 40 * 1e3))
