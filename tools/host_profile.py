"""Host-side cost of one training step: cProfile over the `tiny` workload (10k Gaussians, 400x400: the GPU work is ~0.3 ms, so the
step time IS the host time).  usage: python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    wl, model, cams, grads = bench.build("tiny", dev)
    step = bench.make_step(model, cams, grads, wl, dev)
    for k in range(20):
        step(k % 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        step(k % 8)
    torch.cuda.synchronize()
    print("ms per step (host-bound):", (time.perf_counter() - t0) / n * 1e3, flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for k in range(n):
        step(k % 8)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
