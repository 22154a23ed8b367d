"""Allocator diagnostic (round 4): which requests of a training step reach hipMalloc, and how the caching allocator's
reserved memory settles.  Run on the GPU box: `python tools/diag_alloc.py [steps]`."""
import sys, os, time, gc, collections
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "e-d3dgs_amd")]
import torch, bench
wl, model, cams, grads = bench.build("C3", "cuda")
step = bench.make_step(model, cams, grads, wl, "cuda")
from ed3dgs_amd import dist as D
mine = D.shard_items(400, 0, 1)
item = lambda k: D.visit_item(mine, k, wl["cams"], wl["frames"])
for k in range(10): step(item(k))
torch.cuda.synchronize()
torch.cuda.memory._record_memory_history(max_entries=200000, stacks="python")
for k in range(8): step(item(k))
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
ev = [e for tr in snap["device_traces"] for e in tr]
print("events", collections.Counter(e["action"] for e in ev))
sizes = collections.Counter(e["size"] for e in ev if e["action"] == "alloc")
print("alloc sizes over 8 steps (size: count), those not a multiple of 8 marked *")
for s, c in sorted(sizes.items()):
    print("  %12d %4d %s" % (s, c, "" if c % 8 == 0 else "*"))
for e in ev:
    if e["action"] == "segment_alloc":
        fr = [f for f in e.get("frames", []) if "site-packages" not in f["filename"]][:4]
        print("segment_alloc %d" % e["size"], " <- ".join("%s:%d" % (os.path.basename(f["filename"]), f["line"]) for f in fr))
# what a step leaves for the CYCLIC collector (objects plain reference counting does not free)
gc.collect(); gc.disable(); gc.set_debug(gc.DEBUG_SAVEALL)
for k in range(4): step(item(k))
torch.cuda.synchronize()
n = gc.collect()
print("cyclic garbage after 4 steps: %d objects" % n, collections.Counter(type(o).__name__ for o in gc.garbage).most_common(12))
print("   tensors in it: %.1f MB" % (sum(o.numel() * o.element_size() for o in gc.garbage if isinstance(o, torch.Tensor)) / 1e6))
gc.set_debug(0); gc.garbage.clear(); gc.enable()
for mode in ("gc-on", "gc-off", "gc-off"):
    if mode == "gc-off": gc.collect(); gc.disable()
    m0 = torch.cuda.memory_stats()["num_device_alloc"]; r0 = torch.cuda.memory_reserved()
    for k in range(40): step(item(k))
    torch.cuda.synchronize()
    gc.enable()
    print(mode, "mallocs", torch.cuda.memory_stats()["num_device_alloc"] - m0, "reserved MB %.0f -> %.0f" % (r0 / 1e6, torch.cuda.memory_reserved() / 1e6), "alloc MB %.0f" % (torch.cuda.memory_allocated() / 1e6))
