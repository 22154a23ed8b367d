"""Round 4 probe: the forward of render() (C3) on a stream restricted to a subset of the compute units
(hipExtStreamCreateWithCUMask) -- how long the level-1 sort + binning launches take on 32 CUs and the deformation forward on 224,
i.e. whether the two could share the chip side by side.  Run under `rocprofv3 --kernel-trace --stats`.
usage (GPU box, repo root): python tools/r04_cu_mask_probe.py <n_cus> <contig|strided|all>"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "e-d3dgs_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

n_cus = int(sys.argv[1]); mode = sys.argv[2]
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl, model, cams, grads = bench.build("C3", dev)
step = bench.make_step(model, cams, grads, wl, dev)
stream = None
if mode != "all":
    hip = ctypes.CDLL("libamdhip64.so")
    bits = list(range(n_cus)) if mode == "contig" else [(i * 8) % 256 + (i * 8) // 256 for i in range(n_cus)]
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(8), words)
    print("hipExtStreamCreateWithCUMask rc", rc, "mask", [hex(w) for w in words], flush=True)
    assert rc == 0
    stream = torch.cuda.ExternalStream(h.value, device=dev)
for it in range(3):
    with torch.no_grad():
        step(it, backward=False)
torch.cuda.synchronize()
ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
with ctx:
    ts = []
    for it in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            step(it, backward=False)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("forward render, %s %d CUs: median %.4f ms min %.4f" % (mode, n_cus, ts[len(ts) // 2], ts[0]), flush=True)
torch.cuda.synchronize()
