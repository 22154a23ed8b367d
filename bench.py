#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native E-D3DGS hot path.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a torchrun environment (WORLD_SIZE unset): this process makes NO GPU call and starts
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...` as a child,
  relays rank 0's JSON line and exits with the child's code.  Under torchrun (WORLD_SIZE set) it is one rank of N.

A "step" = one pass of the hot path over one (camera, frame) item: gaussian_renderer.render() forward (fused
deformation MLP -> activations -> preprocess -> binning -> tile forward) + backward (tile backward -> per-Gaussian
backward -> activations -> deformation backward), i.e. one train.py iteration without data loading / optimizer
(SURVEY.md 8d).  Workload = BASELINE.json configs[2] ("C3"): 200k Gaussians, 8 synthetic 1080p cameras x 50
timesteps, deformation on (W=128, D=1, nersemble flags, iter=20000), depth+normal variant.  Items are sharded
i = rank (mod N) (weak scaling: every rank runs K steps) with one 12-byte all-reduce of [loss, psnr, count] per step.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-operand MFMA peak (MI355X_MICROARCH.md, matrix-core table)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same table)
MODE_OPTS = ("DEFORM_FP32_MFMA", "DEFORM_BF16X3")   # library switches (ed3dgs_set_option) that select another MLP multiply mode
WORKLOADS = {
    "C3": dict(P=200_000, W=1920, H=1080, cams=8, frames=50, deform=True,
               name="C3: 200k Gaussians, 8 cams x 50 timesteps, 1080p, deform MLP W=128 D=1, depth+normal (FTT)"),
    "C2": dict(P=100_000, W=1920, H=1080, cams=1, frames=1, deform=False,
               name="C2: 100k Gaussians, 1 cam 1080p, static, depth+normal (FTT)"),
    "C4": dict(P=200_000, W=1100, H=1604, cams=15, frames=300, deform=True,
               name="C4: 200k Gaussians, 15 train cams (16 views, cam00 held out) x 300 timesteps, NeRSemble-shaped 1100x1604, "
                    "deform MLP W=128 D=1, depth+normal (FTT)"),
    "C5": dict(P=500_000, W=1920, H=1080, cams=8, frames=150, deform=True,
               name="C5: 500k Gaussians, SH degree 3, 8 cams x 150 frames, 1080p, full deform, depth+normal (FTT) training step; "
                    "render_fps = all outputs (TTT)"),
    "C1": dict(P=10_000, W=400, H=400, cams=1, frames=1, deform=False,
               name="C1: 10k Gaussians, 1 cam 400x400, 1 timestep, no deformation (plumbing / CPU-baseline point)"),
    "tiny": dict(P=10_000, W=400, H=400, cams=2, frames=4, deform=True, name="tiny: 10k Gaussians 400x400 (plumbing)"),
}


def build(workload, device):
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import SynthGaussianModel, default_hyper
    wl = WORKLOADS[workload]
    scene = S.make_scene(wl["P"], seed=0)
    cams = S.make_cameras(wl["cams"], wl["W"], wl["H"], seed=1, device=device)
    hyper = default_hyper() if wl["deform"] else default_hyper(no_coarse_deform=True, no_fine_deform=True)
    model = SynthGaussianModel(scene, args=hyper, deform_seed=2, device=device)
    grads = {k: v.to(device) for k, v in S.make_upstream_grads(wl["H"], wl["W"], seed=3).items()}
    return wl, model, cams, grads


def make_step(model, cams, grads, wl, device, dp_grads=False):
    from ed3dgs_amd import dist as D
    from ed3dgs_amd.model import PIPE
    from gaussian_renderer import render
    bg = torch.ones(3, device=device)
    params = model.parameters()
    F = wl["frames"]
    inflight = []
    ups = [grads["color"], grads["depth"], grads["mdepth"], grads["normal"]]
    ups_color_flat = grads["color"].reshape(-1).contiguous()
    from ed3dgs_amd import _lib
    L = _lib.lib()
    stats_acc = torch.zeros(272, device=device)   # ED3DGS_STATS_ACC_FLOATS (include/ed3dgs.h)
    stats_out = [torch.zeros(3, device=device), torch.zeros(3, device=device)]
    # opt-in: data-parallel training -- buckets all-reduced from autograd hooks as their gradients land
    reducer = D.BucketedGradReducer(params) if dp_grads else None

    def step(item, backward=True, coord=False):
        ci, fi = D.item_of(item, wl["cams"], F)
        cam = cams[ci].with_time(fi / F)
        t_a = time.perf_counter()
        pkg = render(cam, model, PIPE, bg, kernel_size=0.0, require_coord=coord, require_depth=True, cam_no=None,
                     iter=20000, num_down_emb_c=30, num_down_emb_f=30, disable_filter3D=True)
        t_b = time.perf_counter()
        if not backward:
            return pkg, None
        # fixed upstream gradients stand in for the L1/SSIM + depth-normal losses (the losses are outside the
        # path): they enter the backward directly
        outs = [pkg["render"], pkg["expected_depth"], pkg["median_depth"], pkg["normal"]]
        torch.autograd.backward(outs, ups)
        step.host_split.append((t_b - t_a, time.perf_counter() - t_b))   # host seconds in render() (it waits for K1's count) / in backward()
        # logging stand-in (train.py logs the image loss and PSNR): <image, its upstream gradient> and a PSNR against mid-grey,
        # one fused launch (csrc/stats.hip; ten torch launches, 90 us, as separate ops) into one of two alternating buffers
        # (the previous step's vector may still be in its all-reduce)
        stats = stats_out[step.n & 1]
        step.n += 1
        img = pkg["render"].detach()
        rc = L.ed3dgs_image_stats(ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(ups_color_flat.data_ptr()),
                                  ctypes.c_size_t(img.numel()), ctypes.c_float(0.5), ctypes.c_void_p(stats_acc.data_ptr()),
                                  ctypes.c_void_p(stats.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc < 0:
            raise RuntimeError(_lib.last_error())
        # the path's one collective (RCCL over xGMI): 12 bytes, waited for one step later (the stream, not the host,
        # waits), so that ranks are not re-synchronised every step
        if reducer is not None:  # mean of the ranks' gradients (the collectives were issued during the backward); BEFORE any other
            reducer.finish()     # collective of the group: a rank may issue its last buckets only here (dist.BucketedGradReducer)
        if inflight:
            inflight.pop().wait()
        inflight.append(D.allreduce_sum_async(stats))
        if step.probe is not None:   # untimed passes only: look at the gradients before they are dropped
            step.probe(model)
        for p in params:
            p.grad = None
        pkg["viewspace_points"].grad = None
        return pkg, stats

    def drain():
        while inflight:
            inflight.pop().wait()

    step.drain = drain
    step.host_split = []
    step.n = 0
    step.probe = None
    return step


def r_eff_of_last(wl):
    """R_eff = sum over tiles of the largest last-contributor index in the tile, from the image state."""
    from diff_gaussian_rasterization import _C
    L = _C.LAST
    nc = _C.n_contrib_view(L["P"], L["H"], L["W"], L["R"], L["geom"], L["img"])[0]
    H, W = L["H"], L["W"]
    gy, gx = (H + 15) // 16, (W + 15) // 16
    pad = torch.zeros(gy * 16, gx * 16, dtype=nc.dtype, device=nc.device)
    pad[:H, :W] = nc
    return int(pad.reshape(gy, 16, gx, 16).amax(dim=(1, 3)).sum()), int(nc.sum()), int(L["R"])


def host_cores():
    """Cores this process may actually use: scheduler affinity, capped by the cgroup CPU quota (a GPU box hands each
    job a share of a much larger host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(wl, full=False):
    """PyTorch-CPU autograd restatement (oracle/) of the same step on a bounded sample, all usable host cores
    (full=True: every tile, nothing extrapolated -- the C1 point of SURVEY 8d)."""
    from ed3dgs_amd import synthetic as S
    from ed3dgs_amd.model import default_hyper
    from oracle import deformation_torch as DT
    from oracle import torch_raster as TR
    cores = host_cores()
    torch.set_num_threads(cores)
    scene = S.make_scene(wl["P"], seed=0)
    cam = S.make_cameras(wl["cams"], wl["W"], wl["H"], seed=1)[0]
    hyper = default_hyper()
    leaf = lambda t: t.clone().requires_grad_(True)
    xyz, ls, rot, op = leaf(scene.xyz), leaf(scene.log_scale), leaf(scene.rot), leaf(scene.opacity)
    sh = leaf(torch.cat((scene.f_dc, scene.f_rest), 1))
    emb = leaf(scene.embedding)
    t0 = time.perf_counter()
    if wl["deform"]:
        from scene.deformation import deform_network
        torch.manual_seed(2)
        net = deform_network(W=hyper.net_width, D=hyper.defor_depth, min_embeddings=30, max_embeddings=150, args=hyper)
        with torch.no_grad():
            net.weight.mul_(100.0)
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
        (xyz_f, ls_f, rot_f, op_f, sh_f), _ = DT.forward(sd, hyper, 1, 150, xyz, ls, rot, op, sh, emb, 0.0, None, 20000, 30, 30)
    else:
        xyz_f, ls_f, rot_f, op_f, sh_f = xyz, ls, rot, op, sh
    scales = torch.exp(ls_f); rots = torch.nn.functional.normalize(rot_f); opac = torch.sigmoid(op_f)
    T = ((wl["W"] + 15) // 16) * ((wl["H"] + 15) // 16)
    stride = 1 if full else max(1, T // 1600)   # every 5th tile at 1080p: ~15-20 s of CPU work on 16 cores (round 3: every 20th, 5.5 s)
    subset = list(range(stride // 2, T, stride))
    pp = TR.preprocess(xyz_f, scales, rots, opac, sh_f, cam.world_view_transform, cam.full_proj_transform,
                       cam.camera_center, wl["W"], wl["H"], math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), 0.0, 1.0, 3)
    ids, starts = TR.bin_tiles(pp)
    t_pg = time.perf_counter() - t0            # per-Gaussian part, run in full
    out = TR.render(pp, ids, starts, torch.ones(3), wl["W"], wl["H"], False, True, tile_subset=subset)
    g = S.make_upstream_grads(wl["H"], wl["W"], seed=3)
    loss = ((out["color"] * g["color"]).sum() + (out["depth"] * g["depth"]).sum() + (out["mdepth"] * g["mdepth"]).sum() +
            (out["normal"] * g["normal"]).sum())
    t_tiles = time.perf_counter() - t0 - t_pg  # tile compositing on the sub-sample
    t1 = time.perf_counter()
    loss.backward()
    t_bwd = time.perf_counter() - t1
    # only the tile compositing is sub-sampled (every `stride`-th tile) and scaled by the tile ratio; the backward is
    # not separable, so it is scaled by the same factor the forward's total grows by.
    ratio = T / len(subset)
    fwd_est = t_pg + t_tiles * ratio
    est = fwd_est + t_bwd * fwd_est / (t_pg + t_tiles)
    return dict(value=1.0 / est, unit="iters/s", cores=cores, kind="port",
                sample=(f"1 item of the workload; deformation+preprocess+binning in full ({t_pg:.1f} s), tile compositing "
                        f"on {len(subset)} of {T} tiles (every {stride}th, {t_tiles:.1f} s, scaled x{ratio:.1f}), autograd "
                        f"backward {t_bwd:.1f} s scaled by the forward's growth; estimated {est:.0f} s per step; "
                        f"PyTorch-CPU autograd restatement (oracle/torch_raster.py + oracle/deformation_torch.py), fp32"),
                measured_seconds=round(t_pg + t_tiles + t_bwd, 2))


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mark-every", type=int, default=0, help="steps per step-time window of the timed region (default 4; a mark is a barrier packet on the stream)")
    ap.add_argument("--cpu-baseline-points", action="store_true", help="also time the CPU restatement at C1 (in full) and at the C2 point (SURVEY 8d); minutes of CPU work")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the untimed extra passes in the MLP's other multiply modes (for profiles)")
    ap.add_argument("--sequential-items", action="store_true", help="rank r's k-th item is its k-th (camera 0, frames 0, 1, ... at N=1): round 2's "
                    "schedule, for comparisons with its numbers; the default strides over the whole (camera, frame) list")
    ap.add_argument("--train-only", action="store_true", help="skip the forward-only render-fps passes too, so that every launch of the run is a "
                    "training-step launch (for rocprofv3 --stats: its per-kernel averages then are the training step's)")
    ap.add_argument("--dp-grads", action="store_true", help="also all-reduce the gradients every step (data-parallel training; not the headline configuration)")
    ap.add_argument("--rehearse-launcher", action="store_true",
                    help="NO GPU work and NO measurement: run the launcher / rendezvous / sharding / collectives / JSON plumbing "
                         "with a stub step (CPU, gloo) -- what tests/test_bench_launcher_cpu.py drives")
    return ap.parse_args(argv)


def launch_ranks(a, argv):
    """--gpus N > 1 outside torchrun: start N ranks as a fresh child process tree BEFORE this process touches the GPU (a
    process that has initialised HIP must never exec or be duplicated into ranks), relay the child's output (rank 0 prints
    the JSON line) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["ED3DGS_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("launching %d ranks: %s" % (a.gpus, " ".join(cmd)))
    # rank 0's JSON line goes to stdout; whatever else the ranks' libraries print there (gloo announces its connections on
    # stdout) is passed on to stderr, so that the parent's stdout is the ONE line the contract asks for
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def hbm_ceiling(device, nbytes=1 << 30, reps=10):
    """On-box HBM ceiling, measured in the untimed section (state it beside the 8 TB/s spec): a device-to-device
    copy (read N + write N) and a stream triad a = b + s*c (read 2N + write N) over 1-GiB arrays, hipEvent-timed."""
    n = nbytes // 4
    x, y, z = (torch.empty(n, dtype=torch.float32, device=device).normal_() for _ in range(3))
    out = {}
    for name, fn, moved in (("copy", lambda: z.copy_(x), 2 * nbytes), ("triad", lambda: torch.add(x, y, alpha=3.0, out=z), 3 * nbytes)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        out[name] = moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del x, y, z
    torch.cuda.empty_cache()
    return out


def percentiles(v):
    v = sorted(v)
    q = lambda f: v[min(len(v) - 1, max(0, int(round(f * (len(v) - 1)))))]
    return {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "min": v[0], "max": v[-1], "n": len(v)}


def rehearse(a, D, emit):
    """Launcher rehearsal (no GPU, no measurement): every piece of the multi-rank harness around a stub step."""
    import torch.distributed as dist
    rank, world, local = D.init()
    wl = WORKLOADS[a.workload]
    n_items = wl["cams"] * wl["frames"]
    mine = D.shard_items(max(n_items, world), rank, world)
    inflight = []
    D.barrier()
    t0 = time.perf_counter()
    items = [D.visit_item(mine, k, wl["cams"], wl["frames"]) % n_items for k in range(a.steps)]   # the item schedule of the real path
    t_step = []
    for k in range(a.steps):
        t1 = time.perf_counter()
        stats = torch.tensor([float(items[k]), 0.0, 1.0])
        if inflight:
            inflight.pop().wait()
        inflight.append(D.allreduce_sum_async(stats))
        t_step.append((time.perf_counter() - t1) * 1e3)
    while inflight:
        last = inflight.pop().wait()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, "cpu")
    counts = torch.zeros(world); counts[rank] = a.steps
    D.allreduce_sum_(counts)
    rank_step_ms = D.gather_per_rank(percentiles(t_step)["median"], "cpu")     # the per-rank fields of the real line
    rank_num_rendered = D.gather_per_rank(float(sum(items)) / max(len(items), 1), "cpu")
    cams = sorted({D.item_of(i, wl["cams"], wl["frames"])[0] for i in items})
    ncams = D.gather_per_rank(len(cams), "cpu")
    if rank == 0:
        emit(({"metric": "REHEARSAL of the launcher (stub step, no GPU work) -- not a measurement", "value": None,
                          "rehearsal": True, "n_gpus": world, "gpus_arg": a.gpus, "steps": a.steps, "warmup": a.warmup,
                          "backend": dist.get_backend() if dist.is_initialized() else None, "items_per_rank": counts.tolist(),
                          "last_step_ranks_counted": float(last[2]), "seconds": dt,
                          "step_ms_median_per_rank": rank_step_ms, "mean_num_rendered_per_rank": rank_num_rendered,
                          "rank0_items_timed": items, "rank0_cameras_timed": cams, "cameras_timed_per_rank": ncams,
                          "launched_by_bench": bool(os.environ.get("ED3DGS_BENCH_LAUNCHED"))}))
    D.destroy()


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))            # nothing above this line touches the GPU
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner on stdout when its
    # first communicator is created -- seen on the GPU box in round 4 -- and gloo announces its connections): from here on file
    # descriptor 1 IS stderr, and emit() below writes the JSON line to the real stdout kept aside.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        data = (json.dumps(obj) + "\n").encode()
        while data:                                   # (a pipe may take the ~12 KB line in pieces)
            data = data[os.write(real_stdout, data):]

    from ed3dgs_amd import dist as D
    if a.rehearse_launcher:
        env_world = int(os.environ.get("WORLD_SIZE", "1"))
        if a.gpus != env_world:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, env_world))
        return rehearse(a, D, emit)
    from ed3dgs_amd import _lib
    # the rank's device is chosen BEFORE the process group exists: RCCL binds a rank to the device that is current when its
    # communicator is created (first collective), and `barrier()` must not run on device 0 for every rank
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    rank, world, local = D.init()
    backend_name = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the torchrun environment has WORLD_SIZE=%d -- refusing to report a line "
                         "whose n_gpus is not what was asked for" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    shared_gpu = world > torch.cuda.device_count()         # rehearsal on a 1-GPU box: ranks share the card (gloo)
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    L = _lib.lib()
    from diff_gaussian_rasterization import _C
    log("building workload", a.workload)
    wl, model, cams, grads = build(a.workload, device)
    log("built; warm-up")
    step = make_step(model, cams, grads, wl, device, dp_grads=a.dp_grads)
    n_items = wl["cams"] * wl["frames"]
    my_items = D.shard_items(max(n_items, world), rank, world)
    item_at = (lambda k: my_items[k % len(my_items)] % n_items) if a.sequential_items else \
              (lambda k: D.visit_item(my_items, k, wl["cams"], wl["frames"]) % n_items)   # cameras round-robin, frames strided: any run of `cams` steps visits every camera
    ceiling = hbm_ceiling(device) if rank == 0 else None
    mfma_ceiling = None
    if rank == 0:   # what the matrix pipe sustains on this box (untimed): the spec peak the roofline uses is not a sustained rate
        tf, cms = ctypes.c_double(0.0), ctypes.c_double(0.0)
        torch.cuda.synchronize()
        if L.ed3dgs_measure_mfma_ceiling(ctypes.c_int(4000), ctypes.byref(tf), ctypes.byref(cms)) == 0:
            mfma_ceiling = {"bf16_32x32x16_TFLOPs": tf.value, "ms": cms.value, "spec_dense_peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS,
                            "note": "v_mfma_f32_32x32x16_bf16 back to back on random operands in registers, two waves per SIMD on all 1024 "
                                    "SIMDs, no memory traffic: the rate the chip holds at its power limit (the clock drops under the matrix "
                                    "load); `peak` in the roofline stays the spec figure"}

    # ---- untimed: warm-up + algorithmic-byte bookkeeping of the items the timed region will visit ----
    _C.KEEP_LAST = True
    for k in range(a.warmup):
        step(item_at(k))
    step.drain()
    torch.cuda.synchronize()
    log("warm-up done; bookkeeping pass")
    reff, npairs_ub, rsum = [], [], []
    import contextlib
    with (contextlib.nullcontext() if a.train_only else torch.no_grad()):   # (--train-only: full steps here too, no inference launch in the run)
        for k in range(a.steps):
            step(item_at(k), backward=a.train_only)
            r, n, R = r_eff_of_last(wl)
            reff.append(r); npairs_ub.append(n); rsum.append(R)
    _C.KEEP_LAST = False
    _C.LAST.clear()

    # The interpreter's cyclic collector: run now, then off until the timed region is over (a collection inside it showed as a
    # 6-ms step of 2.2).  HERE and not in front of the timed region: a full collection is ~50 ms of idle GPU, after which the
    # kernels run up to 15 % slower and recover over ~10 steps (kernel trace, round 4: the sum of a step's kernel durations
    # 2.52 -> 2.17 ms over the twelve steps after the pause, no launch gaps -- the clock ramp).  The K-step pass below follows it.
    import gc
    gc.collect(); gc.disable()
    # ---- per-kernel table: an instrumented, UNTIMED pass (five event pairs per step cost stream time) ----
    L.ed3dgs_profile_begin_slots(ctypes.c_int(a.steps + 4), ctypes.c_uint(0x1FF))
    for k in range(a.steps):
        step(item_at(k))
    step.drain()
    torch.cuda.synchronize()
    NS = 9  # ED3DGS_PROF_SLOTS; slot 4 is the three weight-gradient launches together, 5..7 each of them, 8 = K8+K9
    tab_ms, tab_n = (ctypes.c_double * NS)(), (ctypes.c_int * NS)()
    L.ed3dgs_profile_end_slots(tab_ms, tab_n)
    tab_avg = [tab_ms[i] / max(tab_n[i], 1) for i in range(NS)]
    # K7's work counts (visited iterations, blended pairs, ...): a short pass of its own -- counting slows the kernel down
    n_count = 0 if a.train_only else min(4, a.steps)   # (--train-only: no counting launches either -- they are slower ones)
    L.ed3dgs_profile_begin_slots(ctypes.c_int(n_count + 2), ctypes.c_uint(3 | (1 << 30)))
    # ... and the rows the deformation backward walks: Gaussians with a non-zero upstream gradient = non-zero dL/d embedding rows
    active_rows = []
    if wl.get("deform", True):
        step.probe = lambda m: active_rows.append(int((m._embedding.grad.abs().amax(dim=1) > 0).sum()))
    for k in range(n_count):
        step(item_at(k))
    step.probe = None
    step.drain()
    torch.cuda.synchronize()
    cnt_ms, cnt_n = (ctypes.c_double * NS)(), (ctypes.c_int * NS)()
    L.ed3dgs_profile_end_slots(cnt_ms, cnt_n)
    NCNT = 16  # ED3DGS_PROF_COUNTERS
    tile_counts = (ctypes.c_ulonglong * NCNT)()
    L.ed3dgs_profile_tile_counts(tile_counts, ctypes.c_int(NCNT))
    k7_work = [tile_counts[i] / max(cnt_n[1], 1) for i in range(12)]   # per launch: iterations, pairs, staged, kept, quadrant histogram, halves
    k6_work = [tile_counts[12 + i] / max(cnt_n[0], 1) for i in range(4)]  # per launch: iterations, pairs, staged, kept
    k7_count_ms = cnt_ms[1] / max(cnt_n[1], 1)
    k6_count_ms = cnt_ms[0] / max(cnt_n[0], 1)
    dom = max([i for i in range(NS) if i != 4], key=lambda i: tab_avg[i])   # the dominant single KERNEL

    # ---- timed: exactly K steps; events only around the dominant kernel (every event pair is a barrier packet on the launch
    # stream, ~6 us of gap: K7 and the others are timed in the instrumented pass above) ----
    # The W warm-up steps once more, directly before the timed region: the bookkeeping passes above end in host-side reads
    # (an idle GPU, whose clock then ramps through the first timed steps: their window read 2.7-2.9 ms against 2.2).
    for k in range(a.warmup):
        step(item_at(k))
    step.drain()
    torch.cuda.synchronize(); D.barrier()
    log("timed region")
    # (every third launch of it: 7 of 20 steps, a stride coprime to the camera count; an event pair is ~12 us of idle stream)
    L.ed3dgs_profile_begin_slots(ctypes.c_int(a.steps + 4), ctypes.c_uint((1 << dom) | (1 << 29)))
    # step-time spread: a mark every MARK_EVERY steps on the launch stream (a mark is a barrier packet too: ~6 us of idle stream,
    # 0.3 % of a step if taken at every boundary)
    MARK_EVERY = a.mark_every if a.mark_every > 0 else (4 if a.steps >= 8 else 1)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps // MARK_EVERY + 1)]
    n_malloc0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0) if torch.cuda.is_available() else 0
    t0 = time.perf_counter()
    marks[0].record()
    step.host_split.clear()
    host_t = [t0]
    for k in range(a.steps):
        step(item_at(k))
        if (k + 1) % MARK_EVERY == 0:
            marks[(k + 1) // MARK_EVERY].record()
        host_t.append(time.perf_counter())   # when the host had finished enqueueing step k (it waits for K1's count inside it)
    step.drain()
    torch.cuda.synchronize(); D.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    n_malloc = (torch.cuda.memory_stats(device).get("num_device_alloc", 0) - n_malloc0) if torch.cuda.is_available() else 0
    step_ms = [marks[k].elapsed_time(marks[k + 1]) / MARK_EVERY for k in range(len(marks) - 1)]
    slot_ms, slot_n = (ctypes.c_double * NS)(), (ctypes.c_int * NS)()
    L.ed3dgs_profile_end_slots(slot_ms, slot_n)
    # K6, K7, deform fwd, deform dgrad, deform wgrad (+ its three launches): timed-region events where taken, else the
    # instrumented pass
    avg_ms = [slot_ms[i] / slot_n[i] if slot_n[i] else tab_avg[i] for i in range(NS)]
    dt_local = dt
    dt = D.max_over_ranks(dt, device)

    log("timed region done: %.3f ms/step" % (dt / a.steps * 1e3))
    # ---- forward-only render fps (render.py's TTT variant), untimed w.r.t. the headline ----
    torch.cuda.synchronize()
    dt_r, fwd_nokeep_ms = float("inf"), None
    if not a.train_only:
        with torch.no_grad():
            t1 = time.perf_counter()
            for k in range(a.steps):
                step(item_at(k), backward=False, coord=True)
            torch.cuda.synchronize()
            dt_r = time.perf_counter() - t1
            # ... and the deformation forward WITHOUT kept activations (inference), timed in a short pass of its own: what keeping costs
            L.ed3dgs_profile_begin_slots(ctypes.c_int(12), ctypes.c_uint(1 << 2))
            for k in range(min(8, a.steps)):
                step(item_at(k), backward=False, coord=True)
            torch.cuda.synchronize()
            nk_ms, nk_n = (ctypes.c_double * NS)(), (ctypes.c_int * NS)()
            L.ed3dgs_profile_end_slots(nk_ms, nk_n)
            fwd_nokeep_ms = nk_ms[2] / max(nk_n[2], 1)
    dt_r = D.max_over_ranks(dt_r, device)

    # ---- extras, not the headline: the same K steps / K renders with the deformation MLP in its other two modes ----
    def other_mode(var, note):
        # not on a multi-rank run (the extra passes carry barriers of their own and are not part of the scaling measurement)
        if a.no_other_modes or a.train_only or world > 1 or not wl["deform"] or any(_lib.get_option(v) for v in MODE_OPTS):
            return None
        _lib.set_option(var, 1)
        try:
            for k in range(3):
                step(item_at(k))
            step.drain()
            torch.cuda.synchronize(); D.barrier()
            t2 = time.perf_counter()
            for k in range(a.steps):
                step(item_at(k))
            step.drain()
            torch.cuda.synchronize(); D.barrier()
            dt_b = D.max_over_ranks(time.perf_counter() - t2, device)
            with torch.no_grad():
                t3 = time.perf_counter()
                for k in range(a.steps):
                    step(item_at(k), backward=False, coord=True)
                torch.cuda.synchronize()
                dt_br = D.max_over_ranks(time.perf_counter() - t3, device)
            return {"ms_per_step": dt_b / a.steps * 1e3, "value": world * a.steps / dt_b,
                    "render_fps": world * a.steps / dt_br, "note": note}
        finally:
            _lib.set_option(var, 0)

    mode = "fp32_mfma" if _lib.get_option("DEFORM_FP32_MFMA") else \
           "bf16x3" if _lib.get_option("DEFORM_BF16X3") else "exact_split"
    f32m = other_mode("DEFORM_FP32_MFMA",
                      "ED3DGS_DEFORM_FP32_MFMA=1: every MLP contraction on v_mfma_f32_32x32x2_f32 (the round's first kernels; "
                      "same results as the headline mode to fp32 rounding, DESIGN.md section 2)")
    b3 = other_mode("DEFORM_BF16X3",
                    "opt-in ED3DGS_DEFORM_BF16X3=1: two bf16 pieces per operand, three products, fp32 accumulation "
                    "(deformation outputs / gradients within 1e-6 / 2e-5 of the fp32 values, tolerance 1e-4; REDUCED "
                    "precision, never the headline)")

    # per-rank figures (load imbalance must be visible in the one line rank 0 prints): median step, mean instance count
    mean = lambda v: sum(v) / max(len(v), 1)
    rank_step_ms = D.gather_per_rank(percentiles(step_ms)["median"], device)
    rank_num_rendered = D.gather_per_rank(mean(rsum), device)
    rank_wall_ms = D.gather_per_rank(dt_local / a.steps * 1e3, device)
    items_timed = [item_at(k) for k in range(a.steps)]
    cams_timed = sorted({D.item_of(i, wl["cams"], wl["frames"])[0] for i in items_timed})
    D.barrier()
    D.destroy()
    if rank != 0:
        return
    HW, T = wl["H"] * wl["W"], ((wl["W"] + 15) // 16) * ((wl["H"] + 15) // 16)
    bytes_k7 = 128.0 * mean(reff) + 68.0 * HW + 8.0 * T   # FTT: (g_b + 4a) R_eff + r HW + 8T  
    bytes_k6 = 68.0 * mean(reff) + 56.0 * HW + 8.0 * T
    k6_ms, k7_ms = avg_ms[0], avg_ms[1]
    ach = bytes_k7 / (k7_ms * 1e-3) / 1e9 if k7_ms > 0 else 0.0
    # HBM traffic per launch: NOT measured by this run (PMC counters need their own rocprofv3 --pmc passes); read from the
    # newest committed summary of such passes over this same command, and labelled as such (`traffic_source`)
    pmc, traffic_source = {}, None
    for prof_name in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
        prof = os.path.join(ROOT, "profiles", prof_name)
        if not os.path.exists(prof):
            continue
        try:
            pj = json.load(open(prof))
            if pj.get("workload") == a.workload:
                pmc = pj.get("hbm_bytes_per_launch", {})
                traffic_source = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s` (committed file, not "
                                  "collected by this run)" % (prof_name, pj.get("command", "python bench.py")))
                break
        except Exception:
            pmc = {}
    # deformation MLP, algorithmic flops per Gaussian with the per-frame temporal row hoisted (SURVEY 8d counts the
    # un-hoisted 288-wide first layer: 0.505 MFLOP; the 256 broadcast inputs are one GEMV per FRAME here)
    from ed3dgs_amd.model import default_hyper
    hy = default_hyper()
    Wn, En = int(hy.net_width), int(hy.gaussian_embedding_dim)
    dm = wl.get("deform", True)
    outs = 3 + 3 + 4 + 1 + 48
    mac_all = 2 * (En * Wn + 5 * Wn * Wn + Wn * outs) if dm else 0          # forward == data gradient, both stages
    mac_trunk = 2 * En * Wn if dm else 0                                        # dW1
    mac_wide = 2 * (Wn * Wn + Wn * 48) if dm else 0                             # SH head: dW2 + dW3
    mac_narrow = 2 * (4 * Wn * Wn + Wn * (outs - 48)) if dm else 0              # the four narrow heads
    # pieces per operand pair executed on the matrix pipe, per kernel and mode: 8 exact bf16 products (exact_split),
    # 3 (bf16x3), or 1 f32 product
    npr = {"exact_split": 8, "bf16x3": 3, "fp32_mfma": 1}[mode]
    K = {  # slot -> (kernel name, algorithmic MAC per Gaussian, products per MAC, matrix pipe)
        2: ({"exact_split": "deform_forward_b3_kernel<4,3>", "bf16x3": "deform_forward_b3_kernel<4,2>", "fp32_mfma": "deform_forward_pipe_kernel<4>"}[mode], mac_all, npr),
        3: ({"exact_split": "deform_dgrad_kept_bn_kernel<4,3>", "bf16x3": "deform_dgrad_kept_b3_kernel<4>", "fp32_mfma": "deform_dgrad_kept_kernel<4>"}[mode], mac_all, npr),
        5: ("deform_dw1_kernel", mac_trunk, 1),
        6: ({"exact_split": "deform_head_wgrad_tr_kernel<true>", "bf16x3": "deform_head_wgrad_kernel<true,true,false> + deform_dw3_wide_kernel", "fp32_mfma": "deform_head_wgrad_kernel<true,false,true>"}[mode], mac_wide, {"exact_split": 8, "bf16x3": 3, "fp32_mfma": 1}[mode]),
        7: ({"exact_split": "deform_head_wgrad_tr_kernel<false>", "bf16x3": "deform_head_wgrad_kernel<false,true,true>", "fp32_mfma": "deform_head_wgrad_kernel<false,false,true>"}[mode], mac_narrow, npr),
    }
    if mode == "exact_split" and tab_n[6] == 0 and tab_n[7] > 0:
        # default since round 4: the SH head's and the narrow heads' weight gradients are ONE launch (timed under the narrow slot)
        del K[6]
        K[7] = ("deform_head_wgrad_tr_all_kernel", mac_wide + mac_narrow, npr)
    tfl = lambda mac, ms, rows=None: 2.0 * mac * (wl["P"] if rows is None else rows) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    # the backward kernels walk only the rows with a non-zero upstream gradient (csrc/deform.hip, deform_active_rows_body):
    # their rates are priced on the rows they process, not on P
    dense_bwd = bool(_lib.get_option("DEFORM_DENSE_BWD")) or mode != "exact_split"
    rows_bwd = wl["P"] if (dense_bwd or not active_rows) else mean(active_rows)
    kernels = {
        "_source": "separate instrumented pass of the same %d steps (event pairs around every kernel), not the timed region" % a.steps,
        "_mode": mode,
        "render_forward_kernel<false,true> (K6)": {"avg_launch_ms": tab_avg[0], "launches": tab_n[0], "bound": "valu", "GBps_algorithmic": bytes_k6 / (tab_avg[0] * 1e-3) / 1e9 if tab_avg[0] > 0 else 0.0},
        "render_backward_kernel<false,true> (K7)": {"avg_launch_ms": tab_avg[1], "launches": tab_n[1], "bound": "valu", "GBps_algorithmic": bytes_k7 / (tab_avg[1] * 1e-3) / 1e9 if tab_avg[1] > 0 else 0.0},
        "preprocess_backward_kernel (K8+K9)": {"avg_launch_ms": tab_avg[8], "launches": tab_n[8], "bound": "latency"},
    }
    if dm:
        for sl, (nm, mac, pr) in K.items():
            rows = wl["P"] if sl == 2 else rows_bwd
            kernels[nm] = {"avg_launch_ms": tab_avg[sl], "launches": tab_n[sl], "bound": "mfma", "rows_per_launch": rows,
                           "TFLOPs_fp32_equivalent": tfl(mac, tab_avg[sl], rows),
                           "matrix_pipe": "bf16 (%d exact piece products per multiply)" % pr if pr > 1 else "f32"}
        kernels["weight-gradient launches together"] = {"avg_launch_ms": tab_avg[4], "launches": tab_n[4]}
    # K7 against the vector-ALU roof: its inner loop issues this many vector instructions per visited iteration (an iteration
    # now differentiates up to four Gaussians, one per quadrant; counted in the ISA of render_backward_kernel<false,true>,
    # tools/isa.sh, end of round 4: 226 at 4 issue cycles, 4 v_exp_f32 + 4 v_rcp_f32 at 8 -- round 3: 64 + 189 and 8), a SIMD issues one per cycle, 1024 SIMDs
    # at 2.4 GHz; iterations and blended pairs are COUNTED by the kernel in the instrumented pass (popcount of the valid
    # masks), the time is that pass's launch time with the counting on
    K7_ISSUE_CYCLES_PER_ITER = 226 * 4 + 8 * 8
    it_per_s = k7_work[0] / (k7_ms * 1e-3) if k7_ms > 0 else 0.0   # counts are per item (deterministic); time = the instrumented pass's launches
    valu_roof = {"visited_iterations_per_launch": k7_work[0], "blended_pairs_per_launch": k7_work[1],
                 "list_entries_staged_per_launch": k7_work[2], "entries_kept_by_tile_reject_per_launch": k7_work[3],
                 "pairs_per_iteration": k7_work[1] / k7_work[0] if k7_work[0] else 0.0,
                 "issue_cycles_per_iteration": K7_ISSUE_CYCLES_PER_ITER, "iterations_per_s": it_per_s,
                 "roof_iterations_per_s": 1024 * 2.4e9 / K7_ISSUE_CYCLES_PER_ITER,
                 "frac": it_per_s * K7_ISSUE_CYCLES_PER_ITER / (1024 * 2.4e9), "pairs_per_s": k7_work[1] / (k7_ms * 1e-3) if k7_ms > 0 else 0.0,
                 "note": "vector-instruction issue cycles the visited iterations need / cycles the 1024 SIMDs offer in the launch; the "
                         "counts from a separate pass over the first %d items (counting slows K7 to %.3f ms per launch; that is not the time used)" % (n_count, k7_count_ms)}
    valu_roof["quadrant_entries_queued_per_launch"] = k7_work[4]   # (entry, quadrant) pairs the four quadrants' sub-lists hold
    valu_roof["quadrant_entries_per_iteration"] = k7_work[4] / k7_work[0] if k7_work[0] else 0.0   # of 4 slots
    # 64-byte gradient records K7 added to global memory (its atomics / 16): one per (tile, Gaussian, quadrant) that blended in
    # the default build, one per (tile, Gaussian, chunk) with -DED3_K7_LDS_TILE=1 (csrc/render_backward.hip)
    valu_roof["records_added_per_launch"] = k7_work[5]
    valu_roof["record_atomic_bytes_per_launch"] = k7_work[5] * 64
    # K6 against the same roof (ISA of render_forward_kernel<false,true>, tools/isa.sh, round 3: the per-pixel tests of an iteration
    # -- up to four Gaussians, one per quadrant -- cost 49 four-cycle vector instructions + 4 v_exp_f32, the blend 76 more (end of round 4;
    # round 3: 62 + 4 and 98);
    # iterations in which nothing blends pay the tests only and are not counted: a lower bound of the cycles needed)
    K6_TEST_CYCLES, K6_BLEND_CYCLES = 49 * 4 + 4 * 8, 76 * 4
    k6_cycles = k6_work[0] * (K6_TEST_CYCLES + K6_BLEND_CYCLES)
    valu_roof_k6 = {"visited_iterations_per_launch": k6_work[0], "blended_pairs_per_launch": k6_work[1],
                    "list_entries_staged_per_launch": k6_work[2], "entries_kept_by_tile_reject_per_launch": k6_work[3],
                    "pairs_per_iteration": k6_work[1] / k6_work[0] if k6_work[0] else 0.0,
                    "issue_cycles_per_visited_iteration": K6_TEST_CYCLES + K6_BLEND_CYCLES,
                    "issue_cycles_per_launch": k6_cycles,
                    "frac": k6_cycles / (k6_ms * 1e-3 * 1024 * 2.4e9) if k6_ms > 0 else 0.0,
                    "pairs_per_s": k6_work[1] / (k6_ms * 1e-3) if k6_ms > 0 else 0.0,
                    "note": "vector-instruction issue cycles K6's kept entries and visited iterations need / cycles the 1024 SIMDs offer in "
                            "the launch; counts from the same separate pass (counting slows K6 to %.3f ms per launch; not the time used)" % k6_count_ms}
    roof_k6 = {"bound": "hbm", "kernel": "render_forward_kernel<false,true> (K6)", "achieved": bytes_k6 / (k6_ms * 1e-3) / 1e9 if k6_ms > 0 else 0.0,
               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (bytes_k6 / (k6_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if k6_ms > 0 else 0.0,
               "traffic": pmc.get("render_forward_kernel<false,true>"), "traffic_source": traffic_source,
               "traffic_over_algorithmic": (pmc.get("render_forward_kernel<false,true>") / bytes_k6) if (pmc.get("render_forward_kernel<false,true>") and bytes_k6 > 0) else None,
               "algorithmic_bytes_per_launch": bytes_k6, "avg_launch_ms": k6_ms, "valu_roof": valu_roof_k6}
    roof_k7 = {"bound": "hbm", "kernel": "render_backward_kernel<false,true> (K7)", "achieved": ach,
               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
               "traffic": pmc.get("render_backward_kernel<false,true>"), "traffic_source": traffic_source,
               "traffic_over_algorithmic": (pmc.get("render_backward_kernel<false,true>") / bytes_k7) if (pmc.get("render_backward_kernel<false,true>") and bytes_k7 > 0) else None,
               "algorithmic_bytes_per_launch": bytes_k7, "avg_launch_ms": k7_ms, "launches": slot_n[1] or tab_n[1],
               "time_source": "timed region" if slot_n[1] else "instrumented pass of the same steps (hipEvent pairs on the launch stream)",
               "pairs_per_s_upper": mean(npairs_ub) / (k7_ms * 1e-3) if k7_ms > 0 else 0.0,
               "valu_roof": valu_roof,
               "note": "K7 is fp32-VALU-bound (arithmetic intensity >> machine balance, SURVEY 8d); the HBM fraction is "
                       "reported as defined there, next to the pair rate"}
    # HBM bytes per launch from the committed PMC passes beside each kernel's ALGORITHMIC bytes (SURVEY 8d's per-unit figures x the
    # units this run's launches processed): traffic well above the algorithmic bytes is wasted re-reading, the first thing to fix
    alg_bytes = {"render_forward_kernel<false,true> (K6)": bytes_k6, "render_backward_kernel<false,true> (K7)": bytes_k7,
                 "preprocess_backward_kernel (K8+K9)": 8.0 * wl["P"] + 755.0 * (mean(active_rows) if active_rows else wl["P"])}
    if dm:
        kept_b = 2 * 6 * 128 * 4.0   # relu(hid), relu(z_k): 6 x 128 floats per Gaussian and stage (the training forward writes them)
        alg_bytes[K[2][0]] = (364.0 + 236.0) * wl["P"]                        # 364 B in + 236 B out per Gaussian
        alg_bytes[K[3][0]] = (192.0 + 236.0 + 128.0 + 2 * 512.0) * rows_bwd    # sign masks + upstream rows + embedding + g_hid of both stages, active rows
        if 6 in K:
            alg_bytes[K[7][0]] = (4 / 5.0) * kept_b * rows_bwd + 1024.0 * rows_bwd # the four narrow heads' kept tiles + a of both stages
            alg_bytes[K[6][0]] = (1 / 5.0) * kept_b * rows_bwd + 1024.0 * rows_bwd
        else:   # one launch: the blocks of either kind read a of both stages
            alg_bytes[K[7][0]] = kept_b * rows_bwd + 2 * 1024.0 * rows_bwd
        alg_bytes[K[5][0]] = (2 * 512.0 + 128.0) * rows_bwd
    for nm_, kd in kernels.items():
        if not isinstance(kd, dict):
            continue
        tr = pmc.get(nm_.split(" ")[0])
        kd["traffic"] = tr
        if nm_ in alg_bytes:
            kd["algorithmic_bytes_per_launch"] = alg_bytes[nm_]
            kd["traffic_over_algorithmic"] = (tr / alg_bytes[nm_]) if (tr and alg_bytes[nm_] > 0) else None
    kernels["_traffic_source"] = traffic_source
    if dom >= 2 and dm:
        nm, mac, pr = K[dom]
        rows_dom = wl["P"] if dom == 2 else rows_bwd
        eq = tfl(mac, avg_ms[dom], rows_dom)
        peak = MFMA_BF16_PEAK_TFLOPS if pr > 1 else MFMA_F32_PEAK_TFLOPS
        roof = {"bound": "mfma", "kernel": nm, "achieved": eq * pr, "peak": peak, "unit": "TFLOP/s", "frac": eq * pr / peak,
                "traffic": pmc.get(nm.split(" ")[0]), "traffic_source": traffic_source,
                "traffic_over_algorithmic": kernels.get(nm, {}).get("traffic_over_algorithmic"),
                "traffic_note": "the training forward also WRITES the kept activations (6 x 128 floats per Gaussian and stage: 1.23 GB at "
                                "200k) for the backward's weight-gradient kernels; they are not part of the 600 B / Gaussian of algorithmic I/O",
                "algorithmic_flops_per_launch": 2.0 * mac * rows_dom, "executed_matrix_flops_per_launch": 2.0 * mac * rows_dom * pr,
                "fp32_equivalent_TFLOPs": eq, "avg_launch_ms": avg_ms[dom], "launches": slot_n[dom],
                "frac_of_measured_sustained_rate": (eq * pr / mfma_ceiling["bf16_32x32x16_TFLOPs"]) if (mfma_ceiling and pr > 1) else None,
                "note": ("dominant kernel of the step by time.  Algorithmic flops = 2 * %d MAC per Gaussian (fp32 multiplies); " % mac) +
                        ("each fp32 multiply runs as %d exact bf16 piece products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, "
                         "so `achieved` = executed piece-product flops / launch time against the dense bf16 MFMA peak; "
                         "fp32_equivalent_TFLOPs is the algorithmic rate (the f32-operand MFMA peak is 157.3)" % pr if pr > 1 else
                         "fp32 operands on v_mfma_f32_32x32x2_f32 (dense f32 MFMA peak 157.3 TFLOP/s)")}
    else:
        roof = roof_k7
    res = {
        "metric": "train iters/sec @200k Gaussians 1080p (fwd+bwd of render() incl. deformation MLP)",
        "value": world * a.steps / dt, "unit": "iters/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "step_ms_windows": [round(x, 4) for x in step_ms],
        # the HOST's time per step (enqueue + the wait for K1's instance count): a window of the GPU marks far above the median
        # with a host step of the same size beside it is a host stall (a descheduled thread on a shared box), not a slow kernel
        "host_ms_per_step": [round((host_t[k + 1] - host_t[k]) * 1e3, 3) for k in range(len(host_t) - 1)],
        "host_ms_in_render_and_backward": [[round(a_ * 1e3, 3), round(b_ * 1e3, 3)] for a_, b_ in step.host_split[:a.steps]],
        "device_mallocs_in_timed_region": n_malloc,   # hipMalloc calls of the caching allocator inside the timed region (each one stalls the stream)
        "step_ms": dict(percentiles(step_ms), source="rank 0, hipEvent marks every %d steps of the timed region on the launch stream; per-step = window / %d" % (MARK_EVERY, MARK_EVERY)),
        "frames_per_s": world * a.steps / dt,
        "ranks": {"world": world, "backend": backend_name,
                  "rccl_ranks": world if backend_name == "nccl" else 0,
                  "items_per_rank": [a.steps] * world, "gpus_shared": bool(shared_gpu),
                  "launched_by_bench": bool(os.environ.get("ED3DGS_BENCH_LAUNCHED")),
                  "step_ms_median_per_rank": rank_step_ms, "wall_ms_per_step_per_rank": rank_wall_ms,
                  "mean_num_rendered_per_rank": rank_num_rendered,
                  "rank0_items_timed": items_timed, "rank0_cameras_timed": cams_timed},
        "mfma_ceiling_measured": mfma_ceiling,
        "hbm_ceiling_measured_GBps": dict(ceiling, spec=HBM_PEAK_GBPS, note="1-GiB device-to-device copy and stream triad on this box, untimed section; `peak` in the rooflines stays the 8 TB/s spec"),
        "config": {"workload": wl["name"], "gaussians": wl["P"], "resolution": [wl["W"], wl["H"]],
                   "items": n_items, "parallelism": f"frames sharded i = rank mod {world}; 12-byte loss all-reduce/step" + ("; + bucketed gradient all-reduce (--dp-grads)" if a.dp_grads else ""),
                   "mean_num_rendered": mean(rsum), "mean_R_eff": mean(reff), "mean_sum_last_contributor": mean(npairs_ub)},
        "render_fps": None if a.train_only else world * a.steps / dt_r,
        "render_fps_note": "forward only, all outputs (coord+depth+normal), torch.no_grad, incl. deformation",
        "deform_kept_activations": {
            "forward_keeping_ms": tab_avg[2], "forward_not_keeping_ms": fwd_nokeep_ms, "cost_of_keeping_ms": None if fwd_nokeep_ms is None else tab_avg[2] - fwd_nokeep_ms,
            "kept_bytes_per_launch": 2 * 6 * 128 * 4 * wl["P"] if wl.get("deform", True) else 0,
            "rows_read_back": (mean(active_rows) if active_rows else None),
            "note": "the training forward writes relu(hid) and relu(z_k) of every Gaussian (6 x 128 floats per stage) for the weight-gradient "
                    "kernels, which read the ACTIVE rows only; re-forming z_k for the active rows instead is five of the forward's six 128-wide "
                    "products over rows_read_back rows (DESIGN.md section 9)"},
        "deform_backward_rows": {"active_mean": (mean(active_rows) if active_rows else None), "of": wl["P"],
                                 "walked": rows_bwd,
                                 "note": "Gaussians with a non-zero upstream gradient in the counted items (the rest are culled or "
                                         "behind the last contributor of every tile they touch: exact zeros); the default backward "
                                         "walks only these rows -- results identical to the dense walk (ED3DGS_DEFORM_DENSE_BWD=1)"},
        "roofline": roof,
        "roofline_tile_backward": roof_k7,
        "roofline_tile_forward": roof_k6,
        "kernels": kernels,
    }
    res["deform_mode"] = {"mode": mode, "note": {
        "exact_split": "default: fp32 operands split exactly into three bf16 pieces; the eight piece products above 2^-32 "
                       "accumulate in fp32 on the bf16 MFMA (forward, kept data gradient, head weight gradients; the small g_y . W3 "
                       "products and dW1 on the f32 MFMA).  Results at the f32-MFMA kernels' error level (tests/test_deform_parity_gpu.py)",
        "fp32_mfma": "every contraction on the f32-operand MFMA", "bf16x3": "REDUCED precision (two pieces, three products)"}[mode]}
    if f32m is not None:
        res["fp32_mfma_mode"] = f32m
    if b3 is not None:
        res["split_bf16_mode"] = b3
    if world == 1 and not a.no_cpu_baseline:
        log("cpu baseline (bounded sample, ~15-30 s)")
        try:
            res["cpu_baseline"] = cpu_baseline(wl)
        except Exception as ex:  # the baseline is a reported extra; a failure must not lose the GPU measurement
            res["cpu_baseline"] = {"value": None, "error": repr(ex)}
        if a.cpu_baseline_points:
            pts = {}
            for name in ("C1", "C2"):
                log("cpu baseline point", name)
                try:
                    pts[name] = cpu_baseline(WORKLOADS[name], full=(name == "C1"))
                except Exception as ex:
                    pts[name] = {"value": None, "error": repr(ex)}
            res["cpu_baseline_points"] = pts
    emit(res)


if __name__ == "__main__":
    main()
