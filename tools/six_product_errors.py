"""Error table of the deformation MLP's multiply modes against the float64 restatement (the six-product option).

Run once per build / mode on the GPU box and compare the JSON lines:
  python tools/six_product_errors.py eight                                        # default build, 8 piece products
  ED3DGS_DEFORM_FP32_MFMA=1 python tools/six_product_errors.py fp32_mfma            # f32-operand MFMA kernels
  ED3DGS_LIB_PATH=.../variants/libed3dgs_hip_six.so python tools/six_product_errors.py six   # tools/ab_build.sh six -DED3_NP3_SMAX=2
Cases: every reference-generated golden of width <= 128 / depth <= 1 (inputs and state dict from the fixture, reference values
replaced by the float64 restatement) and the 65,836-row case of tests/test_deform_parity_gpu.py.  Per tensor: the largest
absolute error relative to the tensor's largest element.  Also times forward + backward of the 65,836-row case."""
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "e-d3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from oracle import deformation_ref as R   # noqa: E402
from oracle import deformation_torch as T   # noqa: E402
from scene.deformation import deform_network   # noqa: E402

NAMES = ("xyz", "scales", "rot", "opacity", "sh")


class PC:
    def __init__(self, e):
        self.get_embedding = e


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64).reshape(a.shape)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def run(net, a, D, mx, base, t, cam, it, ndc, ndf, seed):
    g = torch.Generator().manual_seed(seed)
    sd64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in net.state_dict().items()}
    b64 = {k: v.double().requires_grad_(True) for k, v in base.items()}
    mg = []
    fin, sub = T.forward(sd64, a, D, mx, b64["xyz"], b64["scales"], b64["rot"], b64["opacity"], b64["sh"], b64["emb"], t, cam, it,
                         ndc, ndf, margin_out=mg)
    off_kink = (mg[0] > 1e-6).float()
    ws = [torch.randn(*x.shape, generator=g) * off_kink.reshape(-1, *([1] * (x.dim() - 1))) for x in list(fin) + list(sub)]
    sum((x * w.double()).sum() for x, w in zip(list(fin) + list(sub), ws)).backward()
    net = net.cuda()
    bg = {k: v.cuda().requires_grad_(True) for k, v in base.items()}
    outs = net(bg["xyz"], bg["scales"], bg["rot"], bg["opacity"], t, cam, PC(bg["emb"]), None, bg["sh"], iter=it, num_down_emb_c=ndc,
               num_down_emb_f=ndf)
    hf, hs = outs[:5], outs[5][0]
    sum((x * w.cuda().reshape(x.shape)).sum() for x, w in zip(list(hf) + list(hs), ws)).backward()
    errs = {}
    for n, x, y in zip(NAMES, hf, fin):
        errs["out_" + n] = rel(x.detach().cpu().numpy(), y.detach().numpy())
    for n, x, y in zip(NAMES, hs, sub):
        errs["sub_" + n] = rel(x.detach().cpu().numpy(), y.detach().numpy())
    for name, p in net.named_parameters():
        ref = sd64[name].grad
        if ref is None or float(ref.abs().max()) == 0 or p.grad is None:
            continue
        errs["g_" + name] = rel(p.grad.cpu().numpy(), ref.numpy())
    for k in base:
        if float(b64[k].grad.abs().max()) > 0:
            errs["gin_" + k] = rel(bg[k].grad.cpu().numpy(), b64[k].grad.numpy())
    return errs


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else "run"
    out = {"label": label, "cases": {}}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "deform_*.npz"))):
        z = np.load(path)
        if int(z["cfg_D"]) > 1:
            continue
        a = R.Args()
        for k in z.files:
            if k.startswith("arg_"):
                setattr(a, k[4:], z[k].item())
        net = deform_network(D=int(z["cfg_D"]), W=int(z["cfg_W"]), min_embeddings=int(z["cfg_min"]), max_embeddings=int(z["cfg_max"]),
                             num_frames=300, args=a)
        net.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")})
        base = {n: torch.from_numpy(z["in_" + n]) for n in ("xyz", "scales", "rot", "opacity", "sh", "emb")}
        cam = int(z["cfg_cam_no"]); cam = None if cam < 0 else cam
        out["cases"][os.path.basename(path)[7:-4]] = run(net, a, int(z["cfg_D"]), int(z["cfg_max"]), base, float(z["cfg_time"]), cam,
                                                          int(z["cfg_iter"]), int(z["cfg_nde_c"]), int(z["cfg_nde_f"]), 77)
    # the 65,836-row case
    a = R.Args(no_do=False, use_coarse_temporal_embedding=True, c2f_temporal_iter=10000, deform_from_iter=5000)
    torch.manual_seed(5)
    net = deform_network(D=1, W=128, min_embeddings=30, max_embeddings=150, num_frames=300, args=a)
    with torch.no_grad():
        net.weight.mul_(100.0)
    g = torch.Generator().manual_seed(6)
    P = 65836
    mk = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    base = dict(xyz=mk(P, 3), scales=mk(P, 3, sc=0.3) - 4, rot=mk(P, 4), opacity=mk(P, 1), sh=mk(P, 16, 3, sc=0.5), emb=mk(P, 32, sc=0.1))
    out["cases"]["rows65836_w128"] = run(net, a, 1, 150, base, 0.43, None, 20000, 30, 30, 78)
    # time: forward + backward of that case
    bg = {k: v.cuda().requires_grad_(True) for k, v in base.items()}
    net = net.cuda()
    def step():
        o = net(bg["xyz"], bg["scales"], bg["rot"], bg["opacity"], 0.43, None, PC(bg["emb"]), None, bg["sh"], iter=20000)
        sum(x.sum() for x in o[:5]).backward()
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    out["ms_fwd_bwd_65836"] = (time.perf_counter() - t0) / 20 * 1e3
    worst = {}
    for c, e in out["cases"].items():
        for k, v in e.items():
            kind = "values" if k.startswith(("out_", "sub_")) else "gradients"
            worst[kind] = max(worst.get(kind, 0.0), v)
    out["worst"] = worst
    print(json.dumps(out))


if __name__ == "__main__":
    main()
