"""tools/gen_golden.py -- generates tests/golden/deform_*.npz by importing the REFERENCE's scene/deformation.py.

Runs only in the authoring container (needs /root/reference).  Two harness-side shims, no edits to the reference:
  * a stub `tkinter` module providing `W` (scene/deformation.py:5 has a stray `from tkinter import W`);
  * `torch.Tensor.cuda` -> identity, because get_temporal_embed hard-codes `.cuda()` (:61) and this host has no GPU.
The module is loaded by file path (importing the `scene` package would pull dataset readers).
Each fixture holds: the constructor/flag configuration, the full state-dict, the inputs, `forward`'s outputs
(final + coarse tuples) and torch-autograd gradients of a fixed random linear functional of all outputs with
respect to every parameter, the Gaussian embedding and the base tensors.  Only data is written -- no reference text.
"""
import importlib.util
import os
import sys
import types
import zlib

import numpy as np
import torch

REF = "/root/reference/scene/deformation.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_reference():
    stub = types.ModuleType("tkinter")
    stub.W = "w"
    sys.modules.setdefault("tkinter", stub)
    torch.Tensor.cuda = lambda self, *a, **k: self
    spec = importlib.util.spec_from_file_location("ref_deformation", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class A:
    pass


def make_args(**kw):
    a = A()
    d = dict(temporal_embedding_dim=256, gaussian_embedding_dim=32, c2f_temporal_iter=20000, zero_temporal=False,
             no_ds=False, no_dr=False, no_do=True, no_dc=False, use_coarse_temporal_embedding=False,
             no_c2f_temporal_embedding=False, no_coarse_deform=False, no_fine_deform=False, use_anneal=True,
             deform_from_iter=0)
    d.update(kw)
    a.__dict__.update(d)
    return a


class PC:
    def __init__(self, emb):
        self.get_embedding = emb


CASES = [
    # name, W, D, min_emb, max_emb, P, iter, cam_no, time, nde_c, nde_f, arg overrides
    ("nersemble_w128", 128, 1, 30, 150, 48, 20000, None, 0.37, 30, 30,
     dict(no_do=False, use_coarse_temporal_embedding=True, c2f_temporal_iter=10000, deform_from_iter=5000)),
    ("default_w64_it500", 64, 1, 30, 150, 40, 500, 3, 0.81, 30, 30, dict()),
    ("d0_w32_it7000", 32, 0, 5, 25, 33, 7000, None, 0.05, 5, 5, dict(no_dr=True, c2f_temporal_iter=10000, deform_from_iter=5000)),
    ("d2_w32_noanneal", 32, 2, 5, 25, 20, 100, 1, 0.5, 7, 9, dict(use_anneal=False, no_do=False, no_ds=True)),
    ("d3_w64_it2500", 64, 3, 5, 25, 37, 2500, 3, 0.4, 6, 8, dict(no_do=False, deform_from_iter=2000)),
    ("w32_nocoarse", 32, 1, 5, 25, 17, 3000, None, 0.999, 5, 5, dict(no_coarse_deform=True, no_dc=True)),
    ("w32_nofine_noc2f", 32, 1, 5, 25, 17, 3000, 2, 0.25, 5, 5, dict(no_fine_deform=True, no_c2f_temporal_embedding=True, no_do=False)),
    ("w32_reflect_time", 32, 1, 5, 25, 9, 12000, 0, 1.2, 5, 5, dict(no_do=False, temporal_embedding_dim=64)),
    ("w32_it0_zero_temporal", 32, 1, 5, 25, 9, 0, None, 0.6, 5, 5, dict(zero_temporal=True)),
    # the layer-by-layer path of the HIP side (csrc/deform_deep.hip): width 256, and a 64-wide Gaussian embedding at width 128
    ("w256_d1_it4000", 256, 1, 5, 25, 29, 4000, 2, 0.63, 6, 7, dict(no_do=False)),
    ("w128_e64_d2", 128, 2, 5, 25, 23, 9000, None, 0.15, 8, 5, dict(no_do=False, gaussian_embedding_dim=64, temporal_embedding_dim=128)),
]


def run_case(mod, name, W, D, mn, mx, P, it, cam_no, time, nde_c, nde_f, over):
    seed = zlib.crc32(name.encode()) % (2 ** 31)
    torch.manual_seed(seed)
    np.random.seed(seed)
    args = make_args(**over)
    net = mod.deform_network(D=D, W=W, min_embeddings=mn, max_embeddings=mx, num_frames=300, args=args)
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        # make deformations visible (the reference initialises the table at ~6e-4) and offsets non-trivial
        if not args.zero_temporal:
            net.weight.mul_(100.0)
        net.offsets.copy_(torch.randn(30, 1, generator=g) * 0.01)
        net.offsets[5:] = 0
    E = args.gaussian_embedding_dim
    xyz = torch.randn(P, 3, generator=g).requires_grad_(True)
    scales = (torch.randn(P, 3, generator=g) * 0.3 - 4).requires_grad_(True)
    rot = torch.randn(P, 4, generator=g).requires_grad_(True)
    opacity = torch.randn(P, 1, generator=g).requires_grad_(True)
    sh = (torch.randn(P, 16, 3, generator=g) * 0.5).requires_grad_(True)
    emb = (torch.randn(P, E, generator=g) * 0.1).requires_grad_(True)
    time_t = torch.tensor(time).repeat(P, 1)
    outs = net(xyz, scales, rot, opacity, time_t, cam_no, PC(emb), None, sh, iter=it, num_down_emb_c=nde_c,
               num_down_emb_f=nde_f)
    final = outs[:5]
    sub = outs[5][0]
    ws = [torch.randn(t.shape, generator=g) for t in list(final) + list(sub)]
    loss = sum((t * w).sum() for t, w in zip(list(final) + list(sub), ws))
    params = dict(net.named_parameters())
    leaves = [xyz, scales, rot, opacity, sh, emb]
    grads = torch.autograd.grad(loss, list(params.values()) + leaves, allow_unused=True)
    rec = {}
    for k, v in vars(args).items():
        rec["arg_" + k] = np.asarray(v)
    rec.update(cfg_W=np.asarray(W), cfg_D=np.asarray(D), cfg_min=np.asarray(mn), cfg_max=np.asarray(mx),
               cfg_iter=np.asarray(it), cfg_cam_no=np.asarray(-1 if cam_no is None else cam_no),
               cfg_time=np.asarray(time, np.float32), cfg_nde_c=np.asarray(nde_c), cfg_nde_f=np.asarray(nde_f))
    for k, v in net.state_dict().items():
        rec["sd_" + k] = v.detach().numpy()
    for n, t in zip(("xyz", "scales", "rot", "opacity", "sh", "emb"), leaves):
        rec["in_" + n] = t.detach().numpy()
    for n, t in zip(("xyz", "scales", "rot", "opacity", "sh"), final):
        rec["out_" + n] = t.detach().numpy()
    for n, t in zip(("xyz", "scales", "rot", "opacity", "sh"), sub):
        rec["sub_" + n] = t.detach().numpy()
    for i, w in enumerate(ws):
        rec[f"lossw_{i}"] = w.numpy()
    for (k, _), gten in zip(list(params.items()) + [("in_" + n, None) for n in ("xyz", "scales", "rot", "opacity", "sh", "emb")], grads):
        key = "gsd_" + k if not k.startswith("in_") else "g" + k
        rec[key] = np.zeros(0, np.float32) if gten is None else gten.detach().numpy()
    path = os.path.join(OUT, f"deform_{name}.npz")
    np.savez_compressed(path, **rec)
    return path


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    mod = load_reference()
    only = set(sys.argv[1:])   # optional: names of the cases to (re)generate; default all
    for c in CASES:
        if only and c[0] not in only:
            continue
        p = run_case(mod, *c)
        print(p, os.path.getsize(p) // 1024, "KiB")
