"""Round 4 probe: what the deformation forward (keeping activations, C3 size) costs per group of heads -- input to the question
whether the appearance heads (opacity, SH) could run beside the level-1 sort + binning instead of in front of them.
usage (GPU box, repo root): python tools/r04_head_split_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "e-d3dgs_amd"))
import torch
from ed3dgs_amd import synthetic as S
from ed3dgs_amd.model import SynthGaussianModel, default_hyper

dev = "cuda:0"
scene = S.make_scene(200_000, seed=0)
cases = {"all heads": {}, "pos+scale+rot (no_do,no_dc)": dict(no_do=True, no_dc=True),
         "pos+opacity+SH (no_ds,no_dr)": dict(no_ds=True, no_dr=True), "pos only": dict(no_ds=True, no_dr=True, no_do=True, no_dc=True),
         "pos+SH": dict(no_ds=True, no_dr=True, no_do=True), "pos+opacity": dict(no_ds=True, no_dr=True, no_dc=True)}
for name, over in cases.items():
    m = SynthGaussianModel(scene, args=default_hyper(**over), deform_seed=2, device=dev)

    def fwd():
        return m._deformation(m._xyz, m._scaling, m._rotation, m._opacity, 0.37, None, m, None, m._features_dc, iter=20000,
                              num_down_emb_c=30, num_down_emb_f=30, sh_coefs_rest=m._features_rest, activated=(None,))
    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fwd(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print("%-34s forward call (prep + MLP, keeping): median %.4f ms  min %.4f" % (name, ts[len(ts) // 2], ts[0]), flush=True)
    del m
