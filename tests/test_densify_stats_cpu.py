"""Sharded densification statistics (second half; scene/gaussian_model.py:452-518, train.py:404-421):
two gloo ranks accumulate the statistics of DIFFERENT views, combine them, and must take identical clone / split / prune
decisions and end with bit-identical post-densify tensors -- equal to what one process seeing all the views decides."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))

COMMON = r'''
import torch
P, VIEWS = 5000, 12
def model_tensors():
    g = torch.Generator().manual_seed(0)
    return dict(xyz=torch.randn(P, 3, generator=g), features_dc=torch.randn(P, 1, 3, generator=g),
                features_rest=torch.randn(P, 15, 3, generator=g), opacity=torch.randn(P, 1, generator=g) * 2,
                scaling=torch.randn(P, 3, generator=g) * 0.7 - 3.0, rotation=torch.randn(P, 4, generator=g),
                embedding=torch.randn(P, 32, generator=g), tongue_class=(torch.rand(P, 1, generator=g) > 0.5).float())
def view(i):
    g = torch.Generator().manual_seed(100 + i)
    grad = torch.randn(P, 3, generator=g) * 2e-4
    grad[:, 2] = grad[:, :2].abs().sum(1)
    radii = (torch.rand(P, generator=g) * 40 - 8).clamp(min=0).floor().int()
    return grad, radii > 0, radii
MAX_GRAD, MIN_OPACITY, EXTENT, MAX_SCREEN = 2.2e-4, 0.05, 6.0, 20
'''

WORKER = COMMON + r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "e-d3dgs_amd"))
from ed3dgs_amd import dist as D
from ed3dgs_amd import densify_stats as DS
rank, world, local = D.init(backend="gloo")
t = model_tensors()
st = DS.DensificationStats(P, "cpu")
for i in D.shard_items(VIEWS, rank, world):
    st.add(*view(i))
local_denom = float(st.denom.sum())
st.all_reduce_()
cm, sm, pm = DS.decide(st, torch.exp(t["scaling"]), torch.sigmoid(t["opacity"]), MAX_GRAD, MIN_OPACITY, EXTENT, MAX_SCREEN)
new = DS.densify_tensors(t, st, MAX_GRAD, EXTENT, iteration=3100)
try:
    st.add(*view(0)); stale = False
except RuntimeError:
    stale = True
print(json.dumps(dict(rank=rank, local_denom=local_denom, denom=float(st.denom.sum()), masks=DS.tensor_hash(cm, sm, pm),
                      counts=[int(cm.sum()), int(sm.sum()), int(pm.sum())], n_new=int(new["xyz"].shape[0]),
                      tensors=DS.tensor_hash(*[new[k] for k in sorted(new)]), stats=DS.tensor_hash(st.xyz_gradient_accum, st.abs_gradient_accum, st.denom, st.max_radii2D),
                      stale_guard=stale)))
'''


def test_two_ranks_take_identical_densify_decisions(tmp_path):
    script = tmp_path / "dworker.py"
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    res = []
    for p in procs:
        o, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
        res.append(json.loads(o.strip().splitlines()[-1]))
    a, b = sorted(res, key=lambda d: d["rank"])
    assert a["local_denom"] != b["local_denom"]                                  # the ranks really saw different views
    for k in ("denom", "masks", "counts", "n_new", "tensors", "stats"):
        assert a[k] == b[k], k                                                   # identical statistics, decisions, tensors
    assert a["stale_guard"] and b["stale_guard"]
    assert min(a["counts"]) > 0                                                  # every branch exercised

    # one process over the union of the views (what the single-GPU reference accumulates)
    ns = {}
    exec(COMMON, ns)
    from ed3dgs_amd import densify_stats as DS
    st = DS.DensificationStats(ns["P"], "cpu")
    for i in range(ns["VIEWS"]):
        st.add(*ns["view"](i))
    assert float(st.denom.sum()) == a["denom"]
    t = ns["model_tensors"]()
    cm, sm, pm = DS.decide(st, torch.exp(t["scaling"]), torch.sigmoid(t["opacity"]), ns["MAX_GRAD"], ns["MIN_OPACITY"],
                           ns["EXTENT"], ns["MAX_SCREEN"])
    assert [int(cm.sum()), int(sm.sum()), int(pm.sum())] == a["counts"]
    new = DS.densify_tensors(t, st, ns["MAX_GRAD"], ns["EXTENT"], iteration=3100)
    assert new["xyz"].shape[0] == a["n_new"] == ns["P"] + a["counts"][0] + a["counts"][1]   # +clones, +2 per split, -split


BATCH_WORKER = COMMON + r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "e-d3dgs_amd"))
from ed3dgs_amd import dist as D
from ed3dgs_amd import densify_stats as DS
rank, world, local = D.init(backend="gloo")
st = DS.DensificationStats(P, "cpu")
for step in range(VIEWS // world):                 # every step: the ranks render DIFFERENT views, one batch of `world`
    st.add_batched_step(*view(step * world + rank))
try:
    st.all_reduce_(); guard = False
except RuntimeError:
    guard = True
print(json.dumps(dict(rank=rank, stats=DS.tensor_hash(st.xyz_gradient_accum, st.abs_gradient_accum, st.denom, st.max_radii2D),
                      denom=float(st.denom.sum()), accum=float(st.xyz_gradient_accum.double().sum()), guard=guard)))
D.destroy()
'''


def test_batched_step_matches_the_reference_batch_semantics(tmp_path):
    """N ranks x one optimizer step = the reference's batch_size N (train.py:166-190, 346-348, 404-407: gradients
    summed over the batch's views, visibility OR-ed, radii maxed, ONE add_densification_stats per step).  ADVICE r3: the
    reference's batch loss is a MEAN over the stacked views (train.py:195-197), so the gradient each view contributes at
    train.py:346-348 is 1 / batch_size of the gradient of that view's own mean loss -- which is what a data-parallel rank
    backpropagates (`view(i)[0]` below).  add_batched_step's default therefore scales the summed gradient by 1 / world."""
    script = tmp_path / "bworker.py"
    script.write_text(BATCH_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    res = []
    for p in procs:
        o, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
        res.append(json.loads(o.strip().splitlines()[-1]))
    a, b = sorted(res, key=lambda d: d["rank"])
    assert a["stats"] == b["stats"] and a["guard"] and b["guard"]
    # the reference's own statements for batch_size = 2, restated: per step sum the views' grads, any() the filters, max the radii
    ns = {}
    exec(COMMON, ns)
    P = ns["P"]
    accum, denom, maxr = torch.zeros(P, 1), torch.zeros(P, 1), torch.zeros(P)
    per_view = torch.zeros(P, 1)
    for step in range(ns["VIEWS"] // 2):
        views = [ns["view"](2 * step + r) for r in range(2)]
        g = torch.zeros_like(views[0][0])
        for v in views:
            g = g + v[0] / 2                                               # train.py:346-348 under the batch-mean loss of :195-197
        vis = torch.stack([v[1] for v in views]).any(dim=0)                  # train.py:190
        rad = torch.stack([v[2] for v in views]).max(dim=0).values.float()   # train.py:189
        maxr[vis] = torch.max(maxr[vis], rad[vis])                           # train.py:406
        accum[vis] += torch.norm(g[vis, :2], dim=-1, keepdim=True)           # scene/gaussian_model.py:517
        denom[vis] += 1                                                      # :518
        for v in views:
            per_view[v[1]] += torch.norm(v[0][v[1], :2], dim=-1, keepdim=True) / 2
    from ed3dgs_amd import densify_stats as DS
    assert float(denom.sum()) == a["denom"]
    assert abs(float(accum.double().sum()) - a["accum"]) <= 1e-6 * a["accum"]
    assert float(per_view.sum()) > 1.1 * a["accum"]          # and it is NOT the per-view accumulation (add() + all_reduce_())


def test_batched_step_loss_scale_argument():
    """world = 1: the default scale is 1; loss_scale = 0.25 stands for a caller whose per-view losses are 4x the batch's share."""
    from ed3dgs_amd import densify_stats as DS
    g = torch.tensor([[3.0, 4.0, 7.0], [0.3, 0.4, 0.7]])
    vis = torch.tensor([True, True])
    rad = torch.tensor([3, 4], dtype=torch.int32)
    a, b = DS.DensificationStats(2, "cpu"), DS.DensificationStats(2, "cpu")
    a.add_batched_step(g, vis, rad)
    b.add_batched_step(g, vis, rad, loss_scale=0.25)
    assert torch.allclose(a.xyz_gradient_accum.reshape(-1), torch.tensor([5.0, 0.5]))
    assert torch.allclose(b.xyz_gradient_accum.reshape(-1), torch.tensor([1.25, 0.125]))
    assert torch.allclose(b.abs_gradient_accum.reshape(-1), torch.tensor([1.75, 0.175]))


def test_decisions_follow_the_reference_formulas():
    from ed3dgs_amd import densify_stats as DS
    st = DS.DensificationStats(4, "cpu")
    grad = torch.tensor([[3e-4, 4e-4, 7e-4], [3e-4, 4e-4, 7e-4], [1e-5, 0, 1e-5], [9.0, 9.0, 18.0]])
    vis = torch.tensor([True, True, True, False])
    st.add(grad, vis, torch.tensor([5, 30, 2, 99], dtype=torch.int32))
    st.add(grad * 0, vis, torch.tensor([7, 1, 2, 99], dtype=torch.int32))
    assert st.denom.reshape(-1).tolist() == [2, 2, 2, 0] and st.max_radii2D.tolist() == [7, 30, 2, 0]
    g = st.mean_grads().reshape(-1)
    assert torch.allclose(g, torch.tensor([2.5e-4, 2.5e-4, 5e-6, 0.0]))          # never-visible row: NaN -> 0
    assert torch.allclose(st.abs_gradient_accum.reshape(-1), torch.tensor([7e-4, 7e-4, 1e-5, 0.0]))
    scaling = torch.tensor([[0.01, 0.01, 0.01], [0.5, 0.01, 0.01], [0.01, 0.01, 0.01], [0.01, 0.01, 0.01]])
    opacity = torch.tensor([[0.9], [0.9], [0.001], [0.9]])
    cm, sm, pm = DS.decide(st, scaling, opacity, max_grad=2e-4, min_opacity=0.005, extent=5.0, max_screen_size=20)
    assert cm.tolist() == [True, False, False, False]                            # hot and small -> clone
    assert sm.tolist() == [False, True, False, False]                            # hot and large -> split
    assert pm.tolist() == [False, True, True, False]                             # radius 30 > 20; opacity < min
    cm2, sm2, pm2 = DS.decide(st, scaling, opacity, 2e-4, 0.005, 5.0, None)
    assert pm2.tolist() == [False, False, True, False]                           # no size threshold before the first reset
