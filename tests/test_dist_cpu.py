"""Multi-process (world_size 2, gloo, CPU) test of the frame-sharding harness: items partition across
ranks with no overlap, and the path's single collective sums [loss, psnr, count] over ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "e-d3dgs_amd"))
import torch
from ed3dgs_amd import dist as D
rank, world, local = D.init(backend="gloo")
n_items = 15 * 7
mine = D.shard_items(n_items, rank, world)
# every rank "renders" its items: loss_i = i, psnr_i = 2 i
loss = sum(float(i) for i in mine); psnr = sum(2.0 * i for i in mine)
t = D.allreduce_stats(loss, psnr, len(mine), "cpu")
D.barrier()
mx = D.max_over_ranks(float(rank + 1), "cpu")
h1 = D.allreduce_sum_async(torch.tensor([float(rank + 1)]))      # two collectives in flight, waited for later
h2 = D.allreduce_sum_async(torch.tensor([10.0 * (rank + 1)]))
asy = [float(h1.wait()[0]), float(h2.wait()[0]), float(h1.wait()[0])]
cams = sorted({D.item_of(i, 15, 7) for i in mine})
print(json.dumps(dict(rank=rank, world=world, mine=mine, total=t.tolist(), mx=mx, n_pairs=len(cams), asy=asy)))
'''


def test_two_rank_sharding_and_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        outs.append(o.strip().splitlines()[-1])
    import json
    res = sorted((json.loads(o) for o in outs), key=lambda d: d["rank"])
    n = 15 * 7
    a, b = set(res[0]["mine"]), set(res[1]["mine"])
    assert a.isdisjoint(b) and a | b == set(range(n))                    # partition, no item rendered twice
    assert abs(len(a) - len(b)) <= 1                                     # balanced
    want = [sum(range(n)), 2.0 * sum(range(n)), float(n)]
    for r in res:
        assert r["world"] == 2 and r["total"] == want and r["mx"] == 2.0 and r["asy"] == [3.0, 30.0, 3.0]
    assert res[0]["n_pairs"] == len(a)                                   # (camera, frame) pairs are distinct


GRAD_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "e-d3dgs_amd"))
import torch
from ed3dgs_amd import dist as D
rank, world, local = D.init(backend="gloo")
torch.manual_seed(0)
ps = [torch.nn.Parameter(torch.zeros(n)) for n in (5, 70000, 3, 1 << 18)] + [torch.nn.Parameter(torch.zeros(4), requires_grad=False)]
for i, p in enumerate(ps[:4]):
    p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
if rank == 1:
    ps[2].grad = None            # a rank without a gradient for a parameter still takes part (zeros)
D.allreduce_gradients_(ps, bucket_bytes=300_000, average=True)
out = [float(p.grad[0]) for p in ps[:4]] + [float(p.grad.min()) == float(p.grad.max()) for p in ps[:4]]
print(json.dumps(dict(rank=rank, out=out, frozen=ps[4].grad is None)))
'''


def test_two_rank_gradient_allreduce(tmp_path):
    """bucketed mean of the ranks' gradients (several buckets, a missing gradient, a frozen one)."""
    script = tmp_path / "gworker.py"
    script.write_text(GRAD_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    import json
    res = []
    for p in procs:
        o, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        res.append(json.loads(o.strip().splitlines()[-1]))
    for r in res:
        # mean over ranks of (rank + 1) * (i + 1): 1.5 (i + 1); parameter 2 had a gradient on rank 0 only: 3 / 2
        assert r["out"][:4] == [1.5, 3.0, 1.5, 6.0] and all(r["out"][4:]) and r["frozen"]


HOOK_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "e-d3dgs_amd"))
import torch
from ed3dgs_amd import dist as D
rank, world, local = D.init(backend="gloo")
ps = [torch.nn.Parameter(torch.full((n,), 0.5)) for n in (5, 70000, 3, 1 << 16)]
red = D.BucketedGradReducer(ps, bucket_bytes=200_000, average=True)
outs = []
for it in range(2):                       # two steps: the reducer re-arms itself
    for p in ps:
        p.grad = None
    # parameter 2 takes no part in rank 1's graph; parameter 0's gradient lands LAST (it is used first)
    x = (ps[0] * (rank + 1)).sum()
    y = (ps[1] * (2 * (rank + 1))).sum() + (ps[3] * (4 * (rank + 1))).sum()
    if rank == 0:
        y = y + (ps[2] * 3.0).sum()
    (x + y).backward()
    early = red.issued_in_backward
    red.finish()
    outs.append([float(p.grad[0]) for p in ps] + [all(float(p.grad.min()) == float(p.grad.max()) for p in ps)])
print(json.dumps(dict(rank=rank, outs=outs, buckets=len(red.buckets), early=early)))
D.destroy()
'''


def test_two_rank_hooked_gradient_reducer(tmp_path):
    """BucketedGradReducer: buckets issued from autograd hooks as their last gradient lands; same means as the after-the-fact
    all-reduce, a missing gradient counts as zeros, and a second step works."""
    script = tmp_path / "hworker.py"
    script.write_text(HOOK_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    import json
    res = []
    for p in procs:
        o, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        res.append(json.loads(o.strip().splitlines()[-1]))
    for r in res:
        assert r["buckets"] >= 3
        for out in r["outs"]:
            # mean over ranks: p0 (1 + 2) / 2, p1 (2 + 4) / 2, p2 (3 + 0) / 2, p3 (4 + 8) / 2
            assert out[:4] == [1.5, 3.0, 1.5, 6.0] and out[4]


def test_hooked_reducer_refuses_a_second_backward_before_finish():
    """gradient accumulation over two backward() calls used to be dropped silently (the bucket had been issued with the
    first gradient and finish() overwrote .grad with it); now the hook raises."""
    import pytest
    import torch
    sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
    from ed3dgs_amd import dist as D
    ps = [torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(3))]
    red = D.BucketedGradReducer(ps, bucket_bytes=8, average=True)
    (ps[0].sum() + 2 * ps[1].sum()).backward()
    with pytest.raises(RuntimeError, match="second backward"):
        (ps[0].sum() + 2 * ps[1].sum()).backward()
    red.remove()
    red2 = D.BucketedGradReducer(ps, bucket_bytes=8, average=True)      # one backward per finish(): fine, twice in a row
    for _ in range(2):
        for p in ps:
            p.grad = None
        (ps[0].sum() + 2 * ps[1].sum()).backward()
        red2.finish()
        assert ps[0].grad.tolist() == [1.0] * 4 and ps[1].grad.tolist() == [2.0] * 3
    red2.remove()


def test_strided_visit_covers_all_cameras():
    sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
    from ed3dgs_amd import dist as D
    for world in (1, 2, 4, 8):
        for rank in range(world):
            mine = D.shard_items(400, rank, world)
            seq = [D.strided_item(mine, k) for k in range(20)]
            assert len(set(seq)) == 20                                           # no item twice in a 20-step run
            assert {D.item_of(i, 8, 50)[0] for i in seq} == set(range(8)), (world, rank)   # every camera
            assert sorted(D.strided_item(mine, k) for k in range(len(mine))) == mine  # a permutation of the shard


def test_visit_order_sees_every_camera_in_any_run_of_as_many_steps():
    """bench.py's item schedule (dist.visit_order): a permutation of the shard, cameras round-robin -- C3 (8 x 50), C4 (15 x 300:
    a single stride over the flat shard left two cameras out of 20 steps on four of eight ranks), C5 (8 x 150)."""
    sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
    from ed3dgs_amd import dist as D
    for cams, frames in ((8, 50), (15, 300), (8, 150)):
        for world in (1, 2, 4, 8):
            for rank in range(world):
                mine = D.shard_items(cams * frames, rank, world)
                order = D.visit_order(mine, cams, frames)
                assert sorted(order) == mine
                for start in (0, 7, 100):
                    run = [D.visit_item(mine, start + k, cams, frames) for k in range(cams)]
                    assert {D.item_of(i, cams, frames)[0] for i in run} == set(range(cams)), (cams, world, rank, start)
                assert len({D.visit_item(mine, k, cams, frames) for k in range(20)}) == 20


def test_item_mapping_is_bijective():
    sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
    from ed3dgs_amd import dist as D
    seen = {D.item_of(i, 8, 50) for i in range(400)}
    assert len(seen) == 400 and all(0 <= c < 8 and 0 <= f < 50 for c, f in seen)
    assert D.shard_items(10, 3, 4) == [3, 7] and D.shard_items(3, 5, 8) == []
