"""The multi-GPU module's collectives through RCCL on the one GPU a test box has.

RCCL refuses two ranks on one device, so this is a ONE-rank communicator with ED3DGS_DIST_COLLECTIVES_AT_WORLD_1=1: every
collective ed3dgs_amd.dist issues at world > 1 is then issued here too, on DEVICE tensors, through backend "nccl" (= RCCL):
communicator creation on the selected device, the stats all-reduce in flight across a step boundary and waited for by the
stream, barrier / max-over-ranks as bench.py brackets its timed region, the bucketed gradient all-reduce issued from autograd
hooks during backward(), and destroy_process_group().  What it cannot show is a second rank; the 2-rank gloo tests
(tests/test_dist_cpu.py) hold the arithmetic, this holds the RCCL plumbing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "e-d3dgs_amd"))
import torch
from ed3dgs_amd import dist as D
torch.cuda.set_device(0)
rank, world, local = D.init()
import torch.distributed as dist
backend = dist.get_backend()
dev = "cuda:0"
t = D.allreduce_stats(1.5, 2.5, 3, dev)
D.barrier()
mx = D.max_over_ranks(7.0, dev)
per = D.gather_per_rank(4.25, dev)
# the stats vector in flight across a "step": the next step's kernels are enqueued before the wait
a = torch.full((3,), 2.0, device=dev); b = torch.full((3,), 5.0, device=dev)
h1 = D.allreduce_sum_async(a); x = torch.randn(1 << 20, device=dev).square().sum(); h2 = D.allreduce_sum_async(b)
h1.wait(); h2.wait()
asy = [float(a[0]), float(b[0])]
# bucketed gradient reducer: buckets issued from the hooks while backward() is still running, several steps
ps = [torch.nn.Parameter(torch.full((n,), 0.5, device=dev)) for n in (5, 70000, 3, 1 << 16)]
red = D.BucketedGradReducer(ps, bucket_bytes=200_000, average=True)
outs = []
for it in range(3):
    for p in ps:
        p.grad = None
    ((ps[0] * 1.0).sum() + (ps[1] * 2.0).sum() + (ps[3] * 4.0).sum()).backward()      # parameter 2 gets no gradient: zeros
    early = red.issued_in_backward
    red.finish()
    outs.append([float(p.grad[0]) for p in ps])
red.remove()
ps2 = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (9, 100000)]
for i, p in enumerate(ps2):
    p.grad = torch.full_like(p, float(i + 1))
D.allreduce_gradients_(ps2, bucket_bytes=100_000, average=True)
torch.cuda.synchronize()
print(json.dumps(dict(backend=backend, world=world, total=t.tolist(), mx=mx, per=per, asy=asy, outs=outs, early=early,
                      buckets=len(red.buckets), g2=[float(p.grad[0]) for p in ps2])))
D.destroy()
'''


@pytest.mark.gpu
def test_collectives_run_through_rccl_on_device_tensors(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               ED3DGS_DIST_COLLECTIVES_AT_WORLD_1="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ED3DGS_DIST_BACKEND", None)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["total"] == [1.5, 2.5, 3.0] and r["mx"] == 7.0 and r["per"] == [4.25] and r["asy"] == [2.0, 5.0]
    assert r["buckets"] >= 3 and r["early"] >= 1
    for out in r["outs"]:
        assert out == [1.0, 2.0, 0.0, 4.0]
    assert r["g2"] == [1.0, 2.0]
