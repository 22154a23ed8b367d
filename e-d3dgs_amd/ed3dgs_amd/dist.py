"""Frame/view sharding across the GPUs of one node.

The path shards embarrassingly: every (camera, frame) render is independent given replicated Gaussian + MLP
parameters, so rank r renders items i = r (mod N) and the only data-path collective is ONE all-reduce(SUM) of
[sum loss, sum psnr, count] per step -- 12 bytes, latency-bound on xGMI.  No tile or Gaussian split.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("ED3DGS_DIST_COLLECTIVES_AT_WORLD_1") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # ED3DGS_DIST_BACKEND=gloo rehearses several ranks on ONE GPU (RCCL refuses duplicate devices)
            backend = os.environ.get("ED3DGS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _active():
    """True when collectives are to be issued: a process group of more than one rank -- or of ONE rank when
    ED3DGS_DIST_COLLECTIVES_AT_WORLD_1=1, which is how tests/test_dist_rccl_gpu.py drives every collective of this module through
    RCCL on a one-GPU box (RCCL refuses two ranks on one device, so a one-rank communicator is the most a single card can show:
    communicator creation, device-tensor all-reduce, stream-ordered wait())."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("ED3DGS_DIST_COLLECTIVES_AT_WORLD_1") == "1"


def _on_wire(t):
    """gloo reduces host tensors; RCCL reduces device tensors in place."""
    if dist.is_initialized() and dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu()
    return t


def allreduce_sum_(t):
    """In-place SUM over ranks of a small tensor (the path's loss/psnr/count vector)."""
    if _active():
        w = _on_wire(t)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        if w is not t:
            t.copy_(w)
    return t


class _Reduced:
    """Handle of an all-reduce in flight.  wait() makes the CURRENT STREAM (not the host) wait for the collective on
    RCCL, so a rank keeps enqueueing its next frame while the 12 bytes travel; gloo completes on the host."""

    def __init__(self, t, wire=None, work=None):
        self.t, self.wire, self.work = t, wire, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
            if self.wire is not self.t:
                self.t.copy_(self.wire)
            self.work = None
        return self.t


def allreduce_sum_async(t):
    """SUM over ranks of a small tensor, not waited for: the caller waits one step later (bench.py), which takes the
    per-step collective out of the ranks' critical path -- frames differ in cost, and a blocking collective per 4 ms
    step would run every rank at the pace of the slowest frame of each step."""
    if _active():
        w = _on_wire(t)
        return _Reduced(t, w, dist.all_reduce(w, op=dist.ReduceOp.SUM, async_op=True))
    return _Reduced(t)


def shard_items(n_items, rank, world):
    """Indices of the (camera, frame) items rank `rank` renders: i = rank (mod world)."""
    return list(range(rank, n_items, world))


def item_of(index, n_cams, n_frames):
    """Flat item index -> (camera index, frame index); frames vary fastest so consecutive ranks take consecutive
    timesteps of one camera."""
    return (index // n_frames) % n_cams, index % n_frames


def visit_stride(n):
    """A stride coprime to n, near 0.37 n: k -> (k * stride) % n visits every index once per n steps and spreads any short
    run of steps over the whole list (a 20-step run over 8 cameras x 50 frames, frames fastest, sees all 8 cameras)."""
    from math import gcd
    if n <= 2:
        return 1
    s = max(1, int(round(0.37 * n)))
    while gcd(s, n) != 1:
        s += 1
    return s


def strided_item(my_items, k):
    """The k-th item a rank renders: its shard visited with visit_stride (train.py:134-187 draws items at random; a fixed
    coprime stride is the deterministic stand-in)."""
    n = len(my_items)
    return my_items[(k * visit_stride(n)) % n]


_ORDER_CACHE = {}


def visit_order(my_items, n_cams, n_frames):
    """The order in which a rank renders its shard: round-robin over the CAMERAS present in the shard, and inside a camera its
    items with visit_stride -- a permutation of the shard whose every run of C consecutive steps sees C different cameras (C = the
    cameras of the shard), so that a short timed run (the driver's 20 steps) covers every camera on every rank whatever the
    workload's shape (C4: 15 cameras x 300 frames over 8 ranks, where a single stride over the flat shard misses two of them).
    train.py:134-187 draws its views at random; this is the deterministic stand-in."""
    key = (len(my_items), my_items[0] if my_items else -1, my_items[-1] if my_items else -1, n_cams, n_frames)
    order = _ORDER_CACHE.get(key)
    if order is None:
        by_cam = {}
        for i in my_items:
            by_cam.setdefault(item_of(i, n_cams, n_frames)[0], []).append(i)
        lists = []
        for c in sorted(by_cam):
            v = by_cam[c]
            st = visit_stride(len(v))
            lists.append([v[(k * st) % len(v)] for k in range(len(v))])
        order, k = [], 0
        while len(order) < len(my_items):
            for v in lists:
                if k < len(v):
                    order.append(v[k])
            k += 1
        _ORDER_CACHE[key] = order
    return order


def visit_item(my_items, k, n_cams, n_frames):
    """The k-th item a rank renders (visit_order, cyclic)."""
    order = visit_order(my_items, n_cams, n_frames)
    return order[k % len(order)]


def gather_per_rank(value, device):
    """[value of rank 0, ..., value of rank N-1] on every rank (one SUM all-reduce of a one-hot vector)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    t = torch.zeros(world, dtype=torch.float64 if device == "cpu" else torch.float32, device=device)
    t[rank] = float(value)
    allreduce_sum_(t)
    return [float(x) for x in t.tolist()]


def destroy():
    if dist.is_initialized():
        dist.destroy_process_group()


def allreduce_stats(loss_sum, psnr_sum, count, device):
    """The path's single collective: SUM of [loss, psnr, count] over ranks.  Returns a tensor of 3 floats."""
    t = torch.tensor([float(loss_sum), float(psnr_sum), float(count)], dtype=torch.float64 if device == "cpu" else torch.float32, device=device)
    return allreduce_sum_(t)


def barrier():
    if _active():
        if dist.get_backend() == "nccl":   # RCCL: on the rank's own device (the caller selected it before init())
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64 if device == "cpu" else torch.float32, device=device)
    if _active():
        w = _on_wire(t)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        t = w
    return float(t[0])


def allreduce_gradients_(params, bucket_bytes=64 << 20, average=True):
    """Data-parallel training on top of the frame sharding (the reference is single-GPU, train.py
    :345-348 steps its optimizer on one rank's gradients): SUM (or mean) the ranks' gradients in place, in flat buckets.
    xGMI is point-to-point (7 links per GPU), so a ring all-reduce is bound per link: few large buckets (64 MB) keep the
    links streaming and the launch count low; each bucket's collective is issued asynchronously and the copies back wait
    on it, so bucket k+1's packing overlaps bucket k's transfer.  Parameters whose .grad is None contribute zeros (a
    rank whose frame saw nothing of a parameter must still take part in the collective)."""
    if not (_active()):
        return
    world = dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nb = p.numel() * p.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
            buckets.append(cur); cur, cur_bytes = [], 0
        cur.append(p); cur_bytes += nb
    if cur:
        buckets.append(cur)
    pending = []
    for b in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
        wire = _on_wire(flat)
        pending.append((b, flat, wire, dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True)))
    for b, flat, wire, work in pending:
        work.wait()
        if wire is not flat:
            flat.copy_(wire)
        if average:
            flat.div_(world)
        off = 0
        for p in b:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n


class BucketedGradReducer:
    """allreduce_gradients_ driven by autograd: every parameter gets a post-accumulate hook, and a bucket's all-reduce is issued
    the moment its last gradient has landed -- while the backward is still producing the others -- instead of after
    backward() has returned.  finish() waits for the buckets, averages and writes the results back into .grad.  Parameters
    that received no gradient by finish() contribute zeros (every rank must issue the same collectives in the same order:
    buckets are therefore ISSUED in bucket order; one that fills early waits for its predecessors).
    In render()'s graph the deformation node delivers most leaf gradients together at the very end, so what overlaps is the
    packing of bucket k+1 with the transfer of bucket k and the early SH / opacity gradients with the MLP backward.

    Ordering contract: the reducer orders its OWN buckets only.  A rank on which some parameter got no gradient issues that
    bucket in finish() while another rank issued it from the hook during backward(), so NO other collective of the same process
    group may be enqueued between backward() and finish() -- call finish() first, then the caller's own collectives (bench.py
    does).  One backward() per finish(): a hook that fires for a bucket already issued (gradient accumulation over two
    backward() calls) raises instead of silently dropping the second gradient."""

    def __init__(self, params, bucket_bytes=64 << 20, average=True):
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.buckets, cur, cur_bytes = [], [], 0
        for p in self.params:
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device):
                self.buckets.append(cur); cur, cur_bytes = [], 0
            cur.append(p); cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self.missing = [len(b) for b in self.buckets]
        self.pending = [None] * len(self.buckets)
        self.next_issue = 0
        self.handles = [p.register_post_accumulate_grad_hook(self._landed) for p in self.params]
        self.issued_in_backward = 0

    def _issue_ready(self, force=False):
        while self.next_issue < len(self.buckets) and (force or self.missing[self.next_issue] == 0):
            i = self.next_issue
            b = self.buckets[i]
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in b])
            wire = _on_wire(flat)
            work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True) if (_active()) else None
            self.pending[i] = (flat, wire, work)
            self.next_issue += 1
            if not force:
                self.issued_in_backward += 1

    def _landed(self, p):
        i = self.bucket_of[id(p)]
        if i < self.next_issue or self.missing[i] <= 0:
            raise RuntimeError("BucketedGradReducer: a gradient landed for a bucket that was already issued or complete -- "
                               "a second backward() before finish() (gradient accumulation) is not supported: call finish() "
                               "after every backward(), or use allreduce_gradients_ after the last one")
        self.missing[i] -= 1
        self._issue_ready()

    def finish(self):
        """Issue what is left (parameters without a gradient count as zeros), wait, average, write back; re-arm."""
        self._issue_ready(force=True)
        world = dist.get_world_size() if dist.is_initialized() else 1
        for b, (flat, wire, work) in zip(self.buckets, self.pending):
            if work is not None:
                work.wait()
            if wire is not flat:
                flat.copy_(wire)
            if self.average and world > 1:
                flat.div_(world)
            off = 0
            for p in b:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
        self.missing = [len(b) for b in self.buckets]
        self.pending = [None] * len(self.buckets)
        self.next_issue = 0

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []
