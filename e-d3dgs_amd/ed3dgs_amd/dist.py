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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # ED3DGS_DIST_BACKEND=gloo rehearses several ranks on ONE GPU (RCCL refuses duplicate devices)
            backend = os.environ.get("ED3DGS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _on_wire(t):
    """gloo reduces host tensors; RCCL reduces device tensors in place."""
    if dist.is_initialized() and dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu()
    return t


def allreduce_sum_(t):
    """In-place SUM over ranks of a small tensor (the path's loss/psnr/count vector)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        w = _on_wire(t)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        if w is not t:
            t.copy_(w)
    return t


def shard_items(n_items, rank, world):
    """Indices of the (camera, frame) items rank `rank` renders: i = rank (mod world)."""
    return list(range(rank, n_items, world))


def item_of(index, n_cams, n_frames):
    """Flat item index -> (camera index, frame index); frames vary fastest so consecutive ranks take consecutive
    timesteps of one camera."""
    return (index // n_frames) % n_cams, index % n_frames


def allreduce_stats(loss_sum, psnr_sum, count, device):
    """The path's single collective: SUM of [loss, psnr, count] over ranks.  Returns a tensor of 3 floats."""
    t = torch.tensor([float(loss_sum), float(psnr_sum), float(count)], dtype=torch.float64 if device == "cpu" else torch.float32, device=device)
    return allreduce_sum_(t)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64 if device == "cpu" else torch.float32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        w = _on_wire(t)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        t = w
    return float(t[0])
