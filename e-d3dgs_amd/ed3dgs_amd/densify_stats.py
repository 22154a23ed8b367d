"""Sharded densification statistics for data-parallel training (second half).

The reference trains on one GPU: after every iteration it folds the rendered view's screen-space gradient norms and radii
into per-Gaussian accumulators (train.py:404-407 -> GaussianModel.add_densification_stats, scene/gaussian_model.py:516-518,
`max_radii2D` train.py:406) and every `densification_interval` iterations turns them into clone / split / prune decisions
(scene/gaussian_model.py:452-514).  With frames sharded over ranks each rank sees only its own views, so the accumulators
are per-rank partial sums: before a decision they are combined over ranks -- SUM for `xyz_gradient_accum` / the abs-grad
accumulator (the `.z` column of dL_dmeans2D) / `denom`, MAX for `max_radii2D` -- in TWO collectives (one flat
fp32 SUM buffer of 3P floats, one MAX of P), after which every rank holds the statistics a single-GPU run with
batch_size = 1 over the union of the views would hold -- one add() per view -- and takes bit-identical decisions.  The
split's random offsets come from a generator seeded by the iteration number, identical on every rank, so the post-densify
tensors are identical too (checked by hash in the tests).

Which single-GPU run that is matters.  N data-parallel ranks that take ONE optimizer step together correspond to the
reference's batch_size = N, and for a batch the reference first sums the views' RAW screen-space gradients (train.py:346-348),
ORs their visibility (train.py:190), maxes their radii (:189) and then calls add_densification_stats ONCE: ||sum_v g_v|| is
added and denom grows by 1 (scene/gaussian_model.py:516-518) -- whereas add() per view accumulates sum_v ||g_v|| and
denom += (views that see the Gaussian).  The mean gradient compared with densify_grad_threshold differs between the two (by
up to a factor N), so clone / split decisions differ.  Both are offered: add() + all_reduce_() = sequential batch_size-1
iterations over the union of the views; add_batched_step() = the reference's batched step (three collectives per step,
accumulators replicated, no all_reduce_() before the decision; the summed gradient is scaled by 1 / world_size by default,
because the reference's batch loss is a mean over the batch while a data-parallel rank's loss is a per-view mean).  tests/test_densify_stats_cpu.py pins each against the
reference's formulas.

Only what the decision needs is here; the optimizer-state surgery of cat_tensors_to_optimizer / _prune_optimizer
(scene/gaussian_model.py:364-423) stays with the caller's optimizer, as SURVEY section 2 scopes it.
"""
import hashlib

import torch
import torch.distributed as dist

from . import dist as D


class DensificationStats:
    def __init__(self, num_points, device):
        self.device = device
        self.reset(num_points)

    def reset(self, num_points):
        """densification_postfix (scene/gaussian_model.py:448-450): all accumulators restart at zero."""
        z = lambda *s: torch.zeros(*s, device=self.device)
        self.xyz_gradient_accum = z(num_points, 1)
        self.abs_gradient_accum = z(num_points, 1)
        self.denom = z(num_points, 1)
        self.max_radii2D = z(num_points)
        self._reduced = False
        self._replicated = False

    @torch.no_grad()
    def add(self, viewspace_point_grad, visibility_filter, radii):
        """One rendered view (train.py:404-407): radii -> running max, |grad[:, :2]| -> running sum, visits -> denom.
        `viewspace_point_grad` is render()'s viewspace_points.grad (P, 3): columns x, y and the abs-grad z."""
        if self._reduced:
            raise RuntimeError("DensificationStats.add after all_reduce_: call reset() (densification_postfix) first")
        vf = visibility_filter
        self.max_radii2D[vf] = torch.max(self.max_radii2D[vf], radii[vf].to(self.max_radii2D.dtype))
        self.xyz_gradient_accum[vf] += torch.norm(viewspace_point_grad[vf, :2], dim=-1, keepdim=True)
        self.abs_gradient_accum[vf] += viewspace_point_grad[vf, 2:3]
        self.denom[vf] += 1

    @torch.no_grad()
    def add_batched_step(self, viewspace_point_grad, visibility_filter, radii, loss_scale=None):
        """One optimizer step taken by all ranks together = ONE batch of the reference (train.py:166-190, 346-348, 404-407):
        SUM the ranks' (P, 3) screen-space gradients, OR their visibility, MAX their radii, then one
        add_densification_stats.  Every rank ends with identical accumulators; do not call all_reduce_() afterwards.

        `loss_scale` multiplies the summed gradient.  The reference's batch loss is a MEAN over the stacked views
        (train.py:195-197: l1_loss(..., keepdim=True).mean()), so each view's gradient at train.py:346-348 already carries
        1 / batch_size.  A data-parallel rank that backpropagates its own per-view mean loss -- what
        dist.BucketedGradReducer(average=True) / allreduce_gradients_ assume -- produces gradients batch_size times larger, and
        the default None = 1 / world_size restores the reference's value (otherwise the mean gradient would meet
        densify_grad_threshold a factor world_size too high).  Pass 1.0 if the ranks already scaled their losses by 1 / world."""
        if self._reduced:
            raise RuntimeError("DensificationStats.add_batched_step after all_reduce_: call reset() first")
        g = viewspace_point_grad.detach().clone()
        vis = visibility_filter.to(torch.float32)
        rad = radii.to(self.max_radii2D.dtype).clone()
        if dist.is_initialized() and dist.get_world_size() > 1:
            wg, wv, wr = D._on_wire(g), D._on_wire(vis), D._on_wire(rad)
            works = [dist.all_reduce(wg, op=dist.ReduceOp.SUM, async_op=True), dist.all_reduce(wv, op=dist.ReduceOp.MAX, async_op=True),
                     dist.all_reduce(wr, op=dist.ReduceOp.MAX, async_op=True)]
            for w in works:
                w.wait()
            for dst, src in ((g, wg), (vis, wv), (rad, wr)):
                if src is not dst:
                    dst.copy_(src)
        world = dist.get_world_size() if dist.is_initialized() else 1
        scale = (1.0 / world) if loss_scale is None else float(loss_scale)
        if scale != 1.0:
            g.mul_(scale)
        vf = vis > 0
        self.max_radii2D[vf] = torch.max(self.max_radii2D[vf], rad[vf])
        self.xyz_gradient_accum[vf] += torch.norm(g[vf, :2], dim=-1, keepdim=True)
        self.abs_gradient_accum[vf] += g[vf, 2:3]
        self.denom[vf] += 1
        self._replicated = True

    @torch.no_grad()
    def all_reduce_(self):
        """Combine the ranks' partial statistics in place: one SUM over a flat (3P) buffer, one MAX over (P).  RCCL on
        the GPUs (device buffers, in place); gloo on the CPU tests."""
        if getattr(self, "_replicated", False):
            raise RuntimeError("DensificationStats.all_reduce_ after add_batched_step: the accumulators are already replicated")
        if dist.is_initialized() and dist.get_world_size() > 1:
            P = self.denom.shape[0]
            flat = torch.cat((self.xyz_gradient_accum.reshape(-1), self.abs_gradient_accum.reshape(-1), self.denom.reshape(-1)))
            wire = D._on_wire(flat)
            w1 = dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True)
            mx = D._on_wire(self.max_radii2D)
            w2 = dist.all_reduce(mx, op=dist.ReduceOp.MAX, async_op=True)
            w1.wait(); w2.wait()
            if wire is not flat:
                flat.copy_(wire)
            if mx is not self.max_radii2D:
                self.max_radii2D.copy_(mx)
            self.xyz_gradient_accum.copy_(flat[:P].view(P, 1))
            self.abs_gradient_accum.copy_(flat[P:2 * P].view(P, 1))
            self.denom.copy_(flat[2 * P:].view(P, 1))
        self._reduced = True
        return self

    @torch.no_grad()
    def mean_grads(self):
        """densify (scene/gaussian_model.py:510-511): accum / denom with NaN (never visible) -> 0."""
        g = self.xyz_gradient_accum / self.denom
        g[g.isnan()] = 0.0
        return g


@torch.no_grad()
def decide(stats, scaling, opacity, max_grad, min_opacity, extent, max_screen_size, percent_dense=0.01):
    """The three masks of one densification round over the CURRENT P Gaussians, as the reference derives them
    (densify_and_clone :479-483, densify_and_split :452-460, prune :495-507 with use_mean=False).  `scaling` / `opacity` are
    the activated values (get_scaling, get_opacity).  Note the reference's split runs after the clone has appended rows
    (`padded_grad` zero-pads them), so clones are never split in the same round; masks here are over the original rows."""
    grads = stats.mean_grads()
    big = torch.max(scaling, dim=1).values > percent_dense * extent
    hot = torch.norm(grads, dim=-1) >= max_grad
    clone_mask = hot & ~big
    split_mask = (grads.squeeze(-1) >= max_grad) & big
    prune_mask = (opacity < min_opacity).reshape(-1)
    if max_screen_size:
        prune_mask = prune_mask | (stats.max_radii2D > max_screen_size) | (scaling.max(dim=1).values > 0.1 * extent)
    return clone_mask, split_mask, prune_mask


def _build_rotation(r):
    """utils/general_utils.py:build_rotation (normalised quaternion -> rotation matrix)."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    R = torch.zeros((q.size(0), 3, 3), device=r.device, dtype=r.dtype)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


@torch.no_grad()
def densify_tensors(t, stats, max_grad, extent, iteration, percent_dense=0.01, N=2):
    """densify = clone then split (scene/gaussian_model.py:509-514) on a dict of per-Gaussian tensors
    {xyz, features_dc, features_rest, opacity, scaling, rotation, embedding, tongue_class} (raw, pre-activation, as
    GaussianModel stores them).  Returns the new dict.  The split's normal samples are drawn on the CPU from a generator
    seeded with `iteration`, so every rank draws the same numbers whatever its device's RNG state."""
    grads = stats.mean_grads()
    P0 = t["xyz"].shape[0]
    act_scale = torch.exp(t["scaling"])
    big = torch.max(act_scale, dim=1).values > percent_dense * extent
    clone_mask = (torch.norm(grads, dim=-1) >= max_grad) & ~big
    t = {k: torch.cat((v, v[clone_mask]), 0) for k, v in t.items()}                      # densification_postfix
    P1 = t["xyz"].shape[0]
    padded = torch.zeros(P1, device=grads.device)
    padded[:P0] = grads.squeeze(-1)
    act_scale = torch.exp(t["scaling"])
    split_mask = (padded >= max_grad) & (torch.max(act_scale, dim=1).values > percent_dense * extent)
    if bool(split_mask.any()):
        stds = act_scale[split_mask].repeat(N, 1)
        gen = torch.Generator().manual_seed(int(iteration))
        samples = (torch.randn(stds.shape, generator=gen, dtype=torch.float32).to(stds.device)) * stds
        rots = _build_rotation(t["rotation"][split_mask]).repeat(N, 1, 1)
        new = {k: v[split_mask].repeat(*([N] + [1] * (v.dim() - 1))) for k, v in t.items()}
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + t["xyz"][split_mask].repeat(N, 1)
        new["scaling"] = torch.log(act_scale[split_mask].repeat(N, 1) / (0.8 * N))
        t = {k: torch.cat((v, new[k]), 0) for k, v in t.items()}
        keep = ~torch.cat((split_mask, torch.zeros(N * int(split_mask.sum()), dtype=torch.bool, device=split_mask.device)))
        t = {k: v[keep] for k, v in t.items()}                                            # prune_points(prune_filter)
    return t


def tensor_hash(*tensors):
    """SHA-256 over the exact bytes of the tensors (rank-identity checks)."""
    h = hashlib.sha256()
    for x in tensors:
        h.update(x.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()
