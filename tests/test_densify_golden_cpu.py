"""SURVEY 8(f) rank 2, pinned by the reference itself: `ed3dgs_amd.densify_stats` against tests/golden/densify_reference.npz, which
tools/gen_densify_golden.py wrote by running the REFERENCE's own `GaussianModel.add_densification_stats` (scene/gaussian_model.py
:516-518, train.py:404-407) and `GaussianModel.densify` (= densify_and_clone + densify_and_split, :452-514, including
densification_postfix / prune_points and their optimizer surgery) on a synthetic 1 200-Gaussian model on the CPU.  The split's normal
samples come from a generator seeded with the iteration number on both sides (densify_tensors' convention; the reference draws
from the device RNG), so the comparison is bit for bit: accumulators, and every post-densify tensor (1 200 -> ~2 000 rows)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "e-d3dgs_amd"))
GOLD = os.path.join(ROOT, "tests", "golden", "densify_reference.npz")


def _load():
    z = np.load(GOLD)
    P, views, iteration = (int(x) for x in z["params"])
    max_grad, extent, percent_dense = (float(x) for x in z["thresholds"])
    return z, P, views, iteration, max_grad, extent, percent_dense


def _stats(z, P, views):
    from ed3dgs_amd import densify_stats as DS
    st = DS.DensificationStats(P, "cpu")
    for i in range(views):
        grad, radii = torch.from_numpy(z["view%d_grad" % i]), torch.from_numpy(z["view%d_radii" % i])
        st.add(grad, radii > 0, radii)
    return st


def test_accumulators_equal_the_references():
    z, P, views, *_ = _load()
    st = _stats(z, P, views)
    assert np.array_equal(st.xyz_gradient_accum.numpy(), z["accum"])
    assert np.array_equal(st.denom.numpy(), z["denom"])
    assert np.array_equal(st.max_radii2D.numpy(), z["max_radii2D"])
    assert (z["denom"] == 0).any() and (z["denom"] > 1).any()          # never-visible rows and several visits both occur


def test_densify_tensors_equal_the_references_post_densify_model():
    from ed3dgs_amd import densify_stats as DS
    z, P, views, iteration, max_grad, extent, percent_dense = _load()
    st = _stats(z, P, views)
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "embedding", "tongue_class")
    t = {k: torch.from_numpy(z["in_" + k]) for k in names}
    out = DS.densify_tensors(t, st, max_grad, extent, iteration, percent_dense=percent_dense)
    n_ref = z["out_xyz"].shape[0]
    assert n_ref > P                                                    # the round cloned and split
    for k in names:
        assert tuple(out[k].shape) == z["out_" + k].shape, k
        assert np.array_equal(out[k].numpy(), z["out_" + k]), k         # bit for bit, the split's new positions and scales included
    # the masks decide() reports are the ones behind those tensors: clones appended after the originals, splits after them, the split
    # originals removed
    cm, sm, _ = DS.decide(st, torch.exp(t["scaling"]), torch.sigmoid(t["opacity"]), max_grad, 0.005, extent, None, percent_dense=percent_dense)
    assert n_ref == P + int(cm.sum()) + 2 * int(sm.sum()) - int(sm.sum())
    kept = (~sm).nonzero().reshape(-1)
    assert np.array_equal(out["xyz"][: kept.numel()].numpy(), t["xyz"][kept].numpy())
