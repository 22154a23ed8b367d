"""The k-NN oracle (oracle/knn_ref.py) against known answers and scipy's KD-tree -- the reference ships no fixture for
simple_knn.distCUDA2 / o3d_knn ("parity unpinned", see the oracle's header)."""
import numpy as np
import pytest

from oracle import knn_ref as K


def lattice(n, h):
    g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)
    return g.astype(np.float32) * np.float32(h)


def test_lattice_known_answer():
    # every lattice point has >= 3 neighbours at distance h
    out = K.mean_dist2(lattice(6, 0.5))
    assert np.array_equal(out, np.full(216, 0.25, np.float32))


def test_tiny_clouds_follow_the_reference_arithmetic():
    p = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32)
    assert np.isinf(K.mean_dist2(p[:1])[0])                                 # FLT_MAX + FLT_MAX overflows
    assert np.isinf(K.mean_dist2(p[:2])[0])
    assert np.allclose(K.mean_dist2(p), K.FLT_MAX / np.float32(3), rtol=1e-6)  # two neighbours + one FLT_MAX slot


def test_duplicates_count_at_distance_zero():
    p = np.array([[0, 0, 0]] * 4 + [[1, 0, 0]], np.float32)
    out = K.mean_dist2(p)
    assert np.array_equal(out[:4], np.zeros(4, np.float32))
    assert out[4] == np.float32(1.0)


@pytest.mark.parametrize("k", [3, 20])
def test_against_scipy_kdtree(k):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    p = rng.normal(size=(5000, 3)).astype(np.float32)
    D, I = K.neighbours(p, k)
    d_ref, i_ref = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k + 1)
    assert np.allclose(D, d_ref[:, 1:] ** 2, rtol=1e-5, atol=1e-9)
    assert (I == i_ref[:, 1:]).mean() > 0.999        # ties / fp32 rounding may swap equal-distance neighbours
