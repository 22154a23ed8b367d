"""simple_knn.distCUDA2 and the k = 20 neighbour lists (csrc/knn.hip through the C ABI) against the CPU restatement
(oracle/knn_ref.py) and, at the full 200k size, scipy's KD-tree.  Tolerance 1e-6 relative on squared distances (an exact
k-NN: only the last-ulp rounding of dx*dx + dy*dy + dz*dz can differ)."""
import numpy as np
import pytest
import torch

from oracle import knn_ref as K

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test collected without a GPU")


def clouds():
    rng = np.random.default_rng(1)
    out = {}
    for P in (1, 2, 3, 4, 5, 63, 64, 65, 129, 1000, 20000):
        out["normal-%d" % P] = rng.normal(size=(P, 3)).astype(np.float32)
    # clusters of very different density + far outliers: the box bounds must prune without losing neighbours
    c = np.concatenate([rng.normal(size=(3000, 3)) * 0.01, rng.normal(size=(3000, 3)) * 5 + 20, rng.uniform(-1e3, 1e3, size=(50, 3))])
    out["clusters"] = c.astype(np.float32)
    out["plane"] = np.concatenate([rng.uniform(size=(4000, 2)), np.zeros((4000, 1))], 1).astype(np.float32)   # zero extent in z
    out["line-dup"] = np.repeat(np.linspace(0, 1, 500, dtype=np.float32)[:, None], 3, 1).repeat(3, 0)        # every point 3 times
    out["same-point"] = np.ones((300, 3), np.float32)
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(12), indexing="ij"), -1).reshape(-1, 3)
    out["lattice"] = (g * 0.25).astype(np.float32)
    return out


CLOUDS = clouds()


@pytest.mark.parametrize("name", list(CLOUDS))
def test_distCUDA2_matches_restatement(name):
    _need_gpu()
    from simple_knn._C import distCUDA2
    p = CLOUDS[name]
    got = distCUDA2(torch.from_numpy(p).cuda()).cpu().numpy()
    ref = K.mean_dist2(p)
    assert got.shape == ref.shape and got.dtype == np.float32
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin)
    assert np.allclose(got[fin], ref[fin], rtol=RTOL, atol=0), np.abs(got[fin] - ref[fin]).max()
    if name == "lattice":
        assert np.array_equal(got, np.full(len(p), 0.0625, np.float32))


@pytest.mark.parametrize("name", ["normal-5", "normal-65", "normal-1000", "normal-20000", "clusters", "plane", "line-dup"])
def test_neighbour_lists_match_restatement(name):
    _need_gpu()
    from ed3dgs_amd.knn import knn_neighbours
    p = CLOUDS[name]
    d, i = knn_neighbours(torch.from_numpy(p).cuda(), 20)
    d, i = d.cpu().numpy(), i.cpu().numpy()
    D, I = K.neighbours(p, 20)
    assert np.allclose(d, D, rtol=RTOL, atol=0)
    assert np.array_equal(i < 0, I < 0)
    assert (np.diff(d, axis=1) >= 0).all()
    # the indices name points at exactly the reported distances (ties may be ordered differently)
    ok = i >= 0
    rows = np.nonzero(ok)[0]
    dd = ((p[rows] - p[i[ok]]) ** 2).sum(-1)
    assert np.allclose(dd, d[ok], rtol=1e-5, atol=1e-12)
    assert (i != np.arange(len(p))[:, None]).all()
    if name != "line-dup":      # every point three times: the order among equal distances is free
        assert (i == I).mean() > 0.99


def test_full_size_against_kdtree():
    """200k points (the C3 cloud size): distCUDA2 and the 20-neighbour lists against scipy's cKDTree in float64."""
    _need_gpu()
    from scipy.spatial import cKDTree
    from ed3dgs_amd.knn import knn_neighbours
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(2)
    p = (rng.normal(size=(200_000, 3)) * np.array([1.0, 0.6, 0.3])).astype(np.float32)
    x = torch.from_numpy(p).cuda()
    d_ref, _ = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), 21, workers=-1)
    ref2 = d_ref[:, 1:] ** 2
    got = distCUDA2(x).cpu().numpy()
    assert np.allclose(got, ref2[:, :3].mean(1), rtol=1e-5)
    d, i = knn_neighbours(x, 20)
    assert np.allclose(d.cpu().numpy(), ref2, rtol=1e-5, atol=1e-12)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); distCUDA2(x); t1.record(); torch.cuda.synchronize()
    print("distCUDA2 200k: %.3f ms" % t0.elapsed_time(t1))
    t0.record(); knn_neighbours(x, 20); t1.record(); torch.cuda.synchronize()
    print("knn_neighbours 200k k=20: %.3f ms" % t0.elapsed_time(t1))


def test_rejects_cpu_tensors_and_bad_shapes():
    _need_gpu()
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(10, 3))
    with pytest.raises(ValueError):
        distCUDA2(torch.zeros(10, 2).cuda())
    assert distCUDA2(torch.zeros(0, 3).cuda()).shape == (0,)
