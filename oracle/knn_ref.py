"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of the reference's nearest-neighbour queries:
  * mean_dist2(points)  -- simple_knn.distCUDA2: submodules/simple-knn/simple_knn.cu:130-181 (updateKBest keeps the 3
    smallest squared distances to the OTHER points -- index != own index, duplicates count at distance 0 -- and the result
    is (best0 + best1 + best2) / 3 in float32, empty slots = FLT_MAX), i.e. exact 3-NN; the Morton order and the boxes of
    :185-220 only prune, they do not change the result.
  * neighbours(points, k) -- utils/extra_utils.py:5-15 (open3d KD-tree, k + 1 nearest, the first dropped).

Brute force in float32, O(P^2), chunked: sizes up to a few 10^4.  Parity unpinned: the reference ships no fixture or test
for either function, the CUDA extension cannot be built here and open3d is not installed; the tests pin the restatement
with known answers (lattices, duplicates, tiny clouds) and, at full size, against scipy's cKDTree in float64.
"""
import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def _sq_dists(a, b):
    d = a[:, None, :] - b[None, :, :]
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def neighbours(points, k, chunk=512):
    """(sq_dists [P,k] float32 ascending, indices [P,k] int64); missing neighbours: FLT_MAX / -1."""
    p = np.ascontiguousarray(points, np.float32)
    P = p.shape[0]
    D = np.full((P, k), FLT_MAX, np.float32)
    I = np.full((P, k), -1, np.int64)
    for s in range(0, P, chunk):
        e = min(P, s + chunk)
        d = _sq_dists(p[s:e], p).astype(np.float32)
        d[np.arange(e - s), np.arange(s, e)] = np.inf          # the point itself (simple_knn.cu:176)
        kk = min(k, P - 1)
        if kk <= 0:
            continue
        idx = np.argpartition(d, kk - 1, axis=1)[:, :kk]
        dd = np.take_along_axis(d, idx, axis=1)
        o = np.argsort(dd, axis=1, kind="stable")
        D[s:e, :kk] = np.take_along_axis(dd, o, axis=1)
        I[s:e, :kk] = np.take_along_axis(idx, o, axis=1)
    return D, I


def mean_dist2(points):
    """simple_knn.cu:181: (best[0] + best[1] + best[2]) / 3.0f in float32."""
    D, _ = neighbours(points, 3)
    with np.errstate(over="ignore"):
        return ((D[:, 0] + D[:, 1]) + D[:, 2]) / np.float32(3.0)
