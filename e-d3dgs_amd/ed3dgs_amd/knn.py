"""k nearest neighbours on the GPU through the C ABI (csrc/knn.hip).

distCUDA2(points)            -- simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:14-25, simple_knn.cu:185-220):
                                mean squared distance to the 3 nearest other points, float32 [P].
knn_neighbours(points, 20)   -- (sq_dists [P,20] float32, indices [P,20] int64), ascending, the point itself excluded.
o3d_knn(pts, 20)             -- utils/extra_utils.py:5-15 with the same signature (numpy in, numpy out), for train.py:219.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _prep(points, who):
    if not isinstance(points, torch.Tensor) or not points.is_cuda:
        raise RuntimeError(f"{who}: points must be a GPU tensor (the MI355X path has no CPU fallback)")
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError(f"{who}: points must have shape [P, 3]")
    x = points.detach().contiguous().float()
    L = _lib.lib()
    P = x.shape[0]
    ws_bytes = L.ed3dgs_knn_workspace_bytes(C.c_int(P))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    return L, x, P, ws, ws_bytes


@torch.no_grad()
def distCUDA2(points):
    L, x, P, ws, ws_bytes = _prep(points, "distCUDA2")
    out = torch.empty(P, dtype=torch.float32, device=x.device)   # spatial.cu:19: torch::full({P}, 0.0)
    rc = L.ed3dgs_knn_mean_dist2(C.c_int(P), C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                 C.c_size_t(ws_bytes), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc < 0:
        raise RuntimeError(_lib.last_error())
    return out


@torch.no_grad()
def knn_neighbours(points, num_knn=20):
    L, x, P, ws, ws_bytes = _prep(points, "knn_neighbours")
    d = torch.empty((P, num_knn), dtype=torch.float32, device=x.device)
    i = torch.empty((P, num_knn), dtype=torch.int64, device=x.device)
    rc = L.ed3dgs_knn_neighbours(C.c_int(P), C.c_int(num_knn), C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()),
                                 C.c_void_p(i.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(ws_bytes),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc < 0:
        raise RuntimeError(_lib.last_error())
    return d, i


def o3d_knn(pts, num_knn, device="cuda"):
    """utils/extra_utils.py:5-15: returns (sq_dists, indices) as numpy arrays of shape [P, num_knn]."""
    x = torch.as_tensor(np.ascontiguousarray(pts, np.float32), device=device)
    d, i = knn_neighbours(x, num_knn)
    return d.cpu().numpy().astype(np.float64), i.cpu().numpy()
