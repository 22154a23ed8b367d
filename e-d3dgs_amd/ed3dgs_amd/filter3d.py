"""compute_3D_filter (scene/gaussian_model.py:538-592) through the C ABI (csrc/filter3d.hip)."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib


def camera_rows(cameras):
    """n x 16 float32 host array: R (row-major, as Camera.R), T, focal_x, focal_y, width, height."""
    rows = np.zeros((len(cameras), 16), np.float32)
    for i, c in enumerate(cameras):
        W, H = c.image_width, c.image_height
        rows[i, :9] = np.asarray(c.R, np.float32).reshape(-1)
        rows[i, 9:12] = np.asarray(c.T, np.float32).reshape(-1)
        rows[i, 12] = W / (2 * math.tan(c.FoVx / 2.0))
        rows[i, 13] = H / (2 * math.tan(c.FoVy / 2.0))
        rows[i, 14], rows[i, 15] = W, H
    return rows


@torch.no_grad()
def compute_3D_filter(xyz, cameras):
    """Returns filter_3D [P, 1] on xyz's device (GPU only: the MI355X path has no CPU fallback)."""
    if not xyz.is_cuda:
        raise RuntimeError("compute_3D_filter: xyz must be on the GPU (the MI355X path has no CPU fallback)")
    L = _lib.lib()
    x = xyz.detach().contiguous().float()
    P = x.shape[0]
    out = torch.empty((P, 1), dtype=torch.float32, device=x.device)
    rows = np.ascontiguousarray(camera_rows(cameras))
    ws_bytes = L.ed3dgs_filter3d_workspace_bytes(C.c_int(P))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    rc = L.ed3dgs_compute_3d_filter(C.c_int(P), C.c_void_p(x.data_ptr()), C.c_int(len(cameras)),
                                    rows.ctypes.data_as(C.c_void_p), C.c_void_p(out.data_ptr()),
                                    C.c_void_p(ws.data_ptr()), C.c_size_t(ws_bytes),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc < 0:
        raise RuntimeError(_lib.last_error())
    return out
