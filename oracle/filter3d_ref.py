"""CPU restatement of GaussianModel.compute_3D_filter (reference scene/gaussian_model.py:538-592) in numpy float32.

TEST INFRASTRUCTURE ONLY (imported by tests/ and nothing else).  Parity unpinned by the reference: the method lives on
GaussianModel, whose module needs simple_knn / open3d / plyfile (none importable here), and the reference ships no
fixture for it; the pins are the hand-derived known answers in tests/test_formats_cpu.py.

`cameras`: objects with R (3x3, the c2w rotation the reference's Camera stores), T (3), FoVx, FoVy, image_width,
image_height."""
import math

import numpy as np


def camera_rows(cameras):
    """n x 16 float32: R row-major, T, focal_x, focal_y, width, height (the layout ed3dgs_compute_3d_filter takes)."""
    rows = []
    for c in cameras:
        W, H = c.image_width, c.image_height
        fx = W / (2 * math.tan(c.FoVx / 2.0))      # :558-559
        fy = H / (2 * math.tan(c.FoVy / 2.0))
        rows.append(np.concatenate([np.asarray(c.R, np.float32).reshape(-1), np.asarray(c.T, np.float32).reshape(-1),
                                    np.array([fx, fy, W, H], np.float32)]))
    return np.stack(rows).astype(np.float32) if rows else np.zeros((0, 16), np.float32)


def compute_3D_filter(xyz, cameras):
    f = np.float32
    xyz = np.asarray(xyz, f)
    P = xyz.shape[0]
    distance = np.full(P, 100000.0, f)                       # :543
    valid_points = np.zeros(P, bool)                         # :544
    focal_length = f(0.0)
    for row in camera_rows(cameras):
        R = row[:9].reshape(3, 3); T = row[9:12]; fx, fy, W, H = row[12:16]
        # xyz @ R + T, accumulated left to right in float32 (:563)
        cam = (xyz[:, 0:1] * R[0][None, :] + xyz[:, 1:2] * R[1][None, :]).astype(f)
        cam = (cam + xyz[:, 2:3] * R[2][None, :]).astype(f)
        cam = (cam + T[None, :]).astype(f)
        valid_depth = cam[:, 2] > f(0.2)                     # :566
        z = np.maximum(cam[:, 2], f(0.001))                  # :569
        x = (cam[:, 0] / z * fx + W / f(2.0)).astype(f)      # :571-572
        y = (cam[:, 1] / z * fy + H / f(2.0)).astype(f)
        in_screen = (x >= f(-0.15) * W) & (x <= W * f(1.15)) & (y >= f(-0.15) * H) & (y <= f(1.15) * H)   # :577-579
        valid = valid_depth & in_screen
        distance[valid] = np.minimum(distance[valid], z[valid])   # :585
        valid_points |= valid
        if focal_length < fx:                                # :587-588
            focal_length = fx
    if valid_points.any():
        distance[~valid_points] = distance[valid_points].max()   # :590
    else:
        return np.zeros((P, 1), f)                           # the reference raises on the empty max
    return (distance / focal_length * f(0.2 ** 0.5)).astype(f)[:, None]   # :594-595
