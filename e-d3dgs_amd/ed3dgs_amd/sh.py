"""Real spherical harmonics up to degree 3 in torch -- the `convert_SHs_python` branch of render()
(gaussian_renderer/__init__.py:85-92 calls utils/sh_utils.py:57 eval_sh) for callers that run outside the reference's
source tree.  Same basis, sign convention and coefficient order as CR/forward.cu:23-74 / CR/auxiliary.h:19-39, so the
colours equal the rasterizer's own SH path (tests/test_reference_paths_gpu.py holds both against the reference-generated
fixture tests/golden/raster_k1_sh.npz)."""
import math

_K0 = 0.5 / math.sqrt(math.pi)
_K1 = math.sqrt(3.0 / (4.0 * math.pi))
_K2 = (0.5 * math.sqrt(15.0 / math.pi), -0.5 * math.sqrt(15.0 / math.pi), 0.25 * math.sqrt(5.0 / math.pi),
       -0.5 * math.sqrt(15.0 / math.pi), 0.25 * math.sqrt(15.0 / math.pi))
_K3 = (-0.25 * math.sqrt(35.0 / (2.0 * math.pi)), 0.5 * math.sqrt(105.0 / math.pi), -0.25 * math.sqrt(21.0 / (2.0 * math.pi)),
       0.25 * math.sqrt(7.0 / math.pi), -0.25 * math.sqrt(21.0 / (2.0 * math.pi)), 0.25 * math.sqrt(105.0 / math.pi),
       -0.25 * math.sqrt(35.0 / (2.0 * math.pi)))


def eval_sh(deg, sh, dirs):
    """sh [..., C, >= (deg+1)^2], dirs [..., 3] unit vectors -> [..., C]"""
    if not 0 <= deg <= 3:
        raise ValueError("eval_sh: degree must be 0..3")
    if sh.shape[-1] < (deg + 1) ** 2:
        raise ValueError("eval_sh: not enough coefficients for the degree")
    res = _K0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - _K1 * y * sh[..., 1] + _K1 * z * sh[..., 2] - _K1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + _K2[0] * xy * sh[..., 4] + _K2[1] * yz * sh[..., 5] + _K2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + _K2[3] * xz * sh[..., 7] + _K2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + _K3[0] * y * (3 * xx - yy) * sh[..., 9] + _K3[1] * xy * z * sh[..., 10]
                       + _K3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + _K3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + _K3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _K3[5] * z * (xx - yy) * sh[..., 14]
                       + _K3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res
