"""oracle/raster_oracle.py -- TEST INFRASTRUCTURE ONLY (see oracle/raster_ref.c header).

ctypes front-end of the plain-C restatement of the reference rasterizer.  It mirrors the orchestration of
`CudaRasterizer::Rasterizer::forward/backward` (CR/rasterizer_impl.cu:255-432, 436-578) and the tensor
shapes of `RasterizeGaussiansCUDA` / `RasterizeGaussiansBackwardCUDA` (DGR/rasterize_points.cu:35-250).
numpy in, numpy out.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
# oracle/raster_oracle64.py executes this same file with _PRECISION = "f64" pre-set: every array / scalar argument is then
# fp64 and the library is libraster_ref64.so (raster_ref.c built with -DED3REF_FP64).  Default: the fp32 restatement.
_F64 = globals().get("_PRECISION") == "f64"
REAL = np.float64 if _F64 else np.float32
CREAL = C.c_double if _F64 else C.c_float
_SO = "libraster_ref64.so" if _F64 else "libraster_ref.so"


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, _SO])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, _SO)
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.ed3ref_higher_msb.restype = C.c_uint32
        _LIB.ed3ref_eig_sym3.restype = C.c_int
        _LIB.ed3ref_get_eig_epsilon.restype = C.c_double
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=REAL)


def set_eig_epsilon(e):
    """Convergence threshold of the eigen-solver (reference: 1e-7 absolute).  Tests only; restore after use."""
    lib().ed3ref_set_eig_epsilon(C.c_double(e))


def set_margin_weights(alpha=1.0, termination=1.0, median=1.0):
    """Per-decision weights of the forward's per-pixel margin (ed3ref_set_margin_weights in raster_ref.c): the margin is the
    minimum over the pixel's decisions of (relative distance to the threshold) / weight; 0 leaves a decision kind out."""
    lib().ed3ref_set_margin_weights(C.c_double(alpha), C.c_double(termination), C.c_double(median))


def get_eig_epsilon():
    return float(lib().ed3ref_get_eig_epsilon())


def eig_sym3(cov6):
    val = np.zeros(3, REAL)
    vec = np.zeros(9, REAL)
    n = lib().ed3ref_eig_sym3(_p(_f32(cov6)), _p(val), _p(vec))
    return n, val, vec.reshape(3, 3)  # vec[c] = eigenvector c


def mark_visible(means3D, viewmatrix, projmatrix):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, np.uint8)
    lib().ed3ref_mark_visible(C.c_int(P), _p(means3D), _p(_f32(viewmatrix).ravel()), _p(_f32(projmatrix).ravel()), _p(out))
    return out.astype(bool)


def forward(bg, means3D, colors_precomp, opacities, tongue_class, scales, rotations, scale_modifier, cov3D_precomp,
            viewmatrix, projmatrix, tanfovx, tanfovy, kernel_size, H, W, sh, degree, campos, require_coord,
            require_depth, with_margin=False):
    """Returns a dict holding the 9 images, radii, and every intermediate state (geometry, binning, image)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    sh = _f32(sh) if sh is not None and sh.size else None
    colors_precomp = _f32(colors_precomp) if colors_precomp is not None and colors_precomp.size else None
    scales = _f32(scales) if scales is not None and scales.size else None
    rotations = _f32(rotations) if rotations is not None and rotations.size else None
    cov3D_precomp = _f32(cov3D_precomp) if cov3D_precomp is not None and cov3D_precomp.size else None
    M = sh.shape[1] if sh is not None else 0
    opacities = _f32(opacities).reshape(-1)
    tongue_class = _f32(tongue_class).reshape(-1)
    view = _f32(viewmatrix).reshape(-1)
    proj = _f32(projmatrix).reshape(-1)
    campos = _f32(campos).reshape(-1)
    bg = _f32(bg).reshape(-1)
    H, W = int(H), int(W)
    tanfovx, tanfovy = REAL(tanfovx), REAL(tanfovy)
    focal_y = REAL(H) / (REAL(2.0) * tanfovy)
    focal_x = REAL(W) / (REAL(2.0) * tanfovx)
    g = dict(
        clamped=np.zeros((P, 3), np.uint8), radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), REAL),
        view_points=np.zeros((P, 3), REAL), depths=np.zeros(P, REAL),
        camera_planes=np.zeros((P, 6), REAL), ray_planes=np.zeros((P, 2), REAL),
        ts=np.zeros(P, REAL), normals=np.zeros((P, 3), REAL), cov3D=np.zeros((P, 6), REAL),
        rgb=np.zeros((P, 3), REAL), conic_opacity=np.zeros((P, 4), REAL),
        is_tongue=np.zeros(P, REAL), tiles_touched=np.zeros(P, np.uint32))
    if P:
        L.ed3ref_preprocess(
            C.c_int(P), C.c_int(int(degree)), C.c_int(M), _p(means3D), _p(scales), CREAL(scale_modifier),
            _p(rotations), _p(opacities), _p(tongue_class), _p(sh), _p(cov3D_precomp), _p(colors_precomp), _p(view),
            _p(proj), _p(campos), C.c_int(W), C.c_int(H), CREAL(tanfovx), CREAL(tanfovy),
            CREAL(kernel_size), _p(g["clamped"]), _p(g["radii"]), _p(g["means2D"]), _p(g["view_points"]),
            _p(g["depths"]), _p(g["camera_planes"]), _p(g["ray_planes"]), _p(g["ts"]), _p(g["normals"]),
            _p(g["cov3D"]), _p(g["rgb"]), _p(g["conic_opacity"]), _p(g["is_tongue"]), _p(g["tiles_touched"]), None, None)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    offsets = np.cumsum(g["tiles_touched"].astype(np.uint64)).astype(np.uint32)  # K2 inclusive scan
    R = int(offsets[-1]) if P else 0
    keys_u = np.zeros(R, np.uint64)
    vals_u = np.zeros(R, np.uint32)
    keys = np.zeros(R, np.uint64)
    point_list = np.zeros(R, np.uint32)
    ranges = np.zeros((T, 2), np.uint32)
    if P:
        L.ed3ref_duplicate_with_keys(C.c_int(P), _p(g["means2D"]), _p(g["depths"]), _p(offsets), _p(g["radii"]),
                                     C.c_int(W), C.c_int(H), _p(keys_u), _p(vals_u))
    bit = int(L.ed3ref_higher_msb(C.c_uint32(T)))
    L.ed3ref_sort_pairs(C.c_int64(R), _p(keys_u), _p(vals_u), _p(keys), _p(point_list), C.c_int(32 + bit))
    L.ed3ref_identify_tile_ranges(C.c_int64(R), _p(keys), _p(ranges), C.c_int(T))
    HW = H * W
    img = dict(
        color=np.zeros((3, H, W), REAL), coord=np.zeros((3, H, W), REAL),
        mcoord=np.zeros((3, H, W), REAL), alpha=np.zeros((1, H, W), REAL),
        tongue=np.zeros((1, H, W), REAL), normal=np.zeros((3, H, W), REAL),
        depth=np.zeros((1, H, W), REAL), mdepth=np.zeros((1, H, W), REAL),
        n_contrib=np.zeros((2, H, W), np.uint32), accum_coord=np.zeros((3, H, W), REAL),
        accum_depth=np.zeros((1, H, W), REAL), normal_length=np.zeros((1, H, W), REAL))
    margin = np.zeros((H, W), REAL) if with_margin else None
    features = colors_precomp if colors_precomp is not None else g["rgb"]
    L.ed3ref_render_forward(
        C.c_int(W), C.c_int(H), _p(ranges), _p(point_list), _p(g["view_points"]), _p(g["means2D"]), _p(features),
        _p(g["ts"]), _p(g["camera_planes"]), _p(g["ray_planes"]), _p(g["normals"]), _p(g["conic_opacity"]),
        _p(g["is_tongue"]), CREAL(focal_x), CREAL(focal_y), _p(bg), C.c_int(bool(require_coord)),
        C.c_int(bool(require_depth)), _p(img["alpha"]), _p(img["tongue"]), _p(img["n_contrib"]), _p(img["color"]),
        _p(img["coord"]), _p(img["mcoord"]), _p(img["normal"]), _p(img["depth"]), _p(img["mdepth"]),
        _p(img["accum_coord"]), _p(img["accum_depth"]), _p(img["normal_length"]), _p(margin))
    out = dict(num_rendered=R, point_offsets=offsets, keys_unsorted=keys_u, values_unsorted=vals_u, keys=keys,
               point_list=point_list, ranges=ranges, margin=margin, focal_x=focal_x, focal_y=focal_y, sort_bits=32 + bit)
    out.update(g)
    out.update(img)
    return out


def backward(fw, bg, means3D, colors_precomp, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
             tanfovx, tanfovy, kernel_size, dL_dcolor, dL_dcoord, dL_dmcoord, dL_ddepth, dL_dmdepth, dL_dalpha,
             dL_dnormal, sh, degree, campos, require_coord, require_depth, reference_q1=True):
    """`fw` is the dict returned by forward().  Returns the 8 gradients of rasterize_gaussians_backward plus the
    intermediate per-Gaussian gradients.  reference_q1=True reproduces quirk Q1 (dL_dconic passed as conic_opacity)."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    H, W = fw["color"].shape[1:]
    sh = _f32(sh) if sh is not None and sh.size else None
    colors_precomp = _f32(colors_precomp) if colors_precomp is not None and colors_precomp.size else None
    scales = _f32(scales) if scales is not None and scales.size else None
    rotations = _f32(rotations) if rotations is not None and rotations.size else None
    cov3D_precomp = _f32(cov3D_precomp) if cov3D_precomp is not None and cov3D_precomp.size else None
    M = sh.shape[1] if sh is not None else 0
    view = _f32(viewmatrix).reshape(-1)
    proj = _f32(projmatrix).reshape(-1)
    campos = _f32(campos).reshape(-1)
    bg = _f32(bg).reshape(-1)
    z = lambda *s: np.zeros(s, np.float64)
    d = dict(view_points=z(P, 3), mean2D=z(P, 3), conic=z(P, 4), opacity=z(P), colors=z(P, 3), ts=z(P),
             camera_planes=z(P, 6), ray_planes=z(P, 2), normals=z(P, 3))
    colors = colors_precomp if colors_precomp is not None else fw["rgb"]
    gz = lambda a, s: _f32(a) if a is not None else np.zeros(s, REAL)
    dL_dcolor = gz(dL_dcolor, (3, H, W)); dL_dcoord = gz(dL_dcoord, (3, H, W)); dL_dmcoord = gz(dL_dmcoord, (3, H, W))
    dL_ddepth = gz(dL_ddepth, (1, H, W)); dL_dmdepth = gz(dL_dmdepth, (1, H, W)); dL_dalpha = gz(dL_dalpha, (1, H, W))
    dL_dnormal = gz(dL_dnormal, (3, H, W))
    L.ed3ref_render_backward(
        C.c_int(W), C.c_int(H), _p(fw["ranges"]), _p(fw["point_list"]), _p(bg), _p(fw["view_points"]),
        _p(fw["means2D"]), _p(fw["conic_opacity"]), _p(colors), _p(fw["ts"]), _p(fw["camera_planes"]),
        _p(fw["ray_planes"]), _p(fw["alpha"]), _p(fw["normals"]), _p(fw["accum_coord"]), _p(fw["accum_depth"]),
        _p(fw["normal_length"]), _p(fw["n_contrib"]), _p(dL_dcolor), _p(dL_dcoord), _p(dL_dmcoord), _p(dL_ddepth),
        _p(dL_dmdepth), _p(dL_dalpha), _p(dL_dnormal), _p(fw["normal"]), CREAL(fw["focal_x"]),
        CREAL(fw["focal_y"]), C.c_int(bool(require_coord)), C.c_int(bool(require_depth)), _p(d["view_points"]),
        _p(d["mean2D"]), _p(d["conic"]), _p(d["opacity"]), _p(d["colors"]), _p(d["ts"]), _p(d["camera_planes"]),
        _p(d["ray_planes"]), _p(d["normals"]))
    f = {k: v.astype(REAL) for k, v in d.items()}  # the single rounding of the order-free sum
    dL_dmeans3D = np.zeros((P, 3), REAL)
    dL_dcov3D = np.zeros((P, 6), REAL)
    dL_dsh = np.zeros((P, M, 3), REAL)
    dL_dscales = np.zeros((P, 3), REAL)
    dL_drot = np.zeros((P, 4), REAL)
    dL_dopacity = f["opacity"].copy()
    cov3D = cov3D_precomp if cov3D_precomp is not None else fw["cov3D"]
    co_arg = f["conic"] if reference_q1 else fw["conic_opacity"]
    if P:
        L.ed3ref_cov2d_backward(
            C.c_int(P), _p(means3D), _p(fw["radii"]), _p(cov3D), CREAL(fw["focal_x"]), CREAL(fw["focal_y"]),
            CREAL(REAL(tanfovx)), CREAL(REAL(tanfovy)), CREAL(kernel_size), _p(view),
            _p(f["conic"]), _p(f["camera_planes"]), _p(f["ray_planes"]), _p(f["normals"]), _p(dL_dmeans3D),
            _p(dL_dcov3D), _p(co_arg), _p(dL_dopacity))
        L.ed3ref_preprocess_backward(
            C.c_int(P), C.c_int(int(degree)), C.c_int(M), _p(means3D), _p(fw["radii"]), _p(sh), _p(fw["clamped"]),
            _p(scales), _p(rotations), CREAL(scale_modifier), _p(view), _p(proj), _p(campos), _p(f["mean2D"]),
            _p(f["view_points"]), _p(dL_dmeans3D), _p(f["colors"]), _p(f["ts"]), _p(dL_dcov3D), _p(dL_dsh),
            _p(dL_dscales), _p(dL_drot))
    return dict(dL_dmeans2D=f["mean2D"], dL_dcolors=f["colors"], dL_dopacity=dL_dopacity.reshape(P, 1),
                dL_dmeans3D=dL_dmeans3D, dL_dcov3D=dL_dcov3D, dL_dsh=dL_dsh, dL_dscales=dL_dscales,
                dL_drotations=dL_drot, inter=f, inter64=d)


def integrate(bg, points3D, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
              viewmatrix, projmatrix, tanfovx, tanfovy, kernel_size, H, W, sh, degree, campos):
    """CudaRasterizer::Rasterizer::integrate (CR/rasterizer_impl.cu:580-851) with the tensor shapes / fill values of
    IntegrateGaussiansToPointsCUDA (DGR/rasterize_points.cu:273-392): K11 (preprocess with the inverse ray-space
    covariance), K2-K5 for the Gaussians, K12 / K13 / sort / K5 for the query points, K14.  Returns a dict with the
    outputs, the intermediate state and per-pixel / per-point decision margins (see ed3ref_integrate)."""
    L = lib()
    means3D = _f32(means3D); points3D = _f32(points3D)
    P, PN = means3D.shape[0], points3D.shape[0]
    sh = _f32(sh) if sh is not None and sh.size else None
    colors_precomp = _f32(colors_precomp) if colors_precomp is not None and colors_precomp.size else None
    scales = _f32(scales) if scales is not None and scales.size else None
    rotations = _f32(rotations) if rotations is not None and rotations.size else None
    cov3D_precomp = _f32(cov3D_precomp) if cov3D_precomp is not None and cov3D_precomp.size else None
    M = sh.shape[1] if sh is not None else 0
    opacities = _f32(opacities).reshape(-1)
    view = _f32(viewmatrix).reshape(-1); proj = _f32(projmatrix).reshape(-1)
    campos = _f32(campos).reshape(-1); bg = _f32(bg).reshape(-1)
    H, W = int(H), int(W)
    tanfovx, tanfovy = REAL(tanfovx), REAL(tanfovy)
    focal_y = REAL(H) / (REAL(2.0) * tanfovy)
    focal_x = REAL(W) / (REAL(2.0) * tanfovx)
    g = dict(
        clamped=np.zeros((P, 3), np.uint8), radii=np.zeros(P, np.int32), means2D=np.zeros((P, 2), REAL),
        view_points=np.zeros((P, 3), REAL), depths=np.zeros(P, REAL),
        camera_planes=np.zeros((P, 6), REAL), ray_planes=np.zeros((P, 2), REAL),
        ts=np.zeros(P, REAL), normals=np.zeros((P, 3), REAL), cov3D=np.zeros((P, 6), REAL),
        rgb=np.zeros((P, 3), REAL), conic_opacity=np.zeros((P, 4), REAL),
        is_tongue=np.zeros(P, REAL), tiles_touched=np.zeros(P, np.uint32),
        invraycov=np.zeros((P, 6), REAL), condition=np.zeros(P, np.uint8))
    L.ed3ref_preprocess(
        C.c_int(P), C.c_int(int(degree)), C.c_int(M), _p(means3D), _p(scales), CREAL(scale_modifier),
        _p(rotations), _p(opacities), None, _p(sh), _p(cov3D_precomp), _p(colors_precomp), _p(view),
        _p(proj), _p(campos), C.c_int(W), C.c_int(H), CREAL(tanfovx), CREAL(tanfovy),
        CREAL(kernel_size), _p(g["clamped"]), _p(g["radii"]), _p(g["means2D"]), _p(g["view_points"]),
        _p(g["depths"]), _p(g["camera_planes"]), _p(g["ray_planes"]), _p(g["ts"]), _p(g["normals"]),
        _p(g["cov3D"]), _p(g["rgb"]), _p(g["conic_opacity"]), _p(g["is_tongue"]), _p(g["tiles_touched"]),
        _p(g["invraycov"]), _p(g["condition"]))
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy
    bit = int(L.ed3ref_higher_msb(C.c_uint32(T)))

    def binned(n, keys_u, vals_u):
        keys = np.zeros(n, np.uint64); lst = np.zeros(n, np.uint32); ranges = np.zeros((T, 2), np.uint32)
        L.ed3ref_sort_pairs(C.c_int64(n), _p(keys_u), _p(vals_u), _p(keys), _p(lst), C.c_int(32 + bit))
        L.ed3ref_identify_tile_ranges(C.c_int64(n), _p(keys), _p(ranges), C.c_int(T))
        return lst, ranges

    offsets = np.cumsum(g["tiles_touched"].astype(np.uint64)).astype(np.uint32)
    R = int(offsets[-1]) if P else 0
    keys_u = np.zeros(R, np.uint64); vals_u = np.zeros(R, np.uint32)
    L.ed3ref_duplicate_with_keys(C.c_int(P), _p(g["means2D"]), _p(g["depths"]), _p(offsets), _p(g["radii"]),
                                 C.c_int(W), C.c_int(H), _p(keys_u), _p(vals_u))
    point_list, ranges = binned(R, keys_u, vals_u)
    # query points
    q = dict(points2D=np.zeros((PN, 2), REAL), depths=np.zeros(PN, REAL), tiles_touched=np.zeros(PN, np.uint32))
    L.ed3ref_preprocess_points(C.c_int(PN), _p(points3D), _p(view), C.c_int(W), C.c_int(H), CREAL(focal_x),
                               CREAL(focal_y), _p(q["points2D"]), _p(q["depths"]), _p(q["tiles_touched"]))
    qoff = np.cumsum(q["tiles_touched"].astype(np.uint64)).astype(np.uint32)
    NI = int(qoff[-1]) if PN else 0
    qk = np.zeros(NI, np.uint64); qv = np.zeros(NI, np.uint32)
    L.ed3ref_create_with_keys(C.c_int(PN), _p(q["points2D"]), _p(q["depths"]), _p(qoff), _p(q["tiles_touched"]),
                              C.c_int(W), C.c_int(H), _p(qk), _p(qv))
    qlist, qranges = binned(NI, qk, qv)
    out = dict(
        out_color=np.zeros((9, H, W), REAL), accum_alpha=np.zeros((1, H, W), REAL),
        n_contrib=np.zeros((H, W), np.uint32), alpha_integrated=np.ones(PN, REAL),
        color_integrated=np.zeros((PN, 3), REAL), coordinate2d=np.zeros((PN, 2), REAL),
        sdf=np.full(PN, -1000.0, REAL), pix_margin=np.zeros((H, W), REAL),
        pt_margin=np.full(PN, 1e30, REAL))
    features = colors_precomp if colors_precomp is not None else g["rgb"]
    L.ed3ref_integrate(
        C.c_int(W), C.c_int(H), _p(ranges), _p(qranges), _p(point_list), _p(qlist), CREAL(focal_x), CREAL(focal_y),
        _p(q["points2D"]), _p(g["means2D"]), _p(features), _p(g["ray_planes"]), _p(g["invraycov"]), _p(q["depths"]),
        _p(g["ts"]), _p(g["conic_opacity"]), _p(g["condition"]), _p(bg), _p(out["accum_alpha"]), _p(out["n_contrib"]),
        _p(out["out_color"]), _p(out["alpha_integrated"]), _p(out["color_integrated"]), _p(out["coordinate2d"]),
        _p(out["sdf"]), _p(out["pix_margin"]), _p(out["pt_margin"]))
    out.update(num_rendered=R, num_integrated=NI, ranges=ranges, point_list=point_list, point_valid=q["tiles_touched"] > 0,
               points2D=q["points2D"], point_depths=q["depths"])
    out.update(g)
    return out
