"""diff_gaussian_rasterization._C -- drop-in for the reference's pybind11 extension module.

Same four names, positional argument order, tuple arities and error behaviour as DGR/ext.cpp:15-20 /
DGR/rasterize_points.cu:35-59,139-172,252-271, implemented over the C-ABI HIP library (include/ed3dgs.h) with
torch used only to allocate device memory and to name the current stream.  No CPU path: tensors must live on the GPU.
"""
import ctypes as C

import torch

from ed3dgs_amd import _lib

NUM_CHANNELS = 3  # CR/config.h:15
# SURVEY quirk Q1 (CR/rasterizer_impl.cu:576): True = behave like the reference binary.
Q1_REFERENCE = True
# bench.py / tests: when KEEP_LAST is set, the most recent forward's (num_rendered, H, W, state buffers) stay reachable
KEEP_LAST = False
LAST = {}
_ZERO = {}   # device -> the 0-d zero behind the planes a variant does not produce


def _ptr(t):
    """Device pointer of a contiguous fp32/int32/u8 tensor; 0-element tensors are the reference's `nullptr`."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _f32c(t, name):
    if t is None:
        return None
    if t.numel() and not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (the MI355X path has no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _num(x, typ):
    # the reference's callers pass 0-d CUDA tensors for scalars (gaussian_renderer/__init__.py:29-37); pybind
    # converts them with an implicit .item().  Plain Python numbers take no device sync.
    return typ(x.item() if torch.is_tensor(x) else x)


# Capacity of the binning state per (device, P, H, W): the one per-frame allocation whose size depends on the DATA (the instance
# count R).  Asked for at its exact size it is a different request every frame, and the caching allocator serves it by splitting
# whatever large free block fits -- e.g. the 1.4-GB deformation workspace the previous step just returned -- after which that
# workspace's next request fits nowhere and goes to hipMalloc: round 4 counted 36 device mallocs inside 40 timed steps and the
# reserved memory growing from 3.6 to 10.7 GB, with steps of 7 ms among the 2.2-ms ones.  Asked for at a capacity that only ever
# grows (1.25 x the largest count seen for the frame size), the request repeats exactly and the allocator hands back the same block.
_BIN_CAPACITY = {}


class _Grow:
    """Growable byte buffer handed to the library as an allocation callback (resizeFunctional,
    DGR/rasterize_points.cu:27-33).  `cap_key`: allocate at the remembered capacity of that key (the binning state, above)."""

    def __init__(self, device, cap_key=None):
        self.device = device
        self.cap_key = cap_key
        self.t = torch.empty(0, dtype=torch.uint8, device=device)

    @property
    def cb(self):
        # Made per call and never stored on the object: a stored callback (-> bound method -> self -> callback) is a reference
        # cycle, and the state tensor in it then lives until the cyclic collector happens to run.  Round 4 measured what that
        # costs: ~245 MB of state per step waiting for the collector, so every step went to hipMalloc for its three buffers
        # (24 segment allocations in 8 steps), the reserved memory swung between 3.6 and 11.9 GB, and the timed region held
        # windows of 7 ms/step.  The caller keeps the returned object alive for the duration of the library call.
        return _lib.ALLOC_FN(self._alloc)

    def _alloc(self, _user, nbytes):
        n = int(nbytes)
        if self.cap_key is not None:
            cap = _BIN_CAPACITY.get(self.cap_key, 0)
            if n > cap:
                cap = _BIN_CAPACITY[self.cap_key] = (int(n * 1.25) + (1 << 21) - 1) & ~((1 << 21) - 1)
            n = cap
        self.t = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self.t.data_ptr()


def _stream():
    return _lib.raw_stream(torch.device("cuda", torch.cuda.current_device()))


def rasterize_gaussians(background, means3D, colors, opacity, tongue_class, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, image_height,
                        image_width, sh, degree, campos, prefiltered, require_coord, require_depth, debug):
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    L = _lib.lib()
    P = means3D.size(0)
    H, W = _num(image_height, int), _num(image_width, int)
    dev = means3D.device
    means3D = _f32c(means3D, "means3D")
    colors = _f32c(colors, "colors_precomp"); opacity = _f32c(opacity, "opacities")
    tongue_class = _f32c(tongue_class, "tongue_class"); scales = _f32c(scales, "scales")
    rotations = _f32c(rotations, "rotations"); cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp")
    sh = _f32c(sh, "sh"); background = _f32c(background, "bg")
    viewmatrix = _f32c(viewmatrix, "viewmatrix"); projmatrix = _f32c(projmatrix, "projmatrix")
    campos = _f32c(campos, "campos")
    rc, rd = bool(require_coord), bool(require_depth)
    fopt = dict(dtype=torch.float32, device=dev)
    geo = rc or rd
    run = P != 0
    def mk(c, written):
        if written and run:
            return torch.empty((c, H, W), **fopt)
        if run:
            # a plane the variant does not produce (the reference returns freshly filled zeros: 25 MB and a launch each at
            # 1080p): one zero (per device, made once) expanded to the shape.  READ-ONLY: reads as zeros everywhere, an in-place
            # write raises instead of aliasing; the library gets NULL for it, never the shared 4 bytes.
            z = _ZERO.get(dev)
            if z is None:
                z = _ZERO[dev] = torch.zeros((), **fopt)
            return z.expand(c, H, W)
        return torch.zeros((c, H, W), **fopt)
    out_color = mk(NUM_CHANNELS, True)
    out_depth, out_mdepth = mk(1, rd), mk(1, rd)
    out_coord, out_mcoord = mk(3, rc), mk(3, rc)
    out_alpha, out_tongue = mk(1, True), mk(1, True)
    out_normal = mk(3, geo)
    radii = (torch.empty if run else torch.zeros)((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Grow(dev), _Grow(dev, (str(dev), P, H, W)), _Grow(dev)
    rendered = 0
    if run:
        M = sh.size(1) if sh.numel() else 0
        rendered = L.ed3dgs_rasterize_forward(
            geom.cb, None, binning.cb, None, img.cb, None, C.c_int(P), C.c_int(_num(degree, int)), C.c_int(M),
            _ptr(background), C.c_int(W), C.c_int(H), _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity),
            _ptr(tongue_class), _ptr(scales), C.c_float(_num(scale_modifier, float)), _ptr(rotations),
            _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), C.c_float(_num(tan_fovx, float)),
            C.c_float(_num(tan_fovy, float)), C.c_float(_num(kernel_size, float)), C.c_int(bool(prefiltered)),
            _ptr(out_color), _ptr(out_coord) if rc else None, _ptr(out_mcoord) if rc else None,
            _ptr(out_depth) if rd else None, _ptr(out_mdepth) if rd else None, _ptr(out_alpha),
            _ptr(out_tongue), _ptr(out_normal) if geo else None, _ptr(radii), C.c_int(rc), C.c_int(rd), C.c_int(bool(debug)), _stream())
        if rendered < 0:
            raise RuntimeError(_lib.last_error())
    if KEEP_LAST:
        LAST.update(P=P, H=H, W=W, R=rendered, geom=geom.t, binning=binning.t, img=img.t)
    return (rendered, out_color, out_coord, out_mcoord, out_alpha, out_tongue, out_normal, out_depth, out_mdepth,
            radii, geom.t, binning.t, img.t)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, kernel_size, dL_dout_color,
                                 dL_dout_coord, dL_dout_mcoord, dL_dout_depth, dL_dout_mdepth, dL_dout_alpha,
                                 dL_dout_normal, normalmap, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, alphas, require_coord, require_depth, debug):
    L = _lib.lib()
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    dev = means3D.device
    means3D = _f32c(means3D, "means3D"); colors = _f32c(colors, "colors_precomp"); scales = _f32c(scales, "scales")
    rotations = _f32c(rotations, "rotations"); cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp")
    sh = _f32c(sh, "sh"); background = _f32c(background, "bg"); viewmatrix = _f32c(viewmatrix, "viewmatrix")
    projmatrix = _f32c(projmatrix, "projmatrix"); campos = _f32c(campos, "campos")
    rc, rd = bool(require_coord), bool(require_depth)
    geo = rc or rd
    # planes the variant does not produce have no upstream gradient the kernels read: NULL for the library, and no
    # .contiguous() copy of an expanded zero plane (3 x H x W floats for the normal map)
    used = (True, rc, rc, rd, rd, True, geo)
    grads = [_f32c(g, "grad") if u else None for g, u in zip((dL_dout_color, dL_dout_coord, dL_dout_mcoord, dL_dout_depth,
                                                              dL_dout_mdepth, dL_dout_alpha, dL_dout_normal), used)]
    normalmap = _f32c(normalmap, "normalmap") if geo else None
    alphas = _f32c(alphas, "alphas")
    M = sh.size(1) if sh.numel() else 0
    fopt = dict(dtype=torch.float32, device=dev)
    run = P != 0
    new = torch.empty if run else torch.zeros
    dL_dmeans3D = new((P, 3), **fopt); dL_dmeans2D = new((P, 3), **fopt); dL_dcolors = new((P, NUM_CHANNELS), **fopt)
    dL_dopacity = new((P, 1), **fopt); dL_dcov3D = new((P, 6), **fopt); dL_dsh = new((P, M, 3), **fopt)
    has_sr = scales.numel() != 0
    dL_dscales = (new if has_sr else torch.zeros)((P, 3), **fopt)
    dL_drotations = (new if has_sr else torch.zeros)((P, 4), **fopt)
    if run:
        ws_bytes = L.ed3dgs_backward_workspace_bytes(C.c_int(P), C.c_int(rc))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rcode = L.ed3dgs_rasterize_backward(
            C.c_int(P), C.c_int(_num(degree, int)), C.c_int(M), C.c_int(int(R)), _ptr(background), C.c_int(W),
            C.c_int(H), _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(alphas), _ptr(scales),
            C.c_float(_num(scale_modifier, float)), _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix),
            _ptr(projmatrix), _ptr(campos), C.c_float(_num(tan_fovx, float)), C.c_float(_num(tan_fovy, float)),
            C.c_float(_num(kernel_size, float)), _ptr(radii), _ptr(normalmap), _ptr(geomBuffer), _ptr(binningBuffer),
            _ptr(imageBuffer), _ptr(grads[0]), _ptr(grads[1]), _ptr(grads[2]), _ptr(grads[3]), _ptr(grads[4]),
            _ptr(grads[5]), _ptr(grads[6]), _ptr(dL_dmeans2D), _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D),
            _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales) if has_sr else None,
            _ptr(dL_drotations) if has_sr else None, _ptr(ws), C.c_size_t(ws_bytes), C.c_int(rc), C.c_int(rd),
            C.c_int(bool(Q1_REFERENCE)), C.c_int(bool(debug)), _stream())
        if rcode < 0:
            raise RuntimeError(_lib.last_error())
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)


def mark_visible(means3D, viewmatrix, projmatrix):
    L = _lib.lib()
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        means3D = _f32c(means3D, "means3D")
        viewmatrix = _f32c(viewmatrix, "viewmatrix"); projmatrix = _f32c(projmatrix, "projmatrix")
        rcode = L.ed3dgs_mark_visible(C.c_int(P), _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix),
                                      C.c_void_p(present.data_ptr()), _stream())
        if rcode < 0:
            raise RuntimeError(_lib.last_error())
    return present


def integrate_gaussians_to_points(background, points3D, means3D, colors, opacity, scales, rotations, scale_modifier,
                                  cov3D_precomp, view2gaussian_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                  kernel_size, subpixel_offset, image_height, image_width, sh, degree, campos,
                                  prefiltered, debug):
    """IntegrateGaussiansToPointsCUDA (DGR/rasterize_points.cu:273-392): same 23 positional arguments, same 10-tuple
    (num_rendered, out_color [9,H,W], alpha_integrated [PN], color_integrated [PN,3], coordinate2d [PN,2], sdf [PN],
    radii [P], geomBuffer, binningBuffer, imgBuffer).  `view2gaussian_precomp` and `subpixel_offset` are accepted and
    ignored, as the reference's kernels ignore their values."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if points3D.dim() != 2 or points3D.size(1) != 3:
        raise RuntimeError("points3D must have dimensions (num_points, 3)")
    L = _lib.lib()
    PN, P = points3D.size(0), means3D.size(0)
    H, W = _num(image_height, int), _num(image_width, int)
    dev = means3D.device
    means3D = _f32c(means3D, "means3D"); points3D = _f32c(points3D, "points3D")
    colors = _f32c(colors, "colors_precomp"); opacity = _f32c(opacity, "opacities"); scales = _f32c(scales, "scales")
    rotations = _f32c(rotations, "rotations"); cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp")
    sh = _f32c(sh, "sh"); background = _f32c(background, "bg")
    viewmatrix = _f32c(viewmatrix, "viewmatrix"); projmatrix = _f32c(projmatrix, "projmatrix")
    campos = _f32c(campos, "campos")
    fopt = dict(dtype=torch.float32, device=dev)
    out_color = torch.zeros((9, H, W), **fopt)
    accum_alpha = torch.zeros((1, H, W), **fopt)
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    out_alpha_integrated = torch.ones((PN,), **fopt)
    out_color_integrated = torch.zeros((PN, 3), **fopt)
    out_coordinate2d = torch.zeros((PN, 2), **fopt)
    out_sdf = torch.full((PN,), -1000.0, **fopt)
    invraycov = torch.zeros((P, 6), **fopt)
    condition = torch.zeros((P,), dtype=torch.uint8, device=dev)
    geom, binning, img, pts, ptsbin = _Grow(dev), _Grow(dev), _Grow(dev), _Grow(dev), _Grow(dev)
    rendered = 0
    if P != 0 and PN != 0:
        M = sh.size(1) if sh.numel() else 0
        rendered = L.ed3dgs_integrate(
            geom.cb, None, binning.cb, None, img.cb, None, pts.cb, None, ptsbin.cb, None, C.c_int(PN), C.c_int(P),
            C.c_int(_num(degree, int)), C.c_int(M), _ptr(background), C.c_int(W), C.c_int(H), _ptr(points3D), _ptr(means3D),
            _ptr(sh), _ptr(colors), _ptr(opacity), _ptr(scales), C.c_float(_num(scale_modifier, float)), _ptr(rotations),
            _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), C.c_float(_num(tan_fovx, float)),
            C.c_float(_num(tan_fovy, float)), C.c_float(_num(kernel_size, float)), C.c_int(bool(prefiltered)),
            _ptr(out_color), _ptr(accum_alpha), _ptr(invraycov), _ptr(radii), _ptr(out_alpha_integrated),
            _ptr(out_color_integrated), _ptr(out_coordinate2d), _ptr(out_sdf), _ptr(condition), C.c_int(bool(debug)),
            _stream())
        if rendered < 0:
            raise RuntimeError(_lib.last_error())
    if KEEP_LAST:
        LAST.update(P=P, H=H, W=W, R=rendered, geom=geom.t, binning=binning.t, img=img.t, invraycov=invraycov,
                    condition=condition, accum_alpha=accum_alpha)
    return (rendered, out_color, out_alpha_integrated, out_color_integrated, out_coordinate2d, out_sdf, radii, geom.t,
            binning.t, img.t)


def n_contrib_view(P, H, W, R, geomBuffer, imageBuffer):
    """(2,H,W) int32 GPU tensor aliasing the image state's n_contrib planes (last / median contributor)."""
    L = _lib.lib()
    sv = _lib.StateView()
    rc = L.ed3dgs_state_view_get(C.c_int(P), C.c_int(W), C.c_int(H), C.c_int(R), _ptr(geomBuffer), None,
                                 _ptr(imageBuffer), C.byref(sv))
    if rc < 0:
        raise RuntimeError(_lib.last_error())
    off = sv.n_contrib - imageBuffer.data_ptr()
    return imageBuffer[off:off + 2 * H * W * 4].view(torch.int32).reshape(2, H, W)


def state_view(P, H, W, R, geomBuffer, binningBuffer, imageBuffer):
    """Parity-test helper: typed tensor copies of the opaque state buffers (tile lists are compared bit-exactly)."""
    import numpy as np

    L = _lib.lib()
    sv = _lib.StateView()
    rc = L.ed3dgs_state_view_get(C.c_int(P), C.c_int(W), C.c_int(H), C.c_int(R), _ptr(geomBuffer),
                                 _ptr(binningBuffer) if R > 0 else None, _ptr(imageBuffer), C.byref(sv))
    if rc < 0:
        raise RuntimeError(_lib.last_error())
    T = ((W + 15) // 16) * ((H + 15) // 16)

    def view(buf, ptr, count, np_dtype):
        off = ptr - buf.data_ptr()
        nbytes = count * np.dtype(np_dtype).itemsize
        return buf[off:off + nbytes].cpu().numpy().view(np_dtype).copy()

    g, b, i = geomBuffer, binningBuffer, imageBuffer
    out = dict(
        rec=view(g, sv.rec, P * 16, np.float32).reshape(P, 16),
        rec_coord=view(g, sv.rec_coord, P * 12, np.float32).reshape(P, 12),
        depths=view(g, sv.depths, P, np.float32), cov3D=view(g, sv.cov3D, P * 6, np.float32).reshape(P, 6),
        clamped=view(g, sv.clamped, P, np.uint8), tiles_touched=view(g, sv.tiles_touched, P, np.uint32),
        point_offsets=view(g, sv.point_offsets, P, np.uint32), depth_order=view(g, sv.depth_order, P, np.uint32),
        ranges=view(i, sv.ranges, T * 2, np.uint32).reshape(T, 2),
        n_contrib=view(i, sv.n_contrib, 2 * H * W, np.uint32).reshape(2, H, W),
        accum_coord=view(i, sv.accum_coord, 3 * H * W, np.float32).reshape(3, H, W),
        accum_depth=view(i, sv.accum_depth, H * W, np.float32).reshape(1, H, W),
        normal_length=view(i, sv.normal_length, H * W, np.float32).reshape(1, H, W))
    if R > 0:
        out["keys"] = view(b, sv.point_list_keys, R, np.uint64)
        out["point_list"] = view(b, sv.point_list, R, np.uint32)
    else:
        out["keys"] = np.zeros(0, np.uint64)
        out["point_list"] = np.zeros(0, np.uint32)
    return out
