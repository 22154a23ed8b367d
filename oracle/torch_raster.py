"""oracle/torch_raster.py -- TEST INFRASTRUCTURE ONLY.

PyTorch-CPU *autograd* reimplementation of the reference rasterizer's math (the "PyTorch-CPU raster reference" of
BASELINE.json configs[0] / north_star): preprocess (CR/forward.cu:23-545), binning (CR/rasterizer_impl.cu:70-173)
and the tile compositing rule (CR/forward.cu:672-821) are written with differentiable torch ops, so torch.autograd
supplies the backward.  It is (a) an independent cross-check of oracle/raster_ref.c (forward values and gradients)
and (b) the timed CPU baseline of bench.py.  Differences from the C restatement, by construction:
  * the inverse of the 3D covariance.  The reference forms it from a TRUNCATED eigen-solver (glm's tred2 / tqli with
    absolute 1e-7 thresholds, CR/auxiliary.h:217-401): at covariance eigenvalues of ~1e-3 its eigenvectors stop ~1e-4..1e-3
    short of convergence, so the reference's plane / normal terms deviate SYSTEMATICALLY from exact linear algebra (measured:
    up to 2e-3 on unit normals, 1e-3 on ray planes; the fp32 and fp64 builds of the solver agree with each other to 5e-6).
    No exact method -- eigh or the closed-form spectrum of S R -- can therefore agree with the reference below ~1e-3.
    Two modes: `sigma_inv=None` uses torch.linalg.eigh (exact algebra: the CPU baseline of bench.py, and the loose
    cross-check); `sigma_inv=(P,3,3)` takes the truncated solver's inverse as DATA (the solver itself is pinned separately,
    tests/test_oracle_pins_cpu.py: an independent Python restatement + numpy) and differentiates through it with the
    perturbation rule d(S^-1) = -S^-1 dS S^-1 -- the rule the reference's K8 applies to its truncated inverse
    (CR/backward.cu:333-336) -- so that everything downstream is an independent autograd check at rounding level;
  * the alpha clamp min(0.99, w G) is straight-through in the backward, as in the reference (CR/backward.cu:852,978: dL_dG =
    w dL_dalpha whether or not the clamp was active);
  * quirk Q1 (CR/rasterizer_impl.cu:576) is NOT reproduced: autograd differentiates the true mip coefficient, which
    is what oracle.raster_oracle.backward(reference_q1=False) computes.
Extra leaves (`extra=`) expose the gradients autograd does not name: an NDC offset of the 2D mean (dL_dmeans2D.xy,
CR/backward.cu:1002-1003), an offset of the final per-Gaussian colour (dL_dcolors) and of the six covariance entries
(dL_dcov3D); the abs-grad column dL_dmeans2D.z (CR/backward.cu:1005-1006) is summed from the per-pair dL/dG that
`render(..., keep_pairs=True)` retains.
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]
TILE = 16


def eval_sh(deg, sh, dirs):
    """CR/forward.cu:23-74 (vectorised): sh (P,16,3), dirs (P,3) normalised -> rgb before clamping."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
        if deg > 2:
            res = (res + SH_C3[0] * y * (3 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                   + SH_C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                   + SH_C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                   + SH_C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res + 0.5


class _TruncatedInverse(torch.autograd.Function):
    """S^-1 taken as data (the reference's truncated solver's value); backward = the perturbation rule
    dL/dS = -S^-T (dL/dS^-1) S^-T that CR/backward.cu:333-336 applies to that same truncated inverse."""

    @staticmethod
    def forward(ctx, Sigma, Sinv):
        ctx.save_for_backward(Sinv)
        return Sinv.clone()

    @staticmethod
    def backward(ctx, g):
        (Sinv,) = ctx.saved_tensors
        St = Sinv.transpose(1, 2)
        return -(St @ g @ St), None


def cov6_to_mat(c):
    return torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)


def preprocess(means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
               kernel_size, scale_modifier, sh_degree, sigma_inv=None, cov3D_precomp=None, colors_precomp=None, extra=None):
    """Returns a dict of per-Gaussian screen-space quantities (differentiable) + integer binning data.
    extra: optional dict of zero leaves {"ndc": (P,2), "rgb": (P,3), "cov6": (P,6)} added to the 2D mean (in NDC), the
    final colour and the covariance entries."""
    dt = means3D.dtype
    extra = extra or {}
    P = means3D.shape[0]
    focal_y = H / (2.0 * tanfovy)
    focal_x = W / (2.0 * tanfovx)
    V = viewmatrix.reshape(4, 4)  # row-major storage of the transposed matrix: p_view = [p,1] @ V
    Pm = projmatrix.reshape(4, 4)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    p_view = (ph @ V)[:, :3]
    p_hom = ph @ Pm
    p_w = 1.0 / (p_hom[:, 3:4] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w
    in_front = p_view[:, 2] > 0.2

    if cov3D_precomp is not None:
        Sigma = cov6_to_mat(cov3D_precomp)                       # CR/forward.cu:484-493: used as given
    else:
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        # standard rotation matrix of the (un-normalised) quaternion; glm fills COLUMNS with these triples, i.e. the
        # glm matrix is this one transposed (CR/forward.cu:286-290)
        Rstd = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        S = torch.diag_embed(scale_modifier * scales)
        Sigma = Rstd @ S @ S @ Rstd.transpose(1, 2)  # = M^T M with M = S R_glm
    if "cov6" in extra:
        Sigma = Sigma + cov6_to_mat(extra["cov6"])   # off-diagonal leaves feed both symmetric entries (CR/backward.cu:426-431)

    t = p_view
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = t[:, 2]
    txtz = torch.clamp(t[:, 0] / tz, -limx, limx)
    tytz = torch.clamp(t[:, 1] / tz, -limy, limy)
    tx, ty = txtz * tz, tytz * tz
    zero = torch.zeros_like(tz)
    Jm = torch.stack([focal_x / tz, zero, -(focal_x * tx) / (tz * tz),
                      zero, focal_y / tz, -(focal_y * ty) / (tz * tz)], 1).reshape(P, 2, 3)
    Rm = V[:3, :3].t()  # world -> view rotation
    Tm = Jm @ Rm
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    c00, c01, c11 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    a, b, c = c00 + kernel_size, c01, c11 + kernel_size
    det0 = torch.clamp(c00 * c11 - c01 * c01, min=1e-6)
    det1 = torch.clamp(a * c - b * b, min=1e-6)
    coef = torch.sqrt(det0 / (det1 + 1e-6) + 1e-6)
    coef = torch.where((det0 <= 1e-6) | (det1 <= 1e-6), torch.zeros_like(coef), coef)

    # ray-space plane / normal (CR/forward.cu:135-260)
    if sigma_inv is not None:
        Sinv = _TruncatedInverse.apply(Sigma, sigma_inv)
    else:
        evals, evecs = torch.linalg.eigh(Sigma)
        well = evals[:, 0] > 0.00000001
        inv_w = evecs @ torch.diag_embed(1.0 / evals) @ evecs.transpose(1, 2)
        emin = evecs[:, :, 0]
        inv_i = emin[:, :, None] * emin[:, None, :]
        Sinv = torch.where(well[:, None, None], inv_w, inv_i)
    Cinv = Rm @ Sinv @ Rm.t()
    uvh = torch.stack([txtz, tytz, torch.ones_like(txtz)], 1)
    m = (Cinv @ uvh[:, :, None])[:, :, 0]
    mn = m / m.norm(dim=1, keepdim=True)
    u2, v2, uv = txtz * txtz, tytz * tytz, txtz * tytz
    l = torch.sqrt(tx * tx + ty * ty + tz * tz)
    nl = u2 + v2 + 1
    vbn = torch.clamp((mn * uvh).sum(1), min=0.0000001)
    aa = mn / vbn[:, None]
    plane0 = (v2 + 1) * aa[:, 0] - uv * aa[:, 1] - txtz * aa[:, 2]
    plane1 = -uv * aa[:, 0] + (u2 + 1) * aa[:, 1] - tytz * aa[:, 2]
    cam_plane = torch.stack([(-(v2 + 1) * tz + plane0 * tx) / nl / focal_x, (uv * tz + plane1 * tx) / nl / focal_y,
                             (uv * tz + plane0 * ty) / nl / focal_x, (-(u2 + 1) * tz + plane1 * ty) / nl / focal_y,
                             (tx + plane0 * tz) / nl / focal_x, (ty + plane1 * tz) / nl / focal_y], 1)
    ray_plane = torch.stack([plane0 * l / nl / focal_x, plane1 * l / nl / focal_y], 1)
    fn = l / nl
    rn = torch.stack([-plane0 * fn, -plane1 * fn, -torch.ones_like(fn)], 1)
    # nJ (glm columns (1/tz,0,-tx/tz^2), (0,1/tz,-ty/tz^2), (tx/l,ty/l,tz/l)) times rn
    cn = torch.stack([rn[:, 0] / tz + rn[:, 2] * tx / l,
                      rn[:, 1] / tz + rn[:, 2] * ty / l,
                      -(tx) / (tz * tz) * rn[:, 0] - (ty) / (tz * tz) * rn[:, 1] + rn[:, 2] * tz / l], 1)
    normal = cn / cn.norm(dim=1, keepdim=True)
    bad = torch.isnan(mn[:, 0])
    cam_plane = torch.where(bad[:, None], torch.zeros_like(cam_plane), cam_plane)
    ray_plane = torch.where(bad[:, None], torch.zeros_like(ray_plane), ray_plane)
    normal = torch.where(bad[:, None], torch.zeros_like(normal), normal)

    ts = p_view.norm(dim=1)
    det = a * c - b * b
    det_inv = 1.0 / det
    conic = torch.stack([c * det_inv, -b * det_inv, a * det_inv], 1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    ndc = p_proj[:, :2] + extra["ndc"] if "ndc" in extra else p_proj[:, :2]
    xy = torch.stack([((ndc[:, 0].double() + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1].double() + 1.0) * H - 1.0) * 0.5], 1).to(dt)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    rad_i = radius.detach().to(torch.int64)
    xyd = xy.detach()
    rminx = torch.clamp(((xyd[:, 0] - rad_i) / TILE).to(torch.int64), 0, gx)
    rminy = torch.clamp(((xyd[:, 1] - rad_i) / TILE).to(torch.int64), 0, gy)
    rmaxx = torch.clamp(((xyd[:, 0] + rad_i + TILE - 1) / TILE).to(torch.int64), 0, gx)
    rmaxy = torch.clamp(((xyd[:, 1] + rad_i + TILE - 1) / TILE).to(torch.int64), 0, gy)
    tiles = (rmaxx - rminx) * (rmaxy - rminy)
    visible = in_front & (det.detach() != 0) & (tiles > 0)

    if colors_precomp is not None:
        rgb = colors_precomp                                       # CR/forward.cu:528-533: no SH, no clamp
    else:
        dirs = means3D - campos[None, :]
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh(sh_degree, shs, dirs), 0.0)
    if "rgb" in extra:
        rgb = rgb + extra["rgb"]
    return dict(visible=visible, radii=torch.where(visible, rad_i, torch.zeros_like(rad_i)), xy=xy, conic=conic,
                w=opacities.reshape(-1) * coef, rgb=rgb, ts=ts, ray_plane=ray_plane, normal=normal, cam_plane=cam_plane,
                view_points=p_view, depth=p_view[:, 2].detach(), rect=(rminx, rminy, rmaxx, rmaxy),
                tiles=torch.where(visible, tiles, torch.zeros_like(tiles)), gx=gx, gy=gy, focal_x=focal_x,
                focal_y=focal_y, Sigma=Sigma)


def bin_tiles(pp):
    """Key emission + stable sort of (tile, depth-bits) keys -> per-tile index lists
    (CR/rasterizer_impl.cu:70-173): instances are emitted Gaussian by Gaussian, row-major inside the rect."""
    vis = torch.nonzero(pp["visible"]).reshape(-1)
    T = pp["gx"] * pp["gy"]
    if vis.numel() == 0:
        return torch.zeros(0, dtype=torch.int64), torch.zeros(T + 1, dtype=torch.int64)
    rminx, rminy, rmaxx, rmaxy = [r[vis] for r in pp["rect"]]
    w = rmaxx - rminx
    n = w * (rmaxy - rminy)
    rep = torch.repeat_interleave(torch.arange(vis.numel()), n)
    first = torch.cumsum(n, 0) - n
    j = torch.arange(int(n.sum())) - first[rep]
    tile_ids = (rminy[rep] + j // w[rep]) * pp["gx"] + rminx[rep] + j % w[rep]
    ids = vis[rep]
    dbits = pp["depth"][ids].float().view(torch.int32).to(torch.int64)
    order = torch.argsort((tile_ids << 32) | dbits, stable=True)  # ties keep emission order = Gaussian index
    ids = ids[order]; tile_ids = tile_ids[order]
    counts = torch.bincount(tile_ids, minlength=T)
    starts = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)])
    return ids, starts


def render(pp, ids, starts, bg, W, H, require_coord, require_depth, tile_subset=None, keep_pairs=False):
    """Tile compositing (CR/forward.cu:672-821) with torch ops; returns the 8 images (variant planes zero).
    keep_pairs: retain, per tile, (Gaussian ids, G = exp(power) with its gradient kept, dx, dy) so that abs_grad_means2D()
    can form the per-pair absolute sums of CR/backward.cu:1005-1006 after loss.backward()."""
    dt = pp["xy"].dtype
    geo = require_coord or require_depth
    gx, gy = pp["gx"], pp["gy"]
    out = {k: torch.zeros(c, H, W, dtype=dt) for k, c in (("color", 3), ("coord", 3), ("mcoord", 3), ("alpha", 1),
                                                          ("normal", 3), ("depth", 1), ("mdepth", 1))}
    out["color"] = out["color"] + bg.reshape(3, 1, 1)
    tiles = range(gx * gy) if tile_subset is None else tile_subset
    pieces = {k: [] for k in out}
    where = []
    npairs = 0
    kept = []
    for tidx in tiles:
        s, e = int(starts[tidx]), int(starts[tidx + 1])
        ty_, tx_ = divmod(tidx, gx)
        y0, x0 = ty_ * TILE, tx_ * TILE
        hh, ww = min(TILE, H - y0), min(TILE, W - x0)
        if s == e:
            continue
        g = ids[s:e]
        ys = torch.arange(y0, y0 + hh, dtype=dt); xs = torch.arange(x0, x0 + ww, dtype=dt)
        py, px = torch.meshgrid(ys, xs, indexing="ij")
        px = px.reshape(1, -1); py = py.reshape(1, -1)
        dx = pp["xy"][g, 0:1] - px
        dy = pp["xy"][g, 1:2] - py
        con = pp["conic"][g]
        power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
        G = torch.exp(power)
        if keep_pairs:
            G.retain_grad()
            kept.append((g, G, dx.detach(), dy.detach(), con.detach()))
        araw = pp["w"][g][:, None] * G
        alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()      # min(0.99, .), straight-through (see header)
        live = (power <= 0) & (alpha >= 1.0 / 255.0)
        a_eff = torch.where(live, alpha, torch.zeros_like(alpha))
        Tcum = torch.cumprod(1 - a_eff, 0)  # T after each entry (if it were blended)
        Tprev = torch.cat([torch.ones(1, a_eff.shape[1], dtype=dt), Tcum[:-1]], 0)
        # the pixel stops at the first live entry whose test_T < 1e-4; that entry and everything behind it is dropped
        stop = live & (Tcum < 0.0001)
        stopped = torch.cumsum(stop.to(torch.int32), 0) > 0
        blend = live & ~stopped
        aT = torch.where(blend, alpha * Tprev, torch.zeros_like(alpha))
        npairs += int(blend.sum())
        Tfinal = torch.where(blend, 1 - alpha, torch.ones_like(alpha)).prod(0)
        weight = aT.sum(0)
        col = aT.t() @ pp["rgb"][g] + Tfinal[:, None] * bg[None, :]
        res = {"color": col.t(), "alpha": weight[None, :]}
        if geo:
            any_c = blend.any(0)
            before = blend & (Tprev > 0.5)
            nb = before.to(torch.int64).sum(0)  # median = last entry with pre-blend T > 0.5 (T is monotone)
            last_idx = torch.clamp(nb - 1, min=0)
            has_med = nb > 0
            # index of the last "before" entry: entries are ordered, before-mask is a prefix of blended entries
            pos = torch.where(before, torch.arange(before.shape[0])[:, None].expand_as(before), torch.full_like(before, -1, dtype=torch.int64)).max(0).values
            pos = torch.clamp(pos, min=0)
            pnx = (px.reshape(-1) - W / 2.0) / pp["focal_x"]; pny = (py.reshape(-1) - H / 2.0) / pp["focal_y"]
            ln = torch.sqrt(pnx * pnx + pny * pny + 1)
            N = aT.t() @ pp["normal"][g]
            nlen = N.norm(dim=1)
            res["normal"] = torch.where(any_c[None, :], (N / torch.clamp(nlen, min=1e-12)[:, None]).t(), torch.zeros(3, N.shape[0], dtype=dt))
            if require_depth:
                tval = pp["ts"][g][:, None] + pp["ray_plane"][g, 0:1] * dx + pp["ray_plane"][g, 1:2] * dy
                D = (tval * aT).sum(0) / ln
                res["depth"] = torch.where(any_c, D / torch.where(any_c, weight, torch.ones_like(weight)), torch.zeros_like(D))[None, :]
                md = torch.gather(tval, 0, pos[None, :])[0]
                res["mdepth"] = (torch.where(has_med, md, torch.zeros_like(md)) / ln)[None, :]
            if require_coord:
                cp = pp["cam_plane"][g]; vp = pp["view_points"][g]
                cs = [vp[:, k:k + 1] + cp[:, 2 * k:2 * k + 1] * dx + cp[:, 2 * k + 1:2 * k + 2] * dy for k in range(3)]
                res["coord"] = torch.stack([torch.where(any_c, (c_ * aT).sum(0) / torch.where(any_c, weight, torch.ones_like(weight)), torch.zeros_like(weight)) for c_ in cs], 0)
                res["mcoord"] = torch.stack([torch.where(has_med, torch.gather(c_, 0, pos[None, :])[0], torch.zeros_like(weight)) for c_ in cs], 0)
        where.append((py.reshape(-1).to(torch.int64) * W + px.reshape(-1).to(torch.int64)))
        for k in res:
            pieces[k].append(res[k].reshape(res[k].shape[0], -1))
    # paste all tiles with one out-of-place index_copy per plane (differentiable w.r.t. the tile values)
    if where:
        idx = torch.cat(where)
        for k in out:
            if pieces[k]:
                C = out[k].shape[0]
                out[k] = torch.index_copy(out[k].reshape(C, -1), 1, idx, torch.cat(pieces[k], 1)).reshape(C, H, W)
    out["npairs"] = npairs
    if keep_pairs:
        out["pairs"] = kept
    return out


def abs_grad_means2D(pairs, P, W, H):
    """dL_dmeans2D[:, 2] (CR/backward.cu:1005-1006): sum over a Gaussian's pairs of |dL/dG dG/ddx W/2| + |dL/dG dG/ddy H/2|
    (the G-only part of the mean gradient).  Call after loss.backward(); `pairs` = render(..., keep_pairs=True)["pairs"]."""
    z = torch.zeros(P, dtype=pairs[0][1].dtype if pairs else torch.float64)
    for g, G, dx, dy, con in pairs:
        if G.grad is None:
            continue
        gdx, gdy = G.detach() * dx, G.detach() * dy
        dGdx = -gdx * con[:, 0:1] - gdy * con[:, 1:2]
        dGdy = -gdy * con[:, 2:3] - gdx * con[:, 1:2]
        z.index_add_(0, g, ((G.grad * dGdx * (0.5 * W)).abs() + (G.grad * dGdy * (0.5 * H)).abs()).sum(1))
    return z


def rasterize(bg, means3D, opacities, scales, rotations, shs, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
              kernel_size, H, W, sh_degree, require_coord, require_depth, scale_modifier=1.0, tile_subset=None,
              sigma_inv=None, cov3D_precomp=None, colors_precomp=None, extra=None, keep_pairs=False):
    pp = preprocess(means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
                    kernel_size, scale_modifier, sh_degree, sigma_inv=sigma_inv, cov3D_precomp=cov3D_precomp,
                    colors_precomp=colors_precomp, extra=extra)
    ids, starts = bin_tiles(pp)
    out = render(pp, ids, starts, bg, W, H, require_coord, require_depth, tile_subset, keep_pairs=keep_pairs)
    out["pp"] = pp
    out["radii"] = pp["radii"]
    out["ids"] = ids
    out["starts"] = starts
    return out
