"""oracle/deformation_ref.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy, float32 with a float64 switch) of the reference's per-Gaussian deformation network,
scene/deformation.py:15-141, including its hand-derived backward (the reference uses torch autograd).
Pinned by tests/golden/deform_*.npz, which tools/gen_golden.py produced by importing the reference's own
scene/deformation.py in the authoring container (values and autograd gradients).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

HEADS = ("pos", "scales", "rotations", "opacity", "rgb")


def _stage_names(s):
    return {"trunk": f"feature_out_{s}", **{h: f"{h}_deform_{s}" for h in HEADS}}


def int_lininterp(t, init_val, final_val, until):
    """scene/deformation.py:69-70"""
    return int(init_val + (final_val - init_val) * min(max(t, 0), until) / until)


def resize_rows(weight, n):
    """F.interpolate(weight[None,None], size=(n, TD), mode='bilinear', align_corners=True)  (:54-56).
    The width is unchanged, so with align_corners the column map is the identity and rows are lerped."""
    E, TD = weight.shape
    dt = weight.dtype
    if n == 1:
        src = np.zeros(1, dt)
    else:
        scale = dt.type(E - 1) / dt.type(n - 1)
        src = (np.arange(n).astype(dt) * scale).astype(dt)
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, E - 1)
    i1 = np.minimum(i0 + 1, E - 1)
    l1 = (src - i0.astype(dt)).astype(dt)
    l0 = (dt.type(1) - l1).astype(dt)
    return (l0[:, None] * weight[i0] + l1[:, None] * weight[i1]).astype(dt)


def _reflect(x, lo, hi):
    """grid_sample padding_mode='reflection', align_corners=True: reflect about [lo, hi] = [0, size-1]."""
    if lo == hi:
        return np.zeros_like(x)
    span = hi - lo
    x = np.abs(x - lo)
    extra = np.mod(x, span)
    flips = np.floor(x / span)
    return np.where(np.mod(flips, 2) == 0, extra + lo, span - extra + lo).astype(x.dtype)


def temporal_embed(weight, t, n):
    """get_temporal_embed (:53-67): row-resize to n rows, then bilinear grid_sample (align_corners=True,
    reflection padding) at y = t, x = j/(TD-1).  Returns the (TD,) vector every Gaussian of the frame shares."""
    dt = weight.dtype
    emb = resize_rows(weight, n)
    TD = weight.shape[1]
    # grid construction of the reference, in the storage dtype: (j/(TD-1) - 0.5) * 2 ; (t - 0.5) * 2
    gx = ((np.arange(TD).astype(dt) / dt.type(TD - 1)) - dt.type(0.5)) * dt.type(2)
    gy = (dt.type(t) - dt.type(0.5)) * dt.type(2)
    ix = ((gx + dt.type(1)) / dt.type(2)) * dt.type(TD - 1)
    iy = ((gy + dt.type(1)) / dt.type(2)) * dt.type(n - 1)
    ix = np.clip(_reflect(ix, 0, TD - 1), 0, TD - 1).astype(dt)
    iy = np.clip(_reflect(np.asarray([iy], dt), 0, n - 1), 0, n - 1).astype(dt)[0]
    x0 = np.floor(ix); y0 = np.floor(iy)
    wx1 = (ix - x0).astype(dt); wx0 = (dt.type(1) - wx1).astype(dt)
    wy1 = dt.type(iy - y0); wy0 = dt.type(1) - wy1
    x0i = x0.astype(np.int64); x1i = x0i + 1
    y0i = int(y0); y1i = y0i + 1

    def at(yy, xx):
        ok = (xx >= 0) & (xx < TD) & (0 <= yy < n)
        return np.where(ok, emb[min(max(yy, 0), n - 1), np.clip(xx, 0, TD - 1)], dt.type(0))

    out = at(y0i, x0i) * wx0 * wy0 + at(y0i, x1i) * wx1 * wy0 + at(y1i, x0i) * wx0 * wy1 + at(y1i, x1i) * wx1 * wy1
    return out.astype(dt)


def temporal_embed_lerp_coefs(E, t, n):
    """The same sample written as a combination of <= 4 rows of the original table (ignoring the ~1e-5 column
    cross-talk of grid_sample's fp32 x coordinate): returns (rows, coefs) with h = sum_k coefs[k] * weight[rows[k]]."""
    y = float(t) * (n - 1)
    if n > 1:
        span = n - 1
        y = abs(y)
        extra = y % span
        y = extra if int(y // span) % 2 == 0 else span - extra
    else:
        y = 0.0
    y = min(max(y, 0.0), n - 1)
    y0 = int(np.floor(y)); y1 = min(y0 + 1, n - 1)
    wy1 = y - y0
    rows, coefs = [], []
    for yy, wy in ((y0, 1.0 - wy1), (y1, wy1)):
        src = yy * (E - 1) / (n - 1) if n > 1 else 0.0
        i0 = min(int(np.floor(src)), E - 1); i1 = min(i0 + 1, E - 1)
        l1 = src - i0
        rows += [i0, i1]
        coefs += [wy * (1.0 - l1), wy * l1]
    return rows, coefs


def _linear(x, w, b):
    return x @ w.T + b


def _relu(x):
    return np.maximum(x, 0)


class Args:
    """The attributes of `args` that deform_network reads (arguments/__init__.py:85-108)."""

    def __init__(self, **kw):
        d = dict(temporal_embedding_dim=256, gaussian_embedding_dim=32, c2f_temporal_iter=20000, zero_temporal=False,
                 no_ds=False, no_dr=False, no_do=True, no_dc=False, use_coarse_temporal_embedding=False,
                 no_c2f_temporal_embedding=False, no_coarse_deform=False, no_fine_deform=False, use_anneal=True,
                 deform_from_iter=0)
        d.update(kw)
        self.__dict__.update(d)


def anneal_coefs(args, it):
    """scene/deformation.py:119-123"""
    if not args.use_anneal:
        return 1.0, 1.0, 1.0, 1.0
    coef = float(np.clip(it / 1000, 0, 1))
    c = float(np.clip((it - args.deform_from_iter) / 1000, 0, 1))
    return coef, c, c, c


def row_counts(args, it, max_embeddings, num_down_emb_c, num_down_emb_f):
    """query_time (:72-80): rows of the resized temporal table for the coarse and the fine stage."""
    def fine(nd):
        if args.no_c2f_temporal_embedding:
            return max_embeddings
        return int_lininterp(it, nd, max_embeddings, args.c2f_temporal_iter)
    n_c = num_down_emb_c if args.use_coarse_temporal_embedding else fine(num_down_emb_c)
    n_f = fine(num_down_emb_f)
    return n_c, n_f


def time_offset(offsets, cam_no):
    """scene/deformation.py:112-117"""
    if cam_no is None:
        nz = offsets[offsets != 0]
        return float(nz.mean()) if nz.size else 0.0
    return float(offsets[cam_no, 0])


def stage_forward(sd, s, args, D, h_t, emb, keep=False):
    """One stage: trunk + 5 heads (:38-51, :85-106). Returns dict of head outputs (None if the head is disabled)
    and, if keep, the activations the backward needs."""
    names = _stage_names(s)
    P = emb.shape[0]
    x = np.concatenate([np.broadcast_to(h_t[None, :], (P, h_t.shape[0])), emb], axis=1)
    acts = {"x": x, "trunk_pre": []}
    hid = _linear(x, sd[names["trunk"] + ".0.weight"], sd[names["trunk"] + ".0.bias"])
    li = 0
    for i in range(max(D - 1, 0)):
        acts["trunk_pre"].append(hid)
        li = 2 * (i + 1)
        hid = _linear(_relu(hid), sd[f"{names['trunk']}.{li}.weight"], sd[f"{names['trunk']}.{li}.bias"])
    acts["hid"] = hid
    out = {}
    enabled = {"pos": True, "scales": not args.no_ds, "rotations": not args.no_dr, "opacity": not args.no_do,
               "rgb": not args.no_dc}
    for hname in HEADS:
        if not enabled[hname]:
            out[hname] = None
            continue
        n = names[hname]
        a = _relu(hid)
        z = _linear(a, sd[n + ".1.weight"], sd[n + ".1.bias"])
        y = _linear(_relu(z), sd[n + ".3.weight"], sd[n + ".3.bias"])
        out[hname] = y
        acts[hname] = z
    return (out, acts) if keep else out


def forward(sd, args, D, max_embeddings, xyz, scales, rot, opacity, sh, emb, time, cam_no, it, num_down_emb_c,
            num_down_emb_f, dtype=np.float32, keep=False):
    """deform_network.forward (:108-141).  sd: state-dict as numpy arrays.  Returns
    (final 5-tuple, coarse 5-tuple, aux)."""
    f = lambda a: None if a is None else np.asarray(a, dtype)
    sd = {k: f(v) for k, v in sd.items()}
    xyz, scales, rot, opacity, sh, emb = f(xyz), f(scales), f(rot), f(opacity).reshape(-1, 1), f(sh), f(emb)
    t = dtype(time) + dtype(time_offset(sd["offsets"], cam_no))
    coef, coef_c, coef_o, coef_s = [dtype(c) for c in anneal_coefs(args, it)]
    n_c, n_f = row_counts(args, it, max_embeddings, num_down_emb_c, num_down_emb_f)
    cur = [xyz, scales, rot, opacity, sh]
    aux = {"n": (n_c, n_f), "t": t, "h": {}, "acts": {}}
    stages = []
    if not args.no_coarse_deform:
        stages.append(("c", n_c))
    sub = None
    for s, n in ([] if args.no_coarse_deform else [("c", n_c)]) + ([] if args.no_fine_deform else [("f", n_f)]):
        h_t = temporal_embed(sd["weight"], t, n)
        aux["h"][s] = h_t
        res = stage_forward(sd, s, args, D, h_t, emb, keep=keep)
        if keep:
            res, aux["acts"][s] = res
        p, sc, r, o, c = cur
        p = p + res["pos"] * coef
        if res["scales"] is not None:
            sc = sc + res["scales"] * coef * coef_s
        if res["rotations"] is not None:
            r = r + res["rotations"] * coef
        if res["opacity"] is not None:
            o = o + res["opacity"] * coef * coef_o
        if res["rgb"] is not None:
            c = c + res["rgb"].reshape(-1, 16, 3) * coef_c
        cur = [p, sc, r, o, c]
        if s == "c":
            sub = list(cur)
    if args.no_coarse_deform:
        sub = [xyz, scales, rot, opacity, sh]
    if args.no_fine_deform:
        cur = list(sub)
    return tuple(cur), tuple(sub), aux
