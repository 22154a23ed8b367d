"""oracle/deformation_torch.py -- TEST INFRASTRUCTURE ONLY.

Differentiable PyTorch-CPU restatement of deform_network.forward (scene/deformation.py:108-141), used as the
gradient checker of the HIP deformation kernels at sizes the golden fixtures do not cover, and as the CPU baseline
of the deformation stage.  Pinned (values AND autograd gradients) by tests/golden/deform_*.npz.
The temporal row is written as the <=4-row lerp of oracle/deformation_ref.temporal_embed_lerp_coefs.
"""
import torch

from . import deformation_ref as R


def temporal_row(weight, t, n):
    """t: 0-d tensor (may require grad).  The row indices are piecewise constant in t; the lerp weight is linear with
    slope +-(n-1) (reflection flips the sign), which is how grid_sample back-propagates into the time offset."""
    tv = float(t)
    rows, coefs = R.temporal_embed_lerp_coefs(weight.shape[0], tv, n)
    c = torch.tensor(coefs, dtype=weight.dtype)
    h = (weight[rows] * c[:, None]).sum(0)
    if n > 1 and torch.is_tensor(t) and t.requires_grad:
        y = tv * (n - 1)
        span = n - 1
        sgn = -1.0 if y < 0 else 1.0
        if int(abs(y) // span) % 2 == 1:
            sgn = -sgn
        wa, wb = c[0] + c[1], c[2] + c[3]
        rowA = (weight[rows[0]] * c[0] + weight[rows[1]] * c[1]) / wa if float(wa) > 0 else weight[rows[0]]
        # resized rows at y0 / y1 (independent of the lerp weight)
        E = weight.shape[0]
        def resized(yy):
            src = yy * (E - 1) / (n - 1)
            i0 = min(int(src // 1), E - 1); i1 = min(i0 + 1, E - 1)
            l1 = src - i0
            return weight[i0] * (1 - l1) + weight[i1] * l1
        yv = abs(y) % span if int(abs(y) // span) % 2 == 0 else span - (abs(y) % span)
        y0 = int(yv // 1); y1 = min(y0 + 1, n - 1)
        dh = (resized(y1) - resized(y0)).detach() * (sgn * (n - 1))
        h = h + dh * (t - t.detach())
    return h


def forward(sd, args, D, max_embeddings, xyz, scales, rot, opacity, sh, emb, time, cam_no, it, num_down_emb_c,
            num_down_emb_f, margin_out=None):
    """sd: dict name -> tensor (requires_grad as the caller wishes).  Returns (final 5-tuple, coarse 5-tuple).
    margin_out (optional list): receives one (P,) tensor = the smallest |ReLU pre-activation| of each Gaussian over
    all hidden units; a Gaussian with a tiny margin sits on a ReLU kink, where its gradient is discontinuous."""
    off = sd["offsets"]
    if cam_no is None:
        nz = off[off != 0]
        offset = nz.mean() if nz.numel() else torch.zeros((), dtype=off.dtype)
    else:
        offset = off[cam_no, 0]
    t = offset + float(time) if torch.is_tensor(offset) else torch.tensor(float(time) + float(offset))
    coef, coef_c, coef_o, coef_s = R.anneal_coefs(args, it)
    n_c, n_f = R.row_counts(args, it, max_embeddings, num_down_emb_c, num_down_emb_f)
    cur = [xyz, scales, rot, opacity.reshape(-1, 1), sh]
    sub = list(cur)
    enabled = {"pos": True, "scales": not args.no_ds, "rotations": not args.no_dr, "opacity": not args.no_do,
               "rgb": not args.no_dc}
    P = emb.shape[0]
    margin = torch.full((P,), float("inf"), dtype=emb.dtype)
    for s, n, on in (("c", n_c, not args.no_coarse_deform), ("f", n_f, not args.no_fine_deform)):
        if on:
            h = temporal_row(sd["weight"], t, n)
            x = torch.cat([h[None, :].expand(P, -1), emb], dim=1)
            hid = torch.nn.functional.linear(x, sd[f"feature_out_{s}.0.weight"], sd[f"feature_out_{s}.0.bias"])
            for i in range(max(D - 1, 0)):
                li = 2 * (i + 1)
                margin = torch.minimum(margin, hid.detach().abs().min(dim=1).values)
                hid = torch.nn.functional.linear(torch.relu(hid), sd[f"feature_out_{s}.{li}.weight"], sd[f"feature_out_{s}.{li}.bias"])
            margin = torch.minimum(margin, hid.detach().abs().min(dim=1).values)
            res = {}
            for hn in R.HEADS:
                if not enabled[hn]:
                    res[hn] = None
                    continue
                z = torch.nn.functional.linear(torch.relu(hid), sd[f"{hn}_deform_{s}.1.weight"], sd[f"{hn}_deform_{s}.1.bias"])
                margin = torch.minimum(margin, z.detach().abs().min(dim=1).values)
                res[hn] = torch.nn.functional.linear(torch.relu(z), sd[f"{hn}_deform_{s}.3.weight"], sd[f"{hn}_deform_{s}.3.bias"])
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
    if margin_out is not None:
        margin_out.append(margin)
    return tuple(cur), tuple(sub)
