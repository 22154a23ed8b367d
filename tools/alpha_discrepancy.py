"""How far apart two CORRECT fp32 evaluations of alpha = w exp(power) are near the 1/255 threshold (round 3).

CPU only (numpy emulation of fp32, FMA = one rounding of the float64 result): samples (Gaussian, pixel) pairs of a C2-like frame from
the oracle's own projected Gaussians and compares
  oracle   power = -0.5f (cx dx dx + cz dy dy) - cy dx dy  (CR/forward.cu:682, every operation rounded), expf
  round-2  power * log2(e) = dx (a dx + b dy) + c dy^2 with log2(e) folded into a, b, c (two FMA), exp2
  round-3  the oracle's operation order + exp2 of a two-word product (csrc/raster_common.h, ED3_EXACT_ALPHA)
against float64.  usage: python tools/alpha_discrepancy.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "e-d3dgs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import util  # noqa: E402

f32 = np.float32
inp = util.scene_inputs(100000, 1920, 1080, scene_seed=0, cam_seed=1)
fw = util.oracle_forward(inp, "FFF", with_margin=False)
co, m2, rad = fw["conic_opacity"], fw["means2D"], fw["radii"]
vis = np.where(rad > 0)[0]
rng = np.random.default_rng(0)
N = 4_000_000
g = vis[rng.integers(0, len(vis), N)]
r = rad[g].astype(f32)
px = np.floor(m2[g, 0] + (rng.random(N, dtype=f32) * 2 - 1) * r).astype(f32)
py = np.floor(m2[g, 1] + (rng.random(N, dtype=f32) * 2 - 1) * r).astype(f32)
cx, cy, cz, w = [co[g, i].astype(f32) for i in range(4)]
dx, dy = (m2[g, 0] - px).astype(f32), (m2[g, 1] - py).astype(f32)
d = lambda a: a.astype(np.float64)
pw_o = (f32(-0.5) * ((cx * dx) * dx + (cz * dy) * dy) - (cy * dx) * dy).astype(f32)
ar_o = (w * np.exp(d(pw_o)).astype(f32)).astype(f32)
L = f32(1.4426950408889634)
a, b, c = ((f32(-0.5) * L) * cx).astype(f32), ((-L) * cy).astype(f32), ((f32(-0.5) * L) * cz).astype(f32)
bdy, cdy2 = (b * dy).astype(f32), ((c * dy).astype(f32) * dy).astype(f32)
inner = (d(a) * d(dx) + d(bdy)).astype(f32)
p2 = (d(dx) * d(inner) + d(cdy2)).astype(f32)
ar_2 = (w * np.exp2(d(p2)).astype(f32)).astype(f32)
hi = (pw_o * L).astype(f32)
lo = (d(pw_o) * float(L) - d(hi)).astype(f32)
lo = (lo + pw_o * f32(1.4426950408889634 - float(L))).astype(f32)
e = np.exp2(d(hi)).astype(f32)
ar_3 = (w * (d(e) * (1 + d(lo) * 0.6931471805599453)).astype(f32)).astype(f32)
pw_t = -0.5 * (d(cx) * d(dx) ** 2 + d(cz) * d(dy) ** 2) - d(cy) * d(dx) * d(dy)
ar_t = d(w) * np.exp(pw_t)
sel = (ar_t > 0.3 / 255) & (ar_t < 3 / 255) & (pw_o <= 0)
print("pairs near the threshold:", int(sel.sum()))
for name, x, y in (("round-2 form vs oracle", ar_2, ar_o), ("oracle vs float64", ar_o, ar_t), ("round-2 form vs float64", ar_2, ar_t),
                   ("round-3 form vs oracle", ar_3, ar_o)):
    v = np.abs(d(x) - d(y))[sel] / ar_t[sel]
    print("%-26s median %.2e  p99 %.2e  p99.99 %.2e  max %.2e" % (name, np.median(v), np.quantile(v, 0.99), np.quantile(v, 0.9999), v.max()))
