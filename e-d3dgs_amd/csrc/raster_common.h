// raster_common.h -- pieces shared by the forward and backward tile kernels.
//
// Tile kernel geometry (both directions): ONE 64-lane wavefront per 16x16 tile; lane l owns the 4 horizontally
// adjacent pixels (4*(l&3) .. +3, l>>2) of the tile, so every image access is a 16-byte vector per lane and a
// 64-byte row segment per 4 lanes.  The tile's depth-sorted instance list is consumed in chunks of 64 records that
// the wave gathers (one 64-byte record per lane) into LDS; the inner loop broadcasts one record per iteration.
#pragma once
#include "common.h"

namespace ed3 {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;

// Per-(lane, Gaussian) setup of the conic quadratic, shared by forward and backward so that both directions take
// bit-identical skip decisions (contraction is pinned: explicit fmaf only).
//   power*log2(e) = dx*(a*dx + b*dy) + c*dy*dy   with a = -0.5*cx*log2e, b = -cy*log2e, c = -0.5*cz*log2e
// (same quadratic as CR/forward.cu:682, evaluated in Horner form with the row term shared by the lane's 4 pixels).
struct ConicRow {
    float a, bdy, cdy2;
};
__device__ __forceinline__ ConicRow conic_row(float cx, float cy, float cz, float dy)
{
#pragma clang fp contract(off)
    ConicRow r;
    r.a = (-0.5f * LOG2E) * cx;
    float b = (-LOG2E) * cy;
    float c = (-0.5f * LOG2E) * cz;
    r.bdy = b * dy;
    r.cdy2 = (c * dy) * dy;
    return r;
}
// returns power*log2e
__device__ __forceinline__ float conic_power2(const ConicRow &r, float dx)
{
    return __builtin_fmaf(dx, __builtin_fmaf(r.a, dx, r.bdy), r.cdy2);
}
// G = exp(power), alpha = min(0.99, w*G)  (CR/forward.cu:692)
__device__ __forceinline__ float gauss_G(float power2) { return __builtin_amdgcn_exp2f(power2); }
__device__ __forceinline__ float gauss_alpha(float w, float G)
{
#pragma clang fp contract(off)
    return fminf(0.99f, w * G);
}

// Tile-level reject, evaluated by ONE lane per list entry while the chunk is staged: can this Gaussian reach
// alpha >= 1/255 anywhere in the pixel box [x0, x1] x [y0, y1]?  alpha = min(0.99, w exp(power)) with
// power(d) = -(0.5 cx dx^2 + cy dx dy + 0.5 cz dy^2), so the question is whether the minimum of the quadratic over
// the box is <= ln(255 w).  The minimum of a quadratic over a box that does not contain its stationary point lies on
// an edge, and on an edge it is a clamped 1-D minimisation (cx, cz > 0).  The answer is CONSERVATIVE (a margin covers
// the rounding of this test and of the per-pixel evaluation; degenerate or non-finite inputs answer yes): entries it
// rejects are exactly entries the per-pixel tests would reject for all 256 pixels, so skipping them changes no output
// bit -- the reference's 3-sigma SQUARE binning (CR/auxiliary.h:46-57) puts ~40 % such entries into the lists.
__device__ __forceinline__ bool tile_may_contribute(float mx, float my, float cx, float cy, float cz, float w,
                                                    float x0, float y0, float x1, float y1)
{
    if (!(cx > 0.f) || !(cz > 0.f) || !(w == w) || !(cy == cy) || !(mx == mx) || !(my == my)) return true;
    if (!(w > 0.f)) return false;                       // alpha <= 0 < 1/255 everywhere
    const float lim = __logf(255.0f * w);               // need min Q <= lim
    if (lim < -1e-3f) return false;                     // w < 1/255: alpha < 1/255 wherever power <= 0 (power > 0 is skipped)
    const float dxlo = mx - x1, dxhi = mx - x0, dylo = my - y1, dyhi = my - y0;
    if (dxlo <= 0.f && dxhi >= 0.f && dylo <= 0.f && dyhi >= 0.f) return true;
    const float ryc = -cy / cz, rxc = -cy / cx;
    float qmin = 3.0e38f;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const float xe = e ? dxhi : dxlo;
        const float ys = fminf(fmaxf(ryc * xe, dylo), dyhi);
        qmin = fminf(qmin, 0.5f * cx * xe * xe + cy * xe * ys + 0.5f * cz * ys * ys);
        const float ye = e ? dyhi : dylo;
        const float xs = fminf(fmaxf(rxc * ye, dxlo), dxhi);
        qmin = fminf(qmin, 0.5f * cx * xs * xs + cy * xs * ye + 0.5f * cz * ye * ye);
    }
    if (!(qmin == qmin)) return true;
    return qmin <= lim + 1e-3f + 1e-4f * fabsf(lim);
}

// 4-wide row-segment load/store for the lane's pixels. `vec` = whole segment inside and 16-byte aligned.
__device__ __forceinline__ void store4(float *__restrict__ plane, size_t pix0, const float v[4], bool vec, int nvalid)
{
    if (vec) {
        *reinterpret_cast<float4 *>(plane + pix0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        for (int p = 0; p < nvalid; p++) plane[pix0 + p] = v[p];
    }
}
__device__ __forceinline__ void store4u(uint32_t *__restrict__ plane, size_t pix0, const uint32_t v[4], bool vec, int nvalid)
{
    if (vec) {
        *reinterpret_cast<uint4 *>(plane + pix0) = make_uint4(v[0], v[1], v[2], v[3]);
    } else {
        for (int p = 0; p < nvalid; p++) plane[pix0 + p] = v[p];
    }
}
__device__ __forceinline__ void load4(const float *__restrict__ plane, size_t pix0, float v[4], bool vec, int nvalid)
{
    if (vec) {
        float4 t = *reinterpret_cast<const float4 *>(plane + pix0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        for (int p = 0; p < 4; p++) v[p] = (p < nvalid) ? plane[pix0 + p] : 0.f;
    }
}
__device__ __forceinline__ void load4u(const uint32_t *__restrict__ plane, size_t pix0, uint32_t v[4], bool vec, int nvalid)
{
    if (vec) {
        uint4 t = *reinterpret_cast<const uint4 *>(plane + pix0);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        for (int p = 0; p < 4; p++) v[p] = (p < nvalid) ? plane[pix0 + p] : 0u;
    }
}

}  // namespace ed3
