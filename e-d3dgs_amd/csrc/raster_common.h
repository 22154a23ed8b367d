// raster_common.h -- pieces shared by the forward and backward tile kernels.
//
// Tile kernel geometry (both directions): ONE 64-lane wavefront per 16x16 tile, QUADRANT-MAJOR: the 16 lanes of DPP row q
// (lanes 16 q .. 16 q + 15) own the 8x8 quadrant (q & 1, q >> 1) of the tile, lane li of the row the 4 horizontally adjacent
// pixels (4 (li & 1) .. + 3, li >> 1) of it -- every image access is a 16-byte vector per lane, a 32-byte row segment per two
// lanes.  The tile's depth-sorted instance list is consumed in chunks of 64 records that the wave gathers (one 64-byte record
// per lane) into LDS.  Round 3: the four quadrants then walk the chunk INDEPENDENTLY -- each its own sub-list of the entries
// that can reach alpha >= 1/255 inside its 8x8 box, in list order -- so that one inner-loop iteration blends up to four
// different Gaussians, one per quadrant.  Measured on C3 before the change (profiles/r03_tile_lane_occupancy.md): an entry the
// tile-level reject keeps blends in 2.74 of the tile's 4 quadrants on average (all four: 42 %), and 115 of the 256 pixel
// slots of a visited iteration blend anything.  Nothing per pixel changes (same operations in the same list order), only
// which lanes idle.
#pragma once
#include "common.h"

namespace ed3 {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Per-(lane, Gaussian) evaluation of alpha, shared by forward and backward so that both directions take bit-identical skip
// decisions (contraction is pinned).  Two forms:
//
// ED3_EXACT_ALPHA = 1 (default).  The blend decisions (alpha >= 1/255, T (1 - alpha) < 1e-4, T > 0.5) are discontinuities of the
// output, and an fp32 evaluation of the conic quadratic is ill-conditioned for elongated Gaussians: two correct evaluations in
// different operation orders differ by up to 1e-5 relative in alpha (measured: tools/alpha_discrepancy.py), so wherever a
// decision sits that close to its threshold two implementations disagree by a whole 1/255-sized term.  This form therefore
// evaluates `power` in EXACTLY the reference's operation order (CR/forward.cu:682:
//     power = -0.5f * (con_o.x * d.x * d.x + con_o.z * d.y * d.y) - con_o.y * d.x * d.y
// every product and sum rounded once, left to right; the factor -0.5 is exact, so the last two operations are one fused
// multiply-subtract with the same single rounding) and exp(power) as exp2 of a two-word product: hi = fl(power * log2e),
// lo = the exact remainder of that product plus power * (log2e - fl(log2e)), exp(power) = 2^hi (1 + lo ln 2) -- within
// 2 ulp of a correctly rounded expf for every power the blend loop sees, which is what the oracle's libm delivers.  alpha then
// agrees with the oracle's to 2.4e-7, decisions are taken identically unless they sit within 1e-6 of their thresholds (the
// parity tests' exclusion margin, tests/test_raster_parity_gpu.py), and T, a product of identical operations on those alphas,
// follows.  Cost: 6 + 5 vector operations per pixel and Gaussian instead of 2.
//
// ED3_EXACT_ALPHA = 0.  The round-1 form: power * log2(e) = dx (a dx + b dy) + c dy^2 with log2(e) folded into the coefficients
// (2 FMA per pixel, the row terms shared by a lane's pixels) and a bare v_exp_f32.  Same quadratic, another rounding sequence:
// needs an exclusion margin of 2e-5.
#ifndef ED3_EXACT_ALPHA
#define ED3_EXACT_ALPHA 1
#endif
#if ED3_EXACT_ALPHA
struct ConicRow {
    float cx, cy, t2, dy;   // t2 = (cz * dy) * dy: the row term of the lane
};
__device__ __forceinline__ ConicRow conic_row(float cx, float cy, float cz, float dy)
{
#pragma clang fp contract(off)
    ConicRow r;
    r.cx = cx; r.cy = cy; r.dy = dy;
    r.t2 = (cz * dy) * dy;
    return r;
}
// returns power (natural-log domain), rounded as the reference rounds it
__device__ __forceinline__ float conic_power2(const ConicRow &r, float dx)
{
#pragma clang fp contract(off)
    const float t1 = (r.cx * dx) * dx;
    const float s = t1 + r.t2;
    const float t3 = (r.cy * dx) * r.dy;
    return __builtin_fmaf(-0.5f, s, -t3);   // -0.5 s is exact: one rounding, as in (-0.5f * s) - t3
}
// G = exp(power) to ~1.5 ulp: 2^hi (1 + lo ln 2), hi + lo = power * log2(e) to 2^-48
__device__ __forceinline__ float gauss_G(float power)
{
    constexpr float L_HI = 1.4426950408889634f;                                    // fl(log2 e)
    constexpr float L_LO = (float)(1.4426950408889634073599 - (double)L_HI);       // log2 e - fl(log2 e)
    constexpr float LN2 = 0.6931471805599453f;
    const float hi = power * L_HI;
    float lo = __builtin_fmaf(power, L_HI, -hi);
    lo = __builtin_fmaf(power, L_LO, lo);
    const float e = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(e, lo * LN2, e);
}
#else
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
__device__ __forceinline__ float gauss_G(float power2) { return __builtin_amdgcn_exp2f(power2); }
#endif
// alpha = min(0.99, w*G)  (CR/forward.cu:692)
__device__ __forceinline__ float gauss_alpha(float w, float G)
{
#pragma clang fp contract(off)
    return fminf(0.99f, w * G);
}

// The same three steps for a PAIR of the lane's pixels, written on f32x2 so that they issue as v_pk_mul / v_pk_add / v_pk_fma_f32
// (full rate: two results per issue slot; only v_exp_f32 and the min stay scalar).  Component for component the operations and
// their order are those of conic_power2 / gauss_G / gauss_alpha above -- IEEE per component, so the results are bit-identical
// to the scalar forms (the forward and the backward both use this one).
struct AlphaPair {
    f32x2 power, G, alpha;
};
// The row constants as the packed operands of alpha_pair, formed ONCE per (Gaussian, lane) -- round 4: inside alpha_pair they were
// formed per pixel pair (the asm statement cannot be merged), four v_mov_b32 per call.
struct ConicSplat {
#if ED3_EXACT_ALPHA
    f32x2 cx2, cy2, t22, dy2;
#else
    ConicRow r;
#endif
};
__device__ __forceinline__ ConicSplat conic_splat(const ConicRow &r)
{
    ConicSplat o;
#if ED3_EXACT_ALPHA
    // explicit splats of values the optimiser cannot trace back to the record's float4 (it otherwise gathers them into a
    // 4-vector and extracts the register pairs of the packed operands THROUGH SCRATCH MEMORY, inside the blend loop)
    float cx = r.cx, cy = r.cy, t2 = r.t2, dy = r.dy;
    asm volatile("" : "+v"(cx), "+v"(cy), "+v"(t2), "+v"(dy));
    o.cx2 = f32x2{cx, cx}; o.cy2 = f32x2{cy, cy}; o.t22 = f32x2{t2, t2}; o.dy2 = f32x2{dy, dy};
#else
    o.r = r;
#endif
    return o;
}
__device__ __forceinline__ AlphaPair alpha_pair(const ConicSplat &c, float w, f32x2 dx)
{
#pragma clang fp contract(off)
    AlphaPair o;
#if ED3_EXACT_ALPHA
    constexpr float L_HI = 1.4426950408889634f;
    constexpr float L_LO = (float)(1.4426950408889634073599 - (double)L_HI);
    constexpr float LN2 = 0.6931471805599453f;
    const f32x2 cx2 = c.cx2, cy2 = c.cy2, t22 = c.t22, dy2 = c.dy2;
    const f32x2 t1 = (cx2 * dx) * dx;
    const f32x2 s = t1 + t22;
    const f32x2 t3 = (cy2 * dx) * dy2;
    o.power = __builtin_elementwise_fma(f32x2{-0.5f, -0.5f}, s, -t3);
    const f32x2 hi = o.power * L_HI;
    f32x2 lo = __builtin_elementwise_fma(o.power, f32x2{L_HI, L_HI}, -hi);
    lo = __builtin_elementwise_fma(o.power, f32x2{L_LO, L_LO}, lo);
    const f32x2 e = {__builtin_amdgcn_exp2f(hi.x), __builtin_amdgcn_exp2f(hi.y)};
    o.G = __builtin_elementwise_fma(e, lo * LN2, e);
#else
    const ConicRow &r = c.r;
    const f32x2 inner = __builtin_elementwise_fma(f32x2{r.a, r.a}, dx, f32x2{r.bdy, r.bdy});
    o.power = __builtin_elementwise_fma(dx, inner, f32x2{r.cdy2, r.cdy2});
    o.G = f32x2{__builtin_amdgcn_exp2f(o.power.x), __builtin_amdgcn_exp2f(o.power.y)};
#endif
    const f32x2 araw = w * o.G;
    o.alpha = f32x2{fminf(0.99f, araw.x), fminf(0.99f, araw.y)};
    return o;
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

// the four quadrant answers of one list entry as a 4-bit mask (bit q = quadrant q), for the entry's staging lane
__device__ __forceinline__ unsigned quadrants_may_contribute(float mx, float my, float cx, float cy, float cz, float w, float tile_x0,
                                                             float tile_y0)
{
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float x0 = tile_x0 + 8.f * (float)(q & 1), y0 = tile_y0 + 8.f * (float)(q >> 1);
        m |= tile_may_contribute(mx, my, cx, cy, cz, w, x0, y0, x0 + 7.f, y0 + 7.f) ? (1u << q) : 0u;
    }
    return m;
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
