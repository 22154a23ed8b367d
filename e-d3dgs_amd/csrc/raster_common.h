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
