// devmath.h -- small fp32 vector/matrix helpers for the per-Gaussian kernels (device only).
// Matrices are column-major (m[c][r]) and products are summed in a fixed left-to-right order, so that with
// -ffp-contract=off the preprocess kernel's results are defined by the source alone (tile lists are compared
// bit-exactly against the CPU oracle).
#pragma once
#include <hip/hip_runtime.h>

namespace ed3 {

struct v3 { float x, y, z; };
struct m3 { float m[3][3]; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len3(v3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ v3 normalize3(v3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return a * inv; }

__device__ __forceinline__ m3 cols3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2)
{
    m3 r;
    r.m[0][0] = a0; r.m[0][1] = a1; r.m[0][2] = a2;
    r.m[1][0] = b0; r.m[1][1] = b1; r.m[1][2] = b2;
    r.m[2][0] = c0; r.m[2][1] = c1; r.m[2][2] = c2;
    return r;
}
__device__ __forceinline__ m3 zero3()
{
    return cols3(0, 0, 0, 0, 0, 0, 0, 0, 0);
}
__device__ __forceinline__ m3 mul3(const m3 &a, const m3 &b)
{
    m3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++)
            r.m[c][q] = a.m[0][q] * b.m[c][0] + a.m[1][q] * b.m[c][1] + a.m[2][q] * b.m[c][2];
    return r;
}
__device__ __forceinline__ m3 tr3(const m3 &a)
{
    m3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++) r.m[c][q] = a.m[q][c];
    return r;
}
__device__ __forceinline__ v3 mulv3(const m3 &a, v3 v)
{
    return mk3(a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z,
               a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z,
               a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z);
}
__device__ __forceinline__ m3 scale3(const m3 &a, float s)
{
    m3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++) r.m[c][q] = a.m[c][q] * s;
    return r;
}
__device__ __forceinline__ m3 div3(const m3 &a, float s)
{
    m3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++) r.m[c][q] = a.m[c][q] / s;
    return r;
}
__device__ __forceinline__ m3 add3(const m3 &a, const m3 &b)
{
    m3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++) r.m[c][q] = a.m[c][q] + b.m[c][q];
    return r;
}
__device__ __forceinline__ m3 neg3(const m3 &a)
{
    m3 r;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int q = 0; q < 3; q++) r.m[c][q] = -a.m[c][q];
    return r;
}
// column j of the result is c * r[j]  (outer product c r^T)
__device__ __forceinline__ m3 outer3(v3 c, v3 r)
{
    m3 o;
    o.m[0][0] = c.x * r.x; o.m[0][1] = c.y * r.x; o.m[0][2] = c.z * r.x;
    o.m[1][0] = c.x * r.y; o.m[1][1] = c.y * r.y; o.m[1][2] = c.z * r.y;
    o.m[2][0] = c.x * r.z; o.m[2][1] = c.y * r.z; o.m[2][2] = c.z * r.z;
    return o;
}
__device__ __forceinline__ v3 col3(const m3 &a, int c) { return mk3(a.m[c][0], a.m[c][1], a.m[c][2]); }

// matrix = 16 floats, column-major 4x4 (CR/auxiliary.h:74-113)
__device__ __forceinline__ v3 xform4x3(v3 p, const float *m)
{
    return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
               m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
               m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(v3 p, const float *m)
{
    float4 r;
    r.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    r.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    r.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    r.w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
    return r;
}
__device__ __forceinline__ v3 xformvec4x3T(v3 p, const float *m)
{
    return mk3(m[0] * p.x + m[1] * p.y + m[2] * p.z,
               m[4] * p.x + m[5] * p.y + m[6] * p.z,
               m[8] * p.x + m[9] * p.y + m[10] * p.z);
}

__device__ __forceinline__ bool feq(float x, float y, float eps) { return fabsf(x - y) <= eps; }
__device__ __forceinline__ float transfer_sign(float v, float s) { return (s >= 0) ? fabsf(v) : -fabsf(v); }
__device__ __forceinline__ float pythag(float a, float b)
{
    const float epsilon = 0.0000001f;
    float absa = fabsf(a), absb = fabsf(b);
    if (absa > absb) { absb /= absa; absb *= absb; return absa * sqrtf(1.0f + absb); }
    if (feq(absb, 0.0f, epsilon)) return 0.0f;
    absa /= absb; absa *= absa; return absb * sqrtf(1.0f + absa);
}

// Symmetric 3x3 eigen-decomposition: Householder reduction to tridiagonal form followed by implicit-shift QL,
// with the absolute 1e-7 convergence thresholds and the 30-iteration cap of the reference's solver
// (glm_modification::findEigenvaluesSymReal, CR/auxiliary.h:217-401).  The thresholds are behavioural (they decide
// how exact the eigenvectors of small covariances are), so the algorithm is kept, specialised here to N = 3 with the
// single Householder step written out.  Returns 3, or 0 if QL did not converge.  vec.m[k] = eigenvector k.
__device__ inline int eig_sym3(const m3 &cov, float val[3], m3 &vec)
{
    const float eps = 0.0000001f;
    // a[r][c] row-major working copy
    float a00 = cov.m[0][0], a01 = cov.m[1][0], a02 = cov.m[2][0];
    float a10 = cov.m[0][1], a11 = cov.m[1][1], a12 = cov.m[2][1];
    float a20 = cov.m[0][2], a21 = cov.m[1][2], a22 = cov.m[2][2];
    float d0, d1, d2, e0, e1, e2;
    // ---- Householder step for row 3 (i = 3, l = 2) ----
    {
        float h = 0.f, scale = 0.f;
        scale += fabsf(a20);
        scale += fabsf(a21);
        if (feq(scale, 0.0f, eps)) {
            e2 = a21;
        } else {
            a20 /= scale; h += a20 * a20;
            a21 /= scale; h += a21 * a21;
            float f = a21;
            float g = ((f >= 0) ? -sqrtf(h) : sqrtf(h));
            e2 = scale * g; h -= f * g; a21 = f - g; f = 0;
            // j = 1
            a02 = a20 / h;
            g = 0; g += a00 * a20; g += a10 * a21;
            e0 = g / h; f += e0 * a20;
            // j = 2
            a12 = a21 / h;
            g = 0; g += a10 * a20; g += a11 * a21;
            e1 = g / h; f += e1 * a21;
            float hh = f / (h + h);
            // j = 1
            f = a20; e0 = g = e0 - hh * f;
            a00 -= (f * e0 + g * a20);
            // j = 2
            f = a21; e1 = g = e1 - hh * f;
            a10 -= (f * e0 + g * a20);
            a11 -= (f * e1 + g * a21);
        }
        d2 = h;
    }
    // ---- i = 2, l = 1 ----
    e1 = a10;
    d1 = 0.f;
    d0 = 0.f; e0 = 0.f;
    // ---- accumulate the transformation ----
    // i = 1: l = 0
    d0 = a00; a00 = 1.f;
    // i = 2: l = 1
    if (!feq(d1, 0.0f, eps)) {
        float g = 0; g += a10 * a00; a00 -= g * a01;
    }
    d1 = a11; a11 = 1.f; a01 = a10 = 0.f;
    // i = 3: l = 2
    if (!feq(d2, 0.0f, eps)) {
        float g;
        g = 0; g += a20 * a00; g += a21 * a10; a00 -= g * a02; a10 -= g * a12;
        g = 0; g += a20 * a01; g += a21 * a11; a01 -= g * a02; a11 -= g * a12;
    }
    d2 = a22; a22 = 1.f; a02 = a20 = 0.f; a12 = a21 = 0.f;

    // ---- QL with implicit shifts on (d, e) ----
    float a[9] = {a00, a01, a02, a10, a11, a12, a20, a21, a22};
    float d[3] = {d0, d1, d2};
    float e[3] = {e1, e2, 0.f};  // shifted: e[i-2] = e[i-1]
    for (int l = 1; l <= 3; l++) {
        int iter = 0, m;
        do {
            for (m = l; m <= 2; m++) {
                if (feq(fabsf(e[m - 1]), 0.0f, eps)) break;
            }
            if (m != l) {
                if (iter++ == 30) return 0;
                float g = (d[l] - d[l - 1]) / (2 * e[l - 1]);
                float r = pythag(g, 1.0f);
                g = d[m - 1] - d[l - 1] + e[l - 1] / (g + transfer_sign(r, g));
                float s = 1.f, c = 1.f, p = 0.f;
                int i;
                for (i = m - 1; i >= l; i--) {
                    float f = s * e[i - 1];
                    float b = c * e[i - 1];
                    e[i] = r = pythag(f, g);
                    if (feq(r, 0.0f, eps)) { d[i] -= p; e[m - 1] = 0; break; }
                    s = f / r; c = g / r; g = d[i] - p;
                    r = (d[i - 1] - g) * s + 2 * c * b;
                    d[i] = g + (p = s * r);
                    g = c * r - b;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        f = a[k * 3 + i];
                        a[k * 3 + i] = s * a[k * 3 + i - 1] + c * f;
                        a[k * 3 + i - 1] = c * a[k * 3 + i - 1] - s * f;
                    }
                }
                if (feq(r, 0.0f, eps) && (i >= l)) continue;
                d[l - 1] -= p; e[l - 1] = g; e[m - 1] = 0;
            }
        } while (m != l);
    }
    val[0] = d[0]; val[1] = d[1]; val[2] = d[2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) vec.m[i][j] = a[j * 3 + i];
    return 3;
}

}  // namespace ed3
