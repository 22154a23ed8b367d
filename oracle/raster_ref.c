/*
 * oracle/raster_ref.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's differentiable EWA-splat rasterizer
 * (vladb99/E-D3DGS, submodules/diff-gaussian-rasterization = "DGR", cuda_rasterizer = "CR").
 * It exists to CHECK the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Nothing in the product path (e-d3dgs_amd/) may include, link or call it.
 *
 * PARITY STATUS: "parity unpinned by the reference" -- the reference holds no golden vectors, KATs or
 * tests for this path and its CUDA sources cannot be built or run in this pipeline (no nvcc, no NVIDIA
 * GPU).  This restatement is pinned instead by (tests/test_oracle_cpu.py, tests/test_oracle_pins_cpu.py):
 *   - hand-derived known answers (single Gaussian, alpha clamp / thresholds, mip coefficient, ordering + median, opaque
 *     stack, culls, quirk Q1 at kernel_size 0.3) and structural invariants;
 *   - its eigen-solver against a second, independent Python restatement (oracle/eig_ql_ref.py): bit-identical;
 *   - an independent PyTorch-autograd restatement (oracle/torch_raster.py) of forward + every returned gradient
 *     (incl. dL_dmeans2D x / y / abs-grad z, dL_dcolors, dL_dcov3D, the precomputed-colour / -covariance variant) on
 *     FFF / FTT / TFT / TTT at kernel_size 0 and 0.3: the fp64 build of this file (-DED3REF_FP64) agrees to < 1e-6, this
 *     fp32 build to < 1e-5 (images) / < 5e-4 (gradients; < 1e-5 once pixels with T_final < 1e-2 are left out -- the
 *     reference's restart from T_final = 1 - alpha_out amplifies fp32 rounding by 1 / T_final);
 *   - fp64 central differences of its own forward against its K7 / K8 / K9 (< 1e-7 with a converged eigen-solver).
 *
 * Every function cites the reference file:line it follows.  Arithmetic is fp32 with the same operand
 * order and the same float/double promotions as the C++ expressions of the reference (double literals such
 * as 1e-6 promote exactly as they do under nvcc); compile with -ffp-contract=off so no FMA is formed.
 * Matrices follow glm's column-major convention (m[c][r]), products are written in glm's summation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float f32_t;         /* the 32-bit type by name (depth bits of the sort keys), whatever `float` means below */
#ifdef ED3REF_FP64
/* libraster_ref64.so: THE SAME TEXT with every fp32 quantity carried in fp64 (arrays, arguments, arithmetic).  It is
 * not a statement about the reference's results -- those are fp32 -- but a tool for the tests: with rounding out of the
 * way, (1) central differences of this forward check the hand-derived backward (K7 / K8 / K9) to ~1e-7, and (2) the
 * independent autograd restatement (oracle/torch_raster.py, fp64) must agree with it to ~1e-9, so structural
 * disagreements cannot hide under fp32 noise.  oracle/raster_oracle64.py is its ctypes front-end. */
#define float double
#define fabsf fabs
#define sqrtf sqrt
#define fmaxf fmax
#define fminf fmin
#define expf exp
#define ceilf ceil
#endif

#define TILE 16              /* CR/config.h:16-17 BLOCK_X = BLOCK_Y = 16 */
#define CHUNK 256            /* CR/auxiliary.h:19 BLOCK_SIZE */
#define NORMALIZE_EPS 1.0E-12F /* CR/auxiliary.h:23 */

/* CR/auxiliary.h:35-52 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

/* bits of the fp32 depth that form the low half of a sort key (CR/rasterizer_impl.cu:99-103) */
static uint32_t depth_bits(float d) { f32_t f = (f32_t)d; uint32_t b; memcpy(&b, &f, 4); return b; }

typedef struct { float x, y, z; } v3;
typedef struct { float m[3][3]; } m3; /* m[c][r], as glm::mat3 */

/* ---- tiny glm-equivalent helpers (summation order as in glm) ---- */
static m3 m3_cols(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2)
{
    m3 r; r.m[0][0]=a0; r.m[0][1]=a1; r.m[0][2]=a2; r.m[1][0]=b0; r.m[1][1]=b1; r.m[1][2]=b2;
    r.m[2][0]=c0; r.m[2][1]=c1; r.m[2][2]=c2; return r;
}
static m3 m3_mul(m3 a, m3 b)
{
    m3 r;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++)
            r.m[c][rr] = a.m[0][rr] * b.m[c][0] + a.m[1][rr] * b.m[c][1] + a.m[2][rr] * b.m[c][2];
    return r;
}
static m3 m3_T(m3 a)
{
    m3 r;
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) r.m[c][rr] = a.m[rr][c];
    return r;
}
static v3 m3_mulv(m3 a, v3 v)
{
    v3 r;
    r.x = a.m[0][0] * v.x + a.m[1][0] * v.y + a.m[2][0] * v.z;
    r.y = a.m[0][1] * v.x + a.m[1][1] * v.y + a.m[2][1] * v.z;
    r.z = a.m[0][2] * v.x + a.m[1][2] * v.y + a.m[2][2] * v.z;
    return r;
}
static m3 m3_scale(m3 a, float s) { m3 r; for (int c=0;c<3;c++) for (int q=0;q<3;q++) r.m[c][q]=a.m[c][q]*s; return r; }
static m3 m3_div(m3 a, float s) { m3 r; for (int c=0;c<3;c++) for (int q=0;q<3;q++) r.m[c][q]=a.m[c][q]/s; return r; }
static m3 m3_add(m3 a, m3 b) { m3 r; for (int c=0;c<3;c++) for (int q=0;q<3;q++) r.m[c][q]=a.m[c][q]+b.m[c][q]; return r; }
static m3 m3_neg(m3 a) { m3 r; for (int c=0;c<3;c++) for (int q=0;q<3;q++) r.m[c][q]=-a.m[c][q]; return r; }
static m3 m3_zero(void) { m3 r; memset(&r, 0, sizeof r); return r; }
/* glm::outerProduct(c, r): column j of the result is c * r[j] */
static m3 m3_outer(v3 c, v3 r)
{
    m3 o; float cc[3] = {c.x, c.y, c.z}, rr[3] = {r.x, r.y, r.z};
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) o.m[j][i] = cc[i] * rr[j];
    return o;
}
static v3 v3_mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static v3 v3_scale(v3 a, float s) { return v3_mk(a.x * s, a.y * s, a.z * s); }
static v3 v3_divs(v3 a, float s) { return v3_mk(a.x / s, a.y / s, a.z / s); }
static v3 v3_add(v3 a, v3 b) { return v3_mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 v3_sub(v3 a, v3 b) { return v3_mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static float v3_len(v3 a) { return sqrtf(v3_dot(a, a)); }
static v3 v3_normalize(v3 a) { float inv = 1.0f / sqrtf(v3_dot(a, a)); return v3_scale(a, inv); }
static v3 m3_col(m3 a, int c) { return v3_mk(a.m[c][0], a.m[c][1], a.m[c][2]); }

/* CR/auxiliary.h:74-113 */
static v3 xform4x3(v3 p, const float *m)
{
    return v3_mk(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
                 m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                 m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
static void xform4x4(v3 p, const float *m, float out[4])
{
    out[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    out[1] = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    out[2] = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    out[3] = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
}
static v3 xformvec4x3T(v3 p, const float *m)
{
    return v3_mk(m[0] * p.x + m[1] * p.y + m[2] * p.z,
                 m[4] * p.x + m[5] * p.y + m[6] * p.z,
                 m[8] * p.x + m[9] * p.y + m[10] * p.z);
}
/* CR/auxiliary.h:123-133 */
static v3 dnormvdv3(v3 v, v3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    v3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}
/* CR/auxiliary.h:57-60 -- the literals are doubles: evaluated in fp64, rounded to fp32 on return */
static float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
/* CR/auxiliary.h:62-72 */
static void get_rect(float px, float py, int max_radius, int gx, int gy, int rmin[2], int rmax[2])
{
    rmin[0] = imin(gx, imax(0, (int)((px - max_radius) / TILE)));
    rmin[1] = imin(gy, imax(0, (int)((py - max_radius) / TILE)));
    rmax[0] = imin(gx, imax(0, (int)((px + max_radius + TILE - 1) / TILE)));
    rmax[1] = imin(gy, imax(0, (int)((py + max_radius + TILE - 1) / TILE)));
}

/* ---- symmetric 3x3 eigen-solver: Householder tridiagonalisation + implicit QL.
 * Follows glm_modification::findEigenvaluesSymReal, CR/auxiliary.h:217-401 (1-based indexing kept through
 * the accessor macros so the control flow can be compared against the reference line by line). ---- */
static int feq(float x, float y, float eps) { return fabsf(x - y) <= eps; }           /* :189-192 */
/* Convergence threshold of the eigen-solver: the reference's ABSOLUTE 1e-7 (CR/auxiliary.h:203,238).  Tests may lower it
 * (ed3ref_set_eig_epsilon) to run the same code with a CONVERGED solver -- used only to separate the solver's truncation
 * from everything else (tests/test_oracle_pins_cpu.py); every parity comparison runs at the reference's value. */
static float g_eig_epsilon = 0.0000001f;
void ed3ref_set_eig_epsilon(double e) { g_eig_epsilon = (float)e; }
/* Which decisions feed the per-pixel margin of ed3ref_render_forward, each divided by its own weight before the minimum is
 * taken: [0] the alpha threshold (and power > 0), [1] termination T(1 - alpha) < 1e-4, [2] the median test T > 0.5.
 * Weight 1 for all = the plain smallest relative distance.  A test that must exclude a WIDER band around the T thresholds than
 * around the alpha threshold (T is a long product of factors, so its discrepancy is larger) passes weights > 1 for [1] and [2]
 * (tests/util.py: 1, 5, 5 -- a T decision within 5 x MARGIN of its threshold then reports a margin below MARGIN); weight 0
 * leaves a decision kind out (diagnostics). */
static float g_margin_weight[3] = {1.0f, 1.0f, 1.0f};
void ed3ref_set_margin_weights(double a, double t, double m) { g_margin_weight[0] = (float)a; g_margin_weight[1] = (float)t; g_margin_weight[2] = (float)m; }
static inline float margin_term(float d, int kind) { return g_margin_weight[kind] > 0.0f ? d / g_margin_weight[kind] : INFINITY; }
double ed3ref_get_eig_epsilon(void) { return (double)g_eig_epsilon; }
static float transfer_sign(float v, float s) { return (s >= 0) ? fabsf(v) : -fabsf(v); } /* :195-198 */
static float pythag(float a, float b)                                                  /* :201-214 */
{
    const float epsilon = g_eig_epsilon;
    float absa = fabsf(a), absb = fabsf(b);
    if (absa > absb) { absb /= absa; absb *= absb; return absa * sqrtf(1.0f + absb); }
    if (feq(absb, 0.0f, epsilon)) return 0.0f;
    absa /= absb; absa *= absa; return absb * sqrtf(1.0f + absa);
}
#define AA(i, j) a[((i) - 1) * 3 + ((j) - 1)]
#define DD(i) d[(i) - 1]
#define EE(i) e[(i) - 1]
static int eig_sym3(m3 cov, float val[3], m3 *vec)
{
    const int N = 3;
    float a[9], d[3], e[3];
    for (int r = 0; r < N; r++) for (int c = 0; c < N; c++) a[r * N + c] = cov.m[c][r];
    int l, k, j, i;
    float scale, hh, h, g, f;
    const float epsilon = g_eig_epsilon;
    for (i = N; i >= 2; i--) {
        l = i - 1; h = scale = 0;
        if (l > 1) {
            for (k = 1; k <= l; k++) scale += fabsf(AA(i, k));
            if (feq(scale, 0.0f, epsilon)) {
                EE(i) = AA(i, l);
            } else {
                for (k = 1; k <= l; k++) { AA(i, k) /= scale; h += AA(i, k) * AA(i, k); }
                f = AA(i, l);
                g = ((f >= 0) ? -sqrtf(h) : sqrtf(h));
                EE(i) = scale * g; h -= f * g; AA(i, l) = f - g; f = 0;
                for (j = 1; j <= l; j++) {
                    AA(j, i) = AA(i, j) / h; g = 0;
                    for (k = 1; k <= j; k++) g += AA(j, k) * AA(i, k);
                    for (k = j + 1; k <= l; k++) g += AA(k, j) * AA(i, k);
                    EE(j) = g / h; f += EE(j) * AA(i, j);
                }
                hh = f / (h + h);
                for (j = 1; j <= l; j++) {
                    f = AA(i, j); EE(j) = g = EE(j) - hh * f;
                    for (k = 1; k <= j; k++) AA(j, k) -= (f * EE(k) + g * AA(i, k));
                }
            }
        } else {
            EE(i) = AA(i, l);
        }
        DD(i) = h;
    }
    DD(1) = 0; EE(1) = 0;
    for (i = 1; i <= N; i++) {
        l = i - 1;
        if (!feq(DD(i), 0.0f, epsilon)) {
            for (j = 1; j <= l; j++) {
                g = 0;
                for (k = 1; k <= l; k++) g += AA(i, k) * AA(k, j);
                for (k = 1; k <= l; k++) AA(k, j) -= g * AA(k, i);
            }
        }
        DD(i) = AA(i, i); AA(i, i) = 1;
        for (j = 1; j <= l; j++) AA(j, i) = AA(i, j) = 0;
    }
    int m, iter;
    float s, r, p, dd, c, b;
    (void)dd;
    for (i = 2; i <= N; i++) EE(i - 1) = EE(i);
    EE(N) = 0;
    for (l = 1; l <= N; l++) {
        iter = 0;
        do {
            for (m = l; m <= N - 1; m++) {
                dd = fabsf(DD(m)) + fabsf(DD(m + 1));
                if (feq(fabsf(EE(m)), 0.0f, epsilon)) break;
            }
            if (m != l) {
                if (iter++ == 30) return 0;
                g = (DD(l + 1) - DD(l)) / (2 * EE(l));
                r = pythag(g, 1.0f);
                g = DD(m) - DD(l) + EE(l) / (g + transfer_sign(r, g));
                s = c = 1; p = 0;
                for (i = m - 1; i >= l; i--) {
                    f = s * EE(i); b = c * EE(i);
                    EE(i + 1) = r = pythag(f, g);
                    if (feq(r, 0.0f, epsilon)) { DD(i + 1) -= p; EE(m) = 0; break; }
                    s = f / r; c = g / r; g = DD(i + 1) - p;
                    r = (DD(i) - g) * s + 2 * c * b;
                    DD(i + 1) = g + (p = s * r);
                    g = c * r - b;
                    for (k = 1; k <= N; k++) {
                        f = AA(k, i + 1);
                        AA(k, i + 1) = s * AA(k, i) + c * f;
                        AA(k, i) = c * AA(k, i) - s * f;
                    }
                }
                if (feq(r, 0.0f, epsilon) && (i >= l)) continue;
                DD(l) -= p; EE(l) = g; EE(m) = 0;
            }
        } while (m != l);
    }
    for (i = 0; i < N; i++) val[i] = d[i];
    for (i = 0; i < N; i++) for (j = 0; j < N; j++) vec->m[i][j] = a[j * N + i];
    return N;
}

/* CR/forward.cu:23-74 */
static v3 color_from_sh(int idx, int deg, int max_coeffs, const float *means, const float *campos, const float *shs,
                        uint8_t *clamped)
{
    v3 pos = v3_mk(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
    v3 dir = v3_sub(pos, v3_mk(campos[0], campos[1], campos[2]));
    dir = v3_divs(dir, v3_len(dir));
    const float *s = shs + (size_t)idx * max_coeffs * 3;
#define SH(k) v3_mk(s[3 * (k)], s[3 * (k) + 1], s[3 * (k) + 2])
    v3 result = v3_scale(SH(0), SH_C0);
    if (deg > 0) {
        float x = dir.x, y = dir.y, z = dir.z;
        result = v3_sub(v3_add(v3_sub(result, v3_scale(SH(1), SH_C1 * y)), v3_scale(SH(2), SH_C1 * z)),
                        v3_scale(SH(3), SH_C1 * x));
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            result = v3_add(v3_add(v3_add(v3_add(v3_add(result, v3_scale(SH(4), SH_C2[0] * xy)),
                                                 v3_scale(SH(5), SH_C2[1] * yz)),
                                          v3_scale(SH(6), SH_C2[2] * (2.0f * zz - xx - yy))),
                                   v3_scale(SH(7), SH_C2[3] * xz)),
                            v3_scale(SH(8), SH_C2[4] * (xx - yy)));
            if (deg > 2) {
                result = v3_add(result, v3_scale(SH(9), SH_C3[0] * y * (3.0f * xx - yy)));
                result = v3_add(result, v3_scale(SH(10), SH_C3[1] * xy * z));
                result = v3_add(result, v3_scale(SH(11), SH_C3[2] * y * (4.0f * zz - xx - yy)));
                result = v3_add(result, v3_scale(SH(12), SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy)));
                result = v3_add(result, v3_scale(SH(13), SH_C3[4] * x * (4.0f * zz - xx - yy)));
                result = v3_add(result, v3_scale(SH(14), SH_C3[5] * z * (xx - yy)));
                result = v3_add(result, v3_scale(SH(15), SH_C3[6] * x * (xx - 3.0f * yy)));
            }
        }
    }
#undef SH
    result.x += 0.5f; result.y += 0.5f; result.z += 0.5f;
    clamped[3 * idx + 0] = (result.x < 0);
    clamped[3 * idx + 1] = (result.y < 0);
    clamped[3 * idx + 2] = (result.z < 0);
    return v3_mk(fmaxf(result.x, 0.0f), fmaxf(result.y, 0.0f), fmaxf(result.z, 0.0f));
}

/* CR/forward.cu:270-304 (quaternion used as given, Q3) */
static void cov3d_from_scale_rot(const float *scale, float mod, const float *rot, float *cov3D)
{
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.m[0][0] = mod * scale[0]; S.m[1][1] = mod * scale[1]; S.m[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 M = m3_mul(S, R);
    m3 Sigma = m3_mul(m3_T(M), M);
    cov3D[0] = Sigma.m[0][0]; cov3D[1] = Sigma.m[0][1]; cov3D[2] = Sigma.m[0][2];
    cov3D[3] = Sigma.m[1][1]; cov3D[4] = Sigma.m[1][2]; cov3D[5] = Sigma.m[2][2];
}

/* CR/forward.cu:77-264; inv6 != NULL is the INTE = true instantiation (:187-235).  Returns well_conditioned. */
static int cov2d_planes(v3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy, float kernel_size,
                        const float *cov3D, const float *view, float cov2D[3], float camera_plane[6], float normal[3],
                        float ray_plane[2], float *coef, float *inv6)
{
    v3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    txtz = t.x / t.z; tytz = t.y / t.z;

    m3 J = m3_cols(focal_x / t.z, 0.0f, -(focal_x * t.x) / (t.z * t.z), 0.0f, focal_y / t.z,
                   -(focal_y * t.y) / (t.z * t.z), 0, 0, 0);
    m3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    m3 T = m3_mul(Wm, J);
    m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 cov = m3_mul(m3_mul(m3_T(T), m3_T(Vrk)), T);

    cov2D[0] = (float)(cov.m[0][0] + kernel_size);
    cov2D[1] = (float)(cov.m[0][1]);
    cov2D[2] = (float)(cov.m[1][1] + kernel_size);
    /* :119-124 -- fmax(double, float): evaluated in fp64 (Q13) */
    const float det_0 = (float)fmax(1e-6, (double)(cov.m[0][0] * cov.m[1][1] - cov.m[0][1] * cov.m[0][1]));
    const float det_1 = (float)fmax(1e-6, (double)((cov.m[0][0] + kernel_size) * (cov.m[1][1] + kernel_size) -
                                                    cov.m[0][1] * cov.m[0][1]));
    *coef = (float)sqrt(det_0 / (det_1 + 1e-6) + 1e-6);
    if (det_0 <= 1e-6 || det_1 <= 1e-6) *coef = 0.0f;

    m3 evec; float eval[3];
    int Dn = eig_sym3(Vrk, eval, &evec);
    unsigned min_id = eval[0] > eval[1] ? (eval[1] > eval[2] ? 2 : 1) : (eval[0] > eval[2] ? 2 : 0);
    m3 Vrk_inv;
    int well_conditioned = eval[min_id] > 0.00000001;
    if (well_conditioned) {
        m3 diag = m3_cols(1 / eval[0], 0, 0, 0, 1 / eval[1], 0, 0, 0, 1 / eval[2]);
        Vrk_inv = m3_mul(m3_mul(evec, diag), m3_T(evec));
    } else {
        v3 emin = m3_col(evec, min_id);
        Vrk_inv = m3_outer(emin, emin);
    }
    m3 cov_cam_inv = m3_mul(m3_mul(m3_T(Wm), Vrk_inv), Wm);
    v3 uvh = v3_mk(txtz, tytz, 1);
    v3 uvh_m = m3_mulv(cov_cam_inv, uvh);
    v3 uvh_mn = v3_normalize(uvh_m);

    if (isnan(uvh_mn.x) || Dn == 0) {
        for (int ch = 0; ch < 6; ch++) camera_plane[ch] = 0;
        normal[0] = normal[1] = normal[2] = 0; ray_plane[0] = ray_plane[1] = 0;
    } else {
        float u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;
        float l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
        m3 nJ = m3_cols(1 / t.z, 0.0f, -(t.x) / (t.z * t.z), 0.0f, 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l,
                        t.z / l);
        m3 nJ_inv = m3_cols(v2 + 1, -uv, 0, -uv, u2 + 1, 0, -txtz, -tytz, 0);
        if (inv6) {
            m3 inv_cov_ray;
            if (well_conditioned) {
                float ltz = u2 + v2 + 1;
                m3 full = m3_scale(m3_cols(v2 + 1, -uv, txtz / l * ltz, -uv, u2 + 1, tytz / l * ltz, -txtz, -tytz, 1 / l * ltz),
                                   t.z / (u2 + v2 + 1));
                m3 T2 = m3_mul(Wm, m3_T(full));
                inv_cov_ray = m3_mul(m3_mul(m3_T(T2), Vrk_inv), T2);
            } else {
                /* :204-232.  The reference stores this branch's result in a block-local `inv_cov_ray` that shadows the
                   outer one (:219), so what it then scales and writes is an UNINITIALISED matrix; the value the branch
                   computes is used here (documented deviation from undefined behaviour). */
                m3 T2 = m3_mul(Wm, nJ);
                m3 cov_ray = m3_mul(m3_mul(m3_T(T2), Vrk_inv), T2);
                m3 cvec; float cval[3];
                eig_sym3(cov_ray, cval, &cvec);
                unsigned mid = cval[0] > cval[1] ? (cval[1] > cval[2] ? 2 : 1) : (cval[0] > cval[2] ? 2 : 0);
                float lambda1 = cval[(mid + 1) % 3], lambda2 = cval[(mid + 2) % 3];
                m3 nv;
                for (int q = 0; q < 3; q++) {
                    nv.m[0][q] = cvec.m[(mid + 1) % 3][q]; nv.m[1][q] = cvec.m[(mid + 2) % 3][q]; nv.m[2][q] = cvec.m[mid][q];
                }
                v3 r3 = v3_mk(nv.m[0][2], nv.m[1][2], nv.m[2][2]);
                m3 c2d = m3_cols(1 / lambda1, 0, -r3.x / r3.z / lambda1, 0, 1 / lambda2, -r3.y / r3.z / lambda2,
                                 -r3.x / r3.z / lambda1, -r3.y / r3.z / lambda2, 0);
                inv_cov_ray = m3_mul(m3_mul(nv, c2d), m3_T(nv));
            }
            m3 sc = m3_cols(1 / focal_x, 0, 0, 0, 1 / focal_y, 0, 0, 0, 1);
            inv_cov_ray = m3_mul(m3_mul(sc, inv_cov_ray), sc);
            inv6[0] = inv_cov_ray.m[0][0]; inv6[1] = inv_cov_ray.m[0][1]; inv6[2] = inv_cov_ray.m[0][2];
            inv6[3] = inv_cov_ray.m[1][1]; inv6[4] = inv_cov_ray.m[1][2]; inv6[5] = inv_cov_ray.m[2][2];
        }
        float vbn = v3_dot(uvh_mn, uvh);
        float factor_normal = l / (u2 + v2 + 1);
        v3 plane = m3_mulv(nJ_inv, v3_divs(uvh_mn, fmaxf(vbn, 0.0000001f)));
        float nl = u2 + v2 + 1;
        camera_plane[0] = (-(v2 + 1) * t.z + plane.x * t.x) / nl / focal_x;
        camera_plane[1] = (uv * t.z + plane.y * t.x) / nl / focal_y;
        camera_plane[2] = (uv * t.z + plane.x * t.y) / nl / focal_x;
        camera_plane[3] = (-(u2 + 1) * t.z + plane.y * t.y) / nl / focal_y;
        camera_plane[4] = (t.x + plane.x * t.z) / nl / focal_x;
        camera_plane[5] = (t.y + plane.y * t.z) / nl / focal_y;
        ray_plane[0] = plane.x * l / nl / focal_x;
        ray_plane[1] = plane.y * l / nl / focal_y;
        v3 ray_normal = v3_mk(-plane.x * factor_normal, -plane.y * factor_normal, -1);
        v3 cam_normal = m3_mulv(nJ, ray_normal);
        v3 n = v3_normalize(cam_normal);
        normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
    }
    return well_conditioned;
}

/* K10: CR/rasterizer_impl.cu:54-66 + CR/auxiliary.h:155-180 */
void ed3ref_mark_visible(int P, const float *means, const float *view, const float *proj, uint8_t *present)
{
    (void)proj;
    for (int i = 0; i < P; i++) {
        v3 pv = xform4x3(v3_mk(means[3 * i], means[3 * i + 1], means[3 * i + 2]), view);
        present[i] = !(pv.z <= 0.2f);
    }
}

/* K1: CR/forward.cu:426-545.  All output arrays must be zero-initialised by the caller. */
void ed3ref_preprocess(int P, int D, int M, const float *means, const float *scales, float scale_modifier,
                       const float *rotations, const float *opacities, const float *tongue_class, const float *shs,
                       const float *cov3D_precomp, const float *colors_precomp, const float *view, const float *proj,
                       const float *campos, int W, int H, float tan_fovx, float tan_fovy, float kernel_size,
                       uint8_t *clamped, int32_t *radii, float *means2D, float *view_points, float *depths,
                       float *camera_planes, float *ray_planes, float *ts, float *normals, float *cov3Ds, float *rgb,
                       float *conic_opacity, float *is_tongue, uint32_t *tiles_touched, float *invraycov /* [P][6] or NULL: K11 */,
                       uint8_t *condition /* [P] or NULL */)
{
    const float focal_y = H / (2.0f * tan_fovy); /* CR/rasterizer_impl.cu:291-292 */
    const float focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0; tiles_touched[idx] = 0;
        v3 p_orig = v3_mk(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
        v3 p_view = xform4x3(p_orig, view);
        if (p_view.z <= 0.2f) continue;
        float p_hom[4]; xform4x4(p_orig, proj, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
        const float *cov3D;
        if (cov3D_precomp) cov3D = cov3D_precomp + 6 * idx;
        else { cov3d_from_scale_rot(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov3Ds + 6 * idx); cov3D = cov3Ds + 6 * idx; }
        float cov2D[3], coef;
        int wc = cov2d_planes(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, kernel_size, cov3D, view, cov2D,
                              camera_planes + 6 * idx, normals + 3 * idx, ray_planes + 2 * idx, &coef,
                              invraycov ? invraycov + 6 * idx : NULL);
        if (condition) condition[idx] = (uint8_t)wc;   /* CR/forward.cu:373-376 */
        ts[idx] = sqrtf(p_view.x * p_view.x + p_view.y * p_view.y + p_view.z * p_view.z);
        float cx = cov2D[0], cy = cov2D[1], cz = cov2D[2];
        float det = (cx * cz - cy * cy);
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cz * det_inv, -cy * det_inv, cx * det_inv};
        float mid = 0.5f * (cx + cz);
        float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        float pix[2] = {ndc2pix(p_proj[0], W), ndc2pix(p_proj[1], H)};
        int rmin[2], rmax[2];
        get_rect(pix[0], pix[1], (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) {
            v3 c = color_from_sh(idx, D, M, means, campos, shs, clamped);
            rgb[3 * idx] = c.x; rgb[3 * idx + 1] = c.y; rgb[3 * idx + 2] = c.z;
        }
        depths[idx] = p_view.z;
        view_points[3 * idx] = p_view.x; view_points[3 * idx + 1] = p_view.y; view_points[3 * idx + 2] = p_view.z;
        radii[idx] = (int)my_radius;
        means2D[2 * idx] = pix[0]; means2D[2 * idx + 1] = pix[1];
        conic_opacity[4 * idx] = conic[0]; conic_opacity[4 * idx + 1] = conic[1]; conic_opacity[4 * idx + 2] = conic[2];
        conic_opacity[4 * idx + 3] = opacities[idx] * coef;
        tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
        is_tongue[idx] = tongue_class ? tongue_class[idx] : 0.f;
    }
}

/* K3: CR/rasterizer_impl.cu:70-111; offsets = inclusive prefix sum of tiles_touched (K2, :355) */
void ed3ref_duplicate_with_keys(int P, const float *means2D, const float *depths, const uint32_t *offsets,
                                const int32_t *radii, int W, int H, uint64_t *keys, uint32_t *values)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : offsets[idx - 1];
            int rmin[2], rmax[2];
            get_rect(means2D[2 * idx], means2D[2 * idx + 1], radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits = depth_bits(depths[idx]);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32; key |= dbits;
                    keys[off] = key; values[off] = (uint32_t)idx; off++;
                }
        }
    }
}

/* CR/rasterizer_impl.cu:35-50 */
uint32_t ed3ref_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

/* K4: stable LSD radix sort of (key,value) on key bits [0,end_bit) -- the contract of
 * cub::DeviceRadixSort::SortPairs at CR/rasterizer_impl.cu:381-386 (CUB is not vendored: CUDA 11.8 toolkit). */
void ed3ref_sort_pairs(int64_t R, const uint64_t *keys_in, const uint32_t *vals_in, uint64_t *keys_out,
                       uint32_t *vals_out, int end_bit)
{
    if (R <= 0) return;
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * R), *kb = (uint64_t *)malloc(sizeof(uint64_t) * R);
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * R), *vb = (uint32_t *)malloc(sizeof(uint32_t) * R);
    memcpy(ka, keys_in, sizeof(uint64_t) * R); memcpy(va, vals_in, sizeof(uint32_t) * R);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1;
        int64_t count[257]; memset(count, 0, sizeof count);
        for (int64_t i = 0; i < R; i++) count[((ka[i] >> shift) & mask) + 1]++;
        for (int b = 0; b < 256; b++) count[b + 1] += count[b];
        for (int64_t i = 0; i < R; i++) { int64_t p = count[(ka[i] >> shift) & mask]++; kb[p] = ka[i]; vb[p] = va[i]; }
        uint64_t *tk = ka; ka = kb; kb = tk; uint32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * R); memcpy(vals_out, va, sizeof(uint32_t) * R);
    free(ka); free(kb); free(va); free(vb);
}

/* K5: CR/rasterizer_impl.cu:151-173 (+ memset :388); ranges[2*t] = start, ranges[2*t+1] = end */
void ed3ref_identify_tile_ranges(int64_t L, const uint64_t *keys, uint32_t *ranges, int num_tiles)
{
    memset(ranges, 0, sizeof(uint32_t) * 2 * num_tiles);
    for (int64_t idx = 0; idx < L; idx++) {
        uint32_t currtile = (uint32_t)(keys[idx] >> 32);
        if (idx == 0) ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(keys[idx - 1] >> 32);
            if (currtile != prevtile) { ranges[2 * prevtile + 1] = (uint32_t)idx; ranges[2 * currtile] = (uint32_t)idx; }
        }
        if (idx == L - 1) ranges[2 * currtile + 1] = (uint32_t)L;
    }
}

static float minf(float a, float b) { return a < b ? a : b; }

/* K6: CR/forward.cu:550-822, one pixel at a time (the reference's 256-entry staging only changes when data is
 * fetched, not what is computed).  NORMAL is on iff COORD or DEPTH (:863-870).
 * margin (optional, H*W): smallest relative distance of any branch decision taken for the pixel to its
 * threshold -- oracle-only diagnostic used by the tests to tell ill-conditioned pixels apart. */
void ed3ref_render_forward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *view_points,
                           const float *means2D, const float *features, const float *ts, const float *camera_planes,
                           const float *ray_planes, const float *normals, const float *conic_opacity,
                           const float *is_tongue, float focal_x, float focal_y, const float *bg, int COORD, int DEPTH,
                           float *out_alpha, float *out_tongue, uint32_t *n_contrib, float *out_color, float *out_coord,
                           float *out_mcoord, float *out_normal, float *out_depth, float *out_mdepth,
                           float *accum_coord, float *accum_depth, float *normal_length, float *margin)
{
    const int NORMAL = COORD || DEPTH, GEO = NORMAL;
    const int gx = (W + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 16)
    for (int py = 0; py < H; py++) {
        for (int px = 0; px < W; px++) {
            const uint32_t pix_id = (uint32_t)(W * py + px);
            float pixfx = (float)px, pixfy = (float)py;
            float pnx = (pixfx - W / 2.f) / focal_x, pny = (pixfy - H / 2.f) / focal_y;
            float ln = sqrtf(pnx * pnx + pny * pny + 1);
            const uint32_t *rg = ranges + 2 * ((py / TILE) * gx + (px / TILE));
            float T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0, max_contributor = (uint32_t)-1;
            float C[3] = {0, 0, 0}, tongue = 0, weight = 0, Coord[3] = {0, 0, 0}, mCoord[3] = {0, 0, 0};
            float Depth = 0, mDepth = 0, Normal[3] = {0, 0, 0};
            float mg = INFINITY;
            for (uint32_t k = rg[0]; k < rg[1]; k++) {
                contributor++;
                uint32_t g = point_list[k];
                float dx = means2D[2 * g] - pixfx, dy = means2D[2 * g + 1] - pixfy;
                const float *co = conic_opacity + 4 * g;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (margin && fabsf(power) < 1e-4f) mg = fminf(mg, margin_term(fabsf(power), 0));
                if (power > 0.0f) continue;
                float araw = co[3] * expf(power);
                float alpha = minf(0.99f, araw);
                if (margin) mg = fminf(mg, margin_term(fabsf(araw - 1.0f / 255.0f) * 255.0f, 0));
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (margin) mg = fminf(mg, margin_term(fabsf(test_T - 0.0001f) / 0.0001f, 1));
                if (test_T < 0.0001f) break; /* done=true ends the pixel (:696-700) */
                const float aT = alpha * T;
                for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * g + ch] * aT;
                tongue += is_tongue[g] * aT;
                int before_median = T > 0.5;
                if (margin && GEO) mg = fminf(mg, margin_term(fabsf(T - 0.5f) / 0.5f, 2));
                if (COORD) {
                    const float *cp = camera_planes + 6 * g;
                    float coord[3] = {view_points[3 * g] + cp[0] * dx + cp[1] * dy,
                                      view_points[3 * g + 1] + cp[2] * dx + cp[3] * dy,
                                      view_points[3 * g + 2] + cp[4] * dx + cp[5] * dy};
                    for (int ch = 0; ch < 3; ch++) Coord[ch] += coord[ch] * aT;
                    if (before_median) for (int ch = 0; ch < 3; ch++) mCoord[ch] = coord[ch];
                }
                if (DEPTH) {
                    float t = ts[g] + (ray_planes[2 * g] * dx + ray_planes[2 * g + 1] * dy);
                    Depth += t * aT;
                    if (before_median) mDepth = t;
                }
                if (NORMAL) for (int ch = 0; ch < 3; ch++) Normal[ch] += normals[3 * g + ch] * aT;
                if (GEO && before_median) max_contributor = contributor;
                weight += aT;
                T = test_T;
                last_contributor = contributor;
            }
            n_contrib[pix_id] = last_contributor;
            n_contrib[pix_id + HW] = max_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix_id] = C[ch] + T * bg[ch];
            out_tongue[pix_id] = tongue;
            out_alpha[pix_id] = weight;
            if (margin) margin[pix_id] = mg;
            if (COORD) {
                for (int ch = 0; ch < 3; ch++) {
                    out_coord[ch * HW + pix_id] = last_contributor ? Coord[ch] / weight : 0;
                    accum_coord[ch * HW + pix_id] = Coord[ch];
                    out_mcoord[ch * HW + pix_id] = mCoord[ch];
                }
            }
            if (DEPTH) {
                float depth_ln = Depth / ln;
                accum_depth[pix_id] = depth_ln;
                out_depth[pix_id] = last_contributor ? depth_ln / weight : 0;
                out_mdepth[pix_id] = mDepth / ln;
            }
            if (NORMAL) {
                if (last_contributor) {
                    float len = sqrtf(Normal[0] * Normal[0] + Normal[1] * Normal[1] + Normal[2] * Normal[2]);
                    normal_length[pix_id] = len;
                    len = fmaxf(len, NORMALIZE_EPS);
                    for (int ch = 0; ch < 3; ch++) out_normal[ch * HW + pix_id] = Normal[ch] / len;
                } else {
                    normal_length[pix_id] = 1;
                    for (int ch = 0; ch < 3; ch++) out_normal[ch * HW + pix_id] = 0;
                }
            }
        }
    }
}

/* K7: CR/backward.cu:631-1016.  The reference accumulates per-Gaussian gradients with float atomicAdd in a
 * non-deterministic order; the oracle accumulates the same per-pair fp32 terms into float64 arrays (caller
 * rounds once), which is the order-free value every summation order approximates.
 * dL_dconic has 4 columns (x, y, unused, w) as the reference's (P,2,2) tensor viewed as float4 (:1008-1010). */
void ed3ref_render_backward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *bg,
                            const float *view_points, const float *means2D, const float *conic_opacity,
                            const float *colors, const float *ts, const float *camera_planes, const float *ray_planes,
                            const float *alphas, const float *normals, const float *accum_coord,
                            const float *accum_depth, const float *normal_length, const uint32_t *n_contrib,
                            const float *dL_dpixels, const float *dL_dpixel_coords, const float *dL_dpixel_mcoords,
                            const float *dL_dpixel_depths, const float *dL_dpixel_mdepths, const float *dL_dalphas,
                            const float *dL_dpixel_normals, const float *normalmap, float focal_x, float focal_y,
                            int COORD, int DEPTH, double *dL_dview_points, double *dL_dmean2D, double *dL_dconic2D,
                            double *dL_dopacity, double *dL_dcolors, double *dL_dts, double *dL_dcamera_planes,
                            double *dL_dray_planes, double *dL_dnormals)
{
    const int NORMAL = COORD || DEPTH, GEO = NORMAL;
    const int gx = (W + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
#define ACC(arr, i, v) do { double v__ = (double)(v); _Pragma("omp atomic") arr[i] += v__; } while (0)
#pragma omp parallel for schedule(dynamic, 16)
    for (int py = 0; py < H; py++) {
        for (int px = 0; px < W; px++) {
            const uint32_t pix_id = (uint32_t)(W * py + px);
            const float pixfx = (float)px, pixfy = (float)py;
            const float pnx = (pixfx - W / 2.f) / focal_x, pny = (pixfy - H / 2.f) / focal_y;
            const float ln = sqrtf(pnx * pnx + pny * pny + 1);
            const uint32_t *rg = ranges + 2 * ((py / TILE) * gx + (px / TILE));
            const int toDo = (int)(rg[1] - rg[0]);
            const float T_final = 1 - alphas[pix_id], w_final = alphas[pix_id];
            float T = T_final;
            uint32_t contributor = (uint32_t)toDo;
            const int last_contributor = (int)n_contrib[pix_id];
            const int max_contributor = (int)n_contrib[pix_id + HW];
            float accum_rec[3] = {0, 0, 0}, dL_dpixel[3], accum_coord_rec[3] = {0, 0, 0}, dL_dpixel_coord[3] = {0, 0, 0};
            float accum_t_rec = 0, dL_dpixel_t = 0, dL_dpixel_mt = 0, accum_alpha_rec = 0, dL_dalpha;
            float accum_normal_rec[3] = {0, 0, 0}, dL_dpixel_normal[3] = {0, 0, 0}, dL_dpixel_mcoord[3] = {0, 0, 0};
            for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * HW + pix_id];
            dL_dalpha = dL_dalphas[pix_id];
            if (GEO) {
                float ww = w_final * w_final;
                if (COORD) {
                    for (int i = 0; i < 3; i++) {
                        float g = dL_dpixel_coords[i * HW + pix_id];
                        dL_dalpha -= g * accum_coord[i * HW + pix_id] / ww;
                        dL_dpixel_coord[i] = g / w_final;
                        dL_dpixel_mcoord[i] = dL_dpixel_mcoords[i * HW + pix_id];
                    }
                }
                if (DEPTH) {
                    float g = dL_dpixel_depths[pix_id];
                    dL_dalpha -= g * accum_depth[pix_id] / ww;
                    dL_dpixel_t = g / w_final / ln;
                    dL_dpixel_mt = dL_dpixel_mdepths[pix_id] / ln;
                }
                if (NORMAL) {
                    v3 gn = v3_mk(dL_dpixel_normals[pix_id], dL_dpixel_normals[HW + pix_id], dL_dpixel_normals[2 * HW + pix_id]);
                    v3 nn = v3_mk(normalmap[pix_id], normalmap[HW + pix_id], normalmap[2 * HW + pix_id]);
                    float nlen = normal_length[pix_id];
                    v3 dL;
                    if (nlen < NORMALIZE_EPS) dL = v3_divs(gn, NORMALIZE_EPS);
                    else dL = v3_divs(v3_sub(gn, v3_scale(nn, v3_dot(gn, nn))), nlen);
                    dL_dpixel_normal[0] = dL.x; dL_dpixel_normal[1] = dL.y; dL_dpixel_normal[2] = dL.z;
                }
            }
            float last_alpha = 0, last_color[3] = {0, 0, 0}, last_coord[3] = {0, 0, 0}, last_t = 0, last_normal[3] = {0, 0, 0};
            for (int64_t k = (int64_t)rg[1] - 1; k >= (int64_t)rg[0]; k--) {
                contributor--;
                if ((int64_t)contributor >= (int64_t)last_contributor) continue;
                const uint32_t gid = point_list[k];
                const float dx = means2D[2 * gid] - pixfx, dy = means2D[2 * gid + 1] - pixfy;
                const float *co = conic_opacity + 4 * gid;
                float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = expf(power);
                const float alpha = minf(0.99f, co[3] * G);
                if (alpha < 1.0f / 255.0f) continue;
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dopa = 0.0f;
                for (int ch = 0; ch < 3; ch++) {
                    const float c = colors[3 * gid + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    dL_dopa += (c - accum_rec[ch]) * dL_dpixel[ch];
                    ACC(dL_dcolors, 3 * gid + ch, dchannel_dcolor * dL_dpixel[ch]);
                }
                float dL_dcoords[3] = {0, 0, 0}, dL_dt = 0;
                const float *cp = camera_planes + 6 * gid;
                const float *rp = ray_planes + 2 * gid;
                if (COORD) {
                    float coord[3] = {view_points[3 * gid] + cp[0] * dx + cp[1] * dy,
                                      view_points[3 * gid + 1] + cp[2] * dx + cp[3] * dy,
                                      view_points[3 * gid + 2] + cp[4] * dx + cp[5] * dy};
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = coord[ch];
                        accum_coord_rec[ch] = last_alpha * last_coord[ch] + (1.f - last_alpha) * accum_coord_rec[ch];
                        last_coord[ch] = c;
                        dL_dopa += (c - accum_coord_rec[ch]) * dL_dpixel_coord[ch];
                        dL_dcoords[ch] = dchannel_dcolor * dL_dpixel_coord[ch];
                        if ((int64_t)contributor == (int64_t)max_contributor - 1) dL_dcoords[ch] += dL_dpixel_mcoord[ch];
                    }
                    for (int ch = 0; ch < 3; ch++) {
                        ACC(dL_dview_points, 3 * gid + ch, dL_dcoords[ch]);
                        ACC(dL_dcamera_planes, 6 * gid + 2 * ch, dL_dcoords[ch] * dx / focal_x);
                        ACC(dL_dcamera_planes, 6 * gid + 2 * ch + 1, dL_dcoords[ch] * dy / focal_y);
                    }
                }
                if (DEPTH) {
                    float t = ts[gid] + (rp[0] * dx + rp[1] * dy);
                    accum_t_rec = last_alpha * last_t + (1.f - last_alpha) * accum_t_rec;
                    last_t = t;
                    dL_dopa += (t - accum_t_rec) * dL_dpixel_t;
                    dL_dt = dchannel_dcolor * dL_dpixel_t;
                    if ((int64_t)contributor == (int64_t)max_contributor - 1) dL_dt += dL_dpixel_mt;
                    ACC(dL_dts, gid, dL_dt);
                    ACC(dL_dray_planes, 2 * gid, dL_dt * dx / focal_x);
                    ACC(dL_dray_planes, 2 * gid + 1, dL_dt * dy / focal_y);
                }
                if (NORMAL) {
                    for (int ch = 0; ch < 3; ch++) {
                        const float c = normals[3 * gid + ch];
                        accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
                        last_normal[ch] = c;
                        dL_dopa += (c - accum_normal_rec[ch]) * dL_dpixel_normal[ch];
                        ACC(dL_dnormals, 3 * gid + ch, dchannel_dcolor * dL_dpixel_normal[ch]);
                    }
                }
                accum_alpha_rec = last_alpha + (1.f - last_alpha) * accum_alpha_rec;
                dL_dopa += (1 - accum_alpha_rec) * dL_dalpha;
                dL_dopa *= T;
                last_alpha = alpha;
                float bg_dot_dpixel = 0;
                for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                dL_dopa += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = co[3] * dL_dopa;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                const float dG_ddely = -gdy * co[2] - gdx * co[1];
                float dL_ddelx = dL_dG * dG_ddelx, dL_ddely = dL_dG * dG_ddely;
                if (COORD) {
                    dL_ddelx += dL_dcoords[0] * cp[0] + dL_dcoords[1] * cp[2] + dL_dcoords[2] * cp[4];
                    dL_ddely += dL_dcoords[0] * cp[1] + dL_dcoords[1] * cp[3] + dL_dcoords[2] * cp[5];
                }
                if (DEPTH) { dL_ddelx += dL_dt * rp[0]; dL_ddely += dL_dt * rp[1]; }
                ACC(dL_dmean2D, 3 * gid, dL_ddelx * ddelx_dx);
                ACC(dL_dmean2D, 3 * gid + 1, dL_ddely * ddely_dy);
                const float abs_dL = fabsf(dL_dG * dG_ddelx * ddelx_dx) + fabsf(dL_dG * dG_ddely * ddely_dy);
                ACC(dL_dmean2D, 3 * gid + 2, abs_dL);
                ACC(dL_dconic2D, 4 * gid, -0.5f * gdx * dx * dL_dG);
                ACC(dL_dconic2D, 4 * gid + 1, -0.5f * gdx * dy * dL_dG);
                ACC(dL_dconic2D, 4 * gid + 3, -0.5f * gdy * dy * dL_dG);
                ACC(dL_dopacity, gid, G * dL_dopa);
            }
        }
    }
#undef ACC
}

/* K8: CR/backward.cu:145-488.  `conic_opacity_arg` is what the reference passes in that slot: its caller hands
 * over (float4*)dL_dconic (CR/rasterizer_impl.cu:576), quirk Q1; pass the true conic_opacity to get the
 * "repaired" behaviour.  dL_dopacity is updated in place (:395,403); dL_dmeans is overwritten (:487). */
void ed3ref_cov2d_backward(int P, const float *means, const int32_t *radii, const float *cov3Ds, float h_x, float h_y,
                           float tan_fovx, float tan_fovy, float kernel_size, const float *view,
                           const float *dL_dconics, const float *dL_dcamera_planes, const float *dL_dray_planes,
                           const float *dL_dnormals, float *dL_dmeans, float *dL_dcov,
                           const float *conic_opacity_arg, float *dL_dopacity)
{
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float *cov3D = cov3Ds + 6 * idx;
        v3 mean = v3_mk(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
        float dLc_x = dL_dconics[4 * idx], dLc_y = dL_dconics[4 * idx + 1], dLc_z = dL_dconics[4 * idx + 3];
        v3 dL_dnormal = v3_mk(dL_dnormals[3 * idx], dL_dnormals[3 * idx + 1], dL_dnormals[3 * idx + 2]);
        const float combined_opacity = conic_opacity_arg[4 * idx + 3];
        const float cp0x = dL_dcamera_planes[6 * idx], cp0y = dL_dcamera_planes[6 * idx + 1];
        const float cp1x = dL_dcamera_planes[6 * idx + 2], cp1y = dL_dcamera_planes[6 * idx + 3];
        const float cp2x = dL_dcamera_planes[6 * idx + 4], cp2y = dL_dcamera_planes[6 * idx + 5];
        const float drx = dL_dray_planes[2 * idx], dry = dL_dray_planes[2 * idx + 1];

        v3 t = xform4x3(mean, view);
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        float txtz = t.x / t.z, tytz = t.y / t.z;
        t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
        t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
        const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
        const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
        txtz = t.x / t.z; tytz = t.y / t.z;

        m3 J = m3_cols(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z), 0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0, 0, 0);
        m3 Wm = m3_cols(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
        m3 Vrk = m3_cols(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
        m3 T = m3_mul(Wm, J);
        m3 cov2D = m3_mul(m3_mul(m3_T(T), m3_T(Vrk)), T);
        const float det_0 = (float)fmax(1e-6, (double)(cov2D.m[0][0] * cov2D.m[1][1] - cov2D.m[0][1] * cov2D.m[0][1]));
        const float det_1 = (float)fmax(1e-6, (double)((cov2D.m[0][0] + kernel_size) * (cov2D.m[1][1] + kernel_size) -
                                                        cov2D.m[0][1] * cov2D.m[0][1]));
        const float coef = (float)sqrt(det_0 / (det_1 + 1e-6) + 1e-6);

        m3 evec; float eval[3];
        int Dn = eig_sym3(Vrk, eval, &evec);
        unsigned min_id = eval[0] > eval[1] ? (eval[1] > eval[2] ? 2 : 1) : (eval[0] > eval[2] ? 2 : 0);
        m3 Vrk_inv; v3 emin = v3_mk(0, 0, 0);
        int well_conditioned = eval[min_id] > 0.00000001;
        if (well_conditioned) {
            m3 diag = m3_cols(1 / eval[0], 0, 0, 0, 1 / eval[1], 0, 0, 0, 1 / eval[2]);
            Vrk_inv = m3_mul(m3_mul(evec, diag), m3_T(evec));
        } else {
            emin = m3_col(evec, min_id);
            Vrk_inv = m3_outer(emin, emin);
        }
        m3 cov_cam_inv = m3_mul(m3_mul(m3_T(Wm), Vrk_inv), Wm);
        v3 uvh = v3_mk(txtz, tytz, 1);
        v3 uvh_m = m3_mulv(cov_cam_inv, uvh);
        v3 uvh_mn = v3_normalize(uvh_m);
        float u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;

        m3 dL_dVrk, dL_dnJ; v3 plane; float dL_du, dL_dv, dL_dl, l, nl;
        if (isnan(uvh_mn.x) || Dn == 0) {
            dL_dVrk = m3_zero(); dL_dnJ = m3_zero(); plane = v3_mk(0, 0, 0);
            nl = 1; l = 1; dL_du = 0; dL_dv = 0; dL_dl = 0;
        } else {
            float vb = v3_dot(uvh_m, uvh), vbn = v3_dot(uvh_mn, uvh);
            l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
            m3 nJ = m3_cols(1 / t.z, 0.0f, -(t.x) / (t.z * t.z), 0.0f, 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l, t.z / l);
            m3 nJ_inv = m3_cols(v2 + 1, -uv, 0, -uv, u2 + 1, 0, -txtz, -tytz, 0);
            float clamp_vb = fmaxf(vb, 0.0000001f), clamp_vbn = fmaxf(vbn, 0.0000001f);
            nl = u2 + v2 + 1;
            float factor_normal = l / nl;
            v3 uvh_m_vb = v3_divs(uvh_mn, clamp_vbn);
            plane = m3_mulv(nJ_inv, uvh_m_vb);
            float c0x = (-(v2 + 1) * t.z + plane.x * t.x) / nl, c0y = (uv * t.z + plane.y * t.x) / nl;
            float c1x = (uv * t.z + plane.x * t.y) / nl, c1y = (-(u2 + 1) * t.z + plane.y * t.y) / nl;
            float c2x = (t.x + plane.x * t.z) / nl, c2y = (t.y + plane.y * t.z) / nl;
            float rpx = plane.x * factor_normal, rpy = plane.y * factor_normal;
            v3 ray_normal = v3_mk(-plane.x * factor_normal, -plane.y * factor_normal, -1);
            v3 cam_normal = m3_mulv(nJ, ray_normal);
            v3 normal_vector = v3_normalize(cam_normal);
            float lv = v3_len(cam_normal);
            v3 dL_dnormal_lv = v3_divs(dL_dnormal, lv);
            v3 dL_dcam_normal = v3_sub(dL_dnormal_lv, v3_scale(normal_vector, v3_dot(normal_vector, dL_dnormal_lv)));
            v3 dL_dray_normal = m3_mulv(m3_T(nJ), dL_dcam_normal);
            dL_dnJ = m3_outer(dL_dcam_normal, ray_normal);
            dL_dl = (-plane.x * dL_dray_normal.x - plane.y * dL_dray_normal.y + plane.x * drx + plane.y * dry) / nl;
            float dpx = (t.x * cp0x + t.y * cp1x + t.z * cp2x - l * dL_dray_normal.x + drx * l) / nl;
            float dpy = (t.x * cp0y + t.y * cp1y + t.z * cp2y - l * dL_dray_normal.y + dry * l) / nl;
            v3 dL_dplane_append = v3_mk(dpx, dpy, 0);
            float dL_dnl = (-cp0x * c0x - cp0y * c0y - cp1x * c1x - cp1y * c1y - cp2x * c2x - cp2y * c2y -
                            dL_dray_normal.x * ray_normal.x - dL_dray_normal.y * ray_normal.y - drx * rpx - dry * rpy) / nl;
            float tmp = dpx * plane.x + dpy * plane.y;
            v3 W_uvh = m3_mulv(Wm, uvh);
            if (well_conditioned) {
                v3 rhs = m3_mulv(m3_div(Vrk_inv, clamp_vb),
                                 v3_add(v3_scale(W_uvh, -tmp), m3_mulv(m3_mul(Wm, m3_T(nJ_inv)), dL_dplane_append)));
                dL_dVrk = m3_neg(m3_outer(m3_mulv(Vrk_inv, W_uvh), rhs));
            } else {
                dL_dVrk = m3_zero();
                float dL_dvb = -tmp / clamp_vb;
                v3 nJ_inv_dL_dplane = m3_mulv(m3_T(nJ_inv), v3_mk(dpx / clamp_vb, dpy / clamp_vb, 0));
                m3 dL_dVrk_inv = m3_outer(W_uvh, v3_add(v3_scale(W_uvh, dL_dvb), m3_mulv(Wm, nJ_inv_dL_dplane)));
                v3 dL_dvv = m3_mulv(m3_add(dL_dVrk_inv, m3_T(dL_dVrk_inv)), emin);
                for (unsigned j = 0; j < 3; j++) {
                    if (j != min_id) {
                        float scale = v3_dot(m3_col(evec, j), dL_dvv) / fminf(eval[min_id] - eval[j], -0.0000001f);
                        dL_dVrk = m3_add(dL_dVrk, m3_outer(v3_scale(m3_col(evec, j), scale), emin));
                    }
                }
            }
            v3 dL_duvh = v3_add(v3_scale(uvh_m_vb, 2 * (-tmp)),
                                m3_mulv(m3_mul(m3_div(cov_cam_inv, clamp_vb), m3_T(nJ_inv)), dL_dplane_append));
            m3 dL_dnJ_inv = m3_outer(dL_dplane_append, uvh_m_vb);
            dL_du = dL_dnl * 2 * txtz + dL_duvh.x + (dL_dnJ_inv.m[0][1] + dL_dnJ_inv.m[1][0]) * (-tytz) +
                    2 * dL_dnJ_inv.m[1][1] * txtz - dL_dnJ_inv.m[2][0] + (cp0y * t.y + cp1x * t.y + cp1y * (-2 * t.x)) / nl;
            dL_dv = dL_dnl * 2 * tytz + dL_duvh.y + (dL_dnJ_inv.m[0][1] + dL_dnJ_inv.m[1][0]) * (-txtz) +
                    2 * dL_dnJ_inv.m[0][0] * tytz - dL_dnJ_inv.m[2][1] + (cp0x * (-2 * t.y) + cp0y * t.x + cp1x * t.x) / nl;
        }

        /* :367-375 -- double literals promote the expressions exactly as below */
        const float opacity = (float)(combined_opacity / (coef + 1e-6));
        const float dL_dcoef = dL_dopacity[idx] * opacity;
        const float dL_dsqrtcoef = (float)(dL_dcoef * 0.5 * 1. / (coef + 1e-6));
        const float dL_ddet0 = (float)(dL_dsqrtcoef / (det_1 + 1e-6));
        const float dL_ddet1 = (float)(dL_dsqrtcoef * det_0 * (-1.f / (det_1 * det_1 + 1e-6)));
        const float dcoef_da = dL_ddet0 * cov2D.m[1][1] + dL_ddet1 * (cov2D.m[1][1] + kernel_size);
        const float dcoef_db = (float)(dL_ddet0 * (-2. * cov2D.m[0][1]) + dL_ddet1 * (-2. * cov2D.m[0][1]));
        const float dcoef_dc = dL_ddet0 * cov2D.m[0][0] + dL_ddet1 * (cov2D.m[0][0] + kernel_size);
        float a = cov2D.m[0][0] + kernel_size, b = cov2D.m[0][1], c = cov2D.m[1][1] + kernel_size;
        float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float *dcov = dL_dcov + 6 * idx;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dLc_x + 2 * b * c * dLc_y + (denom - a * c) * dLc_z);
            dL_dc = denom2inv * (-a * a * dLc_z + 2 * a * b * dLc_y + (denom - a * c) * dLc_x);
            dL_db = denom2inv * 2 * (b * c * dLc_x - (denom + 2 * b * b) * dLc_y + a * b * dLc_z);
            if (det_0 <= 1e-6 || det_1 <= 1e-6) {
                dL_dopacity[idx] = 0;
            } else {
                dL_da += dcoef_da; dL_dc += dcoef_dc; dL_db += dcoef_db;
                dL_dopacity[idx] = dL_dopacity[idx] * coef;
            }
#define TT(c_, r_) T.m[c_][r_]
            dcov[0] = (TT(0,0) * TT(0,0) * dL_da + TT(0,0) * TT(1,0) * dL_db + TT(1,0) * TT(1,0) * dL_dc);
            dcov[3] = (TT(0,1) * TT(0,1) * dL_da + TT(0,1) * TT(1,1) * dL_db + TT(1,1) * TT(1,1) * dL_dc);
            dcov[5] = (TT(0,2) * TT(0,2) * dL_da + TT(0,2) * TT(1,2) * dL_db + TT(1,2) * TT(1,2) * dL_dc);
            dcov[1] = 2 * TT(0,0) * TT(0,1) * dL_da + (TT(0,0) * TT(1,1) + TT(0,1) * TT(1,0)) * dL_db + 2 * TT(1,0) * TT(1,1) * dL_dc;
            dcov[2] = 2 * TT(0,0) * TT(0,2) * dL_da + (TT(0,0) * TT(1,2) + TT(0,2) * TT(1,0)) * dL_db + 2 * TT(1,0) * TT(1,2) * dL_dc;
            dcov[4] = 2 * TT(0,2) * TT(0,1) * dL_da + (TT(0,1) * TT(1,2) + TT(0,2) * TT(1,1)) * dL_db + 2 * TT(1,1) * TT(1,2) * dL_dc;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
        dcov[0] += dL_dVrk.m[0][0]; dcov[3] += dL_dVrk.m[1][1]; dcov[5] += dL_dVrk.m[2][2];
        dcov[1] += dL_dVrk.m[0][1] + dL_dVrk.m[1][0];
        dcov[2] += dL_dVrk.m[0][2] + dL_dVrk.m[2][0];
        dcov[4] += dL_dVrk.m[1][2] + dL_dVrk.m[2][1];

#define VV(c_, r_) Vrk.m[c_][r_]
        float dL_dT00 = 2 * (TT(0,0) * VV(0,0) + TT(0,1) * VV(0,1) + TT(0,2) * VV(0,2)) * dL_da + (TT(1,0) * VV(0,0) + TT(1,1) * VV(0,1) + TT(1,2) * VV(0,2)) * dL_db;
        float dL_dT01 = 2 * (TT(0,0) * VV(1,0) + TT(0,1) * VV(1,1) + TT(0,2) * VV(1,2)) * dL_da + (TT(1,0) * VV(1,0) + TT(1,1) * VV(1,1) + TT(1,2) * VV(1,2)) * dL_db;
        float dL_dT02 = 2 * (TT(0,0) * VV(2,0) + TT(0,1) * VV(2,1) + TT(0,2) * VV(2,2)) * dL_da + (TT(1,0) * VV(2,0) + TT(1,1) * VV(2,1) + TT(1,2) * VV(2,2)) * dL_db;
        float dL_dT10 = 2 * (TT(1,0) * VV(0,0) + TT(1,1) * VV(0,1) + TT(1,2) * VV(0,2)) * dL_dc + (TT(0,0) * VV(0,0) + TT(0,1) * VV(0,1) + TT(0,2) * VV(0,2)) * dL_db;
        float dL_dT11 = 2 * (TT(1,0) * VV(1,0) + TT(1,1) * VV(1,1) + TT(1,2) * VV(1,2)) * dL_dc + (TT(0,0) * VV(1,0) + TT(0,1) * VV(1,1) + TT(0,2) * VV(1,2)) * dL_db;
        float dL_dT12 = 2 * (TT(1,0) * VV(2,0) + TT(1,1) * VV(2,1) + TT(1,2) * VV(2,2)) * dL_dc + (TT(0,0) * VV(2,0) + TT(0,1) * VV(2,1) + TT(0,2) * VV(2,2)) * dL_db;
#define WW(c_, r_) Wm.m[c_][r_]
        float dL_dJ00 = WW(0,0) * dL_dT00 + WW(0,1) * dL_dT01 + WW(0,2) * dL_dT02;
        float dL_dJ02 = WW(2,0) * dL_dT00 + WW(2,1) * dL_dT01 + WW(2,2) * dL_dT02;
        float dL_dJ11 = WW(1,0) * dL_dT10 + WW(1,1) * dL_dT11 + WW(1,2) * dL_dT12;
        float dL_dJ12 = WW(2,0) * dL_dT10 + WW(2,1) * dL_dT11 + WW(2,2) * dL_dT12;
        float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        float l3 = l * l * l;
#define NJ(c_, r_) dL_dnJ.m[c_][r_]
        float dL_dtx = x_grad_mul * (-h_x * tz2 * dL_dJ02 + dL_du * tz - NJ(0,2) * tz2 + NJ(2,0) * (1 / l - t.x * t.x / l3) +
                                     NJ(2,1) * (-t.x * t.y / l3) + NJ(2,2) * (-t.x * t.z / l3) +
                                     (cp0x * plane.x + cp0y * plane.y + cp2x) / nl + dL_dl * t.x / l);
        float dL_dty = y_grad_mul * (-h_y * tz2 * dL_dJ12 + dL_dv * tz - NJ(1,2) * tz2 + NJ(2,0) * (-t.x * t.y / l3) +
                                     NJ(2,1) * (1 / l - t.y * t.y / l3) + NJ(2,2) * (-t.y * t.z / l3) +
                                     (cp1x * plane.x + cp1y * plane.y + cp2y) / nl + dL_dl * t.y / l);
        float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12 -
                       (dL_du * t.x + dL_dv * t.y) * tz2 + (NJ(0,0) + NJ(1,1)) * (-tz2) + NJ(0,2) * (2 * t.x * tz3) +
                       NJ(1,2) * (2 * t.y * tz3) + (NJ(2,0) * t.x + NJ(2,1) * t.y) * (-t.z / l3) + NJ(2,2) * (1 / l - t.z * t.z / l3) +
                       (cp0x * (-(v2 + 1)) + cp0y * uv + cp1x * uv + cp1y * (-(u2 + 1)) + cp2x * plane.x + cp2y * plane.y) / nl +
                       dL_dl * t.z / l;
#undef TT
#undef VV
#undef WW
#undef NJ
        v3 dL_dmean = xformvec4x3T(v3_mk(dL_dtx, dL_dty, dL_dtz), view);
        dL_dmeans[3 * idx] = dL_dmean.x; dL_dmeans[3 * idx + 1] = dL_dmean.y; dL_dmeans[3 * idx + 2] = dL_dmean.z;
    }
}

/* CR/backward.cu:21-140 */
static void sh_backward(int idx, int deg, int max_coeffs, const float *means, const float *campos, const float *shs,
                        const uint8_t *clamped, const float *dL_dcolor, float *dL_dmeans, float *dL_dshs)
{
    v3 pos = v3_mk(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
    v3 dir_orig = v3_sub(pos, v3_mk(campos[0], campos[1], campos[2]));
    v3 dir = v3_divs(dir_orig, v3_len(dir_orig));
    const float *s = shs + (size_t)idx * max_coeffs * 3;
#define SH(k) v3_mk(s[3 * (k)], s[3 * (k) + 1], s[3 * (k) + 2])
    v3 dL_dRGB = v3_mk(dL_dcolor[3 * idx], dL_dcolor[3 * idx + 1], dL_dcolor[3 * idx + 2]);
    dL_dRGB.x *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB.y *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB.z *= clamped[3 * idx + 2] ? 0 : 1;
    v3 dRGBdx = v3_mk(0, 0, 0), dRGBdy = v3_mk(0, 0, 0), dRGBdz = v3_mk(0, 0, 0);
    float x = dir.x, y = dir.y, z = dir.z;
    float *out = dL_dshs + (size_t)idx * max_coeffs * 3;
#define SETSH(k, f) do { v3 q__ = v3_scale(dL_dRGB, (f)); out[3*(k)] = q__.x; out[3*(k)+1] = q__.y; out[3*(k)+2] = q__.z; } while (0)
    SETSH(0, SH_C0);
    if (deg > 0) {
        SETSH(1, -SH_C1 * y); SETSH(2, SH_C1 * z); SETSH(3, -SH_C1 * x);
        dRGBdx = v3_scale(SH(3), -SH_C1); dRGBdy = v3_scale(SH(1), -SH_C1); dRGBdz = v3_scale(SH(2), SH_C1);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            SETSH(4, SH_C2[0] * xy); SETSH(5, SH_C2[1] * yz); SETSH(6, SH_C2[2] * (2.f * zz - xx - yy));
            SETSH(7, SH_C2[3] * xz); SETSH(8, SH_C2[4] * (xx - yy));
            dRGBdx = v3_add(dRGBdx, v3_add(v3_add(v3_add(v3_scale(SH(4), SH_C2[0] * y), v3_scale(SH(6), SH_C2[2] * 2.f * -x)),
                                                  v3_scale(SH(7), SH_C2[3] * z)), v3_scale(SH(8), SH_C2[4] * 2.f * x)));
            dRGBdy = v3_add(dRGBdy, v3_add(v3_add(v3_add(v3_scale(SH(4), SH_C2[0] * x), v3_scale(SH(5), SH_C2[1] * z)),
                                                  v3_scale(SH(6), SH_C2[2] * 2.f * -y)), v3_scale(SH(8), SH_C2[4] * 2.f * -y)));
            dRGBdz = v3_add(dRGBdz, v3_add(v3_add(v3_scale(SH(5), SH_C2[1] * y), v3_scale(SH(6), SH_C2[2] * 2.f * 2.f * z)),
                                           v3_scale(SH(7), SH_C2[3] * x)));
            if (deg > 2) {
                SETSH(9, SH_C3[0] * y * (3.f * xx - yy)); SETSH(10, SH_C3[1] * xy * z);
                SETSH(11, SH_C3[2] * y * (4.f * zz - xx - yy)); SETSH(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                SETSH(13, SH_C3[4] * x * (4.f * zz - xx - yy)); SETSH(14, SH_C3[5] * z * (xx - yy));
                SETSH(15, SH_C3[6] * x * (xx - 3.f * yy));
                v3 ax = v3_scale(SH(9), SH_C3[0] * 3.f * 2.f * xy);
                ax = v3_add(ax, v3_scale(SH(10), SH_C3[1] * yz));
                ax = v3_add(ax, v3_scale(SH(11), SH_C3[2] * -2.f * xy));
                ax = v3_add(ax, v3_scale(SH(12), SH_C3[3] * -3.f * 2.f * xz));
                ax = v3_add(ax, v3_scale(SH(13), SH_C3[4] * (-3.f * xx + 4.f * zz - yy)));
                ax = v3_add(ax, v3_scale(SH(14), SH_C3[5] * 2.f * xz));
                ax = v3_add(ax, v3_scale(SH(15), SH_C3[6] * 3.f * (xx - yy)));
                dRGBdx = v3_add(dRGBdx, ax);
                v3 ay = v3_scale(SH(9), SH_C3[0] * 3.f * (xx - yy));
                ay = v3_add(ay, v3_scale(SH(10), SH_C3[1] * xz));
                ay = v3_add(ay, v3_scale(SH(11), SH_C3[2] * (-3.f * yy + 4.f * zz - xx)));
                ay = v3_add(ay, v3_scale(SH(12), SH_C3[3] * -3.f * 2.f * yz));
                ay = v3_add(ay, v3_scale(SH(13), SH_C3[4] * -2.f * xy));
                ay = v3_add(ay, v3_scale(SH(14), SH_C3[5] * -2.f * yz));
                ay = v3_add(ay, v3_scale(SH(15), SH_C3[6] * -3.f * 2.f * xy));
                dRGBdy = v3_add(dRGBdy, ay);
                v3 az = v3_scale(SH(10), SH_C3[1] * xy);
                az = v3_add(az, v3_scale(SH(11), SH_C3[2] * 4.f * 2.f * yz));
                az = v3_add(az, v3_scale(SH(12), SH_C3[3] * 3.f * (2.f * zz - xx - yy)));
                az = v3_add(az, v3_scale(SH(13), SH_C3[4] * 4.f * 2.f * xz));
                az = v3_add(az, v3_scale(SH(14), SH_C3[5] * (xx - yy)));
                dRGBdz = v3_add(dRGBdz, az);
            }
        }
    }
#undef SH
#undef SETSH
    v3 dL_ddir = v3_mk(v3_dot(dRGBdx, dL_dRGB), v3_dot(dRGBdy, dL_dRGB), v3_dot(dRGBdz, dL_dRGB));
    v3 dm = dnormvdv3(dir_orig, dL_ddir);
    dL_dmeans[3 * idx] += dm.x; dL_dmeans[3 * idx + 1] += dm.y; dL_dmeans[3 * idx + 2] += dm.z;
}

/* CR/backward.cu:492-555 */
static void cov3d_backward(int idx, const float *scale, float mod, const float *rot, const float *dL_dcov3Ds,
                           float *dL_dscales, float *dL_drots)
{
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                   2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                   2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float sx = mod * scale[0], sy = mod * scale[1], sz = mod * scale[2];
    S.m[0][0] = sx; S.m[1][1] = sy; S.m[2][2] = sz;
    m3 M = m3_mul(S, R);
    const float *d = dL_dcov3Ds + 6 * idx;
    m3 dL_dSigma = m3_cols(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
    m3 dL_dM = m3_mul(m3_scale(M, 2.0f), dL_dSigma);
    m3 Rt = m3_T(R), dL_dMt = m3_T(dL_dM);
    dL_dscales[3 * idx] = v3_dot(m3_col(Rt, 0), m3_col(dL_dMt, 0));
    dL_dscales[3 * idx + 1] = v3_dot(m3_col(Rt, 1), m3_col(dL_dMt, 1));
    dL_dscales[3 * idx + 2] = v3_dot(m3_col(Rt, 2), m3_col(dL_dMt, 2));
    for (int q = 0; q < 3; q++) { dL_dMt.m[0][q] *= sx; dL_dMt.m[1][q] *= sy; dL_dMt.m[2][q] *= sz; }
#define MT(c_, r_) dL_dMt.m[c_][r_]
    float *o = dL_drots + 4 * idx;
    o[0] = 2 * z * (MT(0,1) - MT(1,0)) + 2 * y * (MT(2,0) - MT(0,2)) + 2 * x * (MT(1,2) - MT(2,1));
    o[1] = 2 * y * (MT(1,0) + MT(0,1)) + 2 * z * (MT(2,0) + MT(0,2)) + 2 * r * (MT(1,2) - MT(2,1)) - 4 * x * (MT(2,2) + MT(1,1));
    o[2] = 2 * x * (MT(1,0) + MT(0,1)) + 2 * r * (MT(2,0) - MT(0,2)) + 2 * z * (MT(1,2) + MT(2,1)) - 4 * y * (MT(2,2) + MT(0,0));
    o[3] = 2 * r * (MT(0,1) - MT(1,0)) + 2 * x * (MT(2,0) + MT(0,2)) + 2 * y * (MT(1,2) + MT(2,1)) - 4 * z * (MT(1,1) + MT(0,0));
#undef MT
}

/* K9: CR/backward.cu:560-628.  dL_dmeans is accumulated into (+=), as at :615 and :139. */
void ed3ref_preprocess_backward(int P, int D, int M, const float *means, const int32_t *radii, const float *shs,
                                const uint8_t *clamped, const float *scales, const float *rotations,
                                float scale_modifier, const float *view, const float *proj, const float *campos,
                                const float *dL_dmean2D, const float *dL_dview_points, float *dL_dmeans,
                                const float *dL_dcolor, const float *dL_dts, const float *dL_dcov3D, float *dL_dsh,
                                float *dL_dscale, float *dL_drot)
{
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        v3 m = v3_mk(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
        float m_hom[4]; xform4x4(m, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * m.x + proj[4] * m.y + proj[8] * m.z + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;
        float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
        v3 d1;
        d1.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        d1.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        d1.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        v3 mv = xform4x3(m, view);
        float t = sqrtf(mv.x * mv.x + mv.y * mv.y + mv.z * mv.z);
        float dL_dt = dL_dts[idx];
        v3 gv = v3_mk(dL_dview_points[3 * idx], dL_dview_points[3 * idx + 1], dL_dview_points[3 * idx + 2]);
        v3 d2 = xformvec4x3T(v3_mk(gv.x + mv.x / t * dL_dt, gv.y + mv.y / t * dL_dt, gv.z + mv.z / t * dL_dt), view);
        dL_dmeans[3 * idx] += d1.x + d2.x; dL_dmeans[3 * idx + 1] += d1.y + d2.y; dL_dmeans[3 * idx + 2] += d1.z + d2.z;
        if (shs) sh_backward(idx, D, M, means, campos, shs, clamped, dL_dcolor, dL_dmeans, dL_dsh);
        if (scales) cov3d_backward(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
    }
}

/* exposes the eigen-solver and cov2d stage for unit tests */
int ed3ref_eig_sym3(const float cov6[6], float val[3], float vec9[9])
{
    m3 V = m3_cols(cov6[0], cov6[1], cov6[2], cov6[1], cov6[3], cov6[4], cov6[2], cov6[4], cov6[5]);
    m3 E; int n = eig_sym3(V, val, &E);
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) vec9[3 * c + r] = E.m[c][r];
    return n;
}


/* ===================== point integration (mesh-extraction probe) ===================== */
/* K12: CR/forward.cu:1027-1071.  Outputs zero-initialised by the caller. */
void ed3ref_preprocess_points(int PN, const float *pts, const float *view, int W, int H, float focal_x, float focal_y,
                              float *points2D, float *depths, uint32_t *tiles_touched)
{
    for (int i = 0; i < PN; i++) {
        tiles_touched[i] = 0;
        v3 pv = xform4x3(v3_mk(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), view);
        if (pv.z <= 0.2f) continue;
        float px = (float)(focal_x * pv.x / (pv.z + 0.0000001f) + W / 2.);
        float py = (float)(focal_y * pv.y / (pv.z + 0.0000001f) + H / 2.);
        if (px < 0 || px >= W || py < 0 || py >= H) continue;
        depths[i] = sqrtf(pv.x * pv.x + pv.y * pv.y + pv.z * pv.z);
        points2D[2 * i] = px; points2D[2 * i + 1] = py;
        tiles_touched[i] = 1;
    }
}

/* K13: CR/rasterizer_impl.cu:114-145 */
void ed3ref_create_with_keys(int PN, const float *points2D, const float *depths, const uint32_t *offsets,
                             const uint32_t *tiles_touched, int W, int H, uint64_t *keys, uint32_t *vals)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    for (int i = 0; i < PN; i++) {
        if (tiles_touched[i] == 0) continue;
        uint32_t off = (i == 0) ? 0 : offsets[i - 1];
        int x = imin(gx - 1, imax(0, (int)(points2D[2 * i] / TILE)));
        int y = imin(gy - 1, imax(0, (int)(points2D[2 * i + 1] / TILE)));
        uint64_t key = (uint64_t)(y * gx + x);
        key <<= 32;
        uint32_t dbits = depth_bits(depths[i]);
        key |= dbits;
        keys[off] = key; vals[off] = (uint32_t)i;
    }
}

#define MAX_NUM_CONTRIBUTORS 512 /* CR/auxiliary.h:31 */
#define MAX_NUM_PROJECTED 256    /* :32 */

static float relgap(float v, float thr) { return fabsf(v - thr) / thr; }

/* K14: CR/forward.cu:1109-1543, one pixel at a time (the reference's block-wide loops only share loads).
   out_color: 9 planes (0-2 colour, 3 expected ray distance, 4 median, 6 maximal, 7 alpha, 8 number of projected points).
   pix_margin / pt_margin (optional): smallest relative distance of any threshold decision taken for the pixel / point
   from its threshold, so that a test can leave out the cases that a last-ulp difference would flip. */
void ed3ref_integrate(int W, int H, const uint32_t *ranges, const uint32_t *point_ranges, const uint32_t *gaussian_list,
                      const uint32_t *point_list, float focal_x, float focal_y, const float *points2D,
                      const float *gaussians2D, const float *features, const float *ray_planes, const float *invraycov,
                      const float *point_depths, const float *gaussian_depths, const float *conic_opacity,
                      const uint8_t *condition, const float *bg, float *final_T, uint32_t *n_contrib, float *out_color,
                      float *out_alpha_integrated, float *out_color_integrated, float *out_coordinate2d, float *out_sdf,
                      float *pix_margin, float *pt_margin)
{
    (void)focal_x; (void)focal_y;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)H * W;
    static const float offx[5] = {0.0f, -0.5f, 0.5f, -0.5f, 0.5f}, offy[5] = {0.0f, -0.5f, -0.5f, 0.5f, 0.5f};
#pragma omp parallel for schedule(dynamic) collapse(2)
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const int tile = ty * gx + tx;
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const uint32_t q0 = point_ranges[2 * tile], q1 = point_ranges[2 * tile + 1];
            uint16_t *ids = (uint16_t *)malloc(sizeof(uint16_t) * MAX_NUM_CONTRIBUTORS * 4);
            for (int ly = 0; ly < TILE; ly++)
                for (int lx = 0; lx < TILE; lx++) {
                    const int px = tx * TILE + lx, py = ty * TILE + ly;
                    if (px >= W || py >= H) continue;
                    const size_t pix = (size_t)py * W + px;
                    const float pfx = (float)px + 0.5f, pfy = (float)py + 0.5f;
                    float T = 1.0f, cT[5] = {1.f, 1.f, 1.f, 1.f, 1.f};
                    float C[3] = {0, 0, 0}, Cd = 0, Cmed = 0, Cmax = 0, Ca = 0;
                    float mid_dc = 0, mid_plane[2] = {0, 0}, mid_mean[2] = {0, 0};
                    uint32_t contributor = 0, last = 0, n_local = 0;
                    float marg = 1e30f;
                    for (uint32_t i = r0; i < r1; i++) {
                        contributor++;
                        const uint32_t g = gaussian_list[i];
                        const float *co = conic_opacity + 4 * g;
                        const float dc = gaussian_depths[g];
                        const float *dp = ray_planes + 2 * g, *xy = gaussians2D + 2 * g;
                        int used = 0;
                        for (int k = 0; k < 5; k++) {
                            float dx = xy[0] - pfx - offx[k], dy = xy[1] - pfy - offy[k];
                            float depth = dc + (dp[0] * dx + dp[1] * dy);
                            float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                            if (power > 0.0f) continue;
                            float alpha = minf(0.99f, co[3] * expf(power));
                            marg = minf(marg, relgap(alpha, 1.0f / 255.0f));
                            if (alpha < 1.0f / 255.0f) continue;
                            float test_T = cT[k] * (1 - alpha);
                            marg = minf(marg, relgap(test_T, 0.0001f));
                            if (test_T < 0.0001f) continue;
                            if (k == 0) for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * g + ch] * alpha * T;
                            if (depth > Cmax) Cmax = depth;
                            if (k == 0) {
                                Ca += alpha * T;
                                Cd += depth * alpha * T;
                                marg = minf(marg, relgap(T, 0.5f));
                                if (T > 0.5f) { Cmed = depth; mid_dc = dc; mid_plane[0] = dp[0]; mid_plane[1] = dp[1]; mid_mean[0] = xy[0]; mid_mean[1] = xy[1]; }
                                T = test_T;
                            }
                            cT[k] = test_T;
                            used = 1;
                        }
                        if (used) {
                            last = contributor;
                            ids[n_local] = (uint16_t)contributor;
                            n_local++;
                            if (n_local >= MAX_NUM_CONTRIBUTORS * 4) break;
                        }
                    }
                    final_T[pix] = T;
                    n_contrib[pix] = last;
                    for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pix] = C[ch] + T * bg[ch];
                    out_color[3 * HW + pix] = Cd;
                    out_color[4 * HW + pix] = Cmed;
                    out_color[6 * HW + pix] = Cmax;
                    out_color[7 * HW + pix] = Ca;
                    if (pix_margin) pix_margin[pix] = marg;

                    /* the pixel's points, in batches of MAX_NUM_PROJECTED (:1336-1536) */
                    int proj_ids[MAX_NUM_PROJECTED];
                    float proj_xy[MAX_NUM_PROJECTED][2], proj_depth[MAX_NUM_PROJECTED];
                    uint32_t point_counter_last = 0;
                    int total_projected = 0, point_done = 0;
                    while (!point_done) {
                        int num_projected = 0, exceed = 0;
                        uint32_t point_counter = 0;
                        for (uint32_t q = q0; q < q1; q++) {
                            point_counter++;
                            if (point_counter <= point_counter_last) continue;
                            const uint32_t id = point_list[q];
                            const float x = points2D[2 * id], y = points2D[2 * id + 1];
                            if ((x >= (pfx - 0.5)) && (x < (pfx + 0.5)) && (y >= (pfy - 0.5)) && (y < (pfy + 0.5))) {
                                if (num_projected >= MAX_NUM_PROJECTED) { exceed = 1; break; }
                                proj_ids[num_projected] = (int)id; proj_xy[num_projected][0] = x; proj_xy[num_projected][1] = y;
                                proj_depth[num_projected] = point_depths[id];
                                num_projected++;
                            }
                        }
                        point_counter_last = point_counter - 1;
                        point_done = !exceed;
                        total_projected += num_projected;
                        float p_alpha[MAX_NUM_PROJECTED], p_T[MAX_NUM_PROJECTED], p_marg[MAX_NUM_PROJECTED];
                        for (int k = 0; k < num_projected; k++) { p_alpha[k] = 0.f; p_T[k] = 1.f; p_marg[k] = 1e30f; }
                        uint32_t num_iterated = 0;
                        uint16_t second = 0;
                        for (uint32_t i = r0; i < r1; i++) {
                            num_iterated++;
                            if (num_iterated > last) break;
                            if (num_iterated != (uint32_t)ids[second]) continue;
                            second++;
                            const uint32_t g = gaussian_list[i];
                            const float *co = conic_opacity + 4 * g, *dp = ray_planes + 2 * g, *xy = gaussians2D + 2 * g;
                            const float dc = gaussian_depths[g];
                            const float *c6 = invraycov + 6 * g;
                            m3 M = m3_cols(c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]);
                            for (int k = 0; k < num_projected; k++) {
                                float dx = xy[0] - proj_xy[k][0], dy = xy[1] - proj_xy[k][1];
                                float depth = dc + (dp[0] * dx + dp[1] * dy);
                                float alpha;
                                if (condition[g]) {
                                    v3 du = v3_mk(dx, dy, dc - minf(proj_depth[k], depth));
                                    float power = -0.5f * v3_dot(du, m3_mulv(M, du));
                                    alpha = minf(0.99f, co[3] * expf(power));
                                } else {
                                    p_marg[k] = minf(p_marg[k], fabsf(proj_depth[k] - depth) / fmaxf(fabsf(depth), 1e-6f));
                                    if (proj_depth[k] < depth) alpha = 0;
                                    else {
                                        v3 du = v3_mk(dx, dy, dc);
                                        float power = -0.5f * v3_dot(du, m3_mulv(M, du));
                                        alpha = minf(0.99f, co[3] * expf(power));
                                    }
                                }
                                p_marg[k] = minf(p_marg[k], relgap(alpha, 1.0f / 255.0f));
                                if (alpha < 1.0f / 255.0f) continue;
                                float test_T = p_T[k] * (1 - alpha);
                                p_alpha[k] += alpha * p_T[k];
                                p_T[k] = test_T;
                            }
                        }
                        for (int k = 0; k < num_projected; k++) {
                            const int id = proj_ids[k];
                            out_alpha_integrated[id] = p_alpha[k];
                            for (int ch = 0; ch < 3; ch++) out_color_integrated[3 * id + ch] = C[ch] + T * bg[ch];
                            out_coordinate2d[2 * id] = proj_xy[k][0]; out_coordinate2d[2 * id + 1] = proj_xy[k][1];
                            if (proj_depth[k] > 0) {
                                float dx = mid_mean[0] - proj_xy[k][0], dy = mid_mean[1] - proj_xy[k][1];
                                float depth = mid_dc + (mid_plane[0] * dx + mid_plane[1] * dy);
                                out_sdf[id] = depth - proj_depth[k];
                            }
                            if (pt_margin) pt_margin[id] = minf(p_marg[k], marg);
                        }
                    }
                    out_color[8 * HW + pix] = (float)total_projected;
                }
            free(ids);
        }
}
