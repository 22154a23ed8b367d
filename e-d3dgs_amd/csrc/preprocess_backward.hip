// preprocess_backward.hip -- K8 + K9 fused: per-Gaussian chain rule from the tile pass' gradient record back to
// means3D / scales / rotations (or cov3D) / SH / opacity.
// Math follows CR/backward.cu:145-488 (computeCov2DCUDA), :492-555 (computeCov3D bwd), :560-628 (preprocessCUDA bwd),
// :21-140 (SH bwd).  One launch instead of two, and every output element is written (zeros for culled Gaussians),
// so the caller needs no zero-filled tensors (DGR/rasterize_points.cu:184-197 allocates 14 of them).
#include "common.h"
#include "devmath.h"

#ifndef ED3_K89_ABLATE
#define ED3_K89_ABLATE 0   // timing experiments (tools/ab_build.sh; results wrong): 1 no phase 2, 2 no dead-row SH stream, 4 no SH backward, 8 no live-row SH copy-out
#endif

namespace ed3 {

__device__ const float BSH_C0 = 0.28209479177387814f;
__device__ const float BSH_C1 = 0.4886025119029199f;
__device__ const float BSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                    -1.0925484305920792f, 0.5462742152960396f};
__device__ const float BSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                    0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                    -0.5900435899266435f};

__device__ __forceinline__ v3 dnormvdv3(v3 v, v3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    v3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

// returns dL/dmean contribution; writes dL_dsh[0..M) -- IN PLACE over the coefficients: `row` holds the Gaussian's 3 M SH
// coefficients on entry (gathered into the caller's LDS row by the whole block, coalesced) and dL/dsh on return, so every
// coefficient is read (the direction derivatives) before anything is written (CR/backward.cu:24-147)
__device__ inline v3 sh_backward(int deg, int M, v3 pos, v3 campos, float *row, uint8_t clamped, v3 dL_dRGB)
{
    v3 dir_orig = pos - campos;
    v3 dir = dir_orig / len3(dir_orig);
    dL_dRGB.x *= (clamped & 1) ? 0.f : 1.f;
    dL_dRGB.y *= (clamped & 2) ? 0.f : 1.f;
    dL_dRGB.z *= (clamped & 4) ? 0.f : 1.f;
    v3 dx = mk3(0, 0, 0), dy = mk3(0, 0, 0), dz = mk3(0, 0, 0);
    const float x = dir.x, y = dir.y, z = dir.z;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#define SH(k) mk3(row[3 * (k)], row[3 * (k) + 1], row[3 * (k) + 2])
    if (deg > 0) {
        dx = SH(3) * (-BSH_C1); dy = SH(1) * (-BSH_C1); dz = SH(2) * BSH_C1;
        if (deg > 1) {
            dx = dx + (SH(4) * (BSH_C2[0] * y) + SH(6) * (BSH_C2[2] * 2.f * -x) + SH(7) * (BSH_C2[3] * z) + SH(8) * (BSH_C2[4] * 2.f * x));
            dy = dy + (SH(4) * (BSH_C2[0] * x) + SH(5) * (BSH_C2[1] * z) + SH(6) * (BSH_C2[2] * 2.f * -y) + SH(8) * (BSH_C2[4] * 2.f * -y));
            dz = dz + (SH(5) * (BSH_C2[1] * y) + SH(6) * (BSH_C2[2] * 2.f * 2.f * z) + SH(7) * (BSH_C2[3] * x));
            if (deg > 2) {
                dx = dx + (SH(9) * (BSH_C3[0] * 3.f * 2.f * xy) + SH(10) * (BSH_C3[1] * yz) + SH(11) * (BSH_C3[2] * -2.f * xy) +
                           SH(12) * (BSH_C3[3] * -3.f * 2.f * xz) + SH(13) * (BSH_C3[4] * (-3.f * xx + 4.f * zz - yy)) +
                           SH(14) * (BSH_C3[5] * 2.f * xz) + SH(15) * (BSH_C3[6] * 3.f * (xx - yy)));
                dy = dy + (SH(9) * (BSH_C3[0] * 3.f * (xx - yy)) + SH(10) * (BSH_C3[1] * xz) +
                           SH(11) * (BSH_C3[2] * (-3.f * yy + 4.f * zz - xx)) + SH(12) * (BSH_C3[3] * -3.f * 2.f * yz) +
                           SH(13) * (BSH_C3[4] * -2.f * xy) + SH(14) * (BSH_C3[5] * -2.f * yz) + SH(15) * (BSH_C3[6] * -3.f * 2.f * xy));
                dz = dz + (SH(10) * (BSH_C3[1] * xy) + SH(11) * (BSH_C3[2] * 4.f * 2.f * yz) +
                           SH(12) * (BSH_C3[3] * 3.f * (2.f * zz - xx - yy)) + SH(13) * (BSH_C3[4] * 4.f * 2.f * xz) +
                           SH(14) * (BSH_C3[5] * (xx - yy)));
            }
        }
    }
#undef SH
    // (`row` is not restrict-qualified: the stores below stay behind the loads above)
#define SETSH(k, f) do { v3 q__ = dL_dRGB * (f); row[3 * (k)] = q__.x; row[3 * (k) + 1] = q__.y; row[3 * (k) + 2] = q__.z; } while (0)
    SETSH(0, BSH_C0);
    int written = 1;
    if (deg > 0) {
        SETSH(1, -BSH_C1 * y); SETSH(2, BSH_C1 * z); SETSH(3, -BSH_C1 * x);
        written = 4;
        if (deg > 1) {
            SETSH(4, BSH_C2[0] * xy); SETSH(5, BSH_C2[1] * yz); SETSH(6, BSH_C2[2] * (2.f * zz - xx - yy));
            SETSH(7, BSH_C2[3] * xz); SETSH(8, BSH_C2[4] * (xx - yy));
            written = 9;
            if (deg > 2) {
                SETSH(9, BSH_C3[0] * y * (3.f * xx - yy)); SETSH(10, BSH_C3[1] * xy * z);
                SETSH(11, BSH_C3[2] * y * (4.f * zz - xx - yy)); SETSH(12, BSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy));
                SETSH(13, BSH_C3[4] * x * (4.f * zz - xx - yy)); SETSH(14, BSH_C3[5] * z * (xx - yy));
                SETSH(15, BSH_C3[6] * x * (xx - 3.f * yy));
                written = 16;
            }
        }
    }
#undef SETSH
    for (int k = written; k < M; k++) { row[3 * k] = 0.f; row[3 * k + 1] = 0.f; row[3 * k + 2] = 0.f; }
    v3 dL_ddir = mk3(dot3(dx, dL_dRGB), dot3(dy, dL_dRGB), dot3(dz, dL_dRGB));
    return dnormvdv3(dir_orig, dL_ddir);
}

__device__ inline void cov3d_backward(v3 scale, float mod, float4 rot, const float d[6], float *__restrict__ dL_dscale,
                                      float *__restrict__ dL_drot)
{
    float r = rot.x, x = rot.y, y = rot.z, z = rot.w;
    m3 R = cols3(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                 2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                 2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 S = cols3(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float sx = mod * scale.x, sy = mod * scale.y, sz = mod * scale.z;
    S.m[0][0] = sx; S.m[1][1] = sy; S.m[2][2] = sz;
    m3 M = mul3(S, R);
    m3 dL_dSigma = cols3(d[0], 0.5f * d[1], 0.5f * d[2], 0.5f * d[1], d[3], 0.5f * d[4], 0.5f * d[2], 0.5f * d[4], d[5]);
    m3 dL_dM = mul3(scale3(M, 2.0f), dL_dSigma);
    m3 Rt = tr3(R), dMt = tr3(dL_dM);
    dL_dscale[0] = dot3(col3(Rt, 0), col3(dMt, 0));
    dL_dscale[1] = dot3(col3(Rt, 1), col3(dMt, 1));
    dL_dscale[2] = dot3(col3(Rt, 2), col3(dMt, 2));
#pragma unroll
    for (int q = 0; q < 3; q++) { dMt.m[0][q] *= sx; dMt.m[1][q] *= sy; dMt.m[2][q] *= sz; }
#define MT(c_, r_) dMt.m[c_][r_]
    dL_drot[0] = 2 * z * (MT(0,1) - MT(1,0)) + 2 * y * (MT(2,0) - MT(0,2)) + 2 * x * (MT(1,2) - MT(2,1));
    dL_drot[1] = 2 * y * (MT(1,0) + MT(0,1)) + 2 * z * (MT(2,0) + MT(0,2)) + 2 * r * (MT(1,2) - MT(2,1)) - 4 * x * (MT(2,2) + MT(1,1));
    dL_drot[2] = 2 * x * (MT(1,0) + MT(0,1)) + 2 * r * (MT(2,0) - MT(0,2)) + 2 * z * (MT(1,2) + MT(2,1)) - 4 * y * (MT(2,2) + MT(0,0));
    dL_drot[3] = 2 * r * (MT(0,1) - MT(1,0)) + 2 * x * (MT(2,0) + MT(0,2)) + 2 * y * (MT(1,2) + MT(2,1)) - 4 * z * (MT(1,1) + MT(0,0));
#undef MT
}

// every output of one Gaussian as zero (culled, or its record untouched by the tile pass); sh_row / dL_dscale may be null
__device__ __forceinline__ void write_zero_outputs(int idx, int M, float *__restrict__ dL_dmean2D, float *__restrict__ dL_dcolor,
                                                   float *__restrict__ dL_dopacity, float *__restrict__ dL_dmean3D,
                                                   float *__restrict__ dL_dcov3D, float *__restrict__ sh_row,
                                                   float *__restrict__ dL_dscale, float *__restrict__ dL_drot)
{
#pragma unroll
    for (int i = 0; i < 3; i++) { dL_dmean2D[3 * idx + i] = 0.f; dL_dcolor[3 * idx + i] = 0.f; dL_dmean3D[3 * idx + i] = 0.f; }
    dL_dopacity[idx] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = 0.f;
    if (sh_row) for (int i = 0; i < 3 * M; i++) sh_row[i] = 0.f;
    if (dL_dscale) {
#pragma unroll
        for (int i = 0; i < 3; i++) dL_dscale[3 * idx + i] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) dL_drot[4 * idx + i] = 0.f;
    }
}

// one live Gaussian; sh_row: its SH coefficients on entry, dL/dsh on return (an LDS row of the caller, gathered / written out coalesced)
__device__ __forceinline__ void preprocess_backward_body(
    int idx, int P, int D, int M, const float *__restrict__ means, const int *__restrict__ radii, const float *__restrict__ shs,
    const float *__restrict__ scales, const float *__restrict__ rotations, float scale_modifier,
    const float *__restrict__ cov3D_precomp, const float *__restrict__ view, const float *__restrict__ proj,
    const float *__restrict__ campos, float h_x, float h_y, float tan_fovx, float tan_fovy, float kernel_size,
    const float *__restrict__ rec, const float *__restrict__ cov3Ds, const float *__restrict__ eig, const uint8_t *__restrict__ clamped,
    const float *__restrict__ grec, const float *__restrict__ grec_coord, bool has_colors_precomp, bool q1_reference,
    float hW, float hH, float *__restrict__ dL_dmean2D, float *__restrict__ dL_dcolor, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dmean3D, float *__restrict__ dL_dcov3D, float *__restrict__ sh_row,
    float *__restrict__ dL_dscale, float *__restrict__ dL_drot)
{
    // Only LIVE Gaussians get here (radius > 0 and a record the tile pass added to: phase 1 of the kernel).  Everything the
    // chain below reads is requested here, before any of it is used (the reference reads each input at its first use, several
    // behind branches: a dozen memory latencies in a row where one will do).
    float gr[GREC];
    {
        const float4 *g4 = reinterpret_cast<const float4 *>(grec + (size_t)idx * GREC);
#pragma unroll
        for (int i = 0; i < 4; i++) { float4 t = g4[i]; gr[4 * i] = t.x; gr[4 * i + 1] = t.y; gr[4 * i + 2] = t.z; gr[4 * i + 3] = t.w; }
    }
    float gcv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (grec_coord) {
        const float *gc = grec_coord + (size_t)idx * GREC;
#pragma unroll
        for (int i = 0; i < 9; i++) gcv[i] = gc[i];
    }
    float cov3D[6];
    {
        const float *src = cov3D_precomp ? cov3D_precomp + 6 * idx : cov3Ds + 6 * idx;
#pragma unroll
        for (int i = 0; i < 6; i++) cov3D[i] = src[i];
    }
    const v3 mean = mk3(means[3 * idx], means[3 * idx + 1], means[3 * idx + 2]);
    const float rec_w = rec[(size_t)idx * REC + R_W];
    const float4 *e4 = reinterpret_cast<const float4 *>(eig + (size_t)idx * 16);
    const float4 eg0 = e4[0], eg1 = e4[1], eg2 = e4[2];
    const float eg3 = eig[(size_t)idx * 16 + 12];
    const uint8_t clamp_bits = clamped[idx];
    v3 sc_in = mk3(0, 0, 0);
    float4 rot_in = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scales) {
        sc_in = mk3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
        rot_in = reinterpret_cast<const float4 *>(rotations)[idx];
    }
    // ---- unpack the tile pass' record, apply the per-Gaussian linear post-factors ----
    const v3 g_color = mk3(gr[G_R], gr[G_G], gr[G_B]);
    const float dL_dt = gr[G_TS];
    const float drx = gr[G_RPX] / h_x, dry = gr[G_RPY] / h_y;
    const v3 dL_dnormal = mk3(gr[G_NX], gr[G_NY], gr[G_NZ]);
    const float g2x = gr[G_MX] * hW, g2y = gr[G_MY] * hH, g2z = gr[G_MZ];
    const float dLc_x = -0.5f * gr[G_CX], dLc_y = -0.5f * gr[G_CY], dLc_z = -0.5f * gr[G_CW];
    float dLop = gr[G_OP];
    v3 gv = mk3(0, 0, 0);
    float cp0x = 0, cp0y = 0, cp1x = 0, cp1y = 0, cp2x = 0, cp2y = 0;
    if (grec_coord) {
        gv = mk3(gcv[0], gcv[1], gcv[2]);
        cp0x = gcv[3] / h_x; cp0y = gcv[4] / h_y; cp1x = gcv[5] / h_x; cp1y = gcv[6] / h_y; cp2x = gcv[7] / h_x; cp2y = gcv[8] / h_y;
    }
    dL_dmean2D[3 * idx] = g2x; dL_dmean2D[3 * idx + 1] = g2y; dL_dmean2D[3 * idx + 2] = g2z;
    dL_dcolor[3 * idx] = g_color.x; dL_dcolor[3 * idx + 1] = g_color.y; dL_dcolor[3 * idx + 2] = g_color.z;

    // ---- K8: conic / planes / normal -> cov3D, mean ----
    const float combined_opacity = q1_reference ? dLc_z : rec_w;  // Q1
    v3 t = xform4x3(mean, view);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
    txtz = t.x / t.z; tytz = t.y / t.z;

    m3 J = cols3(h_x / t.z, 0.0f, -(h_x * t.x) / (t.z * t.z), 0.0f, h_y / t.z, -(h_y * t.y) / (t.z * t.z), 0, 0, 0);
    m3 Wm = cols3(view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]);
    m3 Vrk = cols3(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 T = mul3(Wm, J);
    m3 cov2D = mul3(mul3(tr3(T), tr3(Vrk)), T);
    const float det_0 = (float)fmax(1e-6, (double)(cov2D.m[0][0] * cov2D.m[1][1] - cov2D.m[0][1] * cov2D.m[0][1]));
    const float det_1 = (float)fmax(1e-6, (double)((cov2D.m[0][0] + kernel_size) * (cov2D.m[1][1] + kernel_size) -
                                                    cov2D.m[0][1] * cov2D.m[0][1]));
    const float coef = (float)sqrt(det_0 / (det_1 + 1e-6) + 1e-6);

    m3 evec; float eval[3];
    // the forward's decomposition of the same matrix (K1 kept it: the iterative solver -- CR/auxiliary.h:217-401, which the
    // reference's backward runs again -- was 14 of this kernel's 63 us)
    const int Dn = (int)eg0.w;
    eval[0] = eg0.x; eval[1] = eg0.y; eval[2] = eg0.z;
    evec = cols3(eg1.x, eg1.y, eg1.z, eg1.w, eg2.x, eg2.y, eg2.z, eg2.w, eg3);
    unsigned min_id = eval[0] > eval[1] ? (eval[1] > eval[2] ? 2 : 1) : (eval[0] > eval[2] ? 2 : 0);
    m3 Vrk_inv; v3 emin = mk3(0, 0, 0);
    const float eval_min = min_id == 0 ? eval[0] : (min_id == 1 ? eval[1] : eval[2]);
    bool well_conditioned = eval_min > 0.00000001;
    if (well_conditioned) {
        m3 diag = cols3(1 / eval[0], 0, 0, 0, 1 / eval[1], 0, 0, 0, 1 / eval[2]);
        Vrk_inv = mul3(mul3(evec, diag), tr3(evec));
    } else {
        emin = min_id == 0 ? col3(evec, 0) : (min_id == 1 ? col3(evec, 1) : col3(evec, 2));
        Vrk_inv = outer3(emin, emin);
    }
    m3 cov_cam_inv = mul3(mul3(tr3(Wm), Vrk_inv), Wm);
    v3 uvh = mk3(txtz, tytz, 1);
    v3 uvh_m = mulv3(cov_cam_inv, uvh);
    v3 uvh_mn = normalize3(uvh_m);
    float u2 = txtz * txtz, v2 = tytz * tytz, uv = txtz * tytz;

    m3 dL_dVrk, dL_dnJ; v3 plane; float dL_du, dL_dv, dL_dl, l, nl;
    if (isnan(uvh_mn.x) || Dn == 0) {
        dL_dVrk = zero3(); dL_dnJ = zero3(); plane = mk3(0, 0, 0);
        nl = 1; l = 1; dL_du = 0; dL_dv = 0; dL_dl = 0;
    } else {
        float vb = dot3(uvh_m, uvh), vbn = dot3(uvh_mn, uvh);
        l = sqrtf(t.x * t.x + t.y * t.y + t.z * t.z);
        m3 nJ = cols3(1 / t.z, 0.0f, -(t.x) / (t.z * t.z), 0.0f, 1 / t.z, -(t.y) / (t.z * t.z), t.x / l, t.y / l, t.z / l);
        m3 nJ_inv = cols3(v2 + 1, -uv, 0, -uv, u2 + 1, 0, -txtz, -tytz, 0);
        float clamp_vb = fmaxf(vb, 0.0000001f), clamp_vbn = fmaxf(vbn, 0.0000001f);
        nl = u2 + v2 + 1;
        float factor_normal = l / nl;
        v3 uvh_m_vb = uvh_mn / clamp_vbn;
        plane = mulv3(nJ_inv, uvh_m_vb);
        float c0x = (-(v2 + 1) * t.z + plane.x * t.x) / nl, c0y = (uv * t.z + plane.y * t.x) / nl;
        float c1x = (uv * t.z + plane.x * t.y) / nl, c1y = (-(u2 + 1) * t.z + plane.y * t.y) / nl;
        float c2x = (t.x + plane.x * t.z) / nl, c2y = (t.y + plane.y * t.z) / nl;
        float rpx = plane.x * factor_normal, rpy = plane.y * factor_normal;
        v3 ray_normal = mk3(-plane.x * factor_normal, -plane.y * factor_normal, -1);
        v3 cam_normal = mulv3(nJ, ray_normal);
        v3 normal_vector = normalize3(cam_normal);
        float lv = len3(cam_normal);
        v3 dL_dnormal_lv = dL_dnormal / lv;
        v3 dL_dcam_normal = dL_dnormal_lv - normal_vector * dot3(normal_vector, dL_dnormal_lv);
        v3 dL_dray_normal = mulv3(tr3(nJ), dL_dcam_normal);
        dL_dnJ = outer3(dL_dcam_normal, ray_normal);
        dL_dl = (-plane.x * dL_dray_normal.x - plane.y * dL_dray_normal.y + plane.x * drx + plane.y * dry) / nl;
        float dpx = (t.x * cp0x + t.y * cp1x + t.z * cp2x - l * dL_dray_normal.x + drx * l) / nl;
        float dpy = (t.x * cp0y + t.y * cp1y + t.z * cp2y - l * dL_dray_normal.y + dry * l) / nl;
        v3 dpa = mk3(dpx, dpy, 0);
        float dL_dnl = (-cp0x * c0x - cp0y * c0y - cp1x * c1x - cp1y * c1y - cp2x * c2x - cp2y * c2y -
                        dL_dray_normal.x * ray_normal.x - dL_dray_normal.y * ray_normal.y - drx * rpx - dry * rpy) / nl;
        float tmp = dpx * plane.x + dpy * plane.y;
        v3 W_uvh = mulv3(Wm, uvh);
        if (well_conditioned) {
            v3 rhs = mulv3(div3(Vrk_inv, clamp_vb), W_uvh * (-tmp) + mulv3(mul3(Wm, tr3(nJ_inv)), dpa));
            dL_dVrk = neg3(outer3(mulv3(Vrk_inv, W_uvh), rhs));
        } else {
            dL_dVrk = zero3();
            float dL_dvb = -tmp / clamp_vb;
            v3 nJ_inv_dL_dplane = mulv3(tr3(nJ_inv), mk3(dpx / clamp_vb, dpy / clamp_vb, 0));
            m3 dL_dVrk_inv = outer3(W_uvh, W_uvh * dL_dvb + mulv3(Wm, nJ_inv_dL_dplane));
            v3 dL_dvv = mulv3(add3(dL_dVrk_inv, tr3(dL_dVrk_inv)), emin);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if ((unsigned)j != min_id) {
                    v3 ej = col3(evec, j);
                    float scale = dot3(ej, dL_dvv) / fminf(eval_min - eval[j], -0.0000001f);
                    dL_dVrk = add3(dL_dVrk, outer3(ej * scale, emin));
                }
            }
        }
        v3 dL_duvh = uvh_m_vb * (2 * (-tmp)) + mulv3(mul3(div3(cov_cam_inv, clamp_vb), tr3(nJ_inv)), dpa);
        m3 dL_dnJ_inv = outer3(dpa, uvh_m_vb);
        dL_du = dL_dnl * 2 * txtz + dL_duvh.x + (dL_dnJ_inv.m[0][1] + dL_dnJ_inv.m[1][0]) * (-tytz) +
                2 * dL_dnJ_inv.m[1][1] * txtz - dL_dnJ_inv.m[2][0] + (cp0y * t.y + cp1x * t.y + cp1y * (-2 * t.x)) / nl;
        dL_dv = dL_dnl * 2 * tytz + dL_duvh.y + (dL_dnJ_inv.m[0][1] + dL_dnJ_inv.m[1][0]) * (-txtz) +
                2 * dL_dnJ_inv.m[0][0] * tytz - dL_dnJ_inv.m[2][1] + (cp0x * (-2 * t.y) + cp0y * t.x + cp1x * t.x) / nl;
    }

    const float opacity = (float)(combined_opacity / (coef + 1e-6));
    const float dL_dcoef = dLop * opacity;
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
    float dcov[6];
#define TT(c_, r_) T.m[c_][r_]
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * dLc_x + 2 * b * c * dLc_y + (denom - a * c) * dLc_z);
        dL_dc = denom2inv * (-a * a * dLc_z + 2 * a * b * dLc_y + (denom - a * c) * dLc_x);
        dL_db = denom2inv * 2 * (b * c * dLc_x - (denom + 2 * b * b) * dLc_y + a * b * dLc_z);
        if (det_0 <= 1e-6 || det_1 <= 1e-6) {
            dLop = 0;
        } else {
            dL_da += dcoef_da; dL_dc += dcoef_dc; dL_db += dcoef_db;
            dLop = dLop * coef;
        }
        dcov[0] = (TT(0,0) * TT(0,0) * dL_da + TT(0,0) * TT(1,0) * dL_db + TT(1,0) * TT(1,0) * dL_dc);
        dcov[3] = (TT(0,1) * TT(0,1) * dL_da + TT(0,1) * TT(1,1) * dL_db + TT(1,1) * TT(1,1) * dL_dc);
        dcov[5] = (TT(0,2) * TT(0,2) * dL_da + TT(0,2) * TT(1,2) * dL_db + TT(1,2) * TT(1,2) * dL_dc);
        dcov[1] = 2 * TT(0,0) * TT(0,1) * dL_da + (TT(0,0) * TT(1,1) + TT(0,1) * TT(1,0)) * dL_db + 2 * TT(1,0) * TT(1,1) * dL_dc;
        dcov[2] = 2 * TT(0,0) * TT(0,2) * dL_da + (TT(0,0) * TT(1,2) + TT(0,2) * TT(1,0)) * dL_db + 2 * TT(1,0) * TT(1,2) * dL_dc;
        dcov[4] = 2 * TT(0,2) * TT(0,1) * dL_da + (TT(0,1) * TT(1,2) + TT(0,2) * TT(1,1)) * dL_db + 2 * TT(1,1) * TT(1,2) * dL_dc;
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) dcov[i] = 0;
    }
    dcov[0] += dL_dVrk.m[0][0]; dcov[3] += dL_dVrk.m[1][1]; dcov[5] += dL_dVrk.m[2][2];
    dcov[1] += dL_dVrk.m[0][1] + dL_dVrk.m[1][0];
    dcov[2] += dL_dVrk.m[0][2] + dL_dVrk.m[2][0];
    dcov[4] += dL_dVrk.m[1][2] + dL_dVrk.m[2][1];
    dL_dopacity[idx] = dLop;
#pragma unroll
    for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = dcov[i];

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
    v3 dmean = xformvec4x3T(mk3(dL_dtx, dL_dty, dL_dtz), view);

    // ---- K9: mean2D / view point / ts -> mean; SH; cov3D -> scale, rotation ----
    float4 m_hom = xform4x4(mean, proj);
    float m_w = 1.0f / (m_hom.w + 0.0000001f);
    float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
    float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
    v3 d1;
    d1.x = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
    d1.y = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
    d1.z = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    v3 mv = xform4x3(mean, view);
    float tl = sqrtf(mv.x * mv.x + mv.y * mv.y + mv.z * mv.z);
    v3 d2 = xformvec4x3T(mk3(gv.x + mv.x / tl * dL_dt, gv.y + mv.y / tl * dL_dt, gv.z + mv.z / tl * dL_dt), view);
    dmean = dmean + mk3(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z);
    if (shs && !has_colors_precomp && !(ED3_K89_ABLATE & 4)) {
        v3 ds = sh_backward(D, M, mean, mk3(campos[0], campos[1], campos[2]), sh_row, clamp_bits, g_color);
        dmean = dmean + ds;
    }
    dL_dmean3D[3 * idx] = dmean.x; dL_dmean3D[3 * idx + 1] = dmean.y; dL_dmean3D[3 * idx + 2] = dmean.z;
    if (scales) {
        cov3d_backward(sc_in, scale_modifier, rot_in, dcov, dL_dscale + 3 * idx, dL_drot + 4 * idx);
    }
}

// Block = KB_ROWS consecutive Gaussians, in two phases.  Half of a dense scene is culled or behind the last contributor of every
// tile it touches (an all-zero record, all-zero outputs), and with thread = Gaussian nearly every wave held a few live lanes
// and ran the whole chain for them (63 us at 200k, ~44 % live).  Phase 1: every thread classifies KB_ROWS / 256 Gaussians from
// radii and the tile pass' record alone, writes the zero outputs of the dead ones, and the block compacts the live ids into
// an LDS list (ballot + one LDS atomic per wave).  Phase 2: dense waves walk the list -- a wave with no entry leaves at once.
// dL_dsh is the largest output (192 B per Gaussian at degree 3); written per thread it is 48 stores of 4 bytes at a 192-byte
// lane stride.  The live rows are staged in LDS (odd row stride: conflict-free) and written out row by row, coalesced within
// a row; the dead rows are zeroed by the whole block as one stream with holes.
// (153 registers = 3 waves per SIMD; asking the allocator for 4 / 5 waves spills and loses, round 3.)
#ifndef ED3_K89_ROWS
#define ED3_K89_ROWS 512   // a multiple of 256
#endif
constexpr int KB_ROWS = ED3_K89_ROWS;
__global__ void __launch_bounds__(256) preprocess_backward_kernel(
    int P, int D, int M, const float *__restrict__ means, const int *__restrict__ radii, const float *__restrict__ shs,
    const float *__restrict__ scales, const float *__restrict__ rotations, float scale_modifier,
    const float *__restrict__ cov3D_precomp, const float *__restrict__ view, const float *__restrict__ proj,
    const float *__restrict__ campos, float h_x, float h_y, float tan_fovx, float tan_fovy, float kernel_size,
    const float *__restrict__ rec, const float *__restrict__ cov3Ds, const float *__restrict__ eig, const uint8_t *__restrict__ clamped,
    const float *__restrict__ grec, const float *__restrict__ grec_coord, bool has_colors_precomp, bool q1_reference,
    float hW, float hH, float *__restrict__ dL_dmean2D, float *__restrict__ dL_dcolor, float *__restrict__ dL_dopacity,
    float *__restrict__ dL_dmean3D, float *__restrict__ dL_dcov3D, float *__restrict__ dL_dsh,
    float *__restrict__ dL_dscale, float *__restrict__ dL_drot)
{
    extern __shared__ float sh_rows[];
    __shared__ int live_list[KB_ROWS];
    __shared__ uint8_t live_flag[KB_ROWS];
    __shared__ int n_live;
    const int w3 = 3 * M, ld = w3 | 1;                // odd stride
    const int tid = threadIdx.x, lane = tid & 63;
    const int first = blockIdx.x * KB_ROWS, nrow = min(KB_ROWS, P - first);
    const bool staged = shs && !has_colors_precomp;   // the case in which every Gaussian's SH row is written
    if (tid == 0) n_live = 0;
    __syncthreads();
    // ---- phase 1: classify, zero the dead Gaussians' small outputs, compact the live ids ----
#pragma unroll
    for (int k = 0; k < KB_ROWS / 256; k++) {
        const int i = k * 256 + tid, idx = first + i;
        bool live = false;
        if (i < nrow) {
            live = radii[idx] > 0;
            if (live) {
                const float4 *g4 = reinterpret_cast<const float4 *>(grec + (size_t)idx * GREC);
                bool untouched = true;
#pragma unroll
                for (int q = 0; q < GREC / 4; q++) { const float4 t = g4[q]; untouched &= t.x == 0.f && t.y == 0.f && t.z == 0.f && t.w == 0.f; }
                if (untouched && grec_coord) {
                    const float *gc = grec_coord + (size_t)idx * GREC;
#pragma unroll
                    for (int q = 0; q < 9; q++) untouched &= gc[q] == 0.f;
                }
                live = !untouched;
            }
            if (!live)
                write_zero_outputs(idx, M, dL_dmean2D, dL_dcolor, dL_dopacity, dL_dmean3D, dL_dcov3D,
                                   (shs && !staged && dL_dsh) ? dL_dsh + (size_t)idx * w3 : nullptr, scales ? dL_dscale : nullptr, dL_drot);
            live_flag[i] = live;
        }
        const unsigned long long m = __ballot(live);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&n_live, __popcll(m));
        wbase = __builtin_amdgcn_readfirstlane(wbase);
        if (live) live_list[wbase + __popcll(m & ((1ull << lane) - 1))] = idx;
    }
    __syncthreads();
    const int n = n_live;
    // ---- dead rows of dL_dsh: one coalesced stream with holes (under way while phase 2 computes) ----
    if (staged && n < nrow && !(ED3_K89_ABLATE & 2)) {
        float *out = dL_dsh + (size_t)first * w3;
        if ((w3 & 3) == 0) {
            const int q4 = w3 >> 2;
            for (int e = tid; e < nrow * q4; e += 256)
                if (!live_flag[e / q4]) reinterpret_cast<float4 *>(out)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int e = tid; e < nrow * w3; e += 256)
                if (!live_flag[e / w3]) out[e] = 0.f;
        }
    }
    // ---- phase 2: the live Gaussians, 256 at a time ----
    for (int r0 = 0; r0 < ((ED3_K89_ABLATE & 1) ? 0 : n); r0 += 256) {
        const int slot = r0 + tid, cnt = min(256, n - r0);
        const int q4 = w3 >> 2, tot4 = cnt * q4;
        if (staged) {
            // the live rows' SH coefficients into the LDS rows: 16-byte chunks, consecutive lanes along a row (one row per thread
            // read where it is used is 3 M loads at a 12 M-byte lane stride)
            if ((w3 & 3) == 0) {
                for (int e0 = tid; e0 < tot4; e0 += 1024) {
                    float4 v[4];
                    int dst[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int e = min(e0 + 256 * u, tot4 - 1), r = e / q4, c4 = e - r * q4;
                        v[u] = reinterpret_cast<const float4 *>(shs + (size_t)live_list[r0 + r] * w3)[c4];
                        dst[u] = r * ld + 4 * c4;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (e0 + 256 * u < tot4) { float *d = sh_rows + dst[u]; d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w; }
                }
            } else {
                for (int e = tid; e < cnt * w3; e += 256) {
                    const int r = e / w3, cc = e - r * w3;
                    sh_rows[r * ld + cc] = shs[(size_t)live_list[r0 + r] * w3 + cc];
                }
            }
            __syncthreads();
        }
        if (slot < n) {
            const int idx = live_list[slot];
            preprocess_backward_body(idx, P, D, M, means, radii, shs, scales, rotations, scale_modifier, cov3D_precomp, view, proj,
                                     campos, h_x, h_y, tan_fovx, tan_fovy, kernel_size, rec, cov3Ds, eig, clamped, grec, grec_coord,
                                     has_colors_precomp, q1_reference, hW, hH, dL_dmean2D, dL_dcolor, dL_dopacity, dL_dmean3D,
                                     dL_dcov3D, staged ? sh_rows + tid * ld : nullptr, dL_dscale, dL_drot);
        }
        if (!staged) continue;
        __syncthreads();
        if (ED3_K89_ABLATE & 8) continue;
        if ((w3 & 3) == 0) {
            for (int e = tid; e < tot4; e += 256) {
                const int r = e / q4, c4 = e - r * q4;
                const float *sr = sh_rows + r * ld + 4 * c4;
                reinterpret_cast<float4 *>(dL_dsh + (size_t)live_list[r0 + r] * w3)[c4] = make_float4(sr[0], sr[1], sr[2], sr[3]);
            }
        } else {
            for (int e = tid; e < cnt * w3; e += 256) {
                const int r = e / w3, cc = e - r * w3;
                dL_dsh[(size_t)live_list[r0 + r] * w3 + cc] = sh_rows[r * ld + cc];
            }
        }
        if (r0 + 256 < n) __syncthreads();
    }
}

void launch_preprocess_backward(int P, int D, int M, const float *means, const int *radii, const float *shs,
                                const float *scales, const float *rotations, float scale_modifier,
                                const float *cov3D_precomp, const float *view, const float *proj, const float *campos,
                                float focal_x, float focal_y, float tan_fovx, float tan_fovy, float kernel_size,
                                GeometryState g, const float *grec, const float *grec_coord, bool colors_precomp,
                                bool q1_reference, int W, int H, float *dL_dmean2D, float *dL_dcolor,
                                float *dL_dopacity, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                                float *dL_dscale, float *dL_drot, hipStream_t s)
{
    const size_t lds = (shs && !colors_precomp) ? (size_t)256 * ((3 * M) | 1) * sizeof(float) : 0;   // SH gradient rows
    hipLaunchKernelGGL(preprocess_backward_kernel, dim3((P + KB_ROWS - 1) / KB_ROWS), dim3(256), lds, s, P, D, M, means, radii, shs,
                       scales, rotations, scale_modifier, cov3D_precomp, view, proj, campos, focal_x, focal_y, tan_fovx,
                       tan_fovy, kernel_size, g.rec, g.cov3D, g.eig, g.clamped, grec, grec_coord, colors_precomp, q1_reference,
                       0.5f * W, 0.5f * H, dL_dmean2D, dL_dcolor, dL_dopacity, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale,
                       dL_drot);
}

}  // namespace ed3
