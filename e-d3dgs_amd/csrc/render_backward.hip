// render_backward.hip -- K7: reverse-order traversal of one 16x16 tile per wavefront, producing per-Gaussian
// gradient records.  Computes what CR/backward.cu:631-1016 (renderCUDA backward) computes; differences in HOW:
//  * the suffix blends (accum_rec etc.) are advanced eagerly at the end of an iteration instead of lazily at the
//    start of the next one (same operands, no last_* copies -> 8 fewer live registers per pixel);
//  * the reference issues 10-25 float atomics per (pixel, Gaussian) pair; here each lane first sums its 4 pixels,
//    the 16 (or 32) partial sums are reduced across the wavefront with a DPP butterfly that leaves sum k in lane k,
//    and ONE 64-byte atomic wave-instruction per (tile, Gaussian) adds the record;
//  * iteration starts at the tile's largest last-contributor instead of the end of the tile list.
// Linear post-factors (1/focal on plane gradients, -0.5 on the conic, W/2,H/2 on mean2D) are applied once per
// Gaussian by the per-Gaussian backward kernel.
#include "raster_common.h"

namespace ed3 {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}

// Butterfly transpose-reduction: on entry every lane holds NV partial sums; on return lane l holds the wave-wide
// total of value (l & (NV-1)).  quad_perm for lane-xor 1 and 2, row_ror:4 / row_ror:8 inside a row of 16,
// ds_bpermute for the cross-row steps.
template <int NV>
__device__ __forceinline__ float wave_transpose_reduce(float (&v)[NV], int lane)
{
    static_assert(NV == 16 || NV == 32, "NV");
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    float a[NV / 2];
#pragma unroll
    for (int i = 0; i < NV / 2; i++) {
        const float keep = b0 ? v[2 * i + 1] : v[2 * i];
        const float send = b0 ? v[2 * i] : v[2 * i + 1];
        a[i] = keep + dpp_mov<0xB1>(send);  // quad_perm [1,0,3,2]
    }
    float b[NV / 4];
#pragma unroll
    for (int i = 0; i < NV / 4; i++) {
        const float keep = b1 ? a[2 * i + 1] : a[2 * i];
        const float send = b1 ? a[2 * i] : a[2 * i + 1];
        b[i] = keep + dpp_mov<0x4E>(send);  // quad_perm [2,3,0,1]
    }
    float c[NV / 8];
#pragma unroll
    for (int i = 0; i < NV / 8; i++) {
        const float keep = b2 ? b[2 * i + 1] : b[2 * i];
        const float send = b2 ? b[2 * i] : b[2 * i + 1];
        c[i] = keep + dpp_mov<0x124>(send);  // row_ror:4
    }
    float d[NV / 16];
#pragma unroll
    for (int i = 0; i < NV / 16; i++) {
        const float keep = b3 ? c[2 * i + 1] : c[2 * i];
        const float send = b3 ? c[2 * i] : c[2 * i + 1];
        d[i] = keep + dpp_mov<0x128>(send);  // row_ror:8
    }
    float z;
    if (NV == 32) {
        const bool b4 = lane & 16;
        const float keep = b4 ? d[NV / 16 - 1] : d[0];
        const float send = b4 ? d[0] : d[NV / 16 - 1];
        z = keep + __shfl_xor(send, 16);
    } else {
        z = d[0];
        z += __shfl_xor(z, 16);
    }
    z += __shfl_xor(z, 32);
    return z;
}

template <bool COORD, bool DEPTH>
__global__ void __launch_bounds__(64) render_backward_kernel(
    int W, int H, int gx, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
    const float4 *__restrict__ rec, const float4 *__restrict__ rec_coord, float focal_x, float focal_y,
    const float *__restrict__ bg, const float *__restrict__ alphas, const float *__restrict__ normalmap,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ accum_coord,
    const float *__restrict__ accum_depth, const float *__restrict__ normal_length,
    const float *__restrict__ dL_dpix, const float *__restrict__ dL_dcoord, const float *__restrict__ dL_dmcoord,
    const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dmdepth, const float *__restrict__ dL_dalpha,
    const float *__restrict__ dL_dnormal, float *__restrict__ grec, float *__restrict__ grec_coord)
{
    constexpr bool GEO = COORD || DEPTH;
    constexpr int NV = COORD ? 32 : 16;
    __shared__ float4 s_rec[64 * 4];
    __shared__ float4 s_recc[COORD ? 64 * 3 : 1];
    __shared__ uint32_t s_id[64];

    const int tile = blockIdx.x;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int px0 = tx * TILE + (lane & 3) * 4;
    const int py = ty * TILE + (lane >> 2);
    const size_t HW = (size_t)H * W;
    const int nvalid = (py < H) ? max(0, min(4, W - px0)) : 0;
    const bool vec = (nvalid == 4) && ((W & 3) == 0);
    const size_t pix0 = (size_t)py * W + px0;
    const float fpy = (float)py;
    float fpx[4];
#pragma unroll
    for (int p = 0; p < 4; p++) fpx[p] = (float)(px0 + p);
    const uint2 range = ranges[tile];

    // ---- per-pixel state ----
    float T[4], PB[4];                        // transmittance; -T_final * (bg . dL_dpixel)
    float g0[4], g1[4], g2[4], gA[4];         // dL_dpixel rgb, adjusted dL_dalpha
    float gT[4], gMT[4], gN0[4], gN1[4], gN2[4];
    float gC0[4], gC1[4], gC2[4], gM0[4], gM1[4], gM2[4];
    float ar0[4], ar1[4], ar2[4], aa[4], at[4], an0[4], an1[4], an2[4], ac0[4], ac1[4], ac2[4];
    uint32_t last[4], maxc[4];
    {
        float al[4];
        uint32_t lc[4], mc[4];
        if (nvalid) {
            load4(alphas, pix0, al, vec, nvalid);
            load4u(n_contrib, pix0, lc, vec, nvalid);
            load4u(n_contrib + HW, pix0, mc, vec, nvalid);
            load4(dL_dpix, pix0, g0, vec, nvalid);
            load4(dL_dpix + HW, pix0, g1, vec, nvalid);
            load4(dL_dpix + 2 * HW, pix0, g2, vec, nvalid);
            load4(dL_dalpha, pix0, gA, vec, nvalid);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const bool in = p < nvalid;
            if (!in) { al[p] = 0.f; lc[p] = 0; mc[p] = 0; g0[p] = g1[p] = g2[p] = gA[p] = 0.f; }
            last[p] = lc[p]; maxc[p] = mc[p];
            T[p] = 1.f - al[p];
            PB[p] = -(T[p]) * (bg[0] * g0[p] + bg[1] * g1[p] + bg[2] * g2[p]);
            ar0[p] = ar1[p] = ar2[p] = aa[p] = at[p] = an0[p] = an1[p] = an2[p] = 0.f;
            ac0[p] = ac1[p] = ac2[p] = 0.f;
            gT[p] = gMT[p] = gN0[p] = gN1[p] = gN2[p] = 0.f;
            gC0[p] = gC1[p] = gC2[p] = gM0[p] = gM1[p] = gM2[p] = 0.f;
        }
        if (GEO && nvalid) {
            float ww[4];
#pragma unroll
            for (int p = 0; p < 4; p++) ww[p] = al[p] * al[p];
            if (COORD) {
                float *gC[3] = {gC0, gC1, gC2};
                float *gM[3] = {gM0, gM1, gM2};
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    float gcw[4], acc[4];
                    load4(dL_dcoord + ch * HW, pix0, gcw, vec, nvalid);
                    load4(accum_coord + ch * HW, pix0, acc, vec, nvalid);
                    load4(dL_dmcoord + ch * HW, pix0, gM[ch], vec, nvalid);
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        gA[p] -= gcw[p] * acc[p] / ww[p];
                        gC[ch][p] = gcw[p] / al[p];
                    }
                }
            }
            if (DEPTH) {
                float gd[4], acd[4], gmd[4];
                load4(dL_ddepth, pix0, gd, vec, nvalid);
                load4(accum_depth, pix0, acd, vec, nvalid);
                load4(dL_dmdepth, pix0, gmd, vec, nvalid);
                const float pny = (fpy - H / 2.f) / focal_y;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float pnx = (fpx[p] - W / 2.f) / focal_x;
                    const float ln = sqrtf(pnx * pnx + pny * pny + 1);
                    gA[p] -= gd[p] * acd[p] / ww[p];
                    gT[p] = gd[p] / al[p] / ln;
                    gMT[p] = gmd[p] / ln;
                }
            }
            {
                float gn[3][4], nn[3][4], nl[4];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    load4(dL_dnormal + ch * HW, pix0, gn[ch], vec, nvalid);
                    load4(normalmap + ch * HW, pix0, nn[ch], vec, nvalid);
                }
                load4(normal_length, pix0, nl, vec, nvalid);
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    if (nl[p] < NORMALIZE_EPS) {
                        gN0[p] = gn[0][p] / NORMALIZE_EPS; gN1[p] = gn[1][p] / NORMALIZE_EPS; gN2[p] = gn[2][p] / NORMALIZE_EPS;
                    } else {
                        const float d = gn[0][p] * nn[0][p] + gn[1][p] * nn[1][p] + gn[2][p] * nn[2][p];
                        gN0[p] = (gn[0][p] - d * nn[0][p]) / nl[p];
                        gN1[p] = (gn[1][p] - d * nn[1][p]) / nl[p];
                        gN2[p] = (gn[2][p] - d * nn[2][p]) / nl[p];
                    }
                }
            }
        }
    }

    // largest last-contributor of the tile: nothing behind it is blended by any pixel
    uint32_t lmax = max(max(last[0], last[1]), max(last[2], last[3]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, off));
    const int tile_max = (int)lmax;
    const float hW = 0.5f * W, hH = 0.5f * H;

    for (int top = tile_max; top > 0; top -= 64) {
        __syncthreads();
        const int cnt = min(64, top);
        if (lane < cnt) {
            const uint32_t id = point_list[range.x + (uint32_t)(top - 1 - lane)];
            const float4 *src = rec + (size_t)id * 4;
            s_id[lane] = id;
            s_rec[lane * 4 + 0] = src[0];
            s_rec[lane * 4 + 1] = src[1];
            s_rec[lane * 4 + 2] = src[2];
            if (GEO) s_rec[lane * 4 + 3] = src[3];
            if (COORD) {
                const float4 *sc = rec_coord + (size_t)id * 3;
                s_recc[lane * 3 + 0] = sc[0]; s_recc[lane * 3 + 1] = sc[1]; s_recc[lane * 3 + 2] = sc[2];
            }
        }
        __syncthreads();
        for (int j = 0; j < cnt; j++) {
            const uint32_t k = (uint32_t)(top - 1 - j);  // 0-based position in the tile list
            const float4 r0 = s_rec[j * 4 + 0];          // x, y, cx, cy
            const float4 r1 = s_rec[j * 4 + 1];          // cz, w, r, g
            const float dy = r0.y - fpy;
            const ConicRow cr = conic_row(r0.z, r0.w, r1.x, dy);
            float dx[4], alpha[4], G[4];
            bool valid[4];
            bool any_valid = false;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                dx[p] = r0.x - fpx[p];
                const float pw2 = conic_power2(cr, dx[p]);
                G[p] = gauss_G(pw2);
                alpha[p] = gauss_alpha(r1.y, G[p]);
                valid[p] = (k < last[p]) && !(pw2 > 0.0f) && !(alpha[p] < ALPHA_MIN);
                any_valid |= valid[p];
            }
            if (!__any(any_valid)) continue;

            const float4 r2 = s_rec[j * 4 + 2];  // b, tongue, ts, rpx
            float4 r3 = make_float4(0, 0, 0, 0); // rpy, nx, ny, nz
            if (GEO) r3 = s_rec[j * 4 + 3];
            float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0;
            if (COORD) { q0 = s_recc[j * 3 + 0]; q1 = s_recc[j * 3 + 1]; q2 = s_recc[j * 3 + 2]; }
            const float t_row = DEPTH ? (r2.z + r3.x * dy) : 0.f;

            float acc[NV];
#pragma unroll
            for (int i = 0; i < NV; i++) acc[i] = 0.f;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const bool vd = valid[p];
                const float a = alpha[p];
                const float om = 1.f - a;
                const float inv = __builtin_amdgcn_rcpf(om);
                const float Tn = T[p] * inv;           // T / (1 - alpha)
                const float wgt = a * Tn;              // dchannel_dcolor
                const bool is_med = vd && (k + 1u == maxc[p]);
                float dopa = (r1.z - ar0[p]) * g0[p] + (r1.w - ar1[p]) * g1[p] + (r2.x - ar2[p]) * g2[p];
                acc[G_R] += vd ? wgt * g0[p] : 0.f;
                acc[G_G] += vd ? wgt * g1[p] : 0.f;
                acc[G_B] += vd ? wgt * g2[p] : 0.f;
                float ddelx_geo = 0.f, ddely_geo = 0.f;
                float c0 = 0.f, c1 = 0.f, c2 = 0.f;
                if (COORD) {
                    c0 = q1.z + q0.x * dx[p] + q0.y * dy;
                    c1 = q1.w + q0.z * dx[p] + q0.w * dy;
                    c2 = q2.x + q1.x * dx[p] + q1.y * dy;
                    dopa += (c0 - ac0[p]) * gC0[p] + (c1 - ac1[p]) * gC1[p] + (c2 - ac2[p]) * gC2[p];
                    float d0 = wgt * gC0[p] + (is_med ? gM0[p] : 0.f);
                    float d1 = wgt * gC1[p] + (is_med ? gM1[p] : 0.f);
                    float d2 = wgt * gC2[p] + (is_med ? gM2[p] : 0.f);
                    d0 = vd ? d0 : 0.f; d1 = vd ? d1 : 0.f; d2 = vd ? d2 : 0.f;
                    acc[16] += d0; acc[17] += d1; acc[18] += d2;
                    acc[19] += d0 * dx[p]; acc[20] += d0 * dy;
                    acc[21] += d1 * dx[p]; acc[22] += d1 * dy;
                    acc[23] += d2 * dx[p]; acc[24] += d2 * dy;
                    ddelx_geo += d0 * q0.x + d1 * q0.z + d2 * q1.x;
                    ddely_geo += d0 * q0.y + d1 * q0.w + d2 * q1.y;
                }
                float tt = 0.f;
                if (DEPTH) {
                    tt = t_row + r2.w * dx[p];
                    dopa += (tt - at[p]) * gT[p];
                    float dLdt = wgt * gT[p] + (is_med ? gMT[p] : 0.f);
                    dLdt = vd ? dLdt : 0.f;
                    acc[G_TS] += dLdt;
                    acc[G_RPX] += dLdt * dx[p];
                    acc[G_RPY] += dLdt * dy;
                    ddelx_geo += dLdt * r2.w;
                    ddely_geo += dLdt * r3.x;
                }
                if (GEO) {
                    dopa += (r3.y - an0[p]) * gN0[p] + (r3.z - an1[p]) * gN1[p] + (r3.w - an2[p]) * gN2[p];
                    acc[G_NX] += vd ? wgt * gN0[p] : 0.f;
                    acc[G_NY] += vd ? wgt * gN1[p] : 0.f;
                    acc[G_NZ] += vd ? wgt * gN2[p] : 0.f;
                }
                dopa += (1.f - aa[p]) * gA[p];
                dopa *= Tn;
                dopa += PB[p] * inv;
                float gd = G[p] * dopa;      // -> dL_dopacity
                gd = vd ? gd : 0.f;
                const float e = r1.y * gd;   // G * dL_dG
                const float gx_ = -e * (dx[p] * r0.z + dy * r0.w);  // dL_dG * dG_ddelx
                const float gy_ = -e * (dy * r1.x + dx[p] * r0.w);  // dL_dG * dG_ddely
                acc[G_MX] += gx_ + ddelx_geo;
                acc[G_MY] += gy_ + ddely_geo;
                acc[G_MZ] += fabsf(gx_ * hW) + fabsf(gy_ * hH);
                acc[G_CX] += e * dx[p] * dx[p];
                acc[G_CY] += e * dx[p] * dy;
                acc[G_CW] += e * dy * dy;
                acc[G_OP] += gd;
                // advance the suffix blends and the transmittance (eager form of :870,:900,:930,:949,:962)
                T[p] = vd ? Tn : T[p];
                ar0[p] = vd ? a * r1.z + om * ar0[p] : ar0[p];
                ar1[p] = vd ? a * r1.w + om * ar1[p] : ar1[p];
                ar2[p] = vd ? a * r2.x + om * ar2[p] : ar2[p];
                aa[p] = vd ? a + om * aa[p] : aa[p];
                if (COORD) {
                    ac0[p] = vd ? a * c0 + om * ac0[p] : ac0[p];
                    ac1[p] = vd ? a * c1 + om * ac1[p] : ac1[p];
                    ac2[p] = vd ? a * c2 + om * ac2[p] : ac2[p];
                }
                if (DEPTH) at[p] = vd ? a * tt + om * at[p] : at[p];
                if (GEO) {
                    an0[p] = vd ? a * r3.y + om * an0[p] : an0[p];
                    an1[p] = vd ? a * r3.z + om * an1[p] : an1[p];
                    an2[p] = vd ? a * r3.w + om * an2[p] : an2[p];
                }
            }
            const float z = wave_transpose_reduce<NV>(acc, lane);
            const uint32_t id = s_id[j];
            if (lane < 16) {
                atomicAdd(grec + (size_t)id * GREC + lane, z);
            } else if (COORD && lane < 16 + 9) {
                atomicAdd(grec_coord + (size_t)id * GREC + (lane - 16), z);
            }
        }
    }
}

void launch_render_backward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                            const float *rec_coord, float focal_x, float focal_y, const float *bg, bool coord,
                            bool depth, const float *alphas, const float *normalmap, ImageState img,
                            const float *dL_dpix, const float *dL_dcoord, const float *dL_dmcoord,
                            const float *dL_ddepth, const float *dL_dmdepth, const float *dL_dalpha,
                            const float *dL_dnormal, float *grec, float *grec_coord, hipStream_t s)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    dim3 grid(gx * gy), block(64);
#define ED3_BWD(C_, D_)                                                                                              \
    hipLaunchKernelGGL((render_backward_kernel<C_, D_>), grid, block, 0, s, W, H, gx,                                \
                       reinterpret_cast<const uint2 *>(ranges), point_list, reinterpret_cast<const float4 *>(rec),  \
                       reinterpret_cast<const float4 *>(rec_coord), focal_x, focal_y, bg, alphas, normalmap,         \
                       img.n_contrib, img.accum_coord, img.accum_depth, img.normal_length, dL_dpix, dL_dcoord,       \
                       dL_dmcoord, dL_ddepth, dL_dmdepth, dL_dalpha, dL_dnormal, grec, grec_coord)
    if (coord && depth) ED3_BWD(true, true);
    else if (coord) ED3_BWD(true, false);
    else if (depth) ED3_BWD(false, true);
    else ED3_BWD(false, false);
#undef ED3_BWD
}

}  // namespace ed3
