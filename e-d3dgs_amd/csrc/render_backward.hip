// render_backward.hip -- K7: reverse-order traversal of one 16x16 tile per wavefront, producing per-Gaussian
// gradient records.  Computes what CR/backward.cu:631-1016 (renderCUDA backward) computes; differences in HOW:
//  * the per-channel suffix blends (accum_rec, accum_alpha_rec, ...; :870-:962) are replaced by ONE per-pixel suffix sum
//    V = sum_{j behind k} alpha_j T_j s_j, s_j = <channel values of Gaussian j at the pixel, upstream channel gradients>:
//    (c - accum_rec) * T_k == c * T_k - U_k / (1 - alpha_k), and contracting over channels first needs a single
//    accumulator instead of one per channel (the background term is the last layer, weight T_final);
//  * a lane's 4 pixels are processed as two f32x2 pairs so the arithmetic issues as v_pk_fma/mul/add_f32; skipped
//    pixels run with alpha = G = 0 (updates are the identity, contributions exact zeros) instead of selects;
//  * sums that are linear in dy (a lane constant) are accumulated without it and scaled once per Gaussian;
//  * the reference issues 10-25 float atomics per (pixel, Gaussian) pair; here each lane first sums its 4 pixels,
//    the 16 (or 32) partial sums are reduced inside the quadrant's DPP row with a butterfly that leaves sum k in lane k, and the
//    row's 16 lanes add the 64-byte record with one atomic wave-instruction per (tile, Gaussian, quadrant); with
//    -DED3_K7_LDS_TILE=1 the quadrants' rows meet in a chunk-local LDS tile first (ds_add_f32, 64 entries x 16 floats) and ONE
//    record per touched (tile, Gaussian) entry leaves at the chunk boundary -- built and measured in round 4, see the macro;
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

// Butterfly transpose-reduction confined to a DPP row of 16 lanes (= one quadrant of the tile, raster_common.h): on entry every
// lane holds NV partial sums; on return lane l holds the ROW's total of value (l & 15) in z[0] (and of value 16 + (l & 15) in z[1]
// for NV = 32).  quad_perm for lane-xor 1 and 2, row_ror:4 / row_ror:8 inside the row: four DPP steps, no cross-row traffic --
// each quadrant reduces the gradients of ITS Gaussian.
template <int NV>
__device__ __forceinline__ void row_transpose_reduce(float (&v)[NV], int lane, float (&z)[NV / 16])
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
#pragma unroll
    for (int i = 0; i < NV / 16; i++) {
        const float keep = b3 ? c[2 * i + 1] : c[2 * i];
        const float send = b3 ? c[2 * i] : c[2 * i + 1];
        z[i] = keep + dpp_mov<0x128>(send);  // row_ror:8
    }
}

// The same reduction with the two 16-lane-rotation steps FIRST, where a step has 8 and 4 results: a rotation step selects by a bank
// bit of the lane ((lane & 4), (lane & 8); a DPP bank = 4 lanes), so "keep one value, send the other" needs no select at all --
// v_add_f32_dpp with bank_mask writes the sum of value 2i into the banks that keep it and the sum of value 2i+1 into the others:
// 2 instructions per result instead of 3 (2 v_cndmask + 1 add).  row_ror:4 has to come before row_ror:8: a rotation by 4 pairs bank
// k with bank k-1, whose (lane & 8) differs for two of the four banks, so it must read values no bank bit has selected yet; the
// rotation by 8 then pairs banks with equal (lane & 4).  The quad_perm steps select by a lane bit inside a bank and keep the select
// form.  24 + 6 + 3 = 33 instructions per 16 values instead of 45.  On return lane l holds value (l & 15) as before: the caller's
// values enter in the order the steps' bit selection undoes (value k at position (k>>2 & 1) | (k>>3 & 1) << 1 | (k>>1 & 1) << 2 |
// (k & 1) << 3).  Inline assembly: the compiler has no builtin for a DPP add with a bank mask (update_dpp + add would be 3 again);
// the s_nop in front covers "VALU write -> DPP read" (2 wait states) for the inputs, the one behind for the compiler's next DPP.
#ifndef ED3_K7_BANK_REDUCE
#define ED3_K7_BANK_REDUCE 1
#endif
__device__ __forceinline__ float row_transpose_reduce16_banks(const float (&v)[16], int lane)
{
    constexpr int pos[16] = {0, 8, 4, 12, 1, 9, 5, 13, 2, 10, 6, 14, 3, 11, 7, 15};   // pos[k]: where value k enters
    float u[16];
#pragma unroll
    for (int k = 0; k < 16; k++) u[pos[k]] = v[k];
    float a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %12, %12 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %13, %13 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %14, %14 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %15, %15 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %16, %16 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %17, %17 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %18, %18 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %19, %19 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %20, %20 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %21, %21 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %22, %22 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %23, %23 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %24, %24 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %25, %25 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %7, %26, %26 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %7, %27, %27 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %8, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %8, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %9, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %9, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %10, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %10, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %11, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %11, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7),
          "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
        : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]),
          "v"(u[8]), "v"(u[9]), "v"(u[10]), "v"(u[11]), "v"(u[12]), "v"(u[13]), "v"(u[14]), "v"(u[15]));
    const bool l1 = lane & 2, l0 = lane & 1;
    const float c0 = (l1 ? b1 : b0) + dpp_mov<0x4E>(l1 ? b0 : b1);   // quad_perm [2,3,0,1]
    const float c1 = (l1 ? b3 : b2) + dpp_mov<0x4E>(l1 ? b2 : b3);
    return (l0 ? c1 : c0) + dpp_mov<0xB1>(l0 ? c0 : c1);             // quad_perm [1,0,3,2]
}

// waves per SIMD asked of the allocator for the headline instantiation (0 = its own choice: 140 registers, 3 waves).  Round 3:
// 4 waves (128 registers, 52 B of scratch per lane) measured 0.460 against 0.438 ms (tools/ab_build.sh k7w4 -DED3_K7_WAVES=4).
#ifndef ED3_K7_WAVES
#define ED3_K7_WAVES 0
#endif
// 1: the quadrants' records meet in a chunk-local LDS tile and ONE 64-byte global atomic per touched (tile, Gaussian) entry is
// issued at the chunk boundary.  0: every quadrant adds its own record to global memory (round 3).  Measured in
// round 4: the tile cuts the records 2.55 M -> 0.94 M per C3 launch and the HBM traffic with them, and is
// SLOWER -- the atomics were never the limiter (365 GB/s of them against ~1.3 TB/s the L2 sustains), the flush's extra
// instructions and its LDS round trip are paid by a kernel that is bound by instruction issue.
#ifndef ED3_K7_LDS_TILE
#define ED3_K7_LDS_TILE 0
#endif
template <bool COORD, bool DEPTH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(((!COORD && DEPTH && ED3_K7_WAVES) ? ED3_K7_WAVES : 1), ((!COORD && DEPTH && ED3_K7_WAVES) ? ED3_K7_WAVES : 8)))) render_backward_kernel(
    int W, int H, int gx, const uint32_t *__restrict__ tile_order, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
    const float4 *__restrict__ rec, const float4 *__restrict__ rec_coord, float focal_x, float focal_y,
    const float *__restrict__ bg, const float *__restrict__ alphas, const float *__restrict__ normalmap,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ accum_coord,
    const float *__restrict__ accum_depth, const float *__restrict__ normal_length,
    const float *__restrict__ dL_dpix, const float *__restrict__ dL_dcoord, const float *__restrict__ dL_dmcoord,
    const float *__restrict__ dL_ddepth, const float *__restrict__ dL_dmdepth, const float *__restrict__ dL_dalpha,
    const float *__restrict__ dL_dnormal, float *__restrict__ grec, float *__restrict__ grec_coord,
    unsigned long long *__restrict__ counters)   // measurement only (bench.py): [0] visited (tile, Gaussian) iterations, [1] blended
                                                  // pairs, [2] staged list entries, [3] entries kept for at least one quadrant,
                                                  // [4] (entry, quadrant) pairs queued, [5] 64-byte records added to global
                                                  // memory (one per touched entry and chunk); NULL = off
{
    constexpr bool GEO = COORD || DEPTH;
    constexpr int NV = COORD ? 32 : 16;
    unsigned n_iter = 0, n_pair = 0, n_staged = 0, n_kept = 0, n_qpairs = 0, n_flushed = 0;
    __shared__ float4 s_rec[64 * 4];
    __shared__ float4 s_recc[COORD ? 64 * 3 : 1];
    __shared__ uint32_t s_id[64];
    // chunk-local gradient tile: entry j's record, summed over the tile's quadrants; zero outside a chunk's flush
    __shared__ float s_g[ED3_K7_LDS_TILE ? 64 * GREC : 1];
    __shared__ float s_gc[(ED3_K7_LDS_TILE && COORD) ? 64 * GREC : 1];
    __shared__ uint32_t s_list[ED3_K7_LDS_TILE ? 64 : 1];   // the chunk's touched entries, compacted (flush)

    const int tile = (int)tile_order[blockIdx.x];   // longest tile lists first (tile_order_kernel)
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int myq = lane >> 4, li = lane & 15;                       // quadrant-major pixel ownership (raster_common.h)
#if ED3_K7_LDS_TILE
#pragma unroll
    for (int i = 0; i < GREC / 4; i++) {
        reinterpret_cast<float4 *>(s_g)[i * 64 + lane] = make_float4(0, 0, 0, 0);
        if (COORD) reinterpret_cast<float4 *>(s_gc)[i * 64 + lane] = make_float4(0, 0, 0, 0);
    }
#endif
    const uint32_t jshift = 8u * (uint32_t)myq;
    const int px0 = tx * TILE + 8 * (myq & 1) + 4 * (li & 1);
    const int py = ty * TILE + 8 * (myq >> 1) + (li >> 1);
    const size_t HW = (size_t)H * W;
    const int nvalid = (py < H) ? max(0, min(4, W - px0)) : 0;
    const bool vec = (nvalid == 4) && ((W & 3) == 0);
    const size_t pix0 = (size_t)py * W + px0;
    const float fpy = (float)py;
    float fpx[4];
#pragma unroll
    for (int p = 0; p < 4; p++) fpx[p] = (float)(px0 + p);
    const uint2 range = ranges[tile];
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);

    // ---- per-pixel state (prologue in scalars, then packed into pixel pairs) ----
    float T[4], V[4];                         // transmittance; suffix sum of w_j * s_j (+ T_final * bg . dL_dpixel)
    float g0[4], g1[4], g2[4], gA[4];         // dL_dpixel rgb, adjusted dL_dalpha
    float gT[4], gMT[4], gN0[4], gN1[4], gN2[4];
    float gC0[4], gC1[4], gC2[4], gM0[4], gM1[4], gM2[4];
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
            last[p] = lc[p]; maxc[p] = mc[p] - 1u;   // the median contributor's 0-based list position (0xFFFFFFFF: none)
            T[p] = 1.f - al[p];
            V[p] = T[p] * (bg[0] * g0[p] + bg[1] * g1[p] + bg[2] * g2[p]);
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
    // a pixel nothing was blended into (alpha_out == 0) has 0/0 in the normalised gradients above; it takes no part in
    // any Gaussian's gradient, and since skipped pixels are multiplied by zero rather than selected away, clear them
#pragma unroll
    for (int p = 0; p < 4; p++) {
        if (last[p] == 0) {
            V[p] = g0[p] = g1[p] = g2[p] = gA[p] = gT[p] = gMT[p] = gN0[p] = gN1[p] = gN2[p] = 0.f;
            gC0[p] = gC1[p] = gC2[p] = gM0[p] = gM1[p] = gM2[p] = 0.f;
        }
    }
    // pixel pairs (2q, 2q+1): every per-pixel quantity below is an f32x2 so the arithmetic issues as v_pk_*_f32
    f32x2 T2[2], V2[2], g0v[2], g1v[2], g2v[2], gAv[2], gTv[2], gMTv[2], gN0v[2], gN1v[2], gN2v[2];
    f32x2 gC0v[2], gC1v[2], gC2v[2], gM0v[2], gM1v[2], gM2v[2], fpxv[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
#define ED3_PAIR(dst, src) dst[q] = f32x2{src[2 * q], src[2 * q + 1]}
        ED3_PAIR(T2, T); ED3_PAIR(V2, V); ED3_PAIR(g0v, g0); ED3_PAIR(g1v, g1); ED3_PAIR(g2v, g2); ED3_PAIR(gAv, gA);
        ED3_PAIR(gTv, gT); ED3_PAIR(gMTv, gMT); ED3_PAIR(gN0v, gN0); ED3_PAIR(gN1v, gN1); ED3_PAIR(gN2v, gN2);
        ED3_PAIR(gC0v, gC0); ED3_PAIR(gC1v, gC1); ED3_PAIR(gC2v, gC2); ED3_PAIR(gM0v, gM0); ED3_PAIR(gM1v, gM1);
        ED3_PAIR(gM2v, gM2); ED3_PAIR(fpxv, fpx);
#undef ED3_PAIR
    }

    // largest last-contributor of the tile: nothing behind it is blended by any pixel
    uint32_t lmax = max(max(last[0], last[1]), max(last[2], last[3]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, off));
    const int tile_max = __builtin_amdgcn_readfirstlane((int)lmax);   // (wave-uniform by construction; tell the compiler: the chunk masks below stay in scalar registers)
    const float hW = 0.5f * W, hH = 0.5f * H;
    // ... and of each quadrant: entries at or behind it are dropped from that quadrant's sub-list
    uint32_t qm = max(max(last[0], last[1]), max(last[2], last[3]));
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) qm = max(qm, (uint32_t)__shfl_xor((int)qm, off));
    const int qmax0 = __builtin_amdgcn_readlane((int)qm, 0), qmax1 = __builtin_amdgcn_readlane((int)qm, 16);
    const int qmax2 = __builtin_amdgcn_readlane((int)qm, 32), qmax3 = __builtin_amdgcn_readlane((int)qm, 48);

    for (int top = tile_max; top > 0; top -= 64) {
        __syncthreads();
        const int cnt = min(64, top);
        unsigned keepq = 0;   // quadrants of the tile in which this lane's entry can reach alpha >= 1/255
        if (lane < cnt) {
            const uint32_t id = point_list[range.x + (uint32_t)(top - 1 - lane)];
            const float4 *src = rec + (size_t)id * 4;
            const float4 q0 = src[0], q1 = src[1];
            s_id[lane] = id;
            s_rec[lane * 4 + 0] = q0;
            s_rec[lane * 4 + 1] = q1;
            keepq = quadrants_may_contribute(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tile_x0, tile_y0);
            if (keepq) {   // the rest of the record only for entries the inner loop will visit
                s_rec[lane * 4 + 2] = src[2];
                if (GEO) s_rec[lane * 4 + 3] = src[3];
            }
            if (COORD && keepq) {
                const float4 *sc = rec_coord + (size_t)id * 3;
                s_recc[lane * 3 + 0] = sc[0]; s_recc[lane * 3 + 1] = sc[1]; s_recc[lane * 3 + 2] = sc[2];
            }
        }
        __syncthreads();
        // per quadrant: the chunk's entries that can contribute inside its box, back to front (raster_common.h); every iteration
        // each quadrant takes the next entry of ITS OWN sub-list, so up to four Gaussians are differentiated per iteration
        unsigned long long live0 = __ballot(keepq & 1u), live1 = __ballot(keepq & 2u), live2 = __ballot(keepq & 4u), live3 = __ballot(keepq & 8u);
        {   // entry j sits at list position top - 1 - j: a quadrant needs only the positions below its own largest last contributor
            auto behind = [&](int qmax) { const int nb = top - qmax; return nb <= 0 ? ~0ull : nb >= 64 ? 0ull : ~((1ull << nb) - 1ull); };
            live0 &= behind(qmax0); live1 &= behind(qmax1); live2 &= behind(qmax2); live3 &= behind(qmax3);
        }
        if (counters) {
            n_staged += (unsigned)cnt; n_kept += (unsigned)__popcll(live0 | live1 | live2 | live3);
            n_qpairs += (unsigned)(__popcll(live0) + __popcll(live1) + __popcll(live2) + __popcll(live3));
        }
#if ED3_K7_LDS_TILE
        unsigned long long touched = 0ull;   // entries of the chunk some quadrant added a record to (scalar)
#endif
        while (live0 | live1 | live2 | live3) {
            // next entry of each quadrant's sub-list (-1: none left), packed into one scalar: a lane picks its byte
            const int j0 = __ffsll(live0) - 1, j1 = __ffsll(live1) - 1, j2 = __ffsll(live2) - 1, j3 = __ffsll(live3) - 1;
            live0 &= live0 - 1; live1 &= live1 - 1; live2 &= live2 - 1; live3 &= live3 - 1;
            const uint32_t jpack = (uint32_t)(j0 & 255) | (uint32_t)(j1 & 255) << 8 | (uint32_t)(j2 & 255) << 16 | (uint32_t)(j3 & 255) << 24;
            const int jsel = (int)(jpack >> jshift & 255u);
            const bool act = jsel != 255;                    // this lane's quadrant still has an entry in the chunk
            // (an idle quadrant reads the record of a quadrant that is not idle: a kept entry, i.e. finite values -- its lanes
            // multiply them by alpha = 0, and a slot nobody staged could hold a NaN)
            const int jany = j0 >= 0 ? j0 : j1 >= 0 ? j1 : j2 >= 0 ? j2 : j3;
            const int j = act ? jsel : jany;
            // 0-based position in the tile list; an idle quadrant gets a position behind every pixel's last contributor, so that the
            // per-pixel test below needs no separate "quadrant has an entry" term (round 4: the && chain compiled to four
            // s_and_saveexec / s_or exec pairs per iteration)
            const uint32_t k = act ? (uint32_t)(top - 1 - j) : 0xFFFFFFFFu;
            const float4 r0 = s_rec[j * 4 + 0];          // x, y, cx, cy
            const float4 r1 = s_rec[j * 4 + 1];          // cz, w, r, g
            const float dy = r0.y - fpy;
            const ConicRow cr = conic_row(r0.z, r0.w, r1.x, dy);
            const ConicSplat cs = conic_splat(cr);
            // skip decisions in scalars, bit-identical to the forward's; a skipped pixel continues with alpha = G = 0,
            // which makes every update below the identity and every contribution an exact zero (no selects needed)
            float dx[4], alpha[4], G[4];
            bool med[4];
            bool any_valid = false;
#pragma unroll
            for (int q = 0; q < 2; q++) {   // pixel pairs: the same packed evaluation as the forward's (raster_common.h)
                dx[2 * q] = r0.x - fpx[2 * q]; dx[2 * q + 1] = r0.x - fpx[2 * q + 1];
                const AlphaPair ap = alpha_pair(cs, r1.y, f32x2{dx[2 * q], dx[2 * q + 1]});
                const bool v0 = (k < last[2 * q]) & !(ap.power.x > 0.0f) & !(ap.alpha.x < ALPHA_MIN);
                const bool v1 = (k < last[2 * q + 1]) & !(ap.power.y > 0.0f) & !(ap.alpha.y < ALPHA_MIN);
                any_valid |= v0 | v1;
                alpha[2 * q] = v0 ? ap.alpha.x : 0.f; alpha[2 * q + 1] = v1 ? ap.alpha.y : 0.f;
                G[2 * q] = v0 ? ap.G.x : 0.f; G[2 * q + 1] = v1 ? ap.G.y : 0.f;
                med[2 * q] = v0 & (k == maxc[2 * q]); med[2 * q + 1] = v1 & (k == maxc[2 * q + 1]);   // (maxc holds the 0-based position)
            }
            const unsigned long long anyb = __ballot(any_valid);
            if (!anyb) continue;
            if (counters) {   // per lane; summed over the wave once, at the end of the tile
                n_iter++;
                n_pair += (alpha[0] > 0.f) + (alpha[1] > 0.f) + (alpha[2] > 0.f) + (alpha[3] > 0.f);
            }

            const float4 r2 = s_rec[j * 4 + 2];  // b, tongue, ts, rpx
            float4 r3 = make_float4(0, 0, 0, 0); // rpy, nx, ny, nz
            if (GEO) r3 = s_rec[j * 4 + 3];
            float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0;
            if (COORD) { q0 = s_recc[j * 3 + 0]; q1 = s_recc[j * 3 + 1]; q2 = s_recc[j * 3 + 2]; }
            const float t_row = DEPTH ? (r2.z + r3.x * dy) : 0.f;
            const float dyb = dy * r0.w, dyc = dy * r1.x;
            float c0_row = 0.f, c1_row = 0.f, c2_row = 0.f;
            if (COORD) { c0_row = q1.z + q0.y * dy; c1_row = q1.w + q0.w * dy; c2_row = q2.x + q1.y * dy; }

            const f32x2 zero2 = {0.f, 0.f};
            f32x2 sR = zero2, sG = zero2, sB = zero2, sTS = zero2, sRPX = zero2, sNX = zero2, sNY = zero2, sNZ = zero2;
            f32x2 sOP = zero2, sGDX = zero2, sGDXX = zero2;
            f32x2 sC0 = zero2, sC1 = zero2, sC2 = zero2, sC0x = zero2, sC1x = zero2, sC2x = zero2;
            float mz = 0.f;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const f32x2 dx2 = {dx[2 * q], dx[2 * q + 1]};
                const f32x2 a2 = {alpha[2 * q], alpha[2 * q + 1]};
                const f32x2 G2 = {G[2 * q], G[2 * q + 1]};
                const f32x2 om = 1.f - a2;
                const f32x2 inv = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                const f32x2 Tn = T2[q] * inv;           // T / (1 - alpha): transmittance in front of this Gaussian
                const f32x2 wgt = a2 * Tn;              // dchannel_dcolor
                // s = sum over channels of (channel value of this Gaussian at the pixel) * (upstream gradient of the channel)
                f32x2 s = g0v[q] * r1.z + gAv[q];
                s += g1v[q] * r1.w;
                s += g2v[q] * r2.x;
                sR += wgt * g0v[q];
                sG += wgt * g1v[q];
                sB += wgt * g2v[q];
                if (COORD) {
                    const f32x2 c0 = c0_row + q0.x * dx2, c1 = c1_row + q0.z * dx2, c2 = c2_row + q1.x * dx2;
                    s += c0 * gC0v[q] + c1 * gC1v[q] + c2 * gC2v[q];
                    f32x2 d0 = wgt * gC0v[q], d1 = wgt * gC1v[q], d2 = wgt * gC2v[q];
                    d0.x += med[2 * q] ? gM0v[q].x : 0.f; d0.y += med[2 * q + 1] ? gM0v[q].y : 0.f;
                    d1.x += med[2 * q] ? gM1v[q].x : 0.f; d1.y += med[2 * q + 1] ? gM1v[q].y : 0.f;
                    d2.x += med[2 * q] ? gM2v[q].x : 0.f; d2.y += med[2 * q + 1] ? gM2v[q].y : 0.f;
                    sC0 += d0; sC1 += d1; sC2 += d2;
                    sC0x += d0 * dx2; sC1x += d1 * dx2; sC2x += d2 * dx2;
                }
                if (DEPTH) {
                    const f32x2 tt = t_row + r2.w * dx2;
                    s += tt * gTv[q];
                    f32x2 dLdt = wgt * gTv[q];
                    dLdt.x += med[2 * q] ? gMTv[q].x : 0.f;
                    dLdt.y += med[2 * q + 1] ? gMTv[q].y : 0.f;
                    sTS += dLdt;
                    sRPX += dLdt * dx2;
                }
                if (GEO) {
                    s += gN0v[q] * r3.y + gN1v[q] * r3.z + gN2v[q] * r3.w;
                    sNX += wgt * gN0v[q];
                    sNY += wgt * gN1v[q];
                    sNZ += wgt * gN2v[q];
                }
                // dL_dalpha of CR/backward.cu:866-999 in suffix-sum form: (c - accum_rec) * T_k == c * T_k - U / (1 - alpha)
                // with U = sum_{j behind} w_j c_j; contracted over channels, one scalar V per pixel carries all of them
                // (and the background term, as the last "layer" of weight T_final)
                const f32x2 dopa = Tn * s - inv * V2[q];
                V2[q] += wgt * s;
                T2[q] = Tn;
                const f32x2 gd = G2 * dopa;             // -> dL_dopacity
                const f32x2 gdx = gd * dx2;
                sOP += gd;
                sGDX += gdx;
                sGDXX += gdx * dx2;
                // |dL_dmean2D| terms need the per-pixel magnitudes
                const f32x2 e = gd * r1.y;              // G * dL_dG
                const f32x2 u1 = e * (dx2 * r0.z + dyb);
                const f32x2 u2 = e * (dx2 * r0.w + dyc);
                mz += fabsf(u1.x) * hW + fabsf(u2.x) * hH;
                mz += fabsf(u1.y) * hW + fabsf(u2.y) * hH;
            }
            float acc[NV];
#pragma unroll
            for (int i = 0; i < NV; i++) acc[i] = 0.f;
            const float OPl = sOP.x + sOP.y;
            const float El = r1.y * OPl;                              // sum of e
            const float Edx = r1.y * (sGDX.x + sGDX.y);               // sum of e * dx
            const float Edxx = r1.y * (sGDXX.x + sGDXX.y);            // sum of e * dx * dx
            acc[G_R] = sR.x + sR.y; acc[G_G] = sG.x + sG.y; acc[G_B] = sB.x + sB.y;
            acc[G_OP] = OPl;
            acc[G_CX] = Edxx;
            acc[G_CY] = Edx * dy;
            acc[G_CW] = El * dy * dy;
            float mx = -(r0.z * Edx + dyb * El);                      // sum of dL_dG * dG_ddelx
            float my = -(dyc * El + r0.w * Edx);
            if (DEPTH) {
                const float TSl = sTS.x + sTS.y;
                acc[G_TS] = TSl;
                acc[G_RPX] = sRPX.x + sRPX.y;
                acc[G_RPY] = TSl * dy;
                mx += TSl * r2.w;
                my += TSl * r3.x;
            }
            if (GEO) { acc[G_NX] = sNX.x + sNX.y; acc[G_NY] = sNY.x + sNY.y; acc[G_NZ] = sNZ.x + sNZ.y; }
            if (COORD) {
                const float C0 = sC0.x + sC0.y, C1 = sC1.x + sC1.y, C2 = sC2.x + sC2.y;
                acc[16] = C0; acc[17] = C1; acc[18] = C2;
                acc[19] = sC0x.x + sC0x.y; acc[20] = C0 * dy;
                acc[21] = sC1x.x + sC1x.y; acc[22] = C1 * dy;
                acc[23] = sC2x.x + sC2x.y; acc[24] = C2 * dy;
                mx += C0 * q0.x + C1 * q0.z + C2 * q1.x;
                my += C0 * q0.y + C1 * q0.w + C2 * q1.y;
            }
            acc[G_MX] = mx;
            acc[G_MY] = my;
            acc[G_MZ] = mz;
            // each quadrant (DPP row) reduces the record of its own Gaussian; a quadrant in which no pixel blended adds nothing.
            // ED3_K7_LDS_TILE: the row goes to the entry's row of the LDS tile (ds_add_f32, no return value; quadrants that meet
            // on one entry in the same iteration are serialised by the LDS)
            float z[NV / 16];
#if ED3_K7_BANK_REDUCE
            {
                float lo[16];
#pragma unroll
                for (int i = 0; i < 16; i++) lo[i] = acc[i];
                z[0] = row_transpose_reduce16_banks(lo, lane);
                if (NV == 32) {
#pragma unroll
                    for (int i = 0; i < 16; i++) lo[i] = acc[(NV - 16) + i];
                    z[NV / 16 - 1] = row_transpose_reduce16_banks(lo, lane);
                }
            }
#else
            row_transpose_reduce<NV>(acc, lane, z);
#endif
            const bool qany = (anyb >> (16 * myq) & 0xFFFFull) != 0ull;
#if ED3_K7_LDS_TILE
            if (act && qany) {
                atomicAdd(&s_g[j * GREC + li], z[0]);
                if (COORD && li < 9) atomicAdd(&s_gc[j * GREC + li], z[NV / 16 - 1]);
            }
            touched |= ((j0 >= 0 && (anyb & 0xFFFFull)) ? 1ull << j0 : 0ull) | ((j1 >= 0 && (anyb >> 16 & 0xFFFFull)) ? 1ull << j1 : 0ull) |
                       ((j2 >= 0 && (anyb >> 32 & 0xFFFFull)) ? 1ull << j2 : 0ull) | ((j3 >= 0 && (anyb >> 48)) ? 1ull << j3 : 0ull);
#else
#ifdef ED3_K7_ABLATE_ATOMICS   // timing experiment (results WRONG): the records are reduced and dropped
            if (act && qany && z[0] == 123456.789f) {
#else
            if (act && qany) {   // round 3: one 64-byte atomic per (tile, Gaussian, quadrant)
#endif
                const uint32_t id = s_id[j];
                atomicAdd(grec + (size_t)id * GREC + li, z[0]);
                if (COORD && li < 9) atomicAdd(grec_coord + (size_t)id * GREC + li, z[NV / 16 - 1]);
            }
            if (counters) n_flushed += (unsigned)(((j0 >= 0 && (anyb & 0xFFFFull)) ? 1 : 0) + ((j1 >= 0 && (anyb >> 16 & 0xFFFFull)) ? 1 : 0) +
                                                  ((j2 >= 0 && (anyb >> 32 & 0xFFFFull)) ? 1 : 0) + ((j3 >= 0 && (anyb >> 48)) ? 1 : 0));
#endif
        }
#if ED3_K7_LDS_TILE
        // chunk boundary: ONE 64-byte global atomic per touched entry.  The touched entries are compacted into s_list (lane i
        // writes i at its rank among the set bits); a DPP row then takes an entry, sixteen entries per batch: the rows read their
        // entries' indices, then the 16 sums and the Gaussian ids, clear the sums for the next chunk and add them to the records
        if (touched) {
            const int n = __popcll(touched);
            if (counters) n_flushed += (unsigned)n;
            if (touched >> lane & 1ull) s_list[__popcll(touched & ((1ull << lane) - 1ull))] = (uint32_t)lane;
            __syncthreads();
            for (int t0 = 0; t0 < n; t0 += 16) {
                int e[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int ix = t0 + 4 * u + myq; ok[u] = ix < n; e[u] = (int)s_list[ok[u] ? ix : 0]; }
                float v[4], vc[4];
                uint32_t gid[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    v[u] = s_g[e[u] * GREC + li];
                    gid[u] = s_id[e[u]];
                    vc[u] = COORD ? s_gc[e[u] * GREC + li] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (ok[u]) {
                        s_g[e[u] * GREC + li] = 0.f;
                        atomicAdd(grec + (size_t)gid[u] * GREC + li, v[u]);
                        if (COORD && li < 9) { s_gc[e[u] * GREC + li] = 0.f; atomicAdd(grec_coord + (size_t)gid[u] * GREC + li, vc[u]); }
                    }
                }
            }
        }
#endif
    }
    if (counters) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n_pair += (unsigned)__shfl_xor((int)n_pair, off);
    }
    if (counters && lane == 0) {
        atomicAdd(counters + 0, (unsigned long long)n_iter); atomicAdd(counters + 1, (unsigned long long)n_pair);
        atomicAdd(counters + 2, (unsigned long long)n_staged); atomicAdd(counters + 3, (unsigned long long)n_kept);
        atomicAdd(counters + 4, (unsigned long long)n_qpairs); atomicAdd(counters + 5, (unsigned long long)n_flushed);
    }
}

void launch_render_backward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                            const float *rec_coord, float focal_x, float focal_y, const float *bg, bool coord,
                            bool depth, const float *alphas, const float *normalmap, ImageState img,
                            const float *dL_dpix, const float *dL_dcoord, const float *dL_dmcoord,
                            const float *dL_ddepth, const float *dL_dmdepth, const float *dL_dalpha,
                            const float *dL_dnormal, float *grec, float *grec_coord, hipStream_t s, unsigned long long *counters)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    dim3 grid(gx * gy), block(64);
#define ED3_BWD(C_, D_)                                                                                              \
    hipLaunchKernelGGL((render_backward_kernel<C_, D_>), grid, block, 0, s, W, H, gx, img.tile_order,                             \
                       reinterpret_cast<const uint2 *>(ranges), point_list, reinterpret_cast<const float4 *>(rec),  \
                       reinterpret_cast<const float4 *>(rec_coord), focal_x, focal_y, bg, alphas, normalmap,         \
                       img.n_contrib, img.accum_coord, img.accum_depth, img.normal_length, dL_dpix, dL_dcoord,       \
                       dL_dmcoord, dL_ddepth, dL_dmdepth, dL_dalpha, dL_dnormal, grec, grec_coord, counters)
    if (coord && depth) ED3_BWD(true, true);
    else if (coord) ED3_BWD(true, false);
    else if (depth) ED3_BWD(false, true);
    else ED3_BWD(false, false);
#undef ED3_BWD
}

}  // namespace ed3
