// render_forward.hip -- K6: front-to-back alpha compositing of one 16x16 tile per wavefront.
// Computes what CR/forward.cu:550-822 (renderCUDA<3,COORD,DEPTH,NORMAL>) computes -- colour, tongue, alpha,
// expected/median coord + depth, normal, n_contrib (last, median) and the accumulators the backward needs --
// with the blend rule of Q5-Q7.  Layout and scheduling are CDNA-native (see raster_common.h).
#include "raster_common.h"

namespace ed3 {

// Round 3: with the exact alpha evaluation and the quadrant scheduling the 96-register bound of round 2 (5 waves per SIMD)
// costs 80 bytes of scratch, some of it inside the blend loop: 4 waves (128 registers, no scratch) measured 0.284 vs 0.298 ms,
// 6 waves 0.574 ms (tools/ab_build.sh k6w4 -DED3_K6_WAVES=4).
#ifndef ED3_K6_WAVES
#define ED3_K6_WAVES 4
#endif
template <bool COORD, bool DEPTH>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(((!COORD && DEPTH) ? ED3_K6_WAVES : 1), ((!COORD && DEPTH) ? ED3_K6_WAVES : 8)))) render_forward_kernel(
    int W, int H, int gx, const uint32_t *__restrict__ tile_order, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
    const float4 *__restrict__ rec, const float4 *__restrict__ rec_coord, float focal_x, float focal_y,
    const float *__restrict__ bg, float *__restrict__ out_color, float *__restrict__ out_coord,
    float *__restrict__ out_mcoord, float *__restrict__ out_depth, float *__restrict__ out_mdepth,
    float *__restrict__ out_alpha, float *__restrict__ out_tongue, float *__restrict__ out_normal,
    uint32_t *__restrict__ n_contrib, float *__restrict__ accum_coord, float *__restrict__ accum_depth,
    float *__restrict__ normal_length,
    unsigned long long *__restrict__ counters)   // measurement only (bench.py): [12] visited (tile, Gaussian) iterations, [13] blended
                                                  // pairs, [14] staged list entries, [15] entries kept by the tile-level reject; NULL = off
{
    constexpr bool GEO = COORD || DEPTH;
    unsigned n_iter = 0, n_pair = 0, n_staged = 0, n_kept = 0;
    __shared__ float4 s_rec[64 * 4];
    __shared__ float4 s_recc[COORD ? 64 * 3 : 1];

    const int tile = (int)tile_order[blockIdx.x];   // longest tile lists first (tile_order_kernel)
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    const int myq = lane >> 4, li = lane & 15;                       // quadrant-major pixel ownership (raster_common.h)
    const uint32_t jshift = 8u * (uint32_t)myq;
    const int px0 = tx * TILE + 8 * (myq & 1) + 4 * (li & 1);
    const int py = ty * TILE + 8 * (myq >> 1) + (li >> 1);
    const size_t HW = (size_t)H * W;
    const int nvalid = (py < H) ? max(0, min(4, W - px0)) : 0;  // pixels of this lane inside the image
    const bool vec = (nvalid == 4) && ((W & 3) == 0);
    const float fpy = (float)py;
    float fpx[4];
#pragma unroll
    for (int p = 0; p < 4; p++) fpx[p] = (float)(px0 + p);

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);
    const float tile_x0 = (float)(tx * TILE), tile_y0 = (float)(ty * TILE);

    float T[4], C0[4], C1[4], C2[4], TG[4], WT[4];
    float DP[4], MD[4], N0[4], N1[4], N2[4];
    float CO0[4], CO1[4], CO2[4], MC0[4], MC1[4], MC2[4];
    uint32_t last[4], maxc[4];
    // "done" lives in the pixel's alpha threshold: +inf once the pixel has terminated (or lies outside the image), so that the
    // per-pixel validity test below is the reference's own two comparisons and nothing else (alpha is never NaN: v_min_f32 returns
    // its other operand, 0.99)
    float amin[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        T[p] = 1.0f; C0[p] = C1[p] = C2[p] = TG[p] = WT[p] = 0.f;
        DP[p] = MD[p] = N0[p] = N1[p] = N2[p] = 0.f;
        CO0[p] = CO1[p] = CO2[p] = MC0[p] = MC1[p] = MC2[p] = 0.f;
        last[p] = 0; maxc[p] = 0xFFFFFFFFu;
        amin[p] = (p < nvalid) ? ALPHA_MIN : __builtin_inff();
    }
#define ED3_ALL_DONE() (amin[0] == __builtin_inff() && amin[1] == __builtin_inff() && amin[2] == __builtin_inff() && amin[3] == __builtin_inff())

    bool finished = false;
    for (int base = 0; base < n && !finished; base += 64) {
        if (__all(ED3_ALL_DONE())) break;
        __syncthreads();
        const int k = base + lane;
        unsigned keepq = 0;   // quadrants of the tile in which this lane's entry can reach alpha >= 1/255
        if (k < n) {
            const uint32_t id = point_list[range.x + k];
            const float4 *src = rec + (size_t)id * 4;
            const float4 q0 = src[0], q1 = src[1];
            s_rec[lane * 4 + 0] = q0;
            s_rec[lane * 4 + 1] = q1;
            keepq = quadrants_may_contribute(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tile_x0, tile_y0);
            if (keepq) {   // the rest of the record only for entries the inner loop will visit
                if (GEO) { s_rec[lane * 4 + 2] = src[2]; s_rec[lane * 4 + 3] = src[3]; }
                else     { s_rec[lane * 4 + 2] = src[2]; }
            }
            if (COORD && keepq) {
                const float4 *sc = rec_coord + (size_t)id * 3;
                s_recc[lane * 3 + 0] = sc[0]; s_recc[lane * 3 + 1] = sc[1]; s_recc[lane * 3 + 2] = sc[2];
            }
        }
        __syncthreads();
        // per quadrant: the chunk's entries that can contribute inside its box, in list order (raster_common.h); every
        // iteration each quadrant takes the next entry of ITS OWN sub-list
        unsigned long long live0 = __ballot(keepq & 1u), live1 = __ballot(keepq & 2u), live2 = __ballot(keepq & 4u), live3 = __ballot(keepq & 8u);
        if (counters) { n_staged += (unsigned)min(64, n - base); n_kept += (unsigned)__popcll(live0 | live1 | live2 | live3); }
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
            const uint32_t contributor = (uint32_t)(base + j + 1);
            const float4 r0 = s_rec[j * 4 + 0];  // x, y, cx, cy
            const float4 r1 = s_rec[j * 4 + 1];  // cz, w, r, g
            const float dy = r0.y - fpy;
            const ConicRow cr = conic_row(r0.z, r0.w, r1.x, dy);
            const ConicSplat cs = conic_splat(cr);
            float dx[4], alpha[4];
            const float op_eff = act ? r1.y : 0.f;
            bool valid[4];
            bool any_valid = false;
#pragma unroll
            for (int q = 0; q < 2; q++) {   // pixel pairs: the alpha evaluation issues as packed fp32 (raster_common.h)
                dx[2 * q] = r0.x - fpx[2 * q]; dx[2 * q + 1] = r0.x - fpx[2 * q + 1];
                // (an idle quadrant evaluates the finite record it re-reads with opacity 0: alpha = 0 fails the threshold test, so the
                // per-pixel test carries no "quadrant has an entry" term)
                const AlphaPair ap = alpha_pair(cs, op_eff, f32x2{dx[2 * q], dx[2 * q + 1]});
                alpha[2 * q] = ap.alpha.x; alpha[2 * q + 1] = ap.alpha.y;
                valid[2 * q] = !(ap.power.x > 0.0f) & !(ap.alpha.x < amin[2 * q]);
                valid[2 * q + 1] = !(ap.power.y > 0.0f) & !(ap.alpha.y < amin[2 * q + 1]);
                any_valid |= valid[2 * q] | valid[2 * q + 1];
            }
            if (!__any(any_valid)) continue;
            if (counters) { n_iter++; n_pair += (unsigned)valid[0] + valid[1] + valid[2] + valid[3]; }

            const float4 r2 = s_rec[j * 4 + 2];  // b, tongue, ts, rpx
            float4 r3 = make_float4(0, 0, 0, 0); // rpy, nx, ny, nz
            if (GEO) r3 = s_rec[j * 4 + 3];
            float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0;
            if (COORD) { q0 = s_recc[j * 3 + 0]; q1 = s_recc[j * 3 + 1]; q2 = s_recc[j * 3 + 2]; }
            const float t_row = DEPTH ? (r2.z + r3.x * dy) : 0.f;  // ts + ray_plane.y*dy
            bool any_term = false;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const float test_T = T[p] * (1.0f - alpha[p]);
                const bool term = valid[p] && (test_T < 0.0001f);
                const bool blend = valid[p] && !term;
                amin[p] = term ? __builtin_inff() : amin[p];
                any_term |= term;
                const float aT = blend ? alpha[p] * T[p] : 0.0f;
                C0[p] += r1.z * aT; C1[p] += r1.w * aT; C2[p] += r2.x * aT;
                TG[p] += r2.y * aT;
                const bool before_median = blend && (T[p] > 0.5f);
                if (COORD) {
                    const float c0 = q1.z + q0.x * dx[p] + q0.y * dy;
                    const float c1 = q1.w + q0.z * dx[p] + q0.w * dy;
                    const float c2 = q2.x + q1.x * dx[p] + q1.y * dy;
                    CO0[p] += c0 * aT; CO1[p] += c1 * aT; CO2[p] += c2 * aT;
                    MC0[p] = before_median ? c0 : MC0[p];
                    MC1[p] = before_median ? c1 : MC1[p];
                    MC2[p] = before_median ? c2 : MC2[p];
                }
                if (DEPTH) {
                    const float t = t_row + r2.w * dx[p];
                    DP[p] += t * aT;
                    MD[p] = before_median ? t : MD[p];
                }
                if (GEO) {
                    N0[p] += r3.y * aT; N1[p] += r3.z * aT; N2[p] += r3.w * aT;
                    maxc[p] = before_median ? contributor : maxc[p];
                }
                WT[p] += aT;
                T[p] = blend ? test_T : T[p];
                last[p] = blend ? contributor : last[p];
            }
            if (__any(any_term)) {   // a quadrant whose 64 pixels are all done drops the rest of its sub-list
                const unsigned long long dn = __ballot(ED3_ALL_DONE());
                if (dn == ~0ull) { finished = true; break; }
                if ((dn & 0xFFFFull) == 0xFFFFull) live0 = 0ull;
                if ((dn >> 16 & 0xFFFFull) == 0xFFFFull) live1 = 0ull;
                if ((dn >> 32 & 0xFFFFull) == 0xFFFFull) live2 = 0ull;
                if ((dn >> 48) == 0xFFFFull) live3 = 0ull;
            }
        }
    }

    if (counters) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n_pair += (unsigned)__shfl_xor((int)n_pair, off);
        if (lane == 0) {
            atomicAdd(counters + 12, (unsigned long long)n_iter); atomicAdd(counters + 13, (unsigned long long)n_pair);
            atomicAdd(counters + 14, (unsigned long long)n_staged); atomicAdd(counters + 15, (unsigned long long)n_kept);
        }
    }
    if (nvalid == 0) return;
    const size_t pix0 = (size_t)py * W + px0;
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    float v[4];
    store4u(n_contrib, pix0, last, vec, nvalid);
    store4u(n_contrib + HW, pix0, maxc, vec, nvalid);
#pragma unroll
    for (int p = 0; p < 4; p++) v[p] = C0[p] + T[p] * b0;
    store4(out_color, pix0, v, vec, nvalid);
#pragma unroll
    for (int p = 0; p < 4; p++) v[p] = C1[p] + T[p] * b1;
    store4(out_color + HW, pix0, v, vec, nvalid);
#pragma unroll
    for (int p = 0; p < 4; p++) v[p] = C2[p] + T[p] * b2;
    store4(out_color + 2 * HW, pix0, v, vec, nvalid);
    store4(out_tongue, pix0, TG, vec, nvalid);
    store4(out_alpha, pix0, WT, vec, nvalid);

    if (COORD) {
        const float *CO[3] = {CO0, CO1, CO2};
        const float *MC[3] = {MC0, MC1, MC2};
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = last[p] ? CO[ch][p] / WT[p] : 0.f;
            store4(out_coord + ch * HW, pix0, v, vec, nvalid);
            store4(accum_coord + ch * HW, pix0, CO[ch], vec, nvalid);
            store4(out_mcoord + ch * HW, pix0, MC[ch], vec, nvalid);
        }
    }
    if (DEPTH) {
        float ln[4], dl[4];
        const float pny = (fpy - H / 2.f) / focal_y;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const float pnx = (fpx[p] - W / 2.f) / focal_x;
            ln[p] = sqrtf(pnx * pnx + pny * pny + 1);
            dl[p] = DP[p] / ln[p];
        }
        store4(accum_depth, pix0, dl, vec, nvalid);
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = last[p] ? dl[p] / WT[p] : 0.f;
        store4(out_depth, pix0, v, vec, nvalid);
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = MD[p] / ln[p];
        store4(out_mdepth, pix0, v, vec, nvalid);
    }
    if (GEO) {
        float len[4], o0[4], o1[4], o2[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (last[p]) {
                float l = sqrtf(N0[p] * N0[p] + N1[p] * N1[p] + N2[p] * N2[p]);
                len[p] = l;
                l = fmaxf(l, NORMALIZE_EPS);
                o0[p] = N0[p] / l; o1[p] = N1[p] / l; o2[p] = N2[p] / l;
            } else {
                len[p] = 1.f; o0[p] = o1[p] = o2[p] = 0.f;
            }
        }
        store4(normal_length, pix0, len, vec, nvalid);
        store4(out_normal, pix0, o0, vec, nvalid);
        store4(out_normal + HW, pix0, o1, vec, nvalid);
        store4(out_normal + 2 * HW, pix0, o2, vec, nvalid);
    }
}

void launch_render_forward(int W, int H, const uint32_t *ranges, const uint32_t *point_list, const float *rec,
                           const float *rec_coord, float focal_x, float focal_y, const float *bg, bool coord,
                           bool depth, float *out_color, float *out_coord, float *out_mcoord, float *out_depth,
                           float *out_mdepth, float *out_alpha, float *out_tongue, float *out_normal, ImageState img,
                           hipStream_t s, unsigned long long *counters)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    dim3 grid(gx * gy), block(64);
#define ED3_FWD(C_, D_)                                                                                              \
    hipLaunchKernelGGL((render_forward_kernel<C_, D_>), grid, block, 0, s, W, H, gx, img.tile_order,                              \
                       reinterpret_cast<const uint2 *>(ranges), point_list, reinterpret_cast<const float4 *>(rec),  \
                       reinterpret_cast<const float4 *>(rec_coord), focal_x, focal_y, bg, out_color, out_coord,      \
                       out_mcoord, out_depth, out_mdepth, out_alpha, out_tongue, out_normal, img.n_contrib,          \
                       img.accum_coord, img.accum_depth, img.normal_length, counters)
    // variant dispatch as CR/forward.cu:863-870 (NORMAL on iff COORD or DEPTH)
    if (coord && depth) ED3_FWD(true, true);
    else if (coord) ED3_FWD(true, false);
    else if (depth) ED3_FWD(false, true);
    else ED3_FWD(false, false);
#undef ED3_FWD
}

}  // namespace ed3
