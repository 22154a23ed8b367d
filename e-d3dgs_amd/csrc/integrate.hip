// integrate.hip -- the point-integration path behind _C.integrate_gaussians_to_points (mesh extraction probes the
// Gaussians' opacity field at tetrahedra vertices): K12 preprocessPointsCUDA (CR/forward.cu:1027-1071), K13 createWithKeys
// (CR/rasterizer_impl.cu:114-145) and K14 integrateCUDA (CR/forward.cu:1109-1543).
//
// K14 in the reference is ONE kernel in which every thread (pixel) keeps a 4-KB list of the positions of the Gaussians
// that touched it (uint16[2048]) and 3 KB of projected-point arrays, and re-walks the tile list once per batch of 256
// points of the pixel.  Both arrays live in scratch memory on any GPU.  Restated here as two kernels with no per-thread
// arrays:
//   pixels kernel -- the reference's first loop (5-sample transmittance per pixel, colour/depth accumulators).  Which
//       list entries "touched" a pixel is a per-wave BALLOT: one 64-bit word per (list entry, wave of 64 pixels),
//       32 bytes per list entry, written by one lane.  The 2048-entry cap of the reference's list is kept (a pixel
//       stops at its 2048th touching entry).
//   points kernel -- the reference's second loop, turned inside out: lane = query POINT (each valid point lies in exactly
//       one pixel: the one containing its projection), the tile's list entries are staged through LDS with their ballot
//       words, and a point accumulates alpha over the entries whose bit is set for its pixel, up to the pixel's last
//       contributor.  A point's result does not depend on the other points of its pixel, so the reference's batching
//       by MAX_NUM_PROJECTED (and the order of points within a tile) has no effect on any output; points are therefore
//       sorted by tile only, and invalid points get a sentinel tile (no scan, no host read-back for the point count).
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "raster_common.h"

namespace ed3 {

constexpr int MAX_CONTRIB_SLOTS = 512 * 4;   // MAX_NUM_CONTRIBUTORS * 4 (CR/auxiliary.h:31, CR/forward.cu:1189)

// ---- K12: project the query points; a point outside the image or behind the near plane gets the sentinel tile ----
__global__ void __launch_bounds__(256) integrate_points_preprocess_kernel(
    int PN, const float *__restrict__ pts, const float *__restrict__ view, int W, int H, float focal_x, float focal_y,
    int gx, int gy, float2 *__restrict__ points2D, float *__restrict__ depths, uint32_t *__restrict__ tile_keys,
    uint32_t *__restrict__ ids)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PN) return;
    const uint32_t none = (uint32_t)(gx * gy);
    uint32_t key = none;
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    const float vx = view[0] * x + view[4] * y + view[8] * z + view[12];
    const float vy = view[1] * x + view[5] * y + view[9] * z + view[13];
    const float vz = view[2] * x + view[6] * y + view[10] * z + view[14];
    float2 pi = make_float2(0.f, 0.f);
    float d = 0.f;
    if (vz > 0.2f) {   // in_frustum: near cull only (CR/auxiliary.h:165-178)
        // `+ W/2.` is a double addition in the reference (CR/forward.cu:1061)
        pi.x = (float)((double)(focal_x * vx / (vz + 0.0000001f)) + W / 2.0);
        pi.y = (float)((double)(focal_y * vy / (vz + 0.0000001f)) + H / 2.0);
        if (!(pi.x < 0 || pi.x >= W || pi.y < 0 || pi.y >= H)) {
            d = sqrtf(vx * vx + vy * vy + vz * vz);
            const int tx = min(gx - 1, max(0, (int)(pi.x / TILE))), ty = min(gy - 1, max(0, (int)(pi.y / TILE)));   // K13
            key = (uint32_t)(ty * gx + tx);
        }
    }
    points2D[i] = pi;
    depths[i] = d;
    tile_keys[i] = key;
    ids[i] = (uint32_t)i;
}

// ---- K14, first loop: per pixel ----
__global__ void __launch_bounds__(256) integrate_pixels_kernel(
    int W, int H, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ point_list, const float *__restrict__ rec,
    const float *__restrict__ bg, float *__restrict__ out_color, float *__restrict__ final_T, uint32_t *__restrict__ last_contrib,
    float *__restrict__ pixaux, unsigned long long *__restrict__ used)
{
    __shared__ float4 srec[256][4];
    __shared__ uint8_t skeep[256];
    const int gx = (W + TILE - 1) / TILE;
    const int tile = blockIdx.y * gx + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int px = blockIdx.x * TILE + (tid & 15), py = blockIdx.y * TILE + (tid >> 4);
    const bool inside = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    const float pfx = (float)px + 0.5f, pfy = (float)py + 0.5f;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    bool done = !inside;
    float T = 1.0f;
    float cT[5] = {1.f, 1.f, 1.f, 1.f, 1.f};
    const float offx[5] = {0.0f, -0.5f, 0.5f, -0.5f, 0.5f}, offy[5] = {0.0f, -0.5f, -0.5f, 0.5f, 0.5f};
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Cdepth = 0.f, Cmed = 0.f, Cmax = 0.f, Calpha = 0.f;
    float mid_dc = 0.f, mid_px = 0.f, mid_py = 0.f, mid_mx = 0.f, mid_my = 0.f;
    uint32_t contributor = 0, last = 0, nlocal = 0;
    for (uint32_t base = r0; base < r1; base += 256) {
        __syncthreads();
        if (base + tid < r1) {
            const float4 *g = reinterpret_cast<const float4 *>(rec + (size_t)point_list[base + tid] * REC);
            const float4 q0 = g[0], q1 = g[1];
            srec[tid][0] = q0; srec[tid][1] = q1; srec[tid][2] = g[2]; srec[tid][3] = g[3];
            // tile-level reject (raster_common.h): can this entry reach alpha >= 1/255 at ANY of the tile's sample
            // positions (pixel centres and corners span [x0, x0 + 16] x [y0, y0 + 16])?  Conservative, so no output
            // changes; the 3-sigma-square binning puts ~40 % such entries into the lists.
            const float x0 = (float)(blockIdx.x * TILE), y0 = (float)(blockIdx.y * TILE);
            skeep[tid] = tile_may_contribute(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, x0, y0, x0 + TILE, y0 + TILE) ? 1 : 0;
        }
        __syncthreads();
        const int n = min(256u, r1 - base);
        for (int j = 0; j < n; j++) {
            bool touched = false;
            if (!done) contributor++;
            if (!done && skeep[j]) {
                const float4 a = srec[j][0], b = srec[j][1], c = srec[j][2], e = srec[j][3];
                const float gxy_x = a.x, gxy_y = a.y, conx = a.z, cony = a.w, conz = b.x, w = b.y;
                const float depth_center = c.z, rpx = c.w, rpy = e.x;
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    const float dx = gxy_x - pfx - offx[k], dy = gxy_y - pfy - offy[k];
                    const float depth = depth_center + (rpx * dx + rpy * dy);
                    const float power = -0.5f * (conx * dx * dx + conz * dy * dy) - cony * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = fminf(0.99f, w * __expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = cT[k] * (1 - alpha);
                    if (test_T < 0.0001f) continue;
                    if (k == 0) { C0 += b.z * alpha * T; C1 += b.w * alpha * T; C2 += c.x * alpha * T; }
                    if (depth > Cmax) Cmax = depth;
                    if (k == 0) {
                        Calpha += alpha * T;
                        Cdepth += depth * alpha * T;
                        if (T > 0.5f) { Cmed = depth; mid_dc = depth_center; mid_px = rpx; mid_py = rpy; mid_mx = gxy_x; mid_my = gxy_y; }
                        T = test_T;
                    }
                    cT[k] = test_T;
                    touched = true;
                }
                if (touched) {
                    last = contributor;
                    nlocal++;
                    if (nlocal >= (uint32_t)MAX_CONTRIB_SLOTS) done = true;   // CR/forward.cu:1290-1294
                }
            }
            const unsigned long long m = __ballot(touched);
            if (lane == 0) used[(size_t)(base + j) * 4 + wave] = m;
        }
    }
    if (inside) {
        const size_t HW = (size_t)H * W;
        final_T[pix] = T;
        last_contrib[pix] = last;
        out_color[0 * HW + pix] = C0 + T * bg[0];
        out_color[1 * HW + pix] = C1 + T * bg[1];
        out_color[2 * HW + pix] = C2 + T * bg[2];
        out_color[3 * HW + pix] = Cdepth;
        out_color[4 * HW + pix] = Cmed;
        out_color[6 * HW + pix] = Cmax;     // DEPTH_OFFSET
        out_color[7 * HW + pix] = Calpha;   // ALPHA_OFFSET
        float *ax = pixaux + pix * 8;
        ax[0] = mid_dc; ax[1] = mid_px; ax[2] = mid_py; ax[3] = mid_mx; ax[4] = mid_my;
    }
}

// ---- K14, second loop: per query point ----
__global__ void __launch_bounds__(256) integrate_points_kernel(
    int W, int H, const uint32_t *__restrict__ ranges, const uint32_t *__restrict__ point_ranges,
    const uint32_t *__restrict__ point_list, const uint32_t *__restrict__ qpoint_list, const float *__restrict__ rec,
    const float *__restrict__ invraycov, const uint8_t *__restrict__ condition, const float2 *__restrict__ points2D,
    const float *__restrict__ pdepths, const uint32_t *__restrict__ last_contrib, const float *__restrict__ pixaux,
    const unsigned long long *__restrict__ used, float *__restrict__ out_color, float *__restrict__ out_alpha_integrated,
    float *__restrict__ out_color_integrated, float *__restrict__ out_coordinate2d, float *__restrict__ out_sdf)
{
    constexpr int CH = 128;   // list entries per staging round
    __shared__ float sg[CH][8];       // x, y, opacity, ray distance of the centre, ray_plane.xy, condition, pad
    __shared__ float sinv[CH][6];
    __shared__ unsigned long long sused[CH][4];
    __shared__ uint32_t smax;
    const int gx = (W + TILE - 1) / TILE;
    const int tile = blockIdx.y * gx + blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t q0 = point_ranges[2 * tile], q1 = point_ranges[2 * tile + 1];
    if (q0 >= q1) return;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const size_t HW = (size_t)H * W;
    for (uint32_t qb = q0; qb < q1; qb += 256) {
        const bool valid = qb + tid < q1;
        uint32_t pid = 0, last = 0;
        float2 xy = make_float2(0.f, 0.f);
        float pdepth = 0.f;
        int lp = 0;
        size_t pix = 0;
        if (valid) {
            pid = qpoint_list[qb + tid];
            xy = points2D[pid];
            pdepth = pdepths[pid];
            // the pixel whose square [px, px + 1) x [py, py + 1) holds the projection (CR/forward.cu:1383-1384)
            const int px = min(W - 1, max(0, (int)floorf(xy.x))), py = min(H - 1, max(0, (int)floorf(xy.y)));
            pix = (size_t)py * W + px;
            lp = (py - blockIdx.y * TILE) * TILE + (px - blockIdx.x * TILE);
            lp = min(255, max(0, lp));
            last = last_contrib[pix];
        }
        __syncthreads();
        if (tid == 0) smax = 0;
        __syncthreads();
        if (last) atomicMax(&smax, last);
        __syncthreads();
        const uint32_t rend = min(r1, r0 + smax);   // nobody needs entries past the largest last contributor
        float acc = 0.f, Tk = 1.f;
        for (uint32_t base = r0; base < rend; base += CH) {
            __syncthreads();
            if (tid < CH && base + tid < rend) {
                const uint32_t g = point_list[base + tid];
                const float *r = rec + (size_t)g * REC;
                sg[tid][0] = r[R_X]; sg[tid][1] = r[R_Y]; sg[tid][2] = r[R_W]; sg[tid][3] = r[R_TS];
                sg[tid][4] = r[R_RPX]; sg[tid][5] = r[R_RPY]; sg[tid][6] = condition[g] ? 1.f : 0.f;
#pragma unroll
                for (int q = 0; q < 6; q++) sinv[tid][q] = invraycov[(size_t)g * 6 + q];
#pragma unroll
                for (int q = 0; q < 4; q++) sused[tid][q] = used[(size_t)(base + tid) * 4 + q];
            }
            __syncthreads();
            const int n = min((uint32_t)CH, rend - base);
            if (valid) {
                for (int j = 0; j < n; j++) {
                    const uint32_t it = base - r0 + j + 1;            // num_iterated
                    if (it > last) break;                             // :1450
                    if (!((sused[j][lp >> 6] >> (lp & 63)) & 1ull)) continue;
                    const float dx = sg[j][0] - xy.x, dy = sg[j][1] - xy.y;
                    const float depth_center = sg[j][3];
                    const float depth = depth_center + (sg[j][4] * dx + sg[j][5] * dy);
                    float dz;
                    if (sg[j][6] != 0.f) dz = depth_center - fminf(pdepth, depth);
                    else {
                        if (pdepth < depth) continue;                 // alpha = 0
                        dz = depth_center;
                    }
                    const float c0 = sinv[j][0], c1 = sinv[j][1], c2 = sinv[j][2], c3 = sinv[j][3], c4 = sinv[j][4], c5 = sinv[j][5];
                    const float vx = c0 * dx + c1 * dy + c2 * dz, vy = c1 * dx + c3 * dy + c4 * dz, vz = c2 * dx + c4 * dy + c5 * dz;
                    const float power = -0.5f * (dx * vx + dy * vy + dz * vz);
                    const float alpha = fminf(0.99f, sg[j][2] * __expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    acc += alpha * Tk;
                    Tk = Tk * (1 - alpha);
                }
            }
        }
        if (valid) {
            out_alpha_integrated[pid] = acc;
            out_color_integrated[3 * (size_t)pid + 0] = out_color[0 * HW + pix];
            out_color_integrated[3 * (size_t)pid + 1] = out_color[1 * HW + pix];
            out_color_integrated[3 * (size_t)pid + 2] = out_color[2 * HW + pix];
            out_coordinate2d[2 * (size_t)pid] = xy.x;
            out_coordinate2d[2 * (size_t)pid + 1] = xy.y;
            if (pdepth > 0) {
                const float *ax = pixaux + pix * 8;
                const float dx = ax[3] - xy.x, dy = ax[4] - xy.y;
                out_sdf[pid] = (ax[0] + (ax[1] * dx + ax[2] * dy)) - pdepth;
            }
            atomicAdd(out_color + 8 * HW + pix, 1.0f);   // DISTORTION_OFFSET: number of points of the pixel
        }
    }
}

// ---- host side ----
struct IntegratePointState {
    float2 *points2D;
    float *depths;
    uint32_t *keys, *keys_sorted, *ids, *ids_sorted, *ranges;   // ranges: (T + 1) pairs (the sentinel tile is the last)
    char *sort_temp;
    size_t sort_bytes;
};

static size_t point_sort_bytes(int PN)
{
    size_t bytes = 0;
    uint32_t *k = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, k, k, PN > 0 ? PN : 1);
    return bytes;
}

static size_t carve_points(int PN, size_t T, char *base, IntegratePointState *out)
{
    char *p = base;
    IntegratePointState t;
    const size_t n = (size_t)(PN > 0 ? PN : 1);
    obtain(p, t.points2D, n, 128);
    obtain(p, t.depths, n, 128);
    obtain(p, t.keys, n, 128);
    obtain(p, t.keys_sorted, n, 128);
    obtain(p, t.ids, n, 128);
    obtain(p, t.ids_sorted, n, 128);
    obtain(p, t.ranges, (T + 1) * 2, 128);
    t.sort_bytes = point_sort_bytes(PN);
    obtain(p, t.sort_temp, t.sort_bytes, 128);
    if (out) *out = t;
    return (size_t)(p - base) + 128;
}

size_t integrate_point_bytes(int PN, int width, int height)
{
    const size_t T = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
    return carve_points(PN, T, nullptr, nullptr);
}

size_t integrate_workspace_bytes(int R, int width, int height)
{
    const size_t HW = (size_t)width * height;
    return (size_t)(R > 0 ? R : 0) * 32 + HW * 4 + HW * 32 + 512;
}

bool launch_integrate(int PN, int R, int W, int H, const float *points3D, const float *view, float focal_x, float focal_y,
                      const uint32_t *ranges, const uint32_t *point_list, const float *rec, const float *invraycov,
                      const uint8_t *condition, const float *bg, char *point_chunk, char *work_chunk, float *out_color,
                      float *accum_alpha, float *out_alpha_integrated, float *out_color_integrated,
                      float *out_coordinate2d, float *out_sdf, int point_end_bit, hipStream_t s)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t T = (size_t)gx * gy, HW = (size_t)W * H;
    IntegratePointState ps;
    carve_points(PN, T, (char *)(((uintptr_t)point_chunk + 127) & ~(uintptr_t)127), &ps);
    char *wp = work_chunk;
    unsigned long long *used = nullptr;
    uint32_t *last_contrib = nullptr;
    float *pixaux = nullptr;
    obtain(wp, used, (size_t)(R > 0 ? R : 0) * 4, 128);
    obtain(wp, last_contrib, HW, 128);
    obtain(wp, pixaux, HW * 8, 128);

    hipLaunchKernelGGL(integrate_points_preprocess_kernel, dim3((PN + 255) / 256), dim3(256), 0, s, PN, points3D, view, W, H,
                       focal_x, focal_y, gx, gy, ps.points2D, ps.depths, ps.keys, ps.ids);
    size_t bytes = ps.sort_bytes;
    if (!check_hip(hipcub::DeviceRadixSort::SortPairs(ps.sort_temp, bytes, ps.keys, ps.keys_sorted, ps.ids, ps.ids_sorted, PN, 0,
                                                      point_end_bit, s), "integrate: sort points")) return false;
    if (!check_hip(hipMemsetAsync(ps.ranges, 0, (T + 1) * 2 * sizeof(uint32_t), s), "integrate: memset point ranges")) return false;
    launch_identify_tile_ranges(PN, ps.keys_sorted, ps.ranges, s);
    hipLaunchKernelGGL(integrate_pixels_kernel, dim3(gx, gy), dim3(256), 0, s, W, H, ranges, point_list, rec, bg, out_color,
                       accum_alpha, last_contrib, pixaux, used);
    hipLaunchKernelGGL(integrate_points_kernel, dim3(gx, gy), dim3(256), 0, s, W, H, ranges, ps.ranges, point_list,
                       ps.ids_sorted, rec, invraycov, condition, ps.points2D, ps.depths, last_contrib, pixaux, used,
                       out_color, out_alpha_integrated, out_color_integrated, out_coordinate2d, out_sdf);
    return check_hip(hipGetLastError(), "integrate");
}

}  // namespace ed3
