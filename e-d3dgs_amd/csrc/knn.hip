// knn.hip -- k nearest neighbours of a point cloud among itself: simple_knn.distCUDA2 (mean squared distance to the 3
// nearest other points, submodules/simple-knn/simple_knn.cu:143-220) and the k = 20 neighbour lists the embedding
// regulariser takes from open3d on the CPU (utils/extra_utils.py:5-15, train.py:218-223).
//
// Same idea as the reference -- order the points along a Morton curve, bound groups of consecutive points by boxes, skip
// boxes that cannot hold a nearer neighbour (exact k-NN, not approximate) -- rebuilt around the 64-lane wavefront:
//   * a box is 64 consecutive points of the Morton order = one wavefront's worth; the points are GATHERED into that
//     order once (16-byte records), so a box is one coalesced 1-KB read;
//   * one wave owns one box of queries (lane = query).  A candidate box is tested ONCE per wave, box against box
//     (a lower bound for all 64 queries) against the wave's largest current k-th distance; the reference tests every box
//     per thread and walks 1024-point boxes per thread, divergently;
//   * a surviving box is read by the wave (lane = candidate) and its 64 points are broadcast lane by lane
//     (v_readlane), so all 64 queries see each candidate from registers: no LDS, no divergence;
//   * candidate boxes are visited outwards along the curve (nearest first), so the bound tightens early;
//   * the bounding box of the cloud is reduced with ordered-integer atomics on the device: no host read-back (the
//     reference synchronises twice for it).
// The result is the exact k-NN (any exact method gives the same distances); only the last-ulp rounding of a squared
// distance may differ from the reference's (contraction of dx*dx + dy*dy + dz*dz).
#include <hipcub/hipcub.hpp>

#include <cfloat>

#include "common.h"

namespace ed3 {

namespace {

constexpr int KBOX = 64;

struct KnnWs {
    uint32_t *bounds;        // 6 ordered-integer floats: min xyz, max xyz
    uint32_t *codes, *codes_sorted, *ids, *ids_sorted;
    float4 *sorted;          // [nb * 64] (x, y, z, bits of the original index); pad entries = +inf
    float4 *boxes;           // [nb][2] min, max
    char *sort_temp;
    size_t sort_bytes;
};

size_t knn_sort_bytes(int P)
{
    size_t bytes = 0;
    uint32_t *k = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, k, k, P > 0 ? P : 1, 0, 30);
    return bytes;
}

size_t knn_carve(int P, char *base, KnnWs *w)
{
    char *p = base;
    KnnWs t;
    const size_t n = (size_t)(P > 0 ? P : 1), nb = (n + KBOX - 1) / KBOX;
    obtain(p, t.bounds, 8, 128);
    obtain(p, t.codes, n, 128);
    obtain(p, t.codes_sorted, n, 128);
    obtain(p, t.ids, n, 128);
    obtain(p, t.ids_sorted, n, 128);
    obtain(p, t.sorted, nb * KBOX, 128);
    obtain(p, t.boxes, nb * 2, 128);
    t.sort_bytes = knn_sort_bytes(P);
    obtain(p, t.sort_temp, t.sort_bytes, 128);
    if (w) *w = t;
    return (size_t)(p - base) + 128;
}

__device__ __forceinline__ uint32_t ordered(float f)
{
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t u)
{
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

__global__ void __launch_bounds__(256) knn_bounds_init_kernel(uint32_t *bounds)
{
    if (threadIdx.x < 6) bounds[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
}

__global__ void __launch_bounds__(256) knn_bounds_kernel(int P, const float *__restrict__ pts, uint32_t *__restrict__ bounds)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float v = pts[3 * (size_t)i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = wave_min(mn[a]), hi = wave_max(mx[a]);
        if ((threadIdx.x & 63) == 0) {
            atomicMin(bounds + a, ordered(lo));
            atomicMax(bounds + 3 + a, ordered(hi));
        }
    }
}

// 10 bits per axis, interleaved (simple_knn.cu:46-62)
__device__ __forceinline__ uint32_t spread10(uint32_t x)
{
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float *__restrict__ pts, const uint32_t *__restrict__ bounds,
                                                         uint32_t *__restrict__ codes, uint32_t *__restrict__ ids)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t q[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float lo = unordered(bounds[a]), hi = unordered(bounds[3 + a]);
        const float ext = hi - lo;
        float t = ext > 0.f ? (pts[3 * (size_t)i + a] - lo) / ext * 1023.0f : 0.f;
        t = fminf(fmaxf(t, 0.f), 1023.0f);       // NaN -> 0
        q[a] = (uint32_t)t;
    }
    codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    ids[i] = (uint32_t)i;
}

// wave = box: gather the box's points into Morton order and bound them
__global__ void __launch_bounds__(256) knn_gather_kernel(int P, int nb, const float *__restrict__ pts, const uint32_t *__restrict__ ids_sorted,
                                                         float4 *__restrict__ sorted, float4 *__restrict__ boxes)
{
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nb) return;
    const int i = b * KBOX + lane;
    const bool valid = i < P;
    float x = INFINITY, y = INFINITY, z = INFINITY;
    uint32_t id = 0xFFFFFFFFu;
    if (valid) {
        id = ids_sorted[i];
        x = pts[3 * (size_t)id]; y = pts[3 * (size_t)id + 1]; z = pts[3 * (size_t)id + 2];
    }
    sorted[i] = make_float4(x, y, z, __uint_as_float(id));
    const float lx = wave_min(valid ? x : FLT_MAX), ly = wave_min(valid ? y : FLT_MAX), lz = wave_min(valid ? z : FLT_MAX);
    const float hx = wave_max(valid ? x : -FLT_MAX), hy = wave_max(valid ? y : -FLT_MAX), hz = wave_max(valid ? z : -FLT_MAX);
    if (lane == 0) {
        boxes[2 * b] = make_float4(lx, ly, lz, 0.f);
        boxes[2 * b + 1] = make_float4(hx, hy, hz, 0.f);
    }
}

__device__ __forceinline__ float bcast(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }

// squared distance between two axis-aligned boxes (0 if they overlap): a lower bound for every pair of points in them
__device__ __forceinline__ float box_box_dist2(const float4 &alo, const float4 &ahi, const float4 &blo, const float4 &bhi)
{
    const float dx = fmaxf(0.f, fmaxf(alo.x - bhi.x, blo.x - ahi.x));
    const float dy = fmaxf(0.f, fmaxf(alo.y - bhi.y, blo.y - ahi.y));
    const float dz = fmaxf(0.f, fmaxf(alo.z - bhi.z, blo.z - ahi.z));
    return dx * dx + dy * dy + dz * dz;
}

// K best (ascending) per lane; IDX: also the candidates' original indices
template <int K, bool IDX>
struct Best {
    float d[K];
    uint32_t id[IDX ? K : 1];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int j = 0; j < K; j++) {
            d[j] = FLT_MAX;
            if constexpr (IDX) id[j] = 0xFFFFFFFFu;
        }
    }
    __device__ __forceinline__ void insert(float c, uint32_t cid)
    {
        if constexpr (IDX) {
#pragma unroll
            for (int j = K - 1; j >= 1; j--) {
                const bool below = c < d[j - 1], here = c < d[j];
                id[j] = below ? id[j - 1] : (here ? cid : id[j]);
                d[j] = below ? d[j - 1] : (here ? c : d[j]);
            }
            if (c < d[0]) { d[0] = c; id[0] = cid; }
        } else {
            // sorted d: the new j-th smallest of d[] + {c} is the median of (d[j-1], d[j], c)
#pragma unroll
            for (int j = K - 1; j >= 1; j--) d[j] = __builtin_amdgcn_fmed3f(d[j - 1], d[j], c);
            d[0] = fminf(d[0], c);
        }
    }
};

template <int K, bool IDX>
__global__ void __launch_bounds__(256) knn_search_kernel(int P, int nb, const float4 *__restrict__ sorted, const float4 *__restrict__ boxes,
                                                         float *__restrict__ out_d, int64_t *__restrict__ out_i)
{
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (q >= nb) return;
    const float4 me4 = sorted[(size_t)q * KBOX + lane];
    const bool valid = q * KBOX + lane < P;
    // pad lanes query lane 0's point (finite arithmetic); their results are not written
    const float qx = valid ? me4.x : bcast(me4.x, 0), qy = valid ? me4.y : bcast(me4.y, 0), qz = valid ? me4.z : bcast(me4.z, 0);
    Best<K, IDX> best;
    best.init();
    auto scan_box = [&](const float4 &c4, bool own) {
#pragma unroll 16
        for (int j = 0; j < KBOX; j++) {
            const float dx = bcast(c4.x, j) - qx, dy = bcast(c4.y, j) - qy, dz = bcast(c4.z, j) - qz;
            float c = dx * dx + dy * dy + dz * dz;      // +inf for a pad candidate
            if (own && j == lane) c = FLT_MAX;          // the query itself (simple_knn.cu:176: i == idx)
            if constexpr (IDX) {
                if (__ballot(c < best.d[K - 1]) == 0) continue;
                best.insert(c, __builtin_amdgcn_readlane(__float_as_int(c4.w), j));
            } else {
                best.insert(c, 0u);
            }
        }
    };
    scan_box(me4, true);
    float rmax = wave_max(valid ? best.d[K - 1] : 0.f);
    const float4 qlo = boxes[2 * q], qhi = boxes[2 * q + 1];
    // candidate boxes outwards along the curve: rank t -> box q + (t/2 + 1) (t even) or q - (t/2 + 1) (t odd)
    const int tmax = 2 * max(q, nb - 1 - q);
    for (int t0 = 0; t0 < tmax; t0 += 64) {
        const int t = t0 + lane;
        const int b = q + ((t & 1) ? -(t / 2 + 1) : (t / 2 + 1));
        float lb = FLT_MAX;
        float4 blo = qlo, bhi = qhi;
        if (t < tmax && b >= 0 && b < nb) {
            blo = boxes[2 * b]; bhi = boxes[2 * b + 1];
            lb = box_box_dist2(qlo, qhi, blo, bhi);
        }
        unsigned long long m = __ballot(lb <= rmax && lb < FLT_MAX);
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            if (bcast(lb, j) > rmax) continue;            // the bound has tightened since the test
            {   // second test, per query: does ANY lane still need this box?  (one far-out query in the box inflates the
                // wave's bound; it must not drag every box near the other 63 in)
                const float dx = fmaxf(0.f, fmaxf(bcast(blo.x, j) - qx, qx - bcast(bhi.x, j)));
                const float dy = fmaxf(0.f, fmaxf(bcast(blo.y, j) - qy, qy - bcast(bhi.y, j)));
                const float dz = fmaxf(0.f, fmaxf(bcast(blo.z, j) - qz, qz - bcast(bhi.z, j)));
                if (__ballot(valid && dx * dx + dy * dy + dz * dz <= best.d[K - 1]) == 0) continue;
            }
            const int tt = t0 + j;
            const int bb = q + ((tt & 1) ? -(tt / 2 + 1) : (tt / 2 + 1));
            const float4 c4 = sorted[(size_t)bb * KBOX + lane];
            scan_box(c4, false);
            rmax = wave_max(valid ? best.d[K - 1] : 0.f);
        }
    }
    if (!valid) return;
    const uint32_t id = __float_as_uint(me4.w);
    if constexpr (IDX) {
#pragma unroll
        for (int j = 0; j < K; j++) {
            out_d[(size_t)id * K + j] = best.d[j];
            out_i[(size_t)id * K + j] = best.id[j] == 0xFFFFFFFFu ? (int64_t)-1 : (int64_t)best.id[j];
        }
    } else {
        static_assert(K == 3, "the mean distance is over 3 neighbours");
        out_d[id] = (best.d[0] + best.d[1] + best.d[2]) / 3.0f;    // simple_knn.cu:181
    }
}

bool knn_prepare(int P, const float *points, const KnnWs &w, hipStream_t s)
{
    const int nb = (P + KBOX - 1) / KBOX;
    hipLaunchKernelGGL(knn_bounds_init_kernel, dim3(1), dim3(64), 0, s, w.bounds);
    hipLaunchKernelGGL(knn_bounds_kernel, dim3(std::min((P + 255) / 256, 1024)), dim3(256), 0, s, P, points, w.bounds);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, w.bounds, w.codes, w.ids);
    size_t bytes = w.sort_bytes;
    if (!check_hip(hipcub::DeviceRadixSort::SortPairs(w.sort_temp, bytes, w.codes, w.codes_sorted, w.ids, w.ids_sorted, P, 0, 30, s),
                   "knn SortPairs")) return false;
    hipLaunchKernelGGL(knn_gather_kernel, dim3((nb + 3) / 4), dim3(256), 0, s, P, nb, points, w.ids_sorted, w.sorted, w.boxes);
    return true;
}

}  // namespace
}  // namespace ed3

using namespace ed3;

extern "C" {

size_t ed3dgs_knn_workspace_bytes(int P) { return knn_carve(P, nullptr, nullptr); }

int ed3dgs_knn_mean_dist2(int P, const float *points, float *mean_dist2, char *workspace, size_t workspace_bytes, void *stream)
{
    if (P < 0) { set_error("ed3dgs_knn_mean_dist2: bad P"); return ED3DGS_ERR_INVALID; }
    if (P == 0) return 0;
    if (!points || !mean_dist2 || !workspace) { set_error("ed3dgs_knn_mean_dist2: null pointer"); return ED3DGS_ERR_INVALID; }
    if (workspace_bytes < ed3dgs_knn_workspace_bytes(P)) { set_error("ed3dgs_knn_mean_dist2: workspace too small"); return ED3DGS_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    KnnWs w;
    knn_carve(P, (char *)(((uintptr_t)workspace + 127) & ~(uintptr_t)127), &w);
    if (!knn_prepare(P, points, w, s)) return ED3DGS_ERR_HIP;
    const int nb = (P + KBOX - 1) / KBOX;
    hipLaunchKernelGGL((knn_search_kernel<3, false>), dim3((nb + 3) / 4), dim3(256), 0, s, P, nb, w.sorted, w.boxes, mean_dist2, (int64_t *)nullptr);
    return check_hip(hipGetLastError(), "knn_mean_dist2") ? 0 : ED3DGS_ERR_HIP;
}

int ed3dgs_knn_neighbours(int P, int K, const float *points, float *sq_dists, int64_t *indices, char *workspace,
                          size_t workspace_bytes, void *stream)
{
    if (P < 0) { set_error("ed3dgs_knn_neighbours: bad P"); return ED3DGS_ERR_INVALID; }
    if (K != 20) { set_error("ed3dgs_knn_neighbours: K must be 20 (train.py:219)"); return ED3DGS_ERR_INVALID; }
    if (P == 0) return 0;
    if (!points || !sq_dists || !indices || !workspace) { set_error("ed3dgs_knn_neighbours: null pointer"); return ED3DGS_ERR_INVALID; }
    if (workspace_bytes < ed3dgs_knn_workspace_bytes(P)) { set_error("ed3dgs_knn_neighbours: workspace too small"); return ED3DGS_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    KnnWs w;
    knn_carve(P, (char *)(((uintptr_t)workspace + 127) & ~(uintptr_t)127), &w);
    if (!knn_prepare(P, points, w, s)) return ED3DGS_ERR_HIP;
    const int nb = (P + KBOX - 1) / KBOX;
    hipLaunchKernelGGL((knn_search_kernel<20, true>), dim3((nb + 3) / 4), dim3(256), 0, s, P, nb, w.sorted, w.boxes, sq_dists, indices);
    return check_hip(hipGetLastError(), "knn_neighbours") ? 0 : ED3DGS_ERR_HIP;
}

}  // extern "C"
