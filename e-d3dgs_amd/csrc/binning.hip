// binning.hip -- K2 (inclusive scans of tiles_touched) and K4 (stable radix sorts).
// Plain library primitives (rocPRIM through hipCUB).  The reference sorts (tile << 32 | depth bits, id) pairs once over
// key bits [0, 32 + msb(T)) (CR/rasterizer_impl.cu:355, :378-386): 6 onesweep passes over 12-byte pairs.  An LSD radix
// sort is a sequence of stable passes from the low bits up, so the same permutation comes out of TWO sorts:
//   level 1: the P Gaussians by their 32 depth bits (value = Gaussian id; tiny, P << R);
//   level 2: the R instances, EMITTED in that depth order, by the msb(T) tile bits only (2 passes over 8-byte pairs).
// Ties behave identically: equal depth bits keep Gaussian-id order (level 1 is stable, as the low-bit passes of the
// reference's sort are over instances emitted in id order), equal tiles keep depth order (level 2 is stable).
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ed3 {

namespace {
struct Gather {
    const uint32_t *src;
    __host__ __device__ __forceinline__ uint32_t operator()(const uint32_t &i) const { return src[i]; }
};
}  // namespace

size_t scan_temp_bytes(int P)
{
    size_t bytes = 0;
    uint32_t *p = nullptr;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, bytes, p, p, P > 0 ? P : 1);
    size_t bytes2 = 0;
    hipcub::TransformInputIterator<uint32_t, Gather, const uint32_t *> it(p, Gather{p});
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, bytes2, it, p, P > 0 ? P : 1);
    return bytes > bytes2 ? bytes : bytes2;
}

size_t sort_temp_bytes(int n)
{
    size_t bytes = 0;
    uint32_t *k = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, k, k, n > 0 ? n : 1);
    return bytes;
}

bool run_scan(char *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, int P, hipStream_t s)
{
    return check_hip(hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, in, out, P, s), "InclusiveSum");
}

bool run_scan_gather(char *temp, size_t temp_bytes, const uint32_t *in, const uint32_t *order, uint32_t *out, int P,
                     hipStream_t s)
{
    hipcub::TransformInputIterator<uint32_t, Gather, const uint32_t *> it(order, Gather{in});
    return check_hip(hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, it, out, P, s), "InclusiveSum (depth order)");
}

// (Level 1, 200k keys: below a million items the library runs a merge sort behind this entry point -- 9 launches, 60 us.
// Forcing its Onesweep radix sort instead (rocprim::radix_sort_config<..., MergeSortLimit = 0>) was measured in round 2:
// 15 launches with its look-back memsets, 155 us.  The library's choice stands.)
bool run_sort(char *temp, size_t temp_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
              int n, int end_bit, hipStream_t s)
{
    if (n <= 0) return true;
    return check_hip(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, kin, kout, vin, vout, n, 0, end_bit, s),
                     "SortPairs");
}


// ------------------------------------------------------------------------------------------------------------
// Binning level 1, hand-written: order[] = the P Gaussians sorted by their 32 depth bits, ties in id order
// (CR/rasterizer_impl.cu:381-386 sorts (tile | depth) keys; the depth half is this sort, see the header of this file).
//
// A stable LSD radix sort in three passes over only the bits in which the frame's keys differ: K1 leaves the smallest and the
// largest key of each of its blocks; k' = key - kmin needs nb = bitlength(kmax - kmin + 1) bits (24 .. 27 for scenes whose depth
// spans 2 .. 16 binades) and the culled Gaussians (key 0xFFFFFFFF, they emit no instance) take the one value above the largest:
// three digits of w = ceil(nb / 3) <= 11 bits.  Per pass two launches:
//   count    tile (4096 keys) x digit histogram in LDS -> counts[pass][tile][digit];
//   scatter  every block sums the count columns for itself (where its tile's keys of each digit start: the digit's start +
//            the earlier tiles' keys of that digit; <= 49 x 2^w words at 200k -- a scan launch of its own would be one block's
//            chain of loads), ranks its keys -- a wave at a time, a row of 64 keys at a time: the lanes holding the same digit
//            are found with w ballots, their rank is a popcount, the row's count goes to the wave's running counter by ONE
//            LDS atomic whose returned value is broadcast to them (no atomic's order decides a position) -- and writes
//            (key, id) to its place.  Waves, rows and lanes are walked in order: the pass is stable.
// 6 launches.  MEASURED (round 3, 200k keys): count 6-7 us, scatter 18-22 us per pass, 81 us in all -- SLOWER than the library's
// merge sort behind hipcub::DeviceRadixSort (9 launches, 60-70 us; its Onesweep radix sort, forced: 15 launches, 155 us): 49
// tiles occupy 49 of 256 CUs, and a scatter block is five dependent phases (frame extremes, column sums, key loads, 16 ranked
// rows, scattered stores) of one or two memory / LDS-atomic latencies each.  It therefore stays OPT-IN (ED3DGS_SORT_HANDWRITTEN=1;
// bit-identical lists: tests/test_binning_stress_gpu.py); what would beat the library is a single launch per pass with a
// decoupled look-back instead of count + column sums, not attempted.
// ------------------------------------------------------------------------------------------------------------
constexpr int DS_TILE = 4096, DS_IPT = 16, DS_MAXBINS = 2048;

size_t depth_sort_count_words(int P)
{
    const size_t tiles = ((size_t)(P > 0 ? P : 0) + DS_TILE - 1) / DS_TILE + 1;
    return 3 * tiles * DS_MAXBINS;
}

struct DepthSort {
    int P, ntiles, nblk_k1, pass;
    const uint32_t *raw;            // K1's keys (pass 0 reads them and forms k')
    const uint32_t *block_kminmax;
    const uint32_t *in;             // [2][P] (k', id) of the previous pass
    uint32_t *out;                  // [2][P]; the last pass writes the ids to `order` instead
    uint32_t *order;
    uint32_t *counts;               // this pass's [ntiles][bins]
    uint32_t *params;
};

// {kmin, nb, w} of the frame from K1's per-block extremes, by every block for itself (a few hundred words)
__device__ __forceinline__ void depth_sort_params(const DepthSort &a, uint32_t *sh, uint32_t &kmin, uint32_t &culled_key, int &w)
{
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    for (int b = threadIdx.x; b < a.nblk_k1; b += blockDim.x) { mn = min(mn, a.block_kminmax[2 * b]); mx = max(mx, a.block_kminmax[2 * b + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, o)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
    if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = mn; sh[2 * (threadIdx.x >> 6) + 1] = mx; }
    __syncthreads();
    for (int q = 0; q < (int)(blockDim.x >> 6); q++) { mn = min(mn, sh[2 * q]); mx = max(mx, sh[2 * q + 1]); }
    __syncthreads();
    if (mn > mx) { mn = 0u; mx = 0u; }                 // no visible Gaussian at all
    kmin = mn;
    culled_key = mx - mn + 1u;                          // one above the largest k'
    const int bits = culled_key ? 32 - __builtin_clz(culled_key) : 32;   // the values 0 .. culled_key (0: the span wrapped, all 32)
    w = (bits + 2) / 3;
    if (w < 1) w = 1;
}

__global__ void __launch_bounds__(256) depth_sort_count_kernel(DepthSort a)
{
    __shared__ uint32_t hist[DS_MAXBINS];
    __shared__ uint32_t red[8];
    uint32_t kmin, culled;
    int w;
    depth_sort_params(a, red, kmin, culled, w);
    const int bins = 1 << w, shift = a.pass * w;
    for (int q = threadIdx.x; q < bins; q += 256) hist[q] = 0u;
    __syncthreads();
    const int base = blockIdx.x * DS_TILE;
    uint32_t k[DS_IPT];
#pragma unroll
    for (int j = 0; j < DS_IPT; j++) {
        const int i = base + j * 256 + threadIdx.x;
        uint32_t v = 0xFFFFFFFFu;
        if (i < a.P) v = a.pass == 0 ? a.raw[i] : a.in[i];
        k[j] = v;
    }
#pragma unroll
    for (int j = 0; j < DS_IPT; j++) {
        const int i = base + j * 256 + threadIdx.x;
        if (i >= a.P) continue;
        const uint32_t kp = a.pass == 0 ? (k[j] == 0xFFFFFFFFu ? culled : k[j] - kmin) : k[j];
        atomicAdd(&hist[(kp >> shift) & (uint32_t)(bins - 1)], 1u);
    }
    __syncthreads();
    uint32_t *row = a.counts + (size_t)blockIdx.x * bins;
    for (int q = threadIdx.x; q < bins; q += 256) row[q] = hist[q];
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.pass == 0) { a.params[0] = kmin; a.params[1] = culled; a.params[2] = (uint32_t)w; }
}

__global__ void __launch_bounds__(256) depth_sort_scatter_kernel(DepthSort a)
{
    __shared__ uint32_t wcnt[4][DS_MAXBINS];   // per wave and digit: running count, then the position of the wave's first key
    __shared__ uint32_t tot[DS_MAXBINS];       // per digit: keys of all tiles, then their exclusive scan
    __shared__ uint32_t red[8], wsum[4];
    uint32_t kmin, culled;
    int w;
    depth_sort_params(a, red, kmin, culled, w);
    const int bins = 1 << w, shift = a.pass * w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < 4 * DS_MAXBINS; q += 256) (&wcnt[0][0])[q] = 0u;
    // column sums: digits in groups of 8 (two 16-byte loads per tile row); the block's 256 threads split the tiles of a group
    // among themselves and keep FOUR rows' loads in flight each (a thread walking all tiles of its group alone is a chain of
    // ntiles memory latencies: 30 us per pass at 200k keys), partial sums meet in LDS
    __shared__ uint32_t pre_s[DS_MAXBINS], all_s[DS_MAXBINS];
    for (int q = tid; q < bins; q += 256) { pre_s[q] = 0u; all_s[q] = 0u; }
    __syncthreads();
    if (bins >= 8) {
        const int ngrp = bins >> 3, parts = max(1, 256 / ngrp);
        const int grp = tid % ngrp, part = tid / ngrp;
        if (part < parts) {
            uint32_t pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, aa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t0 = part; t0 < a.ntiles; t0 += 4 * parts) {
                uint4 lo[4], hi[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = min(t0 + u * parts, a.ntiles - 1);
                    lo[u] = *reinterpret_cast<const uint4 *>(a.counts + (size_t)t * bins + 8 * grp);
                    hi[u] = *reinterpret_cast<const uint4 *>(a.counts + (size_t)t * bins + 8 * grp + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + u * parts;
                    if (t >= a.ntiles) continue;
                    const uint32_t v[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
#pragma unroll
                    for (int q = 0; q < 8; q++) { aa[q] += v[q]; pp[q] += t < (int)blockIdx.x ? v[q] : 0u; }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) { atomicAdd(&all_s[8 * grp + q], aa[q]); atomicAdd(&pre_s[8 * grp + q], pp[q]); }
        }
    } else if (tid == 0) {
        for (int t = 0; t < a.ntiles; t++)
            for (int q = 0; q < bins; q++) { const uint32_t v = a.counts[(size_t)t * bins + q]; all_s[q] += v; pre_s[q] += t < (int)blockIdx.x ? v : 0u; }
    }
    __syncthreads();
    const int d0 = 8 * tid;
    uint32_t pre[8] = {0, 0, 0, 0, 0, 0, 0, 0}, all[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; q++)
        if (d0 + q < bins) { pre[q] = pre_s[d0 + q]; all[q] = all_s[d0 + q]; }
    // exclusive scan of the digit totals over the block: 8 per thread, a wave scan, the four wave sums
    uint32_t mine = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) mine += all[q];
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t nb = __shfl_up(inc, o); if (lane >= o) inc += nb; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (int q = 0; q < wave; q++) run += wsum[q];
    if (d0 < bins) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (d0 + q < bins) tot[d0 + q] = run + pre[q];   // where THIS tile's keys of digit d0 + q start
            run += all[q];
        }
    }
    // the tile's keys: wave `wave` owns rows 16 wave .. 16 wave + 15 of 64 consecutive keys
    const int base = blockIdx.x * DS_TILE + wave * (DS_IPT * 64);
    uint32_t kp[DS_IPT], id[DS_IPT], rank[DS_IPT];
#pragma unroll
    for (int j = 0; j < DS_IPT; j++) {
        const int i = base + j * 64 + lane;
        const int ii = min(i, a.P - 1);
        if (a.pass == 0) {
            const uint32_t raw = a.raw[ii];
            kp[j] = raw == 0xFFFFFFFFu ? culled : raw - kmin;
            id[j] = (uint32_t)ii;
        } else {
            kp[j] = a.in[ii];
            id[j] = a.in[(size_t)a.P + ii];
        }
    }
    __syncthreads();   // wcnt zeroed, tot written
#pragma unroll
    for (int j = 0; j < DS_IPT; j++) {
        const bool in = base + j * 64 + lane < a.P;
        const uint32_t d = (kp[j] >> shift) & (uint32_t)(bins - 1);
        unsigned long long m = __ballot(in);           // lanes of this row with the same digit
        for (int b = 0; b < w; b++) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        if (!in) m = 0ull;
        const int leader = m ? __builtin_ctzll(m) : 0;
        uint32_t prior = 0;
        if (in && lane == leader) prior = atomicAdd(&wcnt[wave][d], (uint32_t)__popcll(m));   // one LDS atomic per (row, digit)
        prior = (uint32_t)__shfl((int)prior, leader);
        rank[j] = prior + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    // per digit: the waves' first positions (tile start of the digit + the lower waves' keys of it)
    for (int d = tid; d < bins; d += 256) {
        uint32_t r = tot[d];
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t v = wcnt[q][d]; wcnt[q][d] = r; r += v; }
    }
    __syncthreads();
    const bool last = a.pass == 2;
#pragma unroll
    for (int j = 0; j < DS_IPT; j++) {
        if (base + j * 64 + lane >= a.P) continue;
        const uint32_t d = (kp[j] >> shift) & (uint32_t)(bins - 1);
        const uint32_t pos = wcnt[wave][d] + rank[j];
        if (last) a.order[pos] = id[j];
        else { a.out[pos] = kp[j]; a.out[(size_t)a.P + pos] = id[j]; }
    }
}

bool launch_depth_sort(const GeometryState &g, int P, hipStream_t s)
{
    if (P <= 0) return true;
    DepthSort a;
    a.P = P; a.ntiles = (P + DS_TILE - 1) / DS_TILE; a.nblk_k1 = (P + 255) / 256;
    a.raw = g.depth_keys; a.block_kminmax = g.block_kminmax; a.order = g.order; a.params = g.sort_params;
    const size_t stride = (size_t)(a.ntiles + 1) * DS_MAXBINS;
    for (int pass = 0; pass < 3; pass++) {
        a.pass = pass;
        a.in = pass == 1 ? g.sort_a : g.sort_b;     // pass 0 reads the raw keys
        a.out = pass == 0 ? g.sort_a : g.sort_b;    // pass 2 writes `order`
        a.counts = g.sort_counts + pass * stride;
        hipLaunchKernelGGL(depth_sort_count_kernel, dim3(a.ntiles), dim3(256), 0, s, a);
        hipLaunchKernelGGL(depth_sort_scatter_kernel, dim3(a.ntiles), dim3(256), 0, s, a);
    }
    return check_hip(hipGetLastError(), "depth sort");
}

}  // namespace ed3
