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
// Binning level 1, hand-written (round 4): order[] = the P Gaussians sorted by their 32 depth bits, ties in id order, culled ones
// (key 0xFFFFFFFF: they emit no instance) last, in id order -- the permutation a stable sort of (key, id) pairs leaves, which is
// what CR/rasterizer_impl.cu:381-386 does to the depth half of its (tile | depth) keys (see the header of this file).
//
// A stable sort of pairs whose input is in id order IS the sort by the 64-bit composite (key, id), and composites are distinct, so
// the output is determined without any notion of stability -- which frees the algorithm from the pass structure of a radix sort
// (rounds 2-3: the library's merge sort 9 launches / 60 us; its Onesweep 15 / 155; two hand-written sorts 6 / 81 and 3 / 86):
//   1  depth_rank_count_kernel    every block finds the frame's smallest / largest key from K1's per-block extremes, maps a key to
//                                 one of NB = 8192 buckets of equal width in KEY-BIT space ((key - kmin) >> shift; key bits of a
//                                 positive float grow like its logarithm, so an outlier in depth costs buckets, not balance) and
//                                 counts with one global atomic per key; culled keys are counted per block of ids; the block
//                                 that finishes last scans the counts into bucket offsets (and the culled counts into a prefix);
//   2  depth_rank_scatter_kernel  drops each (key, id) into its bucket at the next free slot -- a returning atomic on the
//                                 bucket's cursor: the order INSIDE a bucket is whatever the atomics made it; culled ids go
//                                 straight to their final places behind the visible ones;
//   3  depth_rank_place_kernel    thread = slot: the composite's rank inside its bucket is the number of smaller composites in
//                                 it (the block's few hundred neighbouring pairs staged in LDS), and order[bucket start + rank]
//                                 = id.  Deterministic whatever order step 2's atomics produced.
// Cost is linear in P while buckets stay small; a bucket of m pairs costs m^2 compares, so a frame whose depths pile into few
// buckets (many equal depths) degrades gracefully -- 4096 pairs in one bucket are 16 M compares, still microseconds -- and stays
// correct at any size.  counts / cursors / ticket (2 x 32 KB + 256 B) are zeroed by K1's first blocks (the launch in front of step 1).
// ------------------------------------------------------------------------------------------------------------
constexpr int DR_NB = 8192, DR_IPT = 2, DR_TILE = 256 * DR_IPT;
// sort_counts layout (words): counts [NB] | cursors [NB] | ticket + pad [64] -- those zeroed by K1 -- | offs [NB + 1], n_vis = offs[NB],
// kmin, shift [NB + 64 in all] | culled ids per block [nblk] | their exclusive prefix [nblk]
constexpr int DR_TICKET = 2 * DR_NB, DR_OFFS = 2 * DR_NB + 64, DR_BLK = DR_OFFS + DR_NB + 64;
static_assert(DR_TICKET + 64 == DEPTH_SORT_ZERO_WORDS, "K1 zeroes counts, cursors and the ticket");

size_t depth_sort_count_words(int P)
{
    const size_t nblk = ((size_t)(P > 0 ? P : 0) + DR_TILE - 1) / DR_TILE + 1;
    return (size_t)DR_BLK + 2 * nblk + 64;
}

struct DepthRank {
    int P, nblk_k1, nblk;
    const uint32_t *keys;            // K1's depth bits (0xFFFFFFFF: culled)
    const uint32_t *block_kminmax;   // K1's per-block smallest / largest visible key
    uint32_t *w;                     // sort_counts (layout above)
    uint2 *pairs;                    // [P] (key, id), bucket by bucket
    uint32_t *order;
};

// {kmin, shift} of the frame from K1's per-block extremes, by every block of step 1 for itself (a few hundred words from L2)
__device__ __forceinline__ void depth_rank_params(const DepthRank &a, uint32_t *sh, uint32_t &kmin, int &shift)
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
    const uint32_t span = mx - mn;                     // bucket = (key - kmin) >> shift must stay below NB
    const int bits = span ? 32 - __builtin_clz(span) : 0;
    shift = max(0, bits - 13);                         // NB = 2^13
}

__device__ __forceinline__ uint32_t coherent_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Step 1: bucket counts (one global atomic per visible key), culled ids per block; the block that takes the last ticket scans the
// counts into bucket offsets and the per-block culled counts into their prefix, for steps 2 and 3.  Ordering without fences (a
// device-scope release would write back the XCD's L2): everything the last block reads was written by RETURNING atomics whose
// results their threads consumed before the block's barrier, and the ticket is taken behind that barrier (the idiom of
// deform_active_rows_body and image_stats_kernel); the last block reads with agent-scope loads.
__global__ void __launch_bounds__(256) depth_rank_count_kernel(DepthRank a)
{
    __shared__ uint32_t red[8], wsum[4];
    __shared__ bool last_s;
    uint32_t kmin;
    int shift;
    depth_rank_params(a, red, kmin, shift);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * DR_TILE;
    uint32_t k[DR_IPT];
#pragma unroll
    for (int j = 0; j < DR_IPT; j++) { const int i = base + j * 256 + tid; k[j] = i < a.P ? a.keys[i] : 0u; }
    uint32_t nc = 0, seen = 0;
#pragma unroll
    for (int j = 0; j < DR_IPT; j++) {
        const int i = base + j * 256 + tid;
        if (i >= a.P) continue;
        if (k[j] == 0xFFFFFFFFu) nc++;
        else seen += atomicAdd(&a.w[(k[j] - kmin) >> shift], 1u);   // RETURNING: consumed below, before the block's ticket
    }
    asm volatile("" :: "v"(seen) : "memory");   // the results are in registers here: the adds have been performed at the memory side
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nc += (uint32_t)__shfl_xor((int)nc, o);
    if (lane == 0) wsum[wave] = nc;
    __syncthreads();   // every thread of the block is past its atomics' results
    if (tid == 0) {
        const uint32_t old = atomicExch(&a.w[DR_BLK + blockIdx.x], wsum[0] + wsum[1] + wsum[2] + wsum[3]);
        asm volatile("" :: "v"(old) : "memory");        // its result is back: the exchange has been performed
        last_s = atomicAdd(&a.w[DR_TICKET], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last_s) return;
    // exclusive scan of the bucket counts: 32 consecutive buckets per thread, a wave scan, the four wave sums
    constexpr int PER = DR_NB / 256;
    uint32_t c[PER], mine = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { c[q] = coherent_load(&a.w[tid * PER + q]); mine += c[q]; }
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t nb = __shfl_up(inc, o); if (lane >= o) inc += nb; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (int q = 0; q < wave; q++) run += wsum[q];
    const uint32_t n_vis = wsum[0] + wsum[1] + wsum[2] + wsum[3];
#pragma unroll
    for (int q = 0; q < PER; q++) { a.w[DR_OFFS + tid * PER + q] = run; run += c[q]; }
    if (tid == 0) { a.w[DR_OFFS + DR_NB] = n_vis; a.w[DR_OFFS + DR_NB + 1] = kmin; a.w[DR_OFFS + DR_NB + 2] = (uint32_t)shift; }
    __syncthreads();
    // exclusive prefix of the per-block culled counts (blocks of DR_TILE ids)
    uint32_t carry = 0;
    for (int b0 = 0; b0 < a.nblk; b0 += 256) {
        const int b = b0 + tid;
        const uint32_t v = b < a.nblk ? coherent_load(&a.w[DR_BLK + b]) : 0u;
        uint32_t pinc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t nb = __shfl_up(pinc, o); if (lane >= o) pinc += nb; }
        if (lane == 63) wsum[wave] = pinc;
        __syncthreads();
        uint32_t before = carry;
        for (int q = 0; q < wave; q++) before += wsum[q];
        if (b < a.nblk) a.w[DR_BLK + a.nblk + b] = before + pinc - v;
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// Step 2: every (key, id) into its bucket at the next free slot (a returning atomic on the bucket's cursor); culled ids to their
// final places behind the visible ones, in id order
__global__ void __launch_bounds__(256) depth_rank_scatter_kernel(DepthRank a)
{
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *offs = a.w + DR_OFFS;
    const uint32_t n_vis = offs[DR_NB], kmin = offs[DR_NB + 1];
    const int shift = (int)offs[DR_NB + 2];
    const int base = blockIdx.x * DR_TILE;
    uint32_t k[DR_IPT], slot[DR_IPT], start[DR_IPT];
#pragma unroll
    for (int j = 0; j < DR_IPT; j++) { const int i = base + j * 256 + tid; k[j] = i < a.P ? a.keys[i] : 0xFFFFFFFFu; }
#pragma unroll
    for (int j = 0; j < DR_IPT; j++) {   // a thread's returning atomics and offset loads are in flight together
        const bool vis = k[j] != 0xFFFFFFFFu;
        const uint32_t b = vis ? (k[j] - kmin) >> shift : 0u;
        slot[j] = vis ? atomicAdd(&a.w[DR_NB + b], 1u) : 0u;
        start[j] = offs[b];
    }
    uint32_t cull_run = n_vis + a.w[DR_BLK + a.nblk + blockIdx.x];   // ids ascend with (j, wave, lane): the culled ones keep that order
#pragma unroll
    for (int j = 0; j < DR_IPT; j++) {
        const int i = base + j * 256 + tid;
        const bool in = i < a.P, culled = in && k[j] == 0xFFFFFFFFu;
        const unsigned long long bal = __ballot(culled);
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0;
        for (int q = 0; q < wave; q++) before += wsum[q];
        const uint32_t row_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (culled) a.order[cull_run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)i;
        else if (in) a.pairs[start[j] + slot[j]] = make_uint2(k[j], (uint32_t)i);
        cull_run += row_total;
        __syncthreads();
    }
}

// Step 3: thread = slot p of the bucketed array; rank of its composite inside its bucket = the number of smaller composites there;
// order[bucket start + rank] = id.  The block's 256 slots and the buckets they fall in span a few hundred consecutive pairs: those
// are staged in LDS once (coalesced) and ranked from there; a span too long for the buffer (a crowded bucket) is read from memory.
constexpr int DR_SPAN = 1536;
__global__ void __launch_bounds__(256) depth_rank_place_kernel(DepthRank a)
{
    __shared__ uint2 sp[DR_SPAN];
    __shared__ uint32_t lo_s, hi_s;
    const uint32_t *offs = a.w + DR_OFFS;
    const uint32_t n_vis = offs[DR_NB], kmin = offs[DR_NB + 1];
    const int shift = (int)offs[DR_NB + 2];
    const uint32_t p0 = blockIdx.x * 256u;
    if (p0 >= n_vis) return;
    const uint32_t p = p0 + threadIdx.x;
    const bool live = p < n_vis;
    const uint2 me = a.pairs[live ? p : n_vis - 1];
    const uint32_t b = (me.x - kmin) >> shift;
    const uint32_t start = offs[b], end = offs[b + 1];
    // slots ascend with the bucket: the block's span is [start of the first thread's bucket, end of the last live thread's)
    const uint32_t last_live = min(255u, n_vis - 1u - p0);
    if (threadIdx.x == 0) lo_s = start;
    if (threadIdx.x == last_live) hi_s = end;
    __syncthreads();
    const uint32_t lo = lo_s, hi = hi_s;
    uint32_t r = 0;
    if (hi - lo <= (uint32_t)DR_SPAN) {
        for (uint32_t q = lo + threadIdx.x; q < hi; q += 256u) sp[q - lo] = a.pairs[q];
        __syncthreads();
        if (live)
            for (uint32_t j = start; j < end; j++) { const uint2 q = sp[j - lo]; r += (q.x < me.x || (q.x == me.x && q.y < me.y)); }
    } else if (live) {
        uint32_t j = start;
        for (; j + 4 <= end; j += 4) {
            const uint2 q0 = a.pairs[j], q1 = a.pairs[j + 1], q2 = a.pairs[j + 2], q3 = a.pairs[j + 3];
            r += (q0.x < me.x || (q0.x == me.x && q0.y < me.y)) + (q1.x < me.x || (q1.x == me.x && q1.y < me.y)) +
                 (q2.x < me.x || (q2.x == me.x && q2.y < me.y)) + (q3.x < me.x || (q3.x == me.x && q3.y < me.y));
        }
        for (; j < end; j++) { const uint2 q = a.pairs[j]; r += (q.x < me.x || (q.x == me.x && q.y < me.y)); }
    }
    if (live) a.order[start + r] = me.y;
}

bool launch_depth_sort(const GeometryState &g, int P, hipStream_t s)
{
    if (P <= 0) return true;
    DepthRank a;
    a.P = P; a.nblk_k1 = (P + 255) / 256; a.nblk = (P + DR_TILE - 1) / DR_TILE;
    a.keys = g.depth_keys; a.block_kminmax = g.block_kminmax; a.order = g.order;
    a.w = g.sort_counts;
    a.pairs = reinterpret_cast<uint2 *>(g.sort_a);
    hipLaunchKernelGGL(depth_rank_count_kernel, dim3(a.nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(depth_rank_scatter_kernel, dim3(a.nblk), dim3(256), 0, s, a);
    hipLaunchKernelGGL(depth_rank_place_kernel, dim3((P + 255) / 256), dim3(256), 0, s, a);
    return check_hip(hipGetLastError(), "depth sort");
}

}  // namespace ed3
