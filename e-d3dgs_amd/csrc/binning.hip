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

}  // namespace ed3
