// binning.hip -- K2 (inclusive scan of tiles_touched) and K4 (stable radix sort of (tile|depth, id) pairs).
// Plain library primitives (rocPRIM through hipCUB); the contract is the one the reference gets from CUB:
// exact integer scan, stable LSD radix sort on key bits [0, 32+msb(T))  (CR/rasterizer_impl.cu:355, :378-386).
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace ed3 {

size_t scan_temp_bytes(int P)
{
    size_t bytes = 0;
    uint32_t *p = nullptr;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, bytes, p, p, P > 0 ? P : 1);
    return bytes;
}

size_t sort_temp_bytes(int R)
{
    size_t bytes = 0;
    uint64_t *k = nullptr;
    uint32_t *v = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, v, v, R > 0 ? R : 1);
    return bytes;
}

bool run_scan(char *temp, size_t temp_bytes, const uint32_t *in, uint32_t *out, int P, hipStream_t s)
{
    return check_hip(hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, in, out, P, s), "InclusiveSum");
}

bool run_sort(char *temp, size_t temp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
              int R, int end_bit, hipStream_t s)
{
    if (R <= 0) return true;
    return check_hip(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, kin, kout, vin, vout, R, 0, end_bit, s),
                     "SortPairs");
}

}  // namespace ed3
