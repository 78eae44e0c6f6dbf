// sortscan.hip -- the two library primitives of the map-side pipeline: a stable
// LSD radix sort of (cell key, point index) pairs and an exclusive prefix sum,
// both from rocPRIM through the hipCUB front end.  They run once per map update /
// once per increment, never inside the ICP iteration loop.  Isolated in their own
// translation unit because the headers are slow to compile.
#include <cstdlib>
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>
#include "../velo_internal.hpp"

namespace velo {

// rocPRIM sorts up to 2^20 items by a block sort + log2(n / block) merge passes: 12 launches of ~5 us for the 133 k
// returns of a frame's decode, 22 for the 550 k points of an entering tile column -- all latency.  The radix path
// (Onesweep: histogram + scan + one sweep per 8 key bits) is 3 launches for the decode's 7-bit keys and 6 for the
// map's 27-bit ones; it is the one used from 32 k items up.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;

// temp == nullptr: size query (temp_bytes is written).  Stable: equal keys keep
// their input order, which is what makes the map's sorted index deterministic.
hipError_t sort_pairs(void* temp, size_t& temp_bytes, const uint32_t* k_in, uint32_t* k_out,
                      const uint32_t* v_in, uint32_t* v_out, size_t n, int end_bit, hipStream_t s)
{
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 32) end_bit = 32;
    static const bool merge_path = std::getenv("VELO_SORT_MERGE") != nullptr;  // (A/B: rocPRIM's default choice)
    if (merge_path)
        return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k_in, k_out, v_in, v_out, (int)n, 0, end_bit, s);
    return rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, k_in, k_out, v_in, v_out, n, 0u, (unsigned)end_bit, s);
}

// 64-bit keys (voxel coordinates on the unbounded grid: sparse insertion)
hipError_t sort_pairs64(void* temp, size_t& temp_bytes, const uint64_t* k_in, uint64_t* k_out,
                        const uint32_t* v_in, uint32_t* v_out, size_t n, hipStream_t s)
{
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k_in, k_out, v_in, v_out, (int)n, 0, 63, s);
}

hipError_t exclusive_scan_u32(void* temp, size_t& temp_bytes, const uint32_t* in, uint32_t* out,
                              size_t n, hipStream_t s)
{
    return hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, (int)n, s);
}

}  // namespace velo
