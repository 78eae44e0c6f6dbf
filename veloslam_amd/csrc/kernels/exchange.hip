// exchange.hip -- the two pack steps either side of the padded RCCL all-gather of the accepted
// map increments (capi.cpp velo_exchange_increments).  The reference is one process
// on one CPU: no counterpart.
//
//   send side:  three SoA arrays of n points -> one block [x | y | z], each axis padded to `pad`
//               floats (zero filled), i.e. what every rank contributes to the all-gather;
//   recv side:  W such blocks (rank r at recv + r*3*pad) -> ox/oy/oz holding the points of all
//               ranks back to back IN RANK ORDER (rank r at off[r] = sum of the counts below r).
//
// Both are pure streams (4 B per lane, coalesced on both sides: consecutive output indices of
// one rank read consecutive input floats).  The offsets travel by value in the kernel argument
// block (<= 65 words), so a call never reads host memory that a later call may rewrite.
#include "../velo_internal.hpp"

namespace velo {

__global__ __launch_bounds__(256) void k_pack_send(const float* __restrict__ x, const float* __restrict__ y,
                                                   const float* __restrict__ z, uint32_t n, uint32_t pad,
                                                   float* __restrict__ send)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= pad) return;
    const bool live = i < n;
    send[i] = live ? x[i] : 0.0f;
    send[(size_t)pad + i] = live ? y[i] : 0.0f;
    send[2 * (size_t)pad + i] = live ? z[i] : 0.0f;
}

__global__ __launch_bounds__(256) void k_pack_rank_blocks(const float* __restrict__ recv, RankOffsets ro,
                                                          uint32_t pad, float* __restrict__ ox,
                                                          float* __restrict__ oy, float* __restrict__ oz)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= ro.off[ro.world]) return;
    // rank of output index j: the last r with off[r] <= j (empty ranks share an offset with their
    // successor and are skipped by the strict comparison); W <= 64, wave-uniform for all but the
    // wavefronts that straddle a boundary
    int r = 0;
    for (int k = 1; k < ro.world; ++k) r += (ro.off[k] <= j) ? 1 : 0;
    const size_t src = (size_t)r * 3 * pad + (j - ro.off[r]);
    ox[j] = recv[src];
    oy[j] = recv[src + pad];
    oz[j] = recv[src + 2 * (size_t)pad];
}

hipError_t launch_pack_send(const float* x, const float* y, const float* z, uint32_t n, uint32_t pad,
                            float* send, hipStream_t s)
{
    if (!pad) return hipSuccess;
    hipLaunchKernelGGL(k_pack_send, dim3((pad + 255) / 256), dim3(256), 0, s, x, y, z, n, pad, send);
    return hipGetLastError();
}

hipError_t launch_pack_rank_blocks(const float* recv, const RankOffsets& ro, uint32_t pad, float* ox,
                                   float* oy, float* oz, hipStream_t s)
{
    const uint32_t total = ro.off[ro.world];
    if (!total) return hipSuccess;
    hipLaunchKernelGGL(k_pack_rank_blocks, dim3((total + 255) / 256), dim3(256), 0, s, recv, ro, pad, ox, oy, oz);
    return hipGetLastError();
}

}  // namespace velo
