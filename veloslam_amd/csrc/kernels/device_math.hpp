// device_math.hpp -- arithmetic shared by the gfx950 kernels.  Every expression
// that has to agree bit-for-bit with the CPU specification (DESIGN.md "ICP
// semantics") is written with explicit fma()/fmaf(); the translation units are
// compiled with -ffp-contract=off so nothing else is fused.
#pragma once
#include <hip/hip_runtime.h>
#include "../velo_internal.hpp"

namespace velo {

__device__ __forceinline__ int cell_coord(float p, float o, float inv_h, int dim)
{
    float f = floorf((p - o) * inv_h);
    // clamp before float->int; outside [-2, dim+1] no neighbour cell is in range
    f = (f >= -2.0f) ? f : -2.0f;
    f = (f > (float)(dim + 1)) ? (float)(dim + 1) : f;
    return (int)f;
}

// p' = T p in fp64, fixed fma nesting (innermost: translation)
__device__ __forceinline__ void xform(const double* __restrict__ T, float x, float y, float z,
                                      double& px, double& py, double& pz)
{
    const double dx = (double)x, dy = (double)y, dz = (double)z;
    px = fma(T[0], dx, fma(T[1], dy, fma(T[2], dz, T[3])));
    py = fma(T[4], dx, fma(T[5], dy, fma(T[6], dz, T[7])));
    pz = fma(T[8], dx, fma(T[9], dy, fma(T[10], dz, T[11])));
}

__device__ __forceinline__ float dist2(float4 c, float qx, float qy, float qz)
{
    const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// order-preserving float <-> unsigned map (for atomic min/max on floats)
__device__ __forceinline__ unsigned enc_f32(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float dec_f32(unsigned e)
{
    unsigned u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

}  // namespace velo
