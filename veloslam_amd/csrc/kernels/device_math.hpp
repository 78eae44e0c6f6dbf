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

// ---- fine-cell lookups: dense prefix table or sparse hash (MapView) ----------------------
__device__ __forceinline__ uint32_t cell_slot(const MapView& mv, uint32_t key)
{
    // multiplicative hash, then a multiply-shift range reduction: any capacity, no division
    return (uint32_t)(((unsigned long long)(key * 0x9E3779B1u) * mv.hash_cap) >> 32);
}
// v / S for a 32-bit v by the 64-bit reciprocal ceil(2^64 / S): floor((v * magic) / 2^64), exact (the excess is below
// 2^-32 and a fraction of v / S is at most (S - 1) / S); two multiplies instead of a run-time division
__device__ __forceinline__ uint32_t div_S(const MapView& mv, uint32_t v)
{
    if (mv.S == 1) return v;   // (uniform)
    const uint32_t mh = (uint32_t)(mv.s_magic >> 32), ml = (uint32_t)mv.s_magic;
    const unsigned long long t = (unsigned long long)v * mh + __umulhi(v, ml);
    return (uint32_t)(t >> 32);
}

// One ROW PIECE (the S fine cells of a voxel along a fine row) of the sparse table.  found = the piece holds points.
struct Piece {
    int start;        // first sorted index of the piece
    uint32_t o[4];    // o[0] = o1 | o2 << 16, ... : points of the piece in sub-cells < c, c = 1 .. S
};
__device__ __forceinline__ bool piece_find(const MapView& mv, uint32_t pkey, Piece& p)
{
    uint32_t h = cell_slot(mv, pkey);
    for (;;) {
        const int4 e = mv.hash[(size_t)h * mv.hash_stride];
        if ((uint32_t)e.x == pkey) {
            p.start = e.y;
            p.o[0] = (uint32_t)e.z;
            p.o[1] = (uint32_t)e.w;
            p.o[2] = p.o[3] = 0;
            if (mv.hash_stride > 1) {
                const int4 f = mv.hash[(size_t)h * mv.hash_stride + 1];
                p.o[2] = (uint32_t)f.x;
                p.o[3] = (uint32_t)f.y;
            }
            return true;
        }
        if ((uint32_t)e.x == 0xffffffffu) return false;
        h = h + 1 == mv.hash_cap ? 0u : h + 1;
    }
}
// points of the piece in sub-cells < c (c = 0 .. S)
__device__ __forceinline__ int piece_off(const Piece& p, int c)
{
    if (c == 0) return 0;
    const uint32_t w = p.o[(c - 1) >> 1];
    return (int)(((c - 1) & 1) ? (w >> 16) : (w & 0xffffu));
}

// occupied fine cell `key` -> its index range; false = the cell is empty
__device__ __forceinline__ bool cell_find(const MapView& mv, uint32_t key, int& start, int& end)
{
    const uint32_t pk = div_S(mv, key);
    const int c = (int)(key - pk * (uint32_t)mv.S);
    Piece p;
    if (!piece_find(mv, pk, p)) return false;
    start = p.start + piece_off(p, c);
    end = p.start + piece_off(p, c + 1);
    return end > start;
}

// index range [jlo, jhi) of the fine cells x0..x1 (inclusive, x0 <= x1, both inside the row) of
// the row whose first cell has key `row`.  Cells of a row are consecutive in the sorted order,
// so in sparse mode the range runs from the first occupied cell's start to the last one's end:
// one probe per VOXEL the window touches (a 3-cell window: two at most).
template <bool HASH>
__device__ __forceinline__ bool row_range(const MapView& mv, size_t row, int x0, int x1, int& jlo,
                                          int& jhi)
{
    if (!HASH) {
        jlo = mv.cell_start[row + (size_t)x0];
        jhi = mv.cell_start[row + (size_t)x1 + 1];
        return jhi > jlo;
    }
    const uint32_t base = div_S(mv, (uint32_t)row);   // (row is a multiple of fx = nx * S: exact) = row index * nx
    const int S = mv.S;
    const int va = (int)div_S(mv, (uint32_t)x0), vb = (int)div_S(mv, (uint32_t)x1);
    bool any = false;
    for (int v = va; v <= vb; ++v) {
        Piece p;
        if (!piece_find(mv, base + (uint32_t)v, p)) continue;
        const int c0 = max(x0 - v * S, 0), c1 = min(x1 - v * S, S - 1);
        const int a = p.start + piece_off(p, c0), b = p.start + piece_off(p, c1 + 1);
        if (b > a) {
            if (!any) jlo = a;
            jhi = b;
            any = true;
        }
    }
    if (!any) jlo = jhi = 0;
    return any;
}

// the same with the row given as a 32-bit fine key (keys are < 2^32 by construction; modular
// arithmetic on the caller's side): one 64-bit address computation per lookup instead of the
// 64-bit multiplies a size_t row costs
template <bool HASH>
__device__ __forceinline__ bool row_range32(const MapView& mv, uint32_t row, int x0, int x1, int& jlo,
                                            int& jhi)
{
    if (!HASH) {
        jlo = mv.cell_start[row + (uint32_t)x0];
        jhi = mv.cell_start[row + (uint32_t)x1 + 1u];
        return jhi > jlo;
    }
    return row_range<true>(mv, (size_t)row, x0, x1, jlo, jhi);
}

// run-time form for the kernels outside the iteration loop
__device__ __forceinline__ bool row_range_rt(const MapView& mv, size_t row, int x0, int x1, int& jlo,
                                             int& jhi)
{
    return mv.cell_start ? row_range<false>(mv, row, x0, x1, jlo, jhi)
                         : row_range<true>(mv, row, x0, x1, jlo, jhi);
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
