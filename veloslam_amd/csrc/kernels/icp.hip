// icp.hip -- the per-iteration hot path on gfx950:
//   k_linearize     K2+K3 fused: SE(3) transform of the query, nearest neighbour
//                   over the 27 map cells, point-to-plane residual + Jacobian,
//                   block reduction of the 29 normal-equation sums
//   k_reduce_solve  fixed-order sum of the block partials, 6x6 LDLt, pose update
// plus the small kernels around them (compensation K1, frame cell keys, map
// increment).  Semantics: DESIGN.md "ICP semantics" == oracle/icp.c; the
// reference has no counterpart.  Bandwidth/latency-bound integer
// and fp32 work: no MFMA anywhere.
#include "device_math.hpp"
#include <cstdlib>

namespace velo {

// =============================================================== K1 compensate
// type_defs.h:160-166 semantics: ((m0*x + m1*y) + m2*z) + m3, each product
// rounded (the TU is built with -ffp-contract=off), one rounding to float.
__device__ __forceinline__ void apply_affine(const double* __restrict__ M, float x, float y,
                                             float z, float& ox, float& oy, float& oz)
{
    const double dx = (double)x, dy = (double)y, dz = (double)z;
    ox = (float)(M[0] * dx + M[1] * dy + M[2] * dz + M[3]);
    oy = (float)(M[4] * dx + M[5] * dy + M[6] * dz + M[7]);
    oz = (float)(M[8] * dx + M[9] * dy + M[10] * dz + M[11]);
}

// 4 points per lane: 16-byte coalesced SoA loads/stores (26 B/point of traffic)
__global__ __launch_bounds__(256) void k_compensate_v4(
    const float4* __restrict__ x4, const float4* __restrict__ y4, const float4* __restrict__ z4,
    const ushort4* __restrict__ p4, size_t n4, const double* __restrict__ tab, unsigned n_pkt,
    float4* __restrict__ ox4, float4* __restrict__ oy4, float4* __restrict__ oz4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = x4[i], y = y4[i], z = z4[i];
        const ushort4 p = p4[i];
        const unsigned k[4] = {min((unsigned)p.x, n_pkt - 1), min((unsigned)p.y, n_pkt - 1),
                               min((unsigned)p.z, n_pkt - 1), min((unsigned)p.w, n_pkt - 1)};
        const float xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w},
                    zs[4] = {z.x, z.y, z.z, z.w};
        float rx[4], ry[4], rz[4];
        if (k[0] == k[3] && k[1] == k[2] && k[0] == k[1]) {
            // four consecutive returns of a beam almost always share their packet: fetch its
            // 3x4 matrix once (the table loads, not HBM, were what bounded this kernel)
            double M[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) M[e] = tab[12 * (size_t)k[0] + e];
#pragma unroll
            for (int j = 0; j < 4; ++j) apply_affine(M, xs[j], ys[j], zs[j], rx[j], ry[j], rz[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                apply_affine(tab + 12 * (size_t)k[j], xs[j], ys[j], zs[j], rx[j], ry[j], rz[j]);
        }
        ox4[i] = make_float4(rx[0], rx[1], rx[2], rx[3]);
        oy4[i] = make_float4(ry[0], ry[1], ry[2], ry[3]);
        oz4[i] = make_float4(rz[0], rz[1], rz[2], rz[3]);
    }
}

// The same with the table staged per WAVEFRONT: the 256 points of a wavefront are consecutive
// returns of a beam, i.e. ~43 consecutive packets.  Fetched per lane, their 96-byte matrices are
// six 16-byte loads at a 96-byte stride -- 48 cache lines per instruction, and the L1 data path,
// not HBM, is what the kernel waits on (7.4 M points: 31 us without any table fetch, 48 with).
// Here the wavefront finds its
// packet range (two shuffled reductions), loads those matrices ONCE as contiguous 16-byte chunks
// (8 lines per instruction) into its own LDS slice and every lane reads its matrix from there.
// A wavefront that spans more than kK1Span packets (a beam boundary: 1 in 7) fetches per lane
// as before.  No workgroup barrier: nothing is shared between wavefronts.
constexpr int kK1Span = 64;
struct __attribute__((aligned(16))) K1Lds {
    double m[kK1Span * 12];
};

__global__ __launch_bounds__(256) void k_compensate_v4l(
    const float4* __restrict__ x4, const float4* __restrict__ y4, const float4* __restrict__ z4,
    const ushort4* __restrict__ p4, size_t n4, const double* __restrict__ tab, unsigned n_pkt,
    float4* __restrict__ ox4, float4* __restrict__ oy4, float4* __restrict__ oz4)
{
    __shared__ K1Lds s_tab[4];
    const int lane = threadIdx.x & 63;
    K1Lds& L = s_tab[threadIdx.x >> 6];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n4;
    const size_t ic = live ? i : n4 - 1;
    const float4 x = x4[ic], y = y4[ic], z = z4[ic];
    const ushort4 p = p4[ic];
    const unsigned k[4] = {min((unsigned)p.x, n_pkt - 1), min((unsigned)p.y, n_pkt - 1),
                           min((unsigned)p.z, n_pkt - 1), min((unsigned)p.w, n_pkt - 1)};
    unsigned kmin = min(min(k[0], k[1]), min(k[2], k[3])), kmax = max(max(k[0], k[1]), max(k[2], k[3]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, off, 64));
    }
    const float xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w}, zs[4] = {z.x, z.y, z.z, z.w};
    float rx[4], ry[4], rz[4];
    if (kmax - kmin < (unsigned)kK1Span) {  // (uniform)
        const int chunks = (int)(kmax - kmin + 1) * 6;  // 16-byte pieces
        const double2* src = reinterpret_cast<const double2*>(tab + 12 * (size_t)kmin);
        double2* dst = reinterpret_cast<double2*>(L.m);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int c = lane + 64 * t;
            if (c < chunks) dst[c] = src[c];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (k[0] == k[3] && k[1] == k[2] && k[0] == k[1]) {
            double M[12];
            const double2* mp = reinterpret_cast<const double2*>(L.m + 12 * (k[0] - kmin));
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const double2 v = mp[e];
                M[2 * e] = v.x;
                M[2 * e + 1] = v.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) apply_affine(M, xs[j], ys[j], zs[j], rx[j], ry[j], rz[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                apply_affine(L.m + 12 * (k[j] - kmin), xs[j], ys[j], zs[j], rx[j], ry[j], rz[j]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    } else if (k[0] == k[3] && k[1] == k[2] && k[0] == k[1]) {
        double M[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) M[e] = tab[12 * (size_t)k[0] + e];
#pragma unroll
        for (int j = 0; j < 4; ++j) apply_affine(M, xs[j], ys[j], zs[j], rx[j], ry[j], rz[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            apply_affine(tab + 12 * (size_t)k[j], xs[j], ys[j], zs[j], rx[j], ry[j], rz[j]);
    }
    if (live) {
        ox4[i] = make_float4(rx[0], rx[1], rx[2], rx[3]);
        oy4[i] = make_float4(ry[0], ry[1], ry[2], ry[3]);
        oz4[i] = make_float4(rz[0], rz[1], rz[2], rz[3]);
    }
}

__global__ __launch_bounds__(256) void k_compensate_v1(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z,
    const uint16_t* __restrict__ pkt, size_t i0, size_t n, const double* __restrict__ tab,
    unsigned n_pkt, float* __restrict__ ox, float* __restrict__ oy, float* __restrict__ oz)
{
    for (size_t i = i0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const unsigned k = min((unsigned)pkt[i], n_pkt - 1);
        apply_affine(tab + 12 * (size_t)k, x[i], y[i], z[i], ox[i], oy[i], oz[i]);
    }
}

hipError_t launch_compensate(const float* x, const float* y, const float* z, const uint16_t* pkt,
                             size_t n, const double* T3x4, size_t n_pkt, float* ox, float* oy,
                             float* oz, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    size_t done = 0;
    if (al16(x) && al16(y) && al16(z) && al16(ox) && al16(oy) && al16(oz) &&
        (reinterpret_cast<uintptr_t>(pkt) & 7u) == 0 && n >= 4) {
        const size_t n4 = n / 4;
        if (al16(T3x4)) {
            const size_t gl = (n4 + 255) / 256;
            hipLaunchKernelGGL(k_compensate_v4l, dim3((unsigned)gl), dim3(256), 0, s, (const float4*)x,
                               (const float4*)y, (const float4*)z, (const ushort4*)pkt, n4, T3x4,
                               (unsigned)n_pkt, (float4*)ox, (float4*)oy, (float4*)oz);
        } else {
            size_t g = (n4 + 255) / 256;
            const int grid = (int)(g > 16384 ? 16384 : g);
            hipLaunchKernelGGL(k_compensate_v4, dim3(grid), dim3(256), 0, s, (const float4*)x,
                               (const float4*)y, (const float4*)z, (const ushort4*)pkt, n4, T3x4,
                               (unsigned)n_pkt, (float4*)ox, (float4*)oy, (float4*)oz);
        }
        done = n4 * 4;
    }
    if (done < n) {
        size_t g = (n - done + 255) / 256;
        const int grid = (int)(g > 2048 ? 2048 : g);
        hipLaunchKernelGGL(k_compensate_v1, dim3(grid), dim3(256), 0, s, x, y, z, pkt, done, n,
                           T3x4, (unsigned)n_pkt, ox, oy, oz);
    }
    return hipGetLastError();
}

// ============================================================ K2+K3 linearise
// Column k of the 29 sums is v[IA[k]] * v[IB[k]] with v = {J0..J5, r, valid}.
__constant__ unsigned char c_ia[32] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3,
                                       3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6, 7, 7, 7, 7};
__constant__ unsigned char c_ib[32] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3,
                                       4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7};

// Search statistics and byte accounting: counted only by the STATS instantiation of the kernel
// (velo_set_stats), the production instantiation carries none of it.
// [0] live queries, [1] certified (no search), [2] searched, [3] empty-neighbourhood skips,
// [4] stage-A final, [5] stragglers searched per lane, [6] stage-B stragglers (all),
// [7] valid pairs, [8] bytes requested from memory (loads + stores, see Tally),
// [9] candidate points examined, [10] fine-table requests, [11] launches, [12] the query-side
// part of [8] (per-query stream + per-workgroup rows: everything that is not a map-side gather)
__device__ unsigned long long g_lin_stats[16];

// Bytes this lane asked the memory system for: the roofline numerator of the pruned kernel
// (bench.py "roofline.achieved" = these bytes / launch time).  16 B per candidate / hinted /
// matched point, 16 B per normal, 16 B (stage A) or 2 x 4 B (stage B) per fine-table request,
// 12 B query + 4 B hint + 4 B certificate read, 4 + 4 B written back, 1 B voxel flag.
template <bool STATS>
struct Tally {
    unsigned bytes = 0, cand = 0, tab = 0, qbytes = 0;
    __device__ __forceinline__ void add(unsigned b)
    {
        if constexpr (STATS) bytes += b;
    }
    // query side: the per-query stream (coordinates, hint, certificate in; hint, certificate
    // out) and the per-workgroup item / pose / partial row -- bytes no cache can save
    __device__ __forceinline__ void addq(unsigned b)
    {
        if constexpr (STATS) {
            bytes += b;
            qbytes += b;
        }
    }
    __device__ __forceinline__ void candidates(unsigned n)
    {
        if constexpr (STATS) {
            cand += n;
            bytes += 16u * n;
        }
    }
    __device__ __forceinline__ void table(unsigned n_req, unsigned bytes_each)
    {
        if constexpr (STATS) {
            tab += n_req;
            bytes += n_req * bytes_each;
        }
    }
};

__device__ __forceinline__ void stat_add(int slot, unsigned v)
{
    unsigned t = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&g_lin_stats[slot], (unsigned long long)t);
}
#define VELO_COUNT(i, pred)                                                                     \
    do {                                                                                        \
        if constexpr (STATS) {                                                                  \
            const unsigned long long m__ = __ballot(pred);                                      \
            if (m__ && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m__) - 1))           \
                atomicAdd(&g_lin_stats[i], (unsigned long long)__popcll(m__));                  \
        }                                                                                       \
    } while (0)

// ---- fine-grid geometry of a query --------------------------------------------------
// The map is sorted by FINE cell (S sub-cells per voxel edge, DESIGN.md "ICP semantics");
// one row of fine cells (fixed Fy,Fz) is one contiguous index range of the map.
struct QueryCell {
    int cx, cy, cz;     // voxel (clamped to [-2, dim+1])
    int Fx, Fy, Fz;     // fine cell (meaningful when the voxel is within [-1, dim])
    float tx, ty, tz;   // position inside the fine cell, in fine-cell units [0,1)
    bool near;          // voxel within one voxel of the grid: candidates may exist
};

__device__ __forceinline__ void query_axis(float q, float o, float inv_h, int dim, int S, int& c,
                                           int& F, float& t)
{
    const float u = (q - o) * inv_h;
    c = cell_coord(q, o, inv_h, dim);
    const float fr = (u - (float)c) * (float)S;  // exact for power-of-two S
    float sf = floorf(fr);
    sf = fminf(fmaxf(sf, 0.0f), (float)(S - 1));
    t = fr - sf;
    F = c * S + (int)sf;
}

__device__ __forceinline__ QueryCell locate(const MapView& mv, float qx, float qy, float qz)
{
    QueryCell g;
    query_axis(qx, mv.ox, mv.inv_h, mv.nx, mv.S, g.cx, g.Fx, g.tx);
    query_axis(qy, mv.oy, mv.inv_h, mv.ny, mv.S, g.cy, g.Fy, g.ty);
    query_axis(qz, mv.oz, mv.inv_h, mv.nz, mv.S, g.cz, g.Fz, g.tz);
    g.near = g.cx >= -1 && g.cx <= mv.nx && g.cy >= -1 && g.cy <= mv.ny && g.cz >= -1 &&
             g.cz <= mv.nz;
    return g;
}

// ---- variant 0: exhaustive scan ------------------------------------------------------
// All points of the 27 voxels around the query, in ascending sorted index (9*S*S fine
// rows); strict '<' keeps the lowest index among equal distances.  This is the oracle's
// definition executed literally; it is the validation kernel, not the fast path.
__device__ __forceinline__ void nearest_scan(const MapView& mv, float qx, float qy, float qz,
                                             float& bd, int& bj)
{
    const QueryCell g = locate(mv, qx, qy, qz);
    bd = INFINITY;
    bj = -1;
    const int vx0 = max(g.cx - 1, 0), vx1 = min(g.cx + 1, mv.nx - 1);
    const int vy0 = max(g.cy - 1, 0), vy1 = min(g.cy + 1, mv.ny - 1);
    const int vz0 = max(g.cz - 1, 0), vz1 = min(g.cz + 1, mv.nz - 1);
    if (vx0 > vx1 || vy0 > vy1 || vz0 > vz1) return;
    const int S = mv.S;
#pragma unroll 1
    for (int fz = vz0 * S; fz < (vz1 + 1) * S; ++fz) {
#pragma unroll 1
        for (int fy = vy0 * S; fy < (vy1 + 1) * S; ++fy) {
            const size_t row = ((size_t)fz * mv.fy + fy) * mv.fx;
            int j0, j1;
            if (!row_range_rt(mv, row, vx0 * S, (vx1 + 1) * S - 1, j0, j1)) continue;
#pragma unroll 4
            for (int j = j0; j < j1; ++j) {
                const float d2 = dist2(mv.pts[j], qx, qy, qz);
                if (d2 < bd) {
                    bd = d2;
                    bj = j;
                }
            }
        }
    }
}

// ---- variant 1: exact ball search on the fine grid ----------------------------------------
// Same winner as nearest_scan (lexicographic minimum of (d2, sorted index), accepted iff
// d2 <= dmax2) with ~an order of magnitude fewer distance evaluations:
//   stage A  every lane scans the 3x3x3 block of FINE cells around its query: <= 9 index
//            ranges, staged in LDS ([slot][thread]) and walked in one flattened loop, four
//            candidate loads in flight, DESCENDING index with '<=' (the last = lowest index
//            among equal distances wins -- the oracle's tie rule).  If the winner is closer
//            than the distance from the query to the faces of that block (shrunk by a margin
//            that covers the float rounding of the cell-assignment expression), nothing
//            outside the block can beat or tie it: done.  After convergence that is ~98 %
//            of the queries.
//   stage B  the rest are compacted per workgroup (LDS list) and re-searched over the whole
//            ball of radius sqrt(min(best, dmax2)): rows pruned by a conservative bound, the
//            x-extent of each row cut to the ball, descending index, '<='.  Compaction keeps
//            a handful of stragglers from stalling every wavefront of the workgroup.
// Square roots of the search are BOUNDS, never results (distances are compared squared): radii searched, radii
// certified, row half-widths -- each used with an explicit margin (x 1.000001 + 1e-7 upwards, x 0.999999 - 1e-6
// downwards, 1e-5 relative on windows).  sqrtf() is the correctly rounded form, ~14 VALU instructions on gfx950
// (v_sqrt_f32 + a Newton correction + denormal scaling); the bare instruction is good to 1 ulp (1.2e-7 relative;
// a denormal argument may come back as 0, which the absolute terms of the margins cover), an eighth of the
// smallest margin.  Two of them sat on the path of EVERY certified query (|q - hint|, |q - c|): 26 of the 279
// VALU instructions of a converged wavefront-round (round 4).
#ifndef VELO_FAST_SQRT
#define VELO_FAST_SQRT 1
#endif
__device__ __forceinline__ float bsqrt(float x)
{
#if VELO_FAST_SQRT
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}

constexpr int kMaxRanges = 9;
constexpr int kLatItems = 2048;  // launches below this many workgroups use the latency kernel unless the
                                 // caller says which (registrations do: kLatQueries, plan_frames)
#ifndef VELO_CERT_SLACK
#define VELO_CERT_SLACK 0.015f  // measured: 0.005-0.02 within 1%, 0.05 +3%, 0.10 +5% (batch); dense single frame 0.68 vs 0.73 ms
#endif
constexpr float kCertSlack = VELO_CERT_SLACK;  // metres searched beyond the hinted point (tuning only)
#ifndef VELO_OWN_FIRST_S
#define VELO_OWN_FIRST_S 4  // sub-division from which stage A probes the query's own fine cell first (99 = never)
#endif
#ifndef VELO_OWN_FIRST_LAT
#define VELO_OWN_FIRST_LAT 0  // 1: also in the latency kernel (W >= 8) -- measured: stream ICP 0.59 -> 0.63 ms, off
#endif
#ifndef VELO_OWN_FIRST_R
#define VELO_OWN_FIRST_R 1.0f
#endif
#ifndef VELO_BALL_PROBE
#define VELO_BALL_PROBE 1
#endif
#ifndef VELO_UBM
#define VELO_UBM 0
#endif
#ifndef VELO_REXFORM
#define VELO_REXFORM 1  // 0: keep p' live across the search (spills 20 B at 72 registers: converged launch 68 -> 78 us)
#endif
#ifndef VELO_WALK_W
#define VELO_WALK_W 4
#endif
#ifndef VELO_SEED_BOUND
#define VELO_SEED_BOUND 0  // 1: block-less stragglers take a first bound from the sorted-order neighbours of their cell (A/B)
#endif
#ifndef VELO_WALK_W_LAT
#define VELO_WALK_W_LAT 8  // latency kernel: twice the candidate loads in flight per trip
#endif


// per WAVEFRONT: [slot][lane].  Nothing in the search or in the reduction tile below is
// shared between wavefronts, so the query loop needs no workgroup barrier at all.
struct SearchLds {
    int hi[kMaxRanges][64];
    int lo[kMaxRanges][64];
};

// Walk this lane's index ranges (descending, '<=').  W candidates per trip: W loads are in
// flight together; indices are clamped to the range start (a candidate evaluated twice is
// harmless under '<=').  (Carrying the winner's coordinates through the walk to save the
// later re-fetch was measured: the extra selects and registers cost more than the fetch.)
template <int W, bool STATS, bool PAIR = false>
__device__ __forceinline__ void walk_ranges(const MapView& mv, float qx, float qy, float qz,
                                            SearchLds& L, int tid, int nr, float& bd, int& bj,
                                            float& sd, Tally<STATS>& tl, int* sj_io = nullptr, float* td_io = nullptr)
{
    static_assert(!PAIR || W >= 8, "the runner-up is tracked in the latency kernel's branch-free walk only");
    // PAIR (round 6): also the INDEX of the second-smallest candidate and the third-smallest distance -- what a pair
    // certificate needs ({winner, runner-up} are the only map points within sqrt(td)).
    int sj = -1;
    float td = sd;
    if constexpr (PAIR) {
        sj = *sj_io;
        td = *td_io;
    }
    // W candidates per trip drawn ACROSS ranges: a lane with three short ranges needs one or
    // two trips instead of three (measured 54 vs 57.5 us per launch against one range per
    // trip).  Slots past the end repeat the last index and are masked.  sd tracks the
    // second-smallest distance seen (for the uniqueness certificate, see k_linearize).
    int k = 0, j = 0, lo = 0, last = 0;
    bool more = nr > 0;
    while (more) {
        int jj[W];
        float4 c[W];
        bool live[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (j <= lo && k < nr) {
                j = L.hi[k][tid];
                lo = L.lo[k][tid];
                ++k;
            }
            live[u] = j > lo;
            if (live[u]) tl.candidates(1);
            if (live[u]) last = --j;
            jj[u] = last;
            c[u] = mv.pts[(unsigned)last];  // (never negative: no sign extension)
        }
#pragma unroll
        for (int u = 0; u < W; ++u) {
            // bd <= sd always, so the new second-best (the old best if it is beaten, else
            // min(sd, e)) is the median of the three: one instruction instead of three.
            // Latency kernel: branch-free, a dead slot is a candidate at infinity.
            if constexpr (W >= 8) {  // (latency kernel: registers to spare)
                const float e = live[u] ? dist2(c[u], qx, qy, qz) : INFINITY;
                const bool better = e <= bd;
                if constexpr (PAIR) {
                    const bool second = e < sd;                 // enters the best two
                    td = second ? sd : fminf(td, e);            // the old second drops to third, or e competes for third
                    sj = better ? bj : (second ? jj[u] : sj);
                }
                sd = __builtin_amdgcn_fmed3f(bd, e, sd);
                bj = better ? jj[u] : bj;
                bd = better ? e : bd;
            } else {
                const float e = dist2(c[u], qx, qy, qz);
                if (live[u]) {  // (the branch-free form spills in the throughput kernel: +3 %)
                    sd = __builtin_amdgcn_fmed3f(bd, e, sd);
                    const bool better = e <= bd;
                    bj = better ? jj[u] : bj;
                    bd = better ? e : bd;
                }
            }
        }
        more = (j > lo) || (k < nr);
    }
    if constexpr (PAIR) {
        *sj_io = sj;
        *td_io = td;
    }
}

// four consecutive entries of the fine-cell table with one 16-byte request (the table is
// only 4-byte aligned at an arbitrary cell: gfx950 global loads do not need more)
struct __attribute__((packed, aligned(4))) Int4U {
    int v[4];
};

// stage A; returns true when the result is final.
// ub0 (<= dmax2) is an upper bound of the winning distance known before the search: dmax2,
// or the distance to last iteration's correspondence.  Rows and cells of the 3x3x3 block that
// lie further than sqrt(ub0) from the query are not even looked up; with a good hint a query
// costs one table request and a couple of candidates instead of nine and ~14.
// ABL (timing ablations only, results are wrong): 1 = treat stage A as final, 2 = also skip
// the candidate walk, 3 = also skip the fine-table loads
constexpr int kFinal = 0, kStraggler = 1;  // stage A outcome

// stage A epilogue: is the block result final, and what radius does it certify
__device__ __forceinline__ int finish_block(float bd, float sd, float gr, float& cert)
{
    const bool final = bd <= gr * gr * 0.99999f;
    // second-best scanned / pruned-cell bound / block faces, rounded down
    cert = final ? fmaxf(fminf(bsqrt(sd) * 0.999999f, gr) - 1e-6f, 0.0f) : 0.0f;
    return final ? kFinal : kStraggler;
}

// Round 3: the row loop below used to cost ~60 VALU instructions per row, a third of them
// quarter-rate 64-bit multiplies of the row address ((zz * fy + yy) * fx per row) -- with nine rows
// unpruned that was the largest single item of an unhinted launch.  Now: one 32-bit row index per
// query (fine keys are < 2^32 by construction, so modular uint32 arithmetic is exact, also for a
// query one cell outside the grid), +-fx / +-fx*fy per row from scalar registers, squared face
// distances per axis computed once, and the pruning thresholds folded into one compare per test
// (`x > ub0 * 1.00002f` prunes a subset of what `x * 0.99999f > ub0` pruned: still conservative,
// the results -- not the pruning decisions -- are what is exact).
template <int ABL, bool STATS, int W, bool HASH, bool PAIR = false>
__device__ __forceinline__ int search_block(const MapView& mv, const QueryCell& g, float qx,
                                            float qy, float qz, float ub0, SearchLds& L,
                                            int tid, float& bd, int& bj, float& cert,
                                            float& gr_out, Tally<STATS>& tl, int* sj_out = nullptr, float* cert3 = nullptr)
{
    bd = ub0;
    bj = -1;
    cert = 0.0f;  // radius (m) around the query inside which the winner is the only map point
    gr_out = 0.0f;
    if constexpr (PAIR) {
        *sj_out = -1;   // runner-up of a FINAL stage A ...
        *cert3 = 0.0f;  // ... and the radius inside which winner and runner-up are the only map points
    }
    if (!g.near) return kFinal;  // no voxel of the 27 exists: no candidates at all
    // (ub0 may be tightened below by the own-cell probe; bd follows it before the walk)
    const float hf = mv.h / (float)mv.S;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    // conservative distances from the query to the faces of its own fine cell, squared;
    // index 0 / 1 / 2 = offset -1 / 0 / +1 along the axis
    const float lox = fmaxf(g.tx * hf - mg, 0.0f), hix = fmaxf((1.0f - g.tx) * hf - mg, 0.0f);
    const float loy = fmaxf(g.ty * hf - mg, 0.0f), hiy = fmaxf((1.0f - g.ty) * hf - mg, 0.0f);
    const float loz = fmaxf(g.tz * hf - mg, 0.0f), hiz = fmaxf((1.0f - g.tz) * hf - mg, 0.0f);
    const float lox2 = lox * lox, hix2 = hix * hix;
    const float by2[3] = {loy * loy, 0.0f, hiy * hiy};
    const float bz2[3] = {loz * loz, 0.0f, hiz * hiz};
    // guaranteed radius of the 3x3x3 block: one fine cell plus the distance to the nearer
    // face of the query's own fine cell, per axis; shrunk for rounding.  Whatever was pruned
    // inside the block is further than sqrt(ub0) >= sqrt(bd).  (Computed before the row loop: the
    // fractional coordinates need not stay live across it.)
    const float tmin = fminf(fminf(fminf(g.tx, 1.0f - g.tx), fminf(g.ty, 1.0f - g.ty)),
                             fminf(g.tz, 1.0f - g.tz));
    const float gr = fmaxf(hf * (1.0f + tmin) - mg, 0.0f);
    gr_out = gr;
#if VELO_UBM
    const float ubm = ub0 * 1.00002f;  // x > ubm  =>  x * 0.99999f > ub0 (the bound, with its margin)
#define VELO_PRUNED(x) ((x) > ubm)
#else
#define VELO_PRUNED(x) ((x) * 0.99999f > ub0)
#endif
    const int xb = max(g.Fx - 1, 0);  // first table entry fetched for a row
    // 32-bit row arithmetic (mod 2^32): key of cell (Fx = 0, Fy + dy, Fz + dz)
    const uint32_t sy = (uint32_t)mv.fx, sz = (uint32_t)mv.fx * (uint32_t)mv.fy;
    const uint32_t rowc = ((uint32_t)g.Fz * (uint32_t)mv.fy + (uint32_t)g.Fy) * (uint32_t)mv.fx;
#if VELO_OWN_FIRST_S < 99
    // Dense maps (fine cells holding several points each: S >= VELO_OWN_FIRST_S): look into the
    // query's OWN fine cell first.  A point found there bounds the winner, and with that bound most
    // of the 26 cells around are pruned before their candidates are fetched -- on a 10 M-point map
    // an unhinted or loosely hinted query otherwise examines 80-100 candidates.  A bound only (with
    // the certificate slack added, as for a hint): the walk below still finds the winner itself.
    // (only while the bound is loose -- a radius beyond VELO_OWN_FIRST_R fine cells: a tightly
    // hinted query prunes the block by itself, and the probe would cost it candidates and slack)
    if ((VELO_OWN_FIRST_LAT || W < 8) && mv.S >= VELO_OWN_FIRST_S && ub0 > (VELO_OWN_FIRST_R * VELO_OWN_FIRST_R) * hf * hf &&
        (unsigned)g.Fx < (unsigned)mv.fx && (unsigned)g.Fy < (unsigned)mv.fy && (unsigned)g.Fz < (unsigned)mv.fz) {
        int jlo, jhi;
        row_range32<HASH>(mv, rowc, g.Fx, g.Fx, jlo, jhi);
        tl.table(2, 4);
        float pb = ub0;
        for (int j = jhi - 1; j >= jlo; --j) pb = fminf(pb, dist2(mv.pts[(unsigned)j], qx, qy, qz));
        tl.candidates((unsigned)max(jhi - jlo, 0));
        if (pb < ub0) {
            const float rs = bsqrt(pb) * 1.000001f + 1e-7f + kCertSlack;
            ub0 = fminf(ub0, rs * rs * 1.00001f);
        }
    }
#endif
    int nr = 0;
    if (g.Fx + 1 >= 0 && g.Fx - 1 < mv.fx && ABL < 3) {
#pragma unroll
        for (int dz = 1; dz >= -1; --dz) {
            const bool zok = (unsigned)(g.Fz + dz) < (unsigned)mv.fz;
#pragma unroll
            for (int dy = 1; dy >= -1; --dy) {
                const float rb2 = bz2[dz + 1] + by2[dy + 1];
                if (!zok || (unsigned)(g.Fy + dy) >= (unsigned)mv.fy || VELO_PRUNED(rb2)) continue;
                // cells of this row inside the ball: own cell always, neighbours by their bound
                int x0 = g.Fx - (VELO_PRUNED(lox2 + rb2) ? 0 : 1);
                int x1 = g.Fx + (VELO_PRUNED(hix2 + rb2) ? 0 : 1);
                x0 = max(x0, 0);
                x1 = min(x1, mv.fx - 1);
                if (x0 > x1) continue;
                const uint32_t row = rowc + (uint32_t)dz * sz + (uint32_t)dy * sy;
                int jlo, jhi;
                if constexpr (HASH) {
                    tl.table((unsigned)(x1 - x0 + 1), 16);
                    row_range<true>(mv, (size_t)row, x0, x1, jlo, jhi);
                } else {
                    const Int4U e = *reinterpret_cast<const Int4U*>(mv.cell_start + (row + (uint32_t)xb));
                    tl.table(1, 16);
                    const int a = x0 - xb, b = x1 + 1 - xb;  // a in {0,1}, b in {1,2,3}
                    jlo = a == 0 ? e.v[0] : e.v[1];
                    jhi = b == 3 ? e.v[3] : (b == 2 ? e.v[2] : e.v[1]);
                }
                if (jhi > jlo) {
                    L.hi[nr][tid] = jhi;
                    L.lo[nr][tid] = jlo;
                    ++nr;
                }
            }
        }
    }
#undef VELO_PRUNED
    if (ABL >= 2) nr = min(nr, 0);
    bd = ub0;
    float sd = ub0;  // everything not scanned inside the block is further than sqrt(ub0)
    if constexpr (PAIR) {
        int sj = -1;
        float td = ub0;  // (likewise the third)
        walk_ranges<W, STATS, true>(mv, qx, qy, qz, L, tid, nr, bd, bj, sd, tl, &sj, &td);
        const int st = finish_block(bd, sd, gr, cert);
        if (st == kFinal && bj >= 0 && sj >= 0) {
            // third-best scanned / pruned-cell bound / block faces, rounded down: as the uniqueness radius, one rank on
            *sj_out = sj;
            *cert3 = fmaxf(fminf(bsqrt(td) * 0.999999f, gr) - 1e-6f, 0.0f);
        }
        return st;
    } else {
        walk_ranges<W>(mv, qx, qy, qz, L, tid, nr, bd, bj, sd, tl);
        if (ABL >= 1) return kFinal;
        return finish_block(bd, sd, gr, cert);
    }
}

// Round 6, latency kernels: the whole wavefront scans EVERY candidate of one query -- all points of the 27 voxels around
// it, the oracle's candidate set literally -- and returns the smallest squared distance (INFINITY: no candidate at
// all).  For a query that has just been found WITHOUT a match this is what turns "nothing within d_max here" into a
// certificate with room to move: every candidate is at least that far from here, so at least that far minus delta
// from wherever the query goes inside the same voxel.  Rows of fine cells over the lanes, each row at the full width
// of the three voxels.
__device__ __forceinline__ float scan_block_wave(const MapView& mv, float qx, float qy, float qz, int lane)
{
    const QueryCell g = locate(mv, qx, qy, qz);
    float best = INFINITY;
    const int vx0 = max(g.cx - 1, 0), vx1 = min(g.cx + 1, mv.nx - 1);
    const int vy0 = max(g.cy - 1, 0), vy1 = min(g.cy + 1, mv.ny - 1);
    const int vz0 = max(g.cz - 1, 0), vz1 = min(g.cz + 1, mv.nz - 1);
    if (vx0 <= vx1 && vy0 <= vy1 && vz0 <= vz1) {
        const int S = mv.S;
        const int ny = (vy1 - vy0 + 1) * S, nz = (vz1 - vz0 + 1) * S;
        for (int r = lane; r < ny * nz; r += 64) {
            const int fz = vz0 * S + r / ny, fy = vy0 * S + r % ny;
            const size_t row = ((size_t)fz * mv.fy + fy) * mv.fx;
            int j0, j1;
            if (!row_range_rt(mv, row, vx0 * S, (vx1 + 1) * S - 1, j0, j1)) continue;
            for (int j = j0; j < j1; ++j) best = fminf(best, dist2(mv.pts[(unsigned)j], qx, qy, qz));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = fminf(best, __shfl_xor(best, off, 64));
    return best;
}

// stage B, per-lane form (used when most lanes of a wavefront are stragglers, i.e. the first
// iterations of a badly aligned frame): the whole ball of radius sqrt(ub), rows visited in
// descending order in chunks of kMaxRanges.  Per chunk the table entries of all its rows are
// requested together, the surviving ranges staged in LDS and walked like stage A -- two
// memory round trips per nine rows instead of two per row.
// squared radius actually searched around a query whose best distance so far is sqrt(b): a
// little beyond it (so that the outcome certifies a radius, see linearize_body), never past one
// voxel (the row window ends there)
__device__ __forceinline__ float cover_of(float b, float h)
{
    const float r = bsqrt(b) * 1.000001f + 1e-7f + kCertSlack;
    return fminf(r * r * 1.00001f, h * h);
}

// certified radius after a ball search: second-best examined / radius covered / one voxel
__device__ __forceinline__ float ball_certificate(float sd, float cov, float h, float mg)
{
    return fmaxf(fminf(bsqrt(fminf(sd, cov)) * 0.999999f, h - 2.0f * mg) - 1e-6f, 0.0f);
}

// lower bound (rounded down) of the distance from the query to fine row offset d along one axis
__device__ __forceinline__ float axis_gap(int d, float t, float hf, float mg)
{
    return d == 0 ? 0.0f : fmaxf(((float)(abs(d) - 1) + (d > 0 ? 1.0f - t : t)) * hf - mg, 0.0f);
}

// rows of the (2R+1)^2 window a ball of squared radius b can reach (R <= S since b <= h^2)
__device__ __forceinline__ int ball_window(float b, float inv_hf, int S)
{
    return min(S, (int)floorf(bsqrt(b) * 1.00001f * inv_hf + 1.001f));
}

// index range of fine row (Fz+dz, Fy+dy) inside the ball of squared radius `bound` around the
// query (conservative: rounded outwards); clip1: only the cells Fx-1..Fx+1.  false = nothing.
template <bool STATS, bool HASH>
__device__ __forceinline__ bool ball_row(const MapView& mv, const QueryCell& g, int dz, int dy,
                                         float bound, float xf, float hf, float inv_hf, float mg,
                                         bool clip1, int& jlo, int& jhi, Tally<STATS>& tl)
{
    const int zz = g.Fz + dz, yy = g.Fy + dy;
    if (zz < 0 || zz >= mv.fz || yy < 0 || yy >= mv.fy) return false;
    const float bz = axis_gap(dz, g.tz, hf, mg), by = axis_gap(dy, g.ty, hf, mg);
    const float rb2 = (bz * bz + by * by) * 0.99999f;
    if (rb2 > bound) return false;
    // half-width of the ball in this row, in fine cells, rounded outwards
    const float w = (bsqrt(fmaxf(bound - rb2, 0.0f)) * 1.00001f + mg) * inv_hf;
    int x0 = max((int)floorf(xf - w), 0), x1 = min((int)floorf(xf + w), mv.fx - 1);
    if (clip1) {
        x0 = max(x0, g.Fx - 1);
        x1 = min(x1, g.Fx + 1);
    }
    if (x0 > x1) return false;
    const uint32_t row = ((uint32_t)zz * (uint32_t)mv.fy + (uint32_t)yy) * (uint32_t)mv.fx;
    tl.table(2, 4);
    return row_range32<HASH>(mv, row, x0, x1, jlo, jhi);
}

template <bool STATS, bool HASH>
__device__ void search_ball(const MapView& mv, float qx, float qy, float qz, float ub, bool probe,
                            SearchLds& L, int tid, float& bd, int& bj, float& cert, Tally<STATS>& tl)
{
    const QueryCell g = locate(mv, qx, qy, qz);
    const int S = mv.S;
    const float hf = mv.h / (float)S;
    const float inv_hf = (float)S * mv.inv_h;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    const float xf = (float)g.Fx + g.tx;  // fine coordinate of the query along x
    // `cov` is the ball searched: it follows the best distance down, slack included, so every
    // map point within sqrt(cov) at the end has been looked at and `sd` (second-smallest
    // distance seen) bounds everything but the winner -- a certificate, as in the cooperative
    // form.  (Without it a straggler of this path came back as a straggler at every iteration.)
#if VELO_SEED_BOUND
    // A/B of round 5: a SEED for the block-less straggler out of the table itself -- the entry of
    // the query's own (empty) fine cell is the index of the first map point at or beyond it in the sorted order, so
    // that point and the one before it are the nearest along the query's fine row (or, where the row is empty, on a
    // neighbouring row): two candidates, one table entry, and a bound for the probe and the ball below.  A bound only.
    if (probe && !HASH && (unsigned)g.Fx < (unsigned)mv.fx && (unsigned)g.Fy < (unsigned)mv.fy &&
        (unsigned)g.Fz < (unsigned)mv.fz) {
        const uint32_t key = ((uint32_t)g.Fz * (uint32_t)mv.fy + (uint32_t)g.Fy) * (uint32_t)mv.fx + (uint32_t)g.Fx;
        const int js = mv.cell_start[key];
        tl.table(1, 4);
        tl.candidates(2);
        float sb = ub;
        if (js < mv.n) sb = fminf(sb, dist2(mv.pts[(unsigned)js], qx, qy, qz));
        if (js > 0) sb = fminf(sb, dist2(mv.pts[(unsigned)(js - 1)], qx, qy, qz));
        ub = fminf(ub, sb);
    }
#endif
#if VELO_BALL_PROBE
    // A straggler whose 3x3x3 block held NOTHING has no bound but d_max: its ball is a whole voxel
    // wide, all (2S+1)^2 rows, and the wavefront runs for it (a third of the stragglers of an
    // unhinted launch, round 3).  Probe first where a slightly misplaced frame finds its surface:
    // the query's own row at full width and the rows straight above / below / beside it (cells
    // Fx-1..Fx+1), nearest first.  The probe only tightens the bound -- some map point lies that
    // close, so the winner is no further; the search proper below still visits everything within
    // it in index order, so ties resolve as ever.
    if (probe) {
        float pb = ub;
        int nr = 0;
        int jlo, jhi;
        if (ball_row<STATS, HASH>(mv, g, 0, 0, pb, xf, hf, inv_hf, mg, false, jlo, jhi, tl)) {
            L.hi[nr][tid] = jhi;
            L.lo[nr][tid] = jlo;
            ++nr;
        }
        int k = 2, side = 0;
#pragma unroll 1
        while (k <= S) {
            while (nr < kMaxRanges && k <= S) {
                const int dz = side == 0 ? k : (side == 1 ? -k : 0);
                const int dy = side == 2 ? k : (side == 3 ? -k : 0);
                if (ball_row<STATS, HASH>(mv, g, dz, dy, pb, xf, hf, inv_hf, mg, true, jlo, jhi, tl)) {
                    L.hi[nr][tid] = jhi;
                    L.lo[nr][tid] = jlo;
                    ++nr;
                }
                if (++side == 4) {
                    side = 0;
                    ++k;
                }
            }
            int pj = -1;
            float sd_unused = pb;
            walk_ranges<VELO_WALK_W>(mv, qx, qy, qz, L, tid, nr, pb, pj, sd_unused, tl);
            nr = 0;
            const float far = axis_gap(k, 1.0f, hf, mg);  // nearest any probe row at offset >= k can be
            if (far * far * 0.99999f > pb) break;
        }
        ub = fminf(ub, pb);
    }
#else
    (void)probe;
#endif
    float cov = cover_of(ub, mv.h);
    float sd = cov;
    bd = cov;
    bj = -1;
    // Only the (2R+1)^2 rows the ball can reach (R <= S; a straggler with a candidate half a metre
    // away needs 25 rows at S = 3, not 49), in descending order; the row key is carried along in
    // 32-bit arithmetic (see search_block) instead of being multiplied out per row.
    const int R = ball_window(cov, inv_hf, S);
    const uint32_t sy = (uint32_t)mv.fx, sz = (uint32_t)mv.fx * (uint32_t)mv.fy;
    uint32_t rowz = ((uint32_t)(g.Fz + R) * (uint32_t)mv.fy + (uint32_t)g.Fy) * (uint32_t)mv.fx;
    int dz = R, dy = R;
    const int nrows = (2 * R + 1) * (2 * R + 1);
#pragma unroll 1
    for (int r0 = 0; r0 < nrows; r0 += kMaxRanges) {
        int nr = 0;
#pragma unroll
        for (int t = 0; t < kMaxRanges; ++t) {
            if (r0 + t < nrows) {
                const float bz = axis_gap(dz, g.tz, hf, mg), by = axis_gap(dy, g.ty, hf, mg);
                const float rb2 = (bz * bz + by * by) * 0.99999f;
                if ((unsigned)(g.Fz + dz) < (unsigned)mv.fz && (unsigned)(g.Fy + dy) < (unsigned)mv.fy &&
                    !(rb2 > cov)) {
                    // half-width of the ball in this row, in fine cells, rounded outwards
                    const float w = (bsqrt(fmaxf(cov - rb2, 0.0f)) * 1.00001f + mg) * inv_hf;
                    const int x0 = max((int)floorf(xf - w), 0), x1 = min((int)floorf(xf + w), mv.fx - 1);
                    if (x0 <= x1) {
                        int jlo, jhi;
                        row_range32<HASH>(mv, rowz + (uint32_t)dy * sy, x0, x1, jlo, jhi);
                        tl.table(2, 4);
                        if (jhi > jlo) {
                            L.hi[nr][tid] = jhi;
                            L.lo[nr][tid] = jlo;
                            ++nr;
                        }
                    }
                }
                if (--dy < -R) {
                    dy = R;
                    --dz;
                    rowz -= sz;
                }
            }
        }
        walk_ranges<VELO_WALK_W>(mv, qx, qy, qz, L, tid, nr, bd, bj, sd, tl);
        cov = fminf(cov, cover_of(bd, mv.h));
    }
    cert = bj >= 0 ? ball_certificate(sd, cov, mv.h, mg) : 0.0f;
}

// stage B, cooperative form: ONE straggler searched by a whole wavefront.  Lane l takes row
// l (and l+64) of the (2S+1)^2 fine rows around the query, scans the part of it inside the
// ball of radius sqrt(ub) in descending index with '<=', then the 64 lane results are merged
// to the lexicographic minimum of (d2, index) by two wavefront min-reductions (shuffles).
// Every lane returns the same winner.  Rows are independent, so all their loads are in
// flight together: a handful of stragglers no longer costs a serial chain of ~50 dependent
// loads while the other wavefronts of the workgroup wait at the barrier.
template <bool STATS, bool HASH>
__device__ __forceinline__ void search_ball_wave(const MapView& mv, float qx, float qy, float qz,
                                                 float ub, int lane, float& bd, int& bj, float& sd,
                                                 Tally<STATS>& tl)
{
    const QueryCell g = locate(mv, qx, qy, qz);
    float b1 = ub, b2 = ub;  // best and second-best distance seen by this lane
    int j1 = 0x7fffffff;
    const int S = mv.S;
    const int side = 2 * S + 1, nrows = side * side;
    const float hf = mv.h / (float)S;
    const float inv_hf = (float)S * mv.inv_h;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    const float xf = (float)g.Fx + g.tx;
    for (int r = lane; r < nrows; r += 64) {
        const int dz = S - r / side, dy = S - r % side;
        const int zz = g.Fz + dz, yy = g.Fy + dy;
        if (zz < 0 || zz >= mv.fz || yy < 0 || yy >= mv.fy) continue;
        const float bz = dz == 0 ? 0.0f
                                 : fmaxf(((float)(abs(dz) - 1) + (dz > 0 ? 1.0f - g.tz : g.tz)) * hf - mg, 0.0f);
        const float by = dy == 0 ? 0.0f
                                 : fmaxf(((float)(abs(dy) - 1) + (dy > 0 ? 1.0f - g.ty : g.ty)) * hf - mg, 0.0f);
        const float rb2 = (bz * bz + by * by) * 0.99999f;
        if (rb2 > ub) continue;
        const float w = (bsqrt(fmaxf(ub - rb2, 0.0f)) * 1.00001f + mg) * inv_hf;
        const int x0 = max((int)floorf(xf - w), 0), x1 = min((int)floorf(xf + w), mv.fx - 1);
        if (x0 > x1) continue;
        const uint32_t row = ((uint32_t)zz * (uint32_t)mv.fy + (uint32_t)yy) * (uint32_t)mv.fx;
        int jlo, jhi;
        row_range32<HASH>(mv, row, x0, x1, jlo, jhi);
        tl.table(2, 4);
        tl.candidates((unsigned)max(jhi - jlo, 0));
#pragma unroll 4
        for (int j = jhi - 1; j >= jlo; --j) {
            const float d2 = dist2(mv.pts[j], qx, qy, qz);
            // rows are visited in descending order by each lane, so '<=' keeps the lowest
            // index among equal distances inside the lane; lanes are merged below
            if (d2 <= b1) {
                b2 = b1;
                b1 = d2;
                j1 = j;
            } else {
                b2 = fminf(b2, d2);
            }
        }
    }
    float dmin = b1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, off, 64));
    int jm = (b1 == dmin) ? j1 : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) jm = min(jm, __shfl_xor(jm, off, 64));
    // second-smallest distance inside the ball (a tie of the minimum counts): every point
    // within sqrt(ub) has been looked at, so the winner is alone within sqrt(min(sd, ub))
    float s2 = (b1 == dmin && j1 == jm) ? b2 : b1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s2 = fminf(s2, __shfl_xor(s2, off, 64));
    bd = dmin;
    bj = jm == 0x7fffffff ? -1 : jm;
    sd = s2;
}

// ---- stage B for the LATENCY kernel (k_linearize_lat: a single frame, a few hundred
// workgroups -- nothing hides a memory round trip, and registers are plentiful) -------------
// per-lane form.  Called by every lane of the wavefront (`active` = this lane is a straggler):
// the loops synchronise with __any.
template <bool STATS, bool HASH>
__device__ void search_ball_lat(const MapView& mv, float qx, float qy, float qz, float ub, bool active,
                                bool probe, SearchLds& L, int tid, float& bd, int& bj, float& cert,
                                Tally<STATS>& tl)
{
    const QueryCell g = locate(mv, qx, qy, qz);
    bd = ub;
    bj = -1;
    const int S = mv.S;
    const float hf = mv.h / (float)S;
    const float inv_hf = (float)S * mv.inv_h;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    const float xf = (float)g.Fx + g.tx;  // fine coordinate of the query along x
    // Phase 1, only for a query whose 3x3x3 block held nothing: a bound.  The ball of an
    // unmatched query is a whole voxel wide, and on a dense map the rows through the surface it
    // eventually finds hold hundreds of candidates each.  Probe the cells straight above /
    // below / beside the query first (rows (0, +-k) and (+-k, 0), cells Fx-1..Fx+1, nearest
    // first): a frame that is slightly off hangs a fraction of a metre over the ground or in
    // front of a wall, and the foot point is there.  The probe only tightens `bd` as a BOUND
    // (some map point at that distance exists, so the winner is no further); the search proper
    // below still visits everything within it in index order, so ties resolve as ever.
    if (__any(probe)) {
        float pb = bd;
        int k = 2, side = 0;  // next probe row: offset k in direction `side` (+z, -z, +y, -y)
#pragma unroll 1
        while (__any(probe && k <= S)) {
            int nr = 0;
            while (probe && nr < kMaxRanges - 1 && k <= S) {
                const int dz = side == 0 ? k : (side == 1 ? -k : 0);
                const int dy = side == 2 ? k : (side == 3 ? -k : 0);
                int jlo, jhi;
                if (ball_row<STATS, HASH>(mv, g, dz, dy, pb, xf, hf, inv_hf, mg, true, jlo, jhi, tl)) {
                    L.hi[nr][tid] = jhi;
                    L.lo[nr][tid] = jlo;
                    ++nr;
                }
                if (++side == 4) {
                    side = 0;
                    ++k;
                }
            }
            int pj = -1;
            float sd_unused = pb;
            walk_ranges<VELO_WALK_W_LAT>(mv, qx, qy, qz, L, tid, nr, pb, pj, sd_unused, tl);
            const float far = axis_gap(k, 1.0f, hf, mg);  // nearest any row at offset >= k can be
            if (probe && far * far * 0.99999f > pb) k = S + 1;
        }
        bd = fminf(bd, pb);
    }
    // Phase 2: every row the ball can reach, descending (the tie rule), the surviving rows
    // packed nine to a trip: table entries requested together, ranges staged in LDS, one walk.
    // (`cov`, `sd`: the ball searched and the second-smallest distance seen, see search_ball)
    float cov = cover_of(bd, mv.h);
    float sd = cov;
    bd = cov;
    const int R = ball_window(cov, inv_hf, S);
    int dz = active ? R : -R - 1, dy = R;
#pragma unroll 1
    do {
        int nr = 0;
        while (nr < kMaxRanges && dz >= -R) {
            int jlo, jhi;
            if (ball_row<STATS, HASH>(mv, g, dz, dy, cov, xf, hf, inv_hf, mg, false, jlo, jhi, tl)) {
                L.hi[nr][tid] = jhi;
                L.lo[nr][tid] = jlo;
                ++nr;
            }
            if (--dy < -R) {
                dy = R;
                --dz;
            }
        }
        walk_ranges<VELO_WALK_W_LAT>(mv, qx, qy, qz, L, tid, nr, bd, bj, sd, tl);
        cov = fminf(cov, cover_of(bd, mv.h));
    } while (__any(dz >= -R));
    cert = bj >= 0 ? ball_certificate(sd, cov, mv.h, mg) : 0.0f;
}

// cooperative form: as search_ball_wave, over the rows the ball can actually reach, a wide ball
// bounded first by the axis probe (one probe row per lane)
template <bool STATS, bool HASH>
__device__ __forceinline__ void search_ball_wave_lat(const MapView& mv, float qx, float qy, float qz,
                                                     float ub, int lane, float& bd, int& bj, float& sd,
                                                     Tally<STATS>& tl)
{
    const QueryCell g = locate(mv, qx, qy, qz);
    const int S = mv.S;
    const float hf = mv.h / (float)S;
    const float inv_hf = (float)S * mv.inv_h;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    const float xf = (float)g.Fx + g.tx;
    int R = ball_window(ub, inv_hf, S);
    if (R >= 3) {
        float pb = ub;
        if (lane < 4 * (R - 1)) {
            const int k = 2 + lane / 4, side = lane & 3;
            const int dz = side == 0 ? k : (side == 1 ? -k : 0);
            const int dy = side == 2 ? k : (side == 3 ? -k : 0);
            int jlo, jhi;
            if (ball_row<STATS, HASH>(mv, g, dz, dy, ub, xf, hf, inv_hf, mg, true, jlo, jhi, tl)) {
                tl.candidates((unsigned)(jhi - jlo));
                for (int j = jhi - 1; j >= jlo; --j) pb = fminf(pb, dist2(mv.pts[j], qx, qy, qz));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) pb = fminf(pb, __shfl_xor(pb, off, 64));
        if (pb < ub) {
            // a map point at sqrt(pb): search that far plus the certificate slack, no further
            const float rs = bsqrt(pb) * 1.000001f + 1e-7f + kCertSlack;
            ub = fminf(ub, rs * rs * 1.00001f);
            R = ball_window(ub, inv_hf, S);
        }
    }
    float b1 = ub, b2 = ub;  // best and second-best distance seen by this lane
    int j1 = 0x7fffffff;
    const int side = 2 * R + 1, nrows = side * side;
    for (int r = lane; r < nrows; r += 64) {
        const int dz = R - r / side, dy = R - r % side;
        int jlo, jhi;
        if (!ball_row<STATS, HASH>(mv, g, dz, dy, ub, xf, hf, inv_hf, mg, false, jlo, jhi, tl)) continue;
        tl.candidates((unsigned)(jhi - jlo));
#pragma unroll 4
        for (int j = jhi - 1; j >= jlo; --j) {
            const float d2 = dist2(mv.pts[j], qx, qy, qz);
            if (d2 <= b1) {
                b2 = b1;
                b1 = d2;
                j1 = j;
            } else {
                b2 = fminf(b2, d2);
            }
        }
    }
    float dmin = b1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, off, 64));
    int jm = (b1 == dmin) ? j1 : 0x7fffffff;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) jm = min(jm, __shfl_xor(jm, off, 64));
    float s2 = (b1 == dmin && j1 == jm) ? b2 : b1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s2 = fminf(s2, __shfl_xor(s2, off, 64));
    bd = dmin;
    bj = jm == 0x7fffffff ? -1 : jm;
    sd = fminf(s2, ub);
}

// One block = one BlockItem = a run of queries of one frame.  Per round of 256
// queries every thread writes its 8 values {J, r, valid} to LDS; then lane k<29 of
// each 32-lane half sums column k over that half's 32 entries (ascending), the two
// halves are combined by a wavefront shuffle, the four waves through LDS, always in
// the same order: run-to-run deterministic.
struct ReduceLds {
    double v[8][64 + 2];  // per wavefront, SoA; +2 pad: the 8 rows land on distinct banks
};
union LinLds {  // the search ranges and the reduction tile are never live together
    SearchLds s;
    ReduceLds r;
};

#ifndef VELO_COOP_MAX
#define VELO_COOP_MAX 4  // (both forms certify; round 3, per-lane search with ball window + probe: 0 / 1 / 2 / 4 / 8 / 16 -> headline 2234 / 2165 / 2153 / 2120 / 2129 / 2138 us, dense 10 M map 1382 / 1374 / 1378 / 1371 / 1457 / 1698)
#endif
#ifndef VELO_COOP_MAX_LAT
#define VELO_COOP_MAX_LAT 16  // (4 takes 20 us off the second launch on a dense map, but only the
                              // cooperative search leaves a certificate behind: stragglers sent to
                              // the per-lane search come back at every iteration -- converged launch
                              // of a 1 M-point single frame 9 -> 24 us)
#endif
#ifndef VELO_LIN_WAVES
#define VELO_LIN_WAVES 7  // measured: 8 spills (64 VGPRs), 7 = 72 VGPRs no spill, fastest
#endif
// LAT = false: the throughput kernel (batches: tens of thousands of workgroups, bound by VALU
// issue and occupancy: 72 registers, 7 waves per SIMD).  LAT = true: the latency kernel (a
// single frame: < 2 workgroups per CU, bound by dependent memory round trips: registers are
// free, stage B is the packed / probed form above).  Same results bit for bit.
// SEARCH_ONLY (round 5, the split iteration): phase A of an iteration cut in three launches -- certificate test and
// stage A exactly as below, the outcome of every query that is final written to hint / rho (what the full kernel
// would store), every straggler appended to the launch-wide queue sq (k_search_b searches those: one wavefront
// each, or packed 64 to a wavefront), no residual, no sums.  The third launch is this kernel in its ordinary form
// with poses_prev == poses: every query then finds its own certified result and only gathers.
template <bool WRITE_CORR, int VARIANT, bool STATS, bool LAT, bool HASH, int NT, bool SEARCH_ONLY = false>
__device__ __forceinline__ void linearize_body(
    const BlockItem* __restrict__ items, const FrameView& fv, const MapView& mv,
    const double* __restrict__ poses, float dmax2, double* __restrict__ partials,
    int32_t* __restrict__ corr, float* __restrict__ d2out, int32_t* __restrict__ hint,
    float* __restrict__ rho, const double* __restrict__ poses_prev, LinLds* s_uw, double (*s_run)[2][64],
    int lat_lanes = 64, int4* __restrict__ sq = nullptr, unsigned* __restrict__ sq_count = nullptr)
{
    const BlockItem it = items[blockIdx.x];
    const double* __restrict__ T = poses + 12 * (size_t)it.frame;
    const int tid = threadIdx.x;
    // (the wavefront index is uniform: as a scalar it keeps the round loop's bookkeeping out of the VGPRs)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    LinLds& s_u = s_uw[wave];
    const int ia = c_ia[col], ib = c_ib[col];
    Tally<STATS> tl;
    if constexpr (STATS) {
        if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_lin_stats[11], 1ull);
        if (tid == 0) tl.addq(16 + 96 + kAccN * 8);  // work item, pose, partial sums
    }

    // LAT: only the first `lat_lanes` lanes of a wavefront carry a query.  At this launch size the
    // machine is mostly idle, and a wavefront's stragglers are searched one after the other or in
    // lock-step: in the first iteration (a fifth of the queries are stragglers on a dense map) 8
    // queries per wavefront put eight times as many searches in flight -- first launch 304 -> 132 us
    // on a 9 M-point map, 80 -> 62 on 1 M; later iterations use all 64 lanes (9 us against 15).
    //
    // CANONICAL SUMMATION (round 4; DESIGN.md "ICP semantics", Reduction).  The 29 sums of a frame are defined
    // as an aligned binary tree over LEAVES of 8 consecutive queries of the frame (leaf = one fma chain from
    // +0.0 over its entries in order; a node = left child + right child; a child beyond the end of the frame
    // is +0.0).  A work item covers an aligned block of 2^m leaves and writes that node; how a frame is cut
    // into items (rounds per wavefront, which kernel, how many frames share the launch, how many CUs the
    // planner assumed) no longer touches a single bit of the frame's sums, hence of its poses.
    //   wavefront w of an item: the aligned block of 64 R queries at q0 + w 64 R (R = rounds per wavefront,
    //   1 / 2 / 4, in the top bits of `slot`), 64 consecutive queries per round: each half-wavefront's 32
    //   entries are four chains of 8 (leaves) joined as (c0 + c1) + (c2 + c3), the two halves by one cross-lane
    //   exchange (lower + upper); the R rounds are joined by the same tree (running and parked node in LDS);
    //   wavefronts through LDS, pairwise.
    //   Sparse first iteration of the latency kernel (lat_lanes of 8 / 16 / 32): lanes 0 .. lat_lanes-1 of
    //   wavefront w hold the queries q0 + w lat_lanes + lane -- a leaf (or two, four) per wavefront.
    const int logR = (int)((unsigned)it.slot >> 28);
    const int R = 1 << logR;
    const int row_out = it.slot & 0x0fffffff;
    const bool sparse = LAT && lat_lanes < 64;
    const int wblock = sparse ? it.q0 + wave * lat_lanes : it.q0 + wave * (64 * R);  // (wavefront-uniform)
    int rounds_done = 0;
    for (int rr = 0; rr < R; ++rr) {
        if (wblock + rr * 64 >= it.q1) break;  // nothing left for this wavefront (zeros change no sum)
        const int q = wblock + rr * 64 + lane;  // 64 consecutive queries per round: neighbours on the scan line
        const bool live = q < it.q1 && (!sparse || lane < lat_lanes);
        const unsigned uq = (unsigned)q;  // (never negative: no sign extension in the addressing)
        float sxq = 0.f, syq = 0.f, szq = 0.f;
        int hj = -1;
        if (live) {
            sxq = fv.x[uq];
            syq = fv.y[uq];
            szq = fv.z[uq];
            // poses_prev == nullptr marks the first iteration of a registration: whatever the
            // hint / certificate arrays hold is stale (they are not cleared, just overwritten)
            if (hint && poses_prev) hj = hint[uq];
            tl.addq(hint && poses_prev ? 16 : 12);
        }
        double px = 0, py = 0, pz = 0;
        float bd = INFINITY;
        int bj = -1;
        float rho_new_out = 0.0f;  // certified radius for the next iteration (0 = none)
        bool state_same = false;   // hint and rho unchanged: nothing to write back
        // pair certificate left for the next iteration (latency kernels, fv.hint2 != nullptr): runner-up / radius
        constexpr bool PAIR = LAT && VARIANT == 1;
        [[maybe_unused]] int pair_j = -1;
        [[maybe_unused]] float pair_rho = 0.0f;
        [[maybe_unused]] bool scanned_nomatch = false;  // certified WITHOUT a match by the block scan (rho_new_out < 0 holds the radius)
        if (VARIANT >= 1) {
            bool queued = false;
            int st = kFinal;
            float blk_gr = 0.0f;  // guaranteed radius of the stage-A block around this query
            float qx = 0.f, qy = 0.f, qz = 0.f;
            float ub0 = dmax2;
            bool certified = false;
            [[maybe_unused]] bool want_scan = false;
            if (live) {
                xform(T, sxq, syq, szq, px, py, pz);
                qx = (float)px;
                qy = (float)py;
                qz = (float)pz;
                // Temporal coherence, exact:
                //  hint  last iteration's correspondence bounds the search radius;
                //  rho   radius certified last iteration around the query's PREVIOUS position c
                //        inside which the hinted point was the only map point.  The query moved
                //        by delta = |q - c| (c is recomputed from the previous pose), so every
                //        other map point is at least rho - delta away: if the hinted point is
                //        strictly closer than that, it is the unique nearest neighbour and the
                //        search is skipped altogether.  All margins round against skipping.
                if (hj >= 0) {
                    tl.candidates(1);
                    tl.addq(rho ? 4 : 0);
                    const float d1sq = dist2(mv.pts[(unsigned)hj], qx, qy, qz);
                    const float d1 = bsqrt(d1sq) * 1.000001f + 1e-7f;
                    if (rho) {
                        double cx, cy, cz;
                        xform(poses_prev + 12 * (size_t)it.frame, sxq, syq, szq, cx, cy, cz);
                        const float ex = qx - (float)cx, ey = qy - (float)cy, ez = qz - (float)cz;
                        const float rho_in = rho[uq];
                        // a query that has not moved at all (converged pose: bit-identical q)
                        // sees exactly last iteration's distances: the certificate holds verbatim
                        const bool still = (ex == 0.0f) && (ey == 0.0f) && (ez == 0.0f);
                        const float delta = bsqrt(fmaf(ez, ez, fmaf(ey, ey, ex * ex))) * 1.000001f + 1e-7f;
                        const float room = still ? rho_in : (rho_in - delta) * 0.999999f - 1e-7f;
                        if (d1 < room) {
                            certified = true;
                            bd = d1sq;
                            bj = hj;
                            rho_new_out = room;
                            state_same = still;  // hint == hj and rho unchanged
                            if constexpr (PAIR) {
                                // the pair certificate travels along (same winner, same runner-up, delta less room): it
                                // takes over when the uniqueness radius has been used up by the steps of a slowly
                                // converging registration -- without it such a query was searched every few iterations
                                if (fv.hint2 && !still) {
                                    pair_j = fv.hint2[uq];
                                    pair_rho = fmaxf((fv.rho3[uq] - delta) * 0.999999f - 1e-7f, 0.0f);
                                    tl.addq(8);
                                }
                            }
                        }
                        if constexpr (PAIR) {
                            // PAIR CERTIFICATE (round 6).  The uniqueness radius of a query whose two nearest candidates
                            // are nearly equidistant is the distance to the runner-up: no room at all, and on a map made
                            // of scans (consecutive returns of one scan line, centimetres apart) hundreds of queries per
                            // frame were searched at EVERY iteration -- each holding its wavefront for a search while
                            // 99.5 % of the launch had long finished.  rho3 = radius around the previous position inside
                            // which {hint, hint2} are the only map points: both are gathered, the nearer under the
                            // oracle's order (d2, then index) is the nearest neighbour if it lies inside rho3 - delta.
                            if (!certified && fv.hint2) {
                                const int h2 = fv.hint2[uq];
                                const float r3 = fv.rho3[uq];
                                tl.addq(8);
                                if (h2 >= 0 && r3 > 0.0f) {
                                    tl.candidates(1);
                                    const float room3 = still ? r3 : (r3 - delta) * 0.999999f - 1e-7f;
                                    const float d2sq = dist2(mv.pts[(unsigned)h2], qx, qy, qz);
                                    const bool second_wins = d2sq < d1sq || (d2sq == d1sq && h2 < hj);
                                    const float dn_sq = second_wins ? d2sq : d1sq, df_sq = second_wins ? d1sq : d2sq;
                                    const float dn = bsqrt(dn_sq) * 1.000001f + 1e-7f;
                                    if (dn < room3) {
                                        certified = true;
                                        bd = dn_sq;
                                        bj = second_wins ? h2 : hj;
                                        // uniqueness radius of the winner: its partner, or the radius of the pair
                                        rho_new_out = fmaxf(fminf(bsqrt(df_sq) * 0.999999f - 1e-6f, room3), 0.0f);
                                        pair_j = second_wins ? hj : h2;
                                        pair_rho = room3;
                                    }
                                }
                            }
                        }
                    }
                    // search a little beyond the hinted point so that the result certifies a
                    // radius the next iterations can live on
                    const float rs = d1 + kCertSlack;
                    ub0 = fminf(ub0, rs * rs * 1.00001f);
                }
                // A query found WITHOUT a match at exactly this position (rho < 0: written by the search that found
                // nothing within d_max among its candidates) is without one still: the split iteration's third launch
                // meets every such query again at the very pose it was searched at -- they are the costliest searches
                // of a launch (a ball one voxel wide) and come in runs along a scan line, one wavefront's worth.
                // Only for an unmoved query: with d_max = h the radius such a search covers is no larger than d_max.
                if (hj < 0 && rho && hint && poses_prev) {
                    const float rho_in = rho[uq];
                    tl.addq(4);
                    if (rho_in < 0.0f) {
                        double cx, cy, cz;
                        xform(poses_prev + 12 * (size_t)it.frame, sxq, syq, szq, cx, cy, cz);
                        if (qx == (float)cx && qy == (float)cy && qz == (float)cz) {
                            certified = true;  // (bd = INFINITY, bj = -1: no pair)
                            rho_new_out = rho_in;
                            state_same = true;
                        } else if constexpr (PAIR) {
                            // Round 6: -rho is a radius around the previous position c inside which the query's candidate set
                            // (the 27 voxels around c's voxel) holds NO point.  In the same voxel the candidate set is the
                            // same, and every candidate is at least -rho - delta from here: beyond d_max -> still no match.
                            // Otherwise the wavefront scans the block (below): a moved query without a match was the
                            // costliest search of a converged launch, every iteration anew.
                            const float ex = qx - (float)cx, ey = qy - (float)cy, ez = qz - (float)cz;
                            const float delta = bsqrt(fmaf(ez, ez, fmaf(ey, ey, ex * ex))) * 1.000001f + 1e-7f;
                            const float room = (-rho_in - delta) * 0.999999f - 1e-7f;
                            const bool same_voxel =
                                cell_coord(qx, mv.ox, mv.inv_h, mv.nx) == cell_coord((float)cx, mv.ox, mv.inv_h, mv.nx) &&
                                cell_coord(qy, mv.oy, mv.inv_h, mv.ny) == cell_coord((float)cy, mv.oy, mv.inv_h, mv.ny) &&
                                cell_coord(qz, mv.oz, mv.inv_h, mv.nz) == cell_coord((float)cz, mv.oz, mv.inv_h, mv.nz);
                            if (same_voxel && room > 0.0f && room * room * 0.99999f > dmax2) {
                                certified = true;
                                rho_new_out = -room;
                                scanned_nomatch = true;
                            } else {
                                // (scanned once the registration has settled to centimetre steps: a badly placed frame's first
                                //  iterations have thousands of queries without a match, and a radius a 10 cm step uses up)
                                want_scan = fv.hint2 != nullptr && delta < 0.01f;   // (the features of round 6 switch together: VELO_NO_PAIR_CERT)
                            }
                        }
                    }
                }
            }
            if constexpr (PAIR) {
                // the block scans of this wavefront's moved no-match queries, one after the other, EVERY lane of the
                // wavefront on each (outside `if (live)`: the lanes past the end of a frame scan their rows too)
                unsigned long long wm = __ballot(want_scan);
                // (at most 8 per wavefront and round: a run of such queries along a scan line is worked off over a few
                //  iterations -- the others go through the search proper once more, which leaves them "nothing within
                //  d_max here" again)
                for (int n_scan = 0; wm && n_scan < 8; ++n_scan) {
                    const int src = __ffsll((long long)wm) - 1;
                    wm &= wm - 1;
                    const float m2 = scan_block_wave(mv, __shfl(qx, src, 64), __shfl(qy, src, 64), __shfl(qz, src, 64), lane);
                    if (lane == src && m2 * 0.99999f > dmax2) {   // (else: something within reach -- the search proper decides)
                        certified = true;
                        rho_new_out = -fminf(fmaxf(bsqrt(m2) * 0.999999f - 1e-6f, 0.0f), 1.0e30f);
                        scanned_nomatch = true;
                    }
                }
            }
            if (live) {
                VELO_COUNT(1, certified);
                if (!certified) {
                    const QueryCell g = locate(mv, qx, qy, qz);
                    // a query without a previous match whose 27 voxels are all empty has an
                    // empty candidate set: nothing to search (far-range points over a cropped map)
                    bool empty = false;
                    if (hj < 0 && mv.vox_near && g.cx >= 0 && g.cx < mv.nx && g.cy >= 0 &&
                        g.cy < mv.ny && g.cz >= 0 && g.cz < mv.nz)
                        empty = mv.vox_near[((size_t)g.cz * mv.ny + g.cy) * mv.nx + g.cx] == 0;
                    tl.add(hj < 0 && mv.vox_near ? 1 : 0);
                    if (!empty) {
                        if constexpr (PAIR)
                            st = search_block<0, STATS, VELO_WALK_W_LAT, HASH, true>(
                                mv, g, qx, qy, qz, ub0, s_u.s, lane, bd, bj, rho_new_out, blk_gr, tl, &pair_j, &pair_rho);
                        else
                            st = search_block<(VARIANT >= 11 ? VARIANT - 10 : 0), STATS, (LAT ? VELO_WALK_W_LAT : VELO_WALK_W), HASH>(
                                mv, g, qx, qy, qz, ub0, s_u.s, lane, bd, bj, rho_new_out, blk_gr, tl);
                    }
                    VELO_COUNT(3, empty);
                    VELO_COUNT(2, !empty);
                }
            }
            VELO_COUNT(4, live && st == kFinal);
            queued = st == kStraggler;
            if (queued) rho_new_out = 0.0f;
            if constexpr (PAIR) {
                if (queued) {
                    pair_j = -1;
                    pair_rho = 0.0f;
                }
            }
            if constexpr (SEARCH_ONLY) {
                if (live && !queued) {
                    const bool ok = (bj >= 0) && (bd <= dmax2);
                    if (!(state_same && (ok || bj < 0))) {
                        hint[uq] = ok ? bj : -1;
                        // (no match: the search covered the whole d_max -- "no candidate within d_max of here")
                        rho[uq] = ok ? rho_new_out : (PAIR && !scanned_nomatch ? -fmaxf(bsqrt(dmax2) * 0.999999f - 1e-6f, 0.0f) : rho_new_out);
                        if constexpr (PAIR) {
                            if (fv.hint2) {
                                fv.hint2[uq] = ok ? pair_j : -1;
                                fv.rho3[uq] = ok ? pair_rho : 0.0f;
                            }
                        }
                    }
                }
                const unsigned long long qm = __ballot(queued);
                if (qm) {  // one atomic per wavefront-round; the order of the queue does not matter (one slot per query)
                    const int leader = __ffsll((long long)qm) - 1;
                    unsigned base = 0;
                    if (lane == leader) base = atomicAdd(sq_count, (unsigned)__popcll(qm));
                    base = __shfl(base, leader, 64);
                    if (queued)
                        sq[base + (unsigned)__popcll(qm & ((1ull << lane) - 1ull))] =
                            make_int4((int)uq, it.frame | (bj < 0 ? (int)0x40000000 : 0), __float_as_int(bd), 0);
                }
                continue;  // (next round of this wavefront)
            }
            unsigned long long need = __ballot(queued);
            VELO_COUNT(0, live);
            VELO_COUNT(6, queued);
            if (__popcll(need) > (LAT ? VELO_COOP_MAX_LAT : VELO_COOP_MAX)) {
                VELO_COUNT(5, queued);
                if constexpr (LAT) {
                    float rbd = bd, rcert = 0.0f;
                    int rbj = bj;
                    search_ball_lat<STATS, HASH>(mv, qx, qy, qz, queued ? bd : 0.0f, queued, queued && bj < 0,
                                                 s_u.s, lane, rbd, rbj, rcert, tl);
                    if (queued) {
                        bd = rbd;
                        bj = rbj;
                        rho_new_out = rcert;
                    }
                } else if (queued) {
                    const float ub = bd;
                    search_ball<STATS, HASH>(mv, qx, qy, qz, ub, bj < 0, s_u.s, lane, bd, bj, rho_new_out, tl);
                }
            } else {
                while (need) {
                    const int src = __ffsll((long long)need) - 1;
                    need &= need - 1;
                    const float sx = __shfl(qx, src, 64), sy = __shfl(qy, src, 64),
                                sz = __shfl(qz, src, 64);
                    // the ball is searched a little beyond the bound (never past one voxel:
                    // the row window covers that) so that the outcome certifies a radius and
                    // a far-off correspondence is not searched again at every iteration
                    const float rsq = bsqrt(__shfl(bd, src, 64)) + kCertSlack;
                    const float sub = fminf(rsq * rsq, mv.h * mv.h);
                    float rbd, rsd;
                    int rbj;
                    if constexpr (LAT)
                        search_ball_wave_lat<STATS, HASH>(mv, sx, sy, sz, sub, lane, rbd, rbj, rsd, tl);
                    else
                        search_ball_wave<STATS, HASH>(mv, sx, sy, sz, sub, lane, rbd, rbj, rsd, tl);
                    if (lane == src) {
                        bd = rbd;
                        bj = rbj;
                        // (rows outside the window are further than h - mg: same margin as
                        // the cell-assignment rounding everywhere else)
                        const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
                        if (bj >= 0)
                            rho_new_out = fmaxf(fminf(bsqrt(rsd) * 0.999999f, mv.h - 2.0f * mg) - 1e-6f, 0.0f);
                    }
                }
            }
        } else if (live) {
            xform(T, sxq, syq, szq, px, py, pz);
            nearest_scan(mv, (float)px, (float)py, (float)pz, bd, bj);
        }
        // The wavefront's tile {J0..J5, r, valid}[lane] is written INSIDE the branch that computes it, and a
        // lane without a contribution writes zeros in a branch of its own: with the eight values carried out
        // of the nested conditions as variables the compiler zero-initialised them three times over (24
        // v_mov_b64 per wavefront-round, round 4 ISA reading) before the common stores.
        // (The wavefront's own search ranges are dead here: its tile may overwrite them.  DS operations of one
        // wavefront execute in order; the fences only pin the compiler.)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        bool contributes = false;
        if (live) {
            const bool ok = (bj >= 0) && (bd <= dmax2);
            // (a certified, unmoved query with a still-valid match -- or still without one -- keeps its state: no stores)
            if (!(state_same && (ok || bj < 0))) {
                if (hint) hint[uq] = ok ? bj : -1;
                if constexpr (PAIR) {
                    // no match: whichever search said so covered the whole d_max ("no candidate within d_max of here":
                    // the next iteration certifies it if the query has not moved, scans the block if it has); a block
                    // scan's or a moved certificate's own radius stands
                    if (rho) rho[uq] = ok ? rho_new_out : (scanned_nomatch ? rho_new_out : -fmaxf(bsqrt(dmax2) * 0.999999f - 1e-6f, 0.0f));
                    // the pair of a final stage A / of a pair certificate; anything else leaves none (a uniqueness
                    // certificate that moved does not carry the pair's radius along)
                    if (fv.hint2) {
                        fv.hint2[uq] = ok ? pair_j : -1;
                        fv.rho3[uq] = ok ? pair_rho : 0.0f;
                    }
                    tl.addq(fv.hint2 ? 8 : 0);
                } else {
                    if (VARIANT >= 1 && rho) rho[uq] = rho_new_out;
                }
                tl.addq((hint ? 4 : 0) + ((VARIANT >= 1 && rho) ? 4 : 0));
            }
            if (WRITE_CORR) {
                const int qi = fv.order ? fv.order[q] : q;
                if (corr) corr[qi] = ok ? bj : -1;
                if (d2out) d2out[qi] = ok ? bd : INFINITY;
            }
            VELO_COUNT(7, ok);
            if (ok) {
                tl.add(32);
                const float4 nf = mv.nrm[(unsigned)bj];
                const float4 mf = mv.pts[(unsigned)bj];  // issued with the normal: one round trip, not two
#if VELO_REXFORM
                // the unrounded transformed point is needed again only here: recomputing it (9 fp64
                // FMAs, same expression, same bits) keeps six registers free across the whole search
                if (VARIANT >= 1) {
                    asm volatile("" : "+v"(sxq), "+v"(syq), "+v"(szq));
                    xform(T, sxq, syq, szq, px, py, pz);
                }
#endif
                if (!(nf.x == 0.0f && nf.y == 0.0f && nf.z == 0.0f)) {
                    const double nx = nf.x, ny = nf.y, nz = nf.z;
                    const double dx = px - (double)mf.x, dy = py - (double)mf.y,
                                 dz = pz - (double)mf.z;
                    s_u.r.v[6][lane] = fma(nx, dx, fma(ny, dy, nz * dz));
                    s_u.r.v[0][lane] = fma(py, nz, -(pz * ny));
                    s_u.r.v[1][lane] = fma(pz, nx, -(px * nz));
                    s_u.r.v[2][lane] = fma(px, ny, -(py * nx));
                    s_u.r.v[3][lane] = nx;
                    s_u.r.v[4][lane] = ny;
                    s_u.r.v[5][lane] = nz;
                    s_u.r.v[7][lane] = 1.0;
                    contributes = true;
                    asm volatile("" ::: "memory");  // (keeps these stores in this branch)
                }
            }
        }
        if (!contributes) {
            double zero = 0.0;
            asm volatile("" : "+v"(zero));  // one register pair for all eight stores
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) s_u.r.v[k8][lane] = zero;
            asm volatile("" ::: "memory");
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (col < kAccN) {
            const int e0 = half * 32;
            // four leaves of 8 queries, one after the other (all four at once would keep 32 16-byte LDS
            // reads in flight: 80 bytes of scratch per lane at this kernel's 72 registers)
            auto leaf = [&](int eb) {
                double cs = 0.0;
#pragma unroll
                for (int e = 0; e < 8; ++e) cs = fma(s_u.r.v[ia][eb + e], s_u.r.v[ib][eb + e], cs);
                return cs;
            };
            double hsum = leaf(e0) + leaf(e0 + 8);
            asm volatile("" : "+v"(hsum));  // (the second pair's reads start after the first pair is summed)
            const double hs2 = leaf(e0 + 16) + leaf(e0 + 24);
            hsum = hsum + hs2;
            // the two half-wavefronts of a round hold the two 32-query nodes of one 64-query node: joined here
            // (lower + upper, computed on both sides), so that a round stays 64 CONSECUTIVE queries -- with the
            // halves on separate 128-query runs the unhinted launch lost 14 % (392 against 344 us)
            {
                // v_permlane32_swap (gfx950): (a, b) -> a' = {a[0..31], b[0..31]}, b' = {a[32..63], b[32..63]};
                // with a = b = x every lane gets the lower half's value in a' and the upper half's in b'
                const unsigned long long xb = (unsigned long long)__double_as_longlong(hsum);
                const unsigned xl = (unsigned)xb, xh = (unsigned)(xb >> 32);
                const auto pl = __builtin_amdgcn_permlane32_swap(xl, xl, false, false);
                const auto ph = __builtin_amdgcn_permlane32_swap(xh, xh, false, false);
                const double lower = __longlong_as_double((long long)(((unsigned long long)ph[0] << 32) | pl[0]));
                const double upper = __longlong_as_double((long long)(((unsigned long long)ph[1] << 32) | pl[1]));
                hsum = lower + upper;
            }
            // the rounds of this half-wavefront, joined by the same aligned tree (rr is uniform).  The running
            // node and the parked one live in LDS, 16 bytes per lane and round against the 32 KB the column
            // sums read: in a register pair the running node was the one value too many (8 bytes of scratch)
            if (rr & 1) {
                double t = s_run[wave][0][lane] + hsum;
                if (rr & 2) t = s_run[wave][1][lane] + t;
                s_run[wave][0][lane] = t;
            } else {
                if (rr & 2) s_run[wave][1][lane] = s_run[wave][0][lane];
                s_run[wave][0][lane] = hsum;
            }
        }
        rounds_done = rr + 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if constexpr (SEARCH_ONLY) return;
    double colsum = 0.0;
    if (col < kAccN && rounds_done > 0) {
        colsum = s_run[wave][0][lane];
        // a wavefront that ran out of queries after its third round of four still holds (u0 + u1) parked
        if (rounds_done == 3) colsum = s_run[wave][1][lane] + colsum;
    }
    if constexpr (STATS) {
        stat_add(8, tl.bytes);
        stat_add(9, tl.cand);
        stat_add(10, tl.tab);
        stat_add(12, tl.qbytes);
    }
    // halves -> wave (shuffle), waves -> block (LDS): the next levels of the same tree
    if (half == 0) s_run[wave][1][col] = colsum;  // (the parked level is free again)
    __syncthreads();
    if (tid < kAccN) {
        static_assert(NT == 64 || NT == 128 || NT == 256, "one, two or four wavefronts per work item");
        double t = s_run[0][1][tid];
        if constexpr (NT >= 128) t = t + s_run[1][1][tid];
        if constexpr (NT == 256) t = t + (s_run[2][1][tid] + s_run[3][1][tid]);
        partials[(size_t)row_out * kAccStride + tid] = t;
    }
}

// (the hash-table instantiation needs 80 registers: 6 waves per SIMD without spills beat 7 with)
template <bool WRITE_CORR, int VARIANT, bool STATS, bool HASH = false>
__global__ __launch_bounds__(kLinNT, (HASH ? VELO_LIN_WAVES - 1 : VELO_LIN_WAVES)) void k_linearize(
    const BlockItem* __restrict__ items, FrameView fv, MapView mv,
    const double* __restrict__ poses, float dmax2, double* __restrict__ partials,
    int32_t* __restrict__ corr, float* __restrict__ d2out, int32_t* __restrict__ hint,
    float* __restrict__ rho, const double* __restrict__ poses_prev)
{
    __shared__ LinLds s_uw[kLinNT / 64];
    __shared__ double s_run[kLinNT / 64][2][64];
    linearize_body<WRITE_CORR, VARIANT, STATS, false, HASH, kLinNT>(items, fv, mv, poses, dmax2, partials, corr,
                                                                    d2out, hint, rho, poses_prev, s_uw, s_run);
}

template <bool WRITE_CORR, bool STATS, bool HASH = false>
__global__ __launch_bounds__(kLinThreads, 4) void k_linearize_lat(
    const BlockItem* __restrict__ items, FrameView fv, MapView mv,
    const double* __restrict__ poses, float dmax2, double* __restrict__ partials,
    int32_t* __restrict__ corr, float* __restrict__ d2out, int32_t* __restrict__ hint,
    float* __restrict__ rho, const double* __restrict__ poses_prev, int lat_lanes)
{
    __shared__ LinLds s_uw[kLinThreads / 64];
    __shared__ double s_run[kLinThreads / 64][2][64];
    linearize_body<WRITE_CORR, 1, STATS, true, HASH, kLinThreads>(items, fv, mv, poses, dmax2, partials, corr,
                                                                  d2out, hint, rho, poses_prev, s_uw, s_run, lat_lanes);
}

// ---- the split iteration (round 5): phase A = the search of every query up to and including stage A, phase B = the
// stragglers of the whole launch, phase C = the ordinary kernel on certified hints (launch_linearize, poses_prev = poses)
template <bool HASH>
__global__ __launch_bounds__(kLinNT, VELO_LIN_WAVES) void k_search_a(
    const BlockItem* __restrict__ items, FrameView fv, MapView mv, const double* __restrict__ poses, float dmax2,
    int32_t* __restrict__ hint, float* __restrict__ rho, const double* __restrict__ poses_prev,
    int4* __restrict__ sq, unsigned* __restrict__ sq_count)
{
    __shared__ LinLds s_uw[kLinNT / 64];
    linearize_body<false, 1, false, false, HASH, kLinNT, true>(items, fv, mv, poses, dmax2, nullptr, nullptr, nullptr, hint,
                                                                rho, poses_prev, s_uw, nullptr, 64, sq, sq_count);
}
template <bool HASH>
__global__ __launch_bounds__(kLinThreads, 4) void k_search_a_lat(
    const BlockItem* __restrict__ items, FrameView fv, MapView mv, const double* __restrict__ poses, float dmax2,
    int32_t* __restrict__ hint, float* __restrict__ rho, const double* __restrict__ poses_prev, int lat_lanes,
    int4* __restrict__ sq, unsigned* __restrict__ sq_count)
{
    __shared__ LinLds s_uw[kLinThreads / 64];
    linearize_body<false, 1, false, true, HASH, kLinThreads, true>(items, fv, mv, poses, dmax2, nullptr, nullptr, nullptr,
                                                                    hint, rho, poses_prev, s_uw, nullptr, lat_lanes, sq,
                                                                    sq_count);
}

// phase B.  Few stragglers (a single frame: the chip is idle): ONE WAVEFRONT PER STRAGGLER, rows of its ball in
// parallel over the lanes (search_ball_wave_lat) -- inside the full kernel a wavefront works its own stragglers off
// one after the other while four fifths of the machine wait.  Many (a batch): 64 to a wavefront, per lane
// (search_ball_lat) -- inside the full kernel a fifth of the lanes of every wavefront walk while the others idle.
// Either way exact, either way a certificate is left behind (the forms differ in the radius they certify, never in
// the winner).
constexpr int kSearchBThreads = 256;
template <bool HASH>
__global__ __launch_bounds__(kSearchBThreads, 4) void k_search_b(
    const int4* __restrict__ sq, const unsigned* __restrict__ sq_count, FrameView fv, MapView mv,
    const double* __restrict__ poses, float dmax2, int32_t* __restrict__ hint, float* __restrict__ rho,
    unsigned per_wave_max)
{
    __shared__ SearchLds s_l[kSearchBThreads / 64];
    const unsigned n = *sq_count;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned wg = blockIdx.x * (kSearchBThreads / 64) + (unsigned)wave, nw = gridDim.x * (kSearchBThreads / 64);
    Tally<false> tl;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    if (n <= per_wave_max) {
        for (unsigned e = wg; e < n; e += nw) {
            const int4 en = sq[e];
            const unsigned uq = (unsigned)en.x;
            const double* __restrict__ T = poses + 12 * (size_t)(en.y & 0x3fffffff);
            double px, py, pz;
            xform(T, fv.x[uq], fv.y[uq], fv.z[uq], px, py, pz);
            const float qx = (float)px, qy = (float)py, qz = (float)pz;
            // (as the cooperative branch of linearize_body: a little beyond the bound, never past one voxel)
            const float rsq = bsqrt(__int_as_float(en.z)) + kCertSlack;
            const float sub = fminf(rsq * rsq, mv.h * mv.h);
            float rbd, rsd;
            int rbj;
            search_ball_wave_lat<false, HASH>(mv, qx, qy, qz, sub, lane, rbd, rbj, rsd, tl);
            if (lane == 0) {
                const bool ok = (rbj >= 0) && (rbd <= dmax2);
                hint[uq] = ok ? rbj : -1;
                // (nothing within the ball, and the ball was the whole d_max: "no match at this position", see linearize_body)
                // -radius: "no candidate within d_max of here" (linearize_body reads it)
                rho[uq] = rbj >= 0 ? fmaxf(fminf(bsqrt(rsd) * 0.999999f, mv.h - 2.0f * mg) - 1e-6f, 0.0f)
                                   : (__int_as_float(en.z) >= dmax2 ? -fmaxf(bsqrt(dmax2) * 0.999999f - 1e-6f, 1e-30f) : 0.0f);
                if (fv.hint2) {  // (a straggler's result carries no pair)
                    fv.hint2[uq] = -1;
                    fv.rho3[uq] = 0.0f;
                }
            }
        }
    } else {
        for (unsigned b0 = wg * 64u; b0 < n; b0 += nw * 64u) {  // (uniform per wavefront)
            const unsigned e = b0 + (unsigned)lane;
            const bool active = e < n;
            const int4 en = active ? sq[e] : make_int4(0, 0, 0, 0);
            const unsigned uq = (unsigned)en.x;
            float qx = 0.f, qy = 0.f, qz = 0.f;
            if (active) {
                const double* __restrict__ T = poses + 12 * (size_t)(en.y & 0x3fffffff);
                double px, py, pz;
                xform(T, fv.x[uq], fv.y[uq], fv.z[uq], px, py, pz);
                qx = (float)px, qy = (float)py, qz = (float)pz;
            }
            float rbd = 0.0f, rcert = 0.0f;
            int rbj = -1;
            search_ball_lat<false, HASH>(mv, qx, qy, qz, active ? __int_as_float(en.z) : 0.0f, active,
                                         active && (en.y & 0x40000000) != 0, s_l[wave], lane, rbd, rbj, rcert, tl);
            if (active) {
                const bool ok = (rbj >= 0) && (rbd <= dmax2);
                hint[uq] = ok ? rbj : -1;
                rho[uq] = rbj >= 0 ? rcert : (__int_as_float(en.z) >= dmax2 ? -fmaxf(bsqrt(dmax2) * 0.999999f - 1e-6f, 1e-30f) : 0.0f);
                if (fv.hint2) {
                    fv.hint2[uq] = -1;
                    fv.rho3[uq] = 0.0f;
                }
            }
        }
    }
}

hipError_t launch_search_split(const BlockItem* items, int n_items, const FrameView& fv, const MapView& mv,
                               const double* poses, float dmax2, int32_t* hint, float* rho, const double* poses_prev,
                               int force_kernel, int lat_lanes, int4* sq, unsigned* sq_count, unsigned per_wave_max,
                               int grid_b, hipStream_t s)
{
    if (n_items == 0) return hipSuccess;
    if (lat_lanes < 1 || lat_lanes > 64) lat_lanes = 64;
    const bool lat = force_kernel == 2 || (force_kernel != 1 && n_items < kLatItems);
    const bool hash = !mv.cell_start;
    if (lat) {
        if (hash)
            hipLaunchKernelGGL((k_search_a_lat<true>), dim3(n_items), dim3(kLinThreads), 0, s, items, fv, mv, poses, dmax2,
                               hint, rho, poses_prev, lat_lanes, sq, sq_count);
        else
            hipLaunchKernelGGL((k_search_a_lat<false>), dim3(n_items), dim3(kLinThreads), 0, s, items, fv, mv, poses, dmax2,
                               hint, rho, poses_prev, lat_lanes, sq, sq_count);
    } else {
        if (hash)
            hipLaunchKernelGGL((k_search_a<true>), dim3(n_items), dim3(kLinNT), 0, s, items, fv, mv, poses, dmax2, hint,
                               rho, poses_prev, sq, sq_count);
        else
            hipLaunchKernelGGL((k_search_a<false>), dim3(n_items), dim3(kLinNT), 0, s, items, fv, mv, poses, dmax2, hint,
                               rho, poses_prev, sq, sq_count);
    }
    if (hash)
        hipLaunchKernelGGL((k_search_b<true>), dim3(grid_b), dim3(kSearchBThreads), 0, s, sq, sq_count, fv, mv, poses, dmax2,
                           hint, rho, per_wave_max);
    else
        hipLaunchKernelGGL((k_search_b<false>), dim3(grid_b), dim3(kSearchBThreads), 0, s, sq, sq_count, fv, mv, poses, dmax2,
                           hint, rho, per_wave_max);
    return hipGetLastError();
}

hipError_t read_lin_stats(unsigned long long out[16], bool reset, hipStream_t s)
{
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lin_stats), 16 * sizeof(unsigned long long));
    if (e != hipSuccess || !reset) return e;
    const unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lin_stats), z, sizeof z);
}

hipError_t launch_linearize(int variant, const BlockItem* items, int n_items, const FrameView& fv,
                            const MapView& mv, const double* poses, float dmax2, double* partials,
                            int32_t* corr, float* d2, int32_t* hint, float* rho,
                            const double* poses_prev, bool stats, int force_kernel, hipStream_t s, int lat_lanes)
{
    if (n_items == 0) return hipSuccess;
    if (lat_lanes < 1 || lat_lanes > 64) lat_lanes = 64;
    const bool wc = corr || d2;
#define VELO_LAUNCH_LIN(WC, V)                                                                   \
    hipLaunchKernelGGL((k_linearize<WC, V, false>), dim3(n_items), dim3(kLinNT), 0, s,     \
                       items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev)
    // a launch that leaves most of the chip idle is a latency problem: fewer than kLatItems
    // workgroups (~4 frames) go to the latency kernel
    const bool lat = variant == VELO_VARIANT_BALL &&
                     (force_kernel == 2 || (force_kernel != 1 && n_items < kLatItems));
    if (!mv.cell_start) {
        // sparse fine-cell table: the ball search in its two kernels and their counting
        // instantiations (the validation scan reads the table through the run-time form)
        if (variant == VELO_VARIANT_SCAN) {
            hipLaunchKernelGGL((k_linearize<true, 0, false>), dim3(n_items), dim3(kLinNT), 0, s, items,
                               fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev);
        } else if (stats) {
            if (lat)
                hipLaunchKernelGGL((k_linearize_lat<true, true, true>), dim3(n_items), dim3(kLinThreads), 0,
                                   s, items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev, lat_lanes);
            else
                hipLaunchKernelGGL((k_linearize<true, 1, true, true>), dim3(n_items), dim3(kLinNT), 0,
                                   s, items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev);
        } else if (lat) {
            if (wc)
                hipLaunchKernelGGL((k_linearize_lat<true, false, true>), dim3(n_items), dim3(kLinThreads), 0,
                                   s, items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev, lat_lanes);
            else
                hipLaunchKernelGGL((k_linearize_lat<false, false, true>), dim3(n_items), dim3(kLinThreads), 0,
                                   s, items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev, lat_lanes);
        } else {
            if (wc)
                hipLaunchKernelGGL((k_linearize<true, 1, false, true>), dim3(n_items), dim3(kLinNT), 0,
                                   s, items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev);
            else
                hipLaunchKernelGGL((k_linearize<false, 1, false, true>), dim3(n_items), dim3(kLinNT), 0,
                                   s, items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev);
        }
        return hipGetLastError();
    }
    if (stats && variant != VELO_VARIANT_SCAN && variant < 10) {  // counting instantiation
        if (lat)
            hipLaunchKernelGGL((k_linearize_lat<true, true>), dim3(n_items), dim3(kLinThreads), 0, s,
                               items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev, lat_lanes);
        else
            hipLaunchKernelGGL((k_linearize<true, 1, true>), dim3(n_items), dim3(kLinNT), 0, s,
                               items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev);
    } else if (lat) {
        if (wc)
            hipLaunchKernelGGL((k_linearize_lat<true, false>), dim3(n_items), dim3(kLinThreads), 0, s,
                               items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev, lat_lanes);
        else
            hipLaunchKernelGGL((k_linearize_lat<false, false>), dim3(n_items), dim3(kLinThreads), 0, s,
                               items, fv, mv, poses, dmax2, partials, corr, d2, hint, rho, poses_prev, lat_lanes);
    } else if (variant == VELO_VARIANT_SCAN) {
        if (wc) VELO_LAUNCH_LIN(true, 0); else VELO_LAUNCH_LIN(false, 0);
#ifdef VELO_ABLATIONS  // timing ablations (wrong results by design): private builds only
    } else if (variant == 11) {
        VELO_LAUNCH_LIN(false, 11);
    } else if (variant == 12) {
        VELO_LAUNCH_LIN(false, 12);
    } else if (variant == 13) {
        VELO_LAUNCH_LIN(false, 13);
#endif
    } else {  // VELO_VARIANT_BALL and anything unknown: the default kernel
        if (wc) VELO_LAUNCH_LIN(true, 1); else VELO_LAUNCH_LIN(false, 1);
    }
#undef VELO_LAUNCH_LIN
    return hipGetLastError();
}

// ============================================================= reduce + solve
// One lane runs this after the block reduction, so the kernel's duration is the length of the
// dependent instruction chain: one reciprocal per pivot instead of a division per entry (27
// fp64 divisions -> 6), and the rotation coefficients from their series for the small steps ICP
// takes (no sin / cos / sqrt / division below 0.1 rad).  Agrees with oracle/icp.c
// (vo_solve_update, plain divisions and libm) to a few ulp.
__device__ int ldlt6(const double* H, const double* b, double* xs)
{
    double L[36], D[6], Di[6];
#pragma unroll
    for (int i = 0; i < 36; ++i) L[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = H[6 * j + j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        if (!(d > 0.0)) return 1;
        D[j] = d;
        const double inv = 1.0 / d;
        Di[j] = inv;
        L[6 * j + j] = 1.0;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double v = H[6 * i + j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = v * inv;
        }
    }
    double yv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) v -= L[6 * i + k] * yv[k];
        yv[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) yv[i] *= Di[i];
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double v = yv[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) v -= L[6 * k + i] * xs[k];
        xs[i] = v;
    }
    return 0;
}

__device__ void se3_exp_apply(const double* xi, double* T)
{
    const double wx = xi[0], wy = xi[1], wz = xi[2];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double A, B, C;  // sin(t)/t, (1-cos t)/t^2, (t - sin t)/t^3
    if (th2 < 1e-2) {
        // alternating series in t^2, eight terms: the first omitted term is < 1e-31 relative
        // (reciprocal constants: no division in the chain)
        const double x = th2;
        A = 1.0 - x * (1.0 / 6.0) * (1.0 - x * (1.0 / 20.0) * (1.0 - x * (1.0 / 42.0) * (1.0 - x * (1.0 / 72.0) * (1.0 - x * (1.0 / 110.0) * (1.0 - x * (1.0 / 156.0) * (1.0 - x * (1.0 / 210.0)))))));
        B = 0.5 * (1.0 - x * (1.0 / 12.0) * (1.0 - x * (1.0 / 30.0) * (1.0 - x * (1.0 / 56.0) * (1.0 - x * (1.0 / 90.0) * (1.0 - x * (1.0 / 132.0) * (1.0 - x * (1.0 / 182.0) * (1.0 - x * (1.0 / 240.0))))))));
        C = (1.0 / 6.0) * (1.0 - x * (1.0 / 20.0) * (1.0 - x * (1.0 / 42.0) * (1.0 - x * (1.0 / 72.0) * (1.0 - x * (1.0 / 110.0) * (1.0 - x * (1.0 / 156.0) * (1.0 - x * (1.0 / 210.0) * (1.0 - x * (1.0 / 272.0))))))));
    } else {
        const double th = sqrt(th2);
        A = sin(th) / th;
        B = (1.0 - cos(th)) / th2;
        C = (1.0 - A) / th2;
    }
    const double K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double K2[9], Rd[9], Vm[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            K2[3 * i + j] = K[3 * i] * K[j] + K[3 * i + 1] * K[3 + j] + K[3 * i + 2] * K[6 + j];
    for (int i = 0; i < 9; ++i) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        Rd[i] = I + A * K[i] + B * K2[i];
        Vm[i] = I + B * K[i] + C * K2[i];
    }
    double N[12];
    for (int i = 0; i < 3; ++i) {
        const double td = Vm[3 * i] * xi[3] + Vm[3 * i + 1] * xi[4] + Vm[3 * i + 2] * xi[5];
        for (int j = 0; j < 4; ++j)
            N[4 * i + j] = Rd[3 * i] * T[j] + Rd[3 * i + 1] * T[4 + j] + Rd[3 * i + 2] * T[8 + j];
        N[4 * i + 3] += td;
    }
    for (int i = 0; i < 12; ++i) T[i] = N[i];
}

// One 1024-thread workgroup per frame: the upper levels of the frame's CANONICAL tree (see linearize_body).
// A frame's rows are aligned nodes of that tree, in frame order, of two sizes at most: `head` small rows,
// `nbig` rows 2^mlog times as large, small rows again to the end (the planner's one-round items in front of /
// behind the large ones; a uniform decomposition is head = 0, mlog = 0).  On the grid of SMALL nodes ("slots")
// a large row sits in the first slot of its block and the others are +0.0 -- x + 0.0 = x, so the aligned
// binary tree over the slots is the frame's tree whatever the row sizes were.  Thread (g = tid / 32,
// k = tid % 32) reduces column k of the B consecutive slots g B .. (g + 1) B - 1 (B = a power of two >= 16
// with 32 B >= slots): 16 loads in flight per trip (one trip for a single frame's 450 rows), a fixed tree
// over the 16, trips joined through a carry stack in LDS (trip c joins as the binary counter c says);
// then the 32 group nodes pairwise.  Nothing here depends on how many rows there are per slot, per group or
// per launch -- only on the queries of the frame.
#ifndef VELO_SOLVE_SPEC
#define VELO_SOLVE_SPEC 1
#endif
constexpr int kSolveStack = 8;  // trips per group <= 2^8: 8 .. 32 groups x 16 x 256 slots per frame
// (measurement aid: VELO_SOLVE_THREADS = 256 / 512 / 1024 pins the workgroup size of k_reduce_solve)
static int solve_threads_override()
{
    static const int v = [] {
        const char* e = getenv("VELO_SOLVE_THREADS");
        const int t = e ? atoi(e) : 0;
        return (t == 256 || t == 512 || t == 1024) ? t : 0;
    }();
    return v;
}

__device__ __forceinline__ int slot_row(int slot, int head, int nbig, int mlog, int nrows)
{
    int row;
    if (slot < head) {
        row = slot;
    } else {
        const int d = slot - head;
        if (d < (nbig << mlog))
            row = (d & ((1 << mlog) - 1)) == 0 ? head + (d >> mlog) : -1;
        else
            row = head + nbig + (d - (nbig << mlog));
    }
    return row < nrows ? row : -1;
}

// MIXED = false: every frame's rows are of one size (a single frame, the latency kernel's items): the
// instantiation carries none of the two-size bookkeeping -- it is the kernel a single-frame iteration waits for
template <bool MIXED, int kSolveThreads>
__global__ __launch_bounds__(kSolveThreads) void k_reduce_solve(
    const double* __restrict__ partials, const int32_t* __restrict__ fbs, const int4* __restrict__ layout,
    double* __restrict__ poses, velo_icp_iter* __restrict__ stats, int iter,
    double* __restrict__ acc_out, int do_update, double* __restrict__ poses_prev,
    unsigned long long* __restrict__ pairs_total, int spec_rows, int4 lay0)
{
    constexpr int kSolveGroups = kSolveThreads / 32;
    __shared__ double s_g[kSolveGroups][32];
    __shared__ double s_acc[32];
    __shared__ double s_pose[12];
    __shared__ double s_stack[kSolveStack][kSolveThreads];
    const int f = blockIdx.x, k = threadIdx.x & 31, g = threadIdx.x >> 5;
    // the frame's pose travels with the partial sums (one memory round trip, not two in a row)
    double pose_k = 0.0;
    if (threadIdx.x < 12) pose_k = poses[12 * (size_t)f + threadIdx.x];
    {
        // Frame 0's rows start at row 0 and its layout comes with the launch (lay0, spec_rows = rows the
        // caller vouches for): its loads go out without waiting for anything -- one memory round trip per
        // iteration of a single frame's registration.  The other frames fetch their row range and layout first.
        int b0 = 0, nrows, head, nbig, mlog, nslots;
        if (VELO_SOLVE_SPEC && f == 0 && spec_rows > 0) {
            head = lay0.x, nbig = lay0.y, mlog = lay0.z, nslots = lay0.w;
            nrows = min(spec_rows, head + nbig + max(nslots - head - (nbig << mlog), 0));
        } else {
            b0 = fbs[f];
            nrows = fbs[f + 1] - b0;
            const int4 l = layout[f];
            head = l.x, nbig = l.y, mlog = l.z, nslots = l.w;
        }
        int B = 16;
        while (B * kSolveGroups < nslots) B <<= 1;
        // 16 slots = one fixed tree; with B >= 32 two of them (an aligned 32-slot node) per trip, ALL their loads
        // in flight together: a batch's frames are 900 slots (B = 32) -- as two trips of 16 the solve of a
        // 64-frame batch went from 6.5 to 9.8 us, two memory round trips in a row instead of one
        auto load16 = [&](int s0, double (&v)[16]) {
            if (!MIXED || nbig == 0) {  // uniform rows (a single frame's items): row = slot, nothing to work out
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    v[u] = partials[(size_t)(b0 + min(s0 + u, max(nrows - 1, 0))) * kAccStride + k];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (s0 + u >= nrows) v[u] = 0.0;
            } else {
                // (1 024 threads are 4 wavefronts per SIMD: 16 x slot_row per thread was 2 us of integer work.  A
                // 16-slot span lies inside ONE region of the layout except at the two region borders: inside the
                // large rows it is 16 >> mlog loads at a constant stride, inside the small ones 16 consecutive rows)
                const int big0 = head, big1 = head + (nbig << mlog);
                if (s0 >= big0 && s0 + 16 <= big1 && mlog >= 1 && mlog <= 2) {
                    const int r0 = head + ((s0 - big0) >> mlog);
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = 0.0;
                    // (unconditional loads on clamped rows, masked afterwards: a load under a branch of its own
                    // is waited for at the branch's end -- a chain of round trips instead of one)
                    const int last = max(nrows - 1, 0);
                    if (mlog == 2) {
                        double w[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[j] = partials[(size_t)(b0 + min(r0 + j, last)) * kAccStride + k];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[4 * j] = r0 + j < nrows ? w[j] : 0.0;
                    } else {
                        double w[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) w[j] = partials[(size_t)(b0 + min(r0 + j, last)) * kAccStride + k];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[2 * j] = r0 + j < nrows ? w[j] : 0.0;
                    }
                } else if (s0 + 16 <= big0 || s0 >= big1) {
                    const int r0 = s0 < big0 ? s0 : s0 - (nbig << mlog) + nbig;
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        v[u] = partials[(size_t)(b0 + min(r0 + u, max(nrows - 1, 0))) * kAccStride + k];
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (r0 + u >= nrows) v[u] = 0.0;
                } else {
                    double w[16];
                    int rr[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        rr[u] = slot_row(s0 + u, head, nbig, mlog, nrows);
                        w[u] = partials[(size_t)(b0 + max(rr[u], 0)) * kAccStride + k];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = rr[u] >= 0 ? w[u] : 0.0;
                }
            }
        };
        auto tree16 = [](const double (&v)[16]) {
            return (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                   (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        };
        const bool wide = B >= 32;
        const int trips = wide ? B >> 5 : 1;
        double node = 0.0;
        for (int c = 0; c < trips; ++c) {  // (uniform)
            const int s0 = g * B + c * (wide ? 32 : 16);
            double t = 0.0;
            if (k < kAccN && s0 < nslots && nrows > 0) {
                double v0[16], v1[16];
                load16(s0, v0);
                if (wide) load16(s0 + 16, v1);
                t = tree16(v0);
                if (wide) t = t + tree16(v1);
            }
            if (trips > 1) {
                int cc = c, lvl = 0;
                while (cc & 1) {  // (uniform) carry: join with the waiting node of each completed level
                    t = s_stack[lvl][threadIdx.x] + t;
                    cc >>= 1;
                    ++lvl;
                }
                if (c + 1 < trips) s_stack[lvl][threadIdx.x] = t;  // (own slot; the last trip's node is the result)
            }
            node = t;
        }
        s_g[g][k] = node;  // after the last trip (trips is a power of two) t is the group's node
    }
    if (threadIdx.x < 12) {
        s_pose[threadIdx.x] = pose_k;
        if (do_update && poses_prev) poses_prev[12 * (size_t)f + threadIdx.x] = pose_k;
    }
    __syncthreads();
    if (threadIdx.x < kAccN) {
        double w[kSolveGroups];
#pragma unroll
        for (int gg = 0; gg < kSolveGroups; ++gg) w[gg] = s_g[gg][threadIdx.x];
#pragma unroll
        for (int span = 1; span < kSolveGroups; span <<= 1)
#pragma unroll
            for (int gg = 0; gg < kSolveGroups; gg += 2 * span) w[gg] = w[gg] + w[gg + span];
        const double a = w[0];
        s_acc[threadIdx.x] = a;
        if (acc_out) acc_out[(size_t)f * kAccStride + threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // (poses_prev, written above: the pose this iteration was linearised at -- the next
    // k_linearize recomputes every query's previous position from it, uniqueness certificate)
    const double cnt = s_acc[28];
    velo_icp_iter st;
    st.n_pairs = (uint32_t)cnt;
    st.rmse = cnt > 0.0 ? sqrt(s_acc[27] / cnt) : 0.0;
    st.solve_flag = 0;
    // running count of the pairs every registration iteration on this ctx has processed
    // (integer adds: exact and order independent)
    if (pairs_total && do_update) atomicAdd(pairs_total, (unsigned long long)cnt);
    if (do_update) {
        if (cnt < 6.0) {
            st.solve_flag = 2;
        } else {
            double H[36], b[6], xi[6];
            int c = 0;
            for (int a = 0; a < 6; ++a)
                for (int e = a; e < 6; ++e, ++c) H[6 * a + e] = H[6 * e + a] = s_acc[c];
            for (int a = 0; a < 6; ++a) b[a] = -s_acc[21 + a];
            int bad = ldlt6(H, b, xi);
            if (bad) {
                st.solve_flag = 1;
                for (int a = 0; a < 6; ++a) H[7 * a] += 1e-9;
                bad = ldlt6(H, b, xi);
            }
            if (bad) {
                st.solve_flag = 2;
            } else {
                double T[12];
                for (int i = 0; i < 12; ++i) T[i] = s_pose[i];
                se3_exp_apply(xi, T);
                for (int i = 0; i < 12; ++i) poses[12 * (size_t)f + i] = T[i];
            }
        }
    }
    if (stats) stats[(size_t)f * VELO_MAX_ITERS + iter] = st;
}

hipError_t launch_reduce_solve(const double* partials, const int32_t* frame_block_start, const RowLayout* layout,
                               int n_frames, double* poses, velo_icp_iter* stats, int iter,
                               int solve_threads, double* acc_out, int do_update, double* poses_prev,
                               unsigned long long* pairs_total, hipStream_t s, int spec_rows,
                               const RowLayout* layout0, bool mixed)
{
    if (n_frames == 0) return hipSuccess;
    static_assert(sizeof(RowLayout) == sizeof(int4), "RowLayout is read as an int4");
    int4 l0 = make_int4(0, 0, 0, 0);
    if (layout0 && spec_rows > 0)
        l0 = make_int4(layout0->head, layout0->nbig, layout0->mlog, layout0->nslots);
    else
        spec_rows = 0;
    // 1 024 threads (32 groups): a frame's 450 rows in ONE trip of loads.  The workgroup size is a template parameter
    // since round 5 -- the tree over the slots is the same aligned one for 8, 16 or 32 groups (B x groups >= slots),
    // the whole GPU suite passes bit for bit with 256 threads for single frames -- because a workgroup of 1 024 threads
    // x 112 registers needs a nearly empty CU, and beside a map roll on another stream it waited 0.9 - 4.7 ms for
    // one.  What cured that was not a smaller solve (256 threads still waited 0.2 - 0.35 ms per launch and cost the
    // stream 7 % by its four trips) but keeping two CUs of every shader engine free of the roll (capi.cpp,
    // velo_map_roll_begin; profiles/r05/roll_begin_trace_*.txt).  VELO_SOLVE_THREADS pins another size (measurement).
    // (cfg.solve_threads; 0 = the default, which the measurement variable VELO_SOLVE_THREADS may override)
    const int threads = (solve_threads == 256 || solve_threads == 512 || solve_threads == 1024)
                            ? solve_threads : (solve_threads_override() ? solve_threads_override() : 1024);
#define VELO_LAUNCH_SOLVE(MX, TT)                                                                                      \
    hipLaunchKernelGGL((k_reduce_solve<MX, TT>), dim3(n_frames), dim3(TT), 0, s, partials, frame_block_start,          \
                       reinterpret_cast<const int4*>(layout), poses, stats, iter, acc_out, do_update, poses_prev,       \
                       pairs_total, spec_rows, l0)
    if (mixed) {
        if (threads == 256) VELO_LAUNCH_SOLVE(true, 256);
        else if (threads == 512) VELO_LAUNCH_SOLVE(true, 512);
        else VELO_LAUNCH_SOLVE(true, 1024);
    } else {
        if (threads == 256) VELO_LAUNCH_SOLVE(false, 256);
        else if (threads == 512) VELO_LAUNCH_SOLVE(false, 512);
        else VELO_LAUNCH_SOLVE(false, 1024);
    }
#undef VELO_LAUNCH_SOLVE
    return hipGetLastError();
}

// ====================================================== frame query ordering
// key = frame * (ncell + 1) + cell key of the transformed query (ncell = outside)
__global__ __launch_bounds__(256) void k_frame_cellkeys(FrameView fv,
                                                        const int64_t* __restrict__ frame_start,
                                                        int n_frames, size_t n_total, MapView mv,
                                                        const double* __restrict__ poses,
                                                        uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ idx)
{
    const size_t ncell = (size_t)mv.nx * mv.ny * mv.nz;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_total;
         i += (size_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_frames;  // frame of query i: last f with frame_start[f] <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((size_t)frame_start[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        double px, py, pz;
        xform(poses + 12 * (size_t)lo, fv.x[i], fv.y[i], fv.z[i], px, py, pz);
        const int cx = cell_coord((float)px, mv.ox, mv.inv_h, mv.nx);
        const int cy = cell_coord((float)py, mv.oy, mv.inv_h, mv.ny);
        const int cz = cell_coord((float)pz, mv.oz, mv.inv_h, mv.nz);
        size_t key = ncell;
        if (cx >= 0 && cx < mv.nx && cy >= 0 && cy < mv.ny && cz >= 0 && cz < mv.nz)
            key = ((size_t)cz * mv.ny + cy) * mv.nx + cx;
        keys[i] = (uint32_t)((size_t)lo * (ncell + 1) + key);
        idx[i] = (uint32_t)i;
    }
}

hipError_t launch_frame_cellkeys(const FrameView& fv, const int64_t* d_frame_start, int n_frames,
                                 size_t n_total, const MapView& mv, const double* poses,
                                 uint32_t* keys, uint32_t* idx, hipStream_t s)
{
    if (n_total == 0) return hipSuccess;
    size_t g = (n_total + 255) / 256;
    const int grid = (int)(g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(k_frame_cellkeys, dim3(grid), dim3(256), 0, s, fv, d_frame_start, n_frames,
                       n_total, mv, poses, keys, idx);
    return hipGetLastError();
}

// gather a frame batch into cell-sorted order (once per registration)
__global__ __launch_bounds__(256) void k_permute3(const float* __restrict__ x,
                                                  const float* __restrict__ y,
                                                  const float* __restrict__ z,
                                                  const uint32_t* __restrict__ order, size_t n,
                                                  float* __restrict__ ox, float* __restrict__ oy,
                                                  float* __restrict__ oz)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t s = order[i];
        ox[i] = x[s];
        oy[i] = y[s];
        oz[i] = z[s];
    }
}

hipError_t launch_permute3(const float* x, const float* y, const float* z, const uint32_t* order,
                           size_t n, float* ox, float* oy, float* oz, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t g = (n + 255) / 256;
    const int grid = (int)(g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(k_permute3, dim3(grid), dim3(256), 0, s, x, y, z, order, n, ox, oy, oz);
    return hipGetLastError();
}

// ============================================================= map increment
// points in voxel (cx,cy,cz), counted up to `enough` (the S*S row pieces are summed until the
// answer to "fewer than min_count?" is known: on a mapped surface that is one or two pieces)
__device__ __forceinline__ int voxel_count(const MapView& mv, int cx, int cy, int cz, int enough)
{
    int occ = 0;
    for (int fz = cz * mv.S; fz < (cz + 1) * mv.S && occ < enough; ++fz)
        for (int fy = cy * mv.S; fy < (cy + 1) * mv.S && occ < enough; ++fy) {
            const size_t row = ((size_t)fz * mv.fy + fy) * mv.fx;
            int j0, j1;
            if (row_range_rt(mv, row, cx * mv.S, (cx + 1) * mv.S - 1, j0, j1)) occ += j1 - j0;
        }
    return occ;
}

__global__ __launch_bounds__(256) void k_increment_flags(const float* __restrict__ x,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ z, size_t n,
                                                         MapView mv,
                                                         const double* __restrict__ pose,
                                                         int min_count,
                                                         uint32_t* __restrict__ flags)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        double px, py, pz;
        xform(pose, x[i], y[i], z[i], px, py, pz);
        const int cx = cell_coord((float)px, mv.ox, mv.inv_h, mv.nx);
        const int cy = cell_coord((float)py, mv.oy, mv.inv_h, mv.ny);
        const int cz = cell_coord((float)pz, mv.oz, mv.inv_h, mv.nz);
        int occ = 0;
        if (cx >= 0 && cx < mv.nx && cy >= 0 && cy < mv.ny && cz >= 0 && cz < mv.nz)
            occ = voxel_count(mv, cx, cy, cz, min_count);
        flags[i] = occ < min_count ? 1u : 0u;
    }
}

hipError_t launch_increment_flags(const float* x, const float* y, const float* z, size_t n,
                                  const MapView& mv, const double* pose, int min_count,
                                  uint32_t* flags, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t g = (n + 255) / 256;
    const int grid = (int)(g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(k_increment_flags, dim3(grid), dim3(256), 0, s, x, y, z, n, mv, pose,
                       min_count, flags);
    return hipGetLastError();
}

// ---- the same in two launches instead of six (flags, memset, two scan kernels, count copy,
// scatter) for frame-sized inputs: pass 1 flags kIncTile points per workgroup and counts them; pass 2
// gives every workgroup its base (the sum of the counts in front of it -- a few hundred numbers,
// summed again by every workgroup rather than scanned by a launch of its own), scans its own
// flags and scatters.  Same flags, same order: the same increment bit for bit.
static_assert(kIncTilePoints % 256 == 0, "a tile is a whole number of 256-point strips");
constexpr int kIncTile = (int)kIncTilePoints;  // (1 024: the flag pass takes 20 us instead of 14 on a frame -- a quarter of the workgroups)
__global__ __launch_bounds__(256) void k_increment_flags_count(const float* __restrict__ x,
                                                               const float* __restrict__ y,
                                                               const float* __restrict__ z, uint32_t n,
                                                               MapView mv, const double* __restrict__ pose,
                                                               int min_count, uint32_t* __restrict__ flags,
                                                               uint32_t* __restrict__ block_cnt)
{
    __shared__ uint32_t s_c[4];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kIncTile / 256; ++k) {
        const uint32_t i = blockIdx.x * kIncTile + k * 256 + threadIdx.x;
        bool take = false;
        if (i < n) {
            double px, py, pz;
            xform(pose, x[i], y[i], z[i], px, py, pz);
            const int cx = cell_coord((float)px, mv.ox, mv.inv_h, mv.nx);
            const int cy = cell_coord((float)py, mv.oy, mv.inv_h, mv.ny);
            const int cz = cell_coord((float)pz, mv.oz, mv.inv_h, mv.nz);
            int occ = 0;
            if (cx >= 0 && cx < mv.nx && cy >= 0 && cy < mv.ny && cz >= 0 && cz < mv.nz)
                occ = voxel_count(mv, cx, cy, cz, min_count);
            take = occ < min_count;
            flags[i] = take ? 1u : 0u;
        }
        mine += (uint32_t)__popcll(__ballot(take));  // (the same number in every lane of the wavefront)
    }
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}

__global__ __launch_bounds__(256) void k_increment_scatter_tiles(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, uint32_t n,
    const double* __restrict__ pose, const uint32_t* __restrict__ flags,
    const uint32_t* __restrict__ block_cnt, float* __restrict__ ox, float* __restrict__ oy,
    float* __restrict__ oz, uint32_t* __restrict__ total)
{
    __shared__ uint32_t s_w[4], s_base;
    // base of this tile = the counts of the tiles in front of it (block_cnt sits in L2)
    uint32_t part = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 256) part += block_cnt[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (blockIdx.x == gridDim.x - 1) *total = s_base + block_cnt[blockIdx.x];
    }
    __syncthreads();
    uint32_t run = s_base;  // kept points in front of the 256-point strip being placed
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
    for (int k = 0; k < kIncTile / 256; ++k) {
        const uint32_t i = blockIdx.x * kIncTile + k * 256 + threadIdx.x;
        const bool take = i < n && flags[i];
        const unsigned long long m = __ballot(take);
        __syncthreads();  // (s_w is reused from strip to strip)
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0;  // kept in the wavefronts of this strip in front of mine
        for (int w = 0; w < wave; ++w) before += s_w[w];
        if (take) {
            const uint32_t o = run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            double px, py, pz;
            xform(pose, x[i], y[i], z[i], px, py, pz);
            ox[o] = (float)px;
            oy[o] = (float)py;
            oz[o] = (float)pz;
        }
        run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    }
}

// flags + order-preserving scatter of one frame's increment in two launches; n_tiles workgroups
// each; d_block_cnt: n_tiles counters; d_total: the count, on the device
hipError_t launch_increment_fused(const float* x, const float* y, const float* z, uint32_t n, const MapView& mv,
                                  const double* pose, int min_count, uint32_t* flags, uint32_t* d_block_cnt,
                                  float* ox, float* oy, float* oz, uint32_t* d_total, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const uint32_t tiles = (n + kIncTile - 1) / kIncTile;
    hipLaunchKernelGGL(k_increment_flags_count, dim3(tiles), dim3(256), 0, s, x, y, z, n, mv, pose, min_count, flags,
                       d_block_cnt);
    hipLaunchKernelGGL(k_increment_scatter_tiles, dim3(tiles), dim3(256), 0, s, x, y, z, n, pose, flags, d_block_cnt,
                       ox, oy, oz, d_total);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_increment_scatter(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, size_t n,
    const double* __restrict__ pose, const uint32_t* __restrict__ flags,
    const uint32_t* __restrict__ offs, float* __restrict__ ox, float* __restrict__ oy,
    float* __restrict__ oz)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        if (!flags[i]) continue;
        double px, py, pz;
        xform(pose, x[i], y[i], z[i], px, py, pz);
        const uint32_t o = offs[i];
        ox[o] = (float)px;
        oy[o] = (float)py;
        oz[o] = (float)pz;
    }
}

hipError_t launch_increment_scatter(const float* x, const float* y, const float* z, size_t n,
                                    const double* pose, const uint32_t* flags,
                                    const uint32_t* offs, float* ox, float* oy, float* oz,
                                    hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t g = (n + 255) / 256;
    const int grid = (int)(g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(k_increment_scatter, dim3(grid), dim3(256), 0, s, x, y, z, n, pose, flags,
                       offs, ox, oy, oz);
    return hipGetLastError();
}

// ---- the same for every resident frame at once (one workgroup per work item of the
// linearise decomposition, each frame at its own pose): the concatenation, in frame order,
// of the per-frame increments against the same map snapshot
__global__ __launch_bounds__(kLinThreads) void k_increment_flags_items(
    const BlockItem* __restrict__ items, FrameView fv, MapView mv, const double* __restrict__ poses,
    int min_count, uint32_t* __restrict__ flags)
{
    const BlockItem it = items[blockIdx.x];
    const double* __restrict__ T = poses + 12 * (size_t)it.frame;
    for (int q = it.q0 + (int)threadIdx.x; q < it.q1; q += kLinThreads) {
        double px, py, pz;
        xform(T, fv.x[q], fv.y[q], fv.z[q], px, py, pz);
        const int cx = cell_coord((float)px, mv.ox, mv.inv_h, mv.nx);
        const int cy = cell_coord((float)py, mv.oy, mv.inv_h, mv.ny);
        const int cz = cell_coord((float)pz, mv.oz, mv.inv_h, mv.nz);
        int occ = 0;
        if (cx >= 0 && cx < mv.nx && cy >= 0 && cy < mv.ny && cz >= 0 && cz < mv.nz)
            occ = voxel_count(mv, cx, cy, cz, min_count);
        flags[q] = occ < min_count ? 1u : 0u;
    }
}

__global__ __launch_bounds__(kLinThreads) void k_increment_scatter_items(
    const BlockItem* __restrict__ items, FrameView fv, const double* __restrict__ poses,
    const uint32_t* __restrict__ flags, const uint32_t* __restrict__ offs, float* __restrict__ ox,
    float* __restrict__ oy, float* __restrict__ oz)
{
    const BlockItem it = items[blockIdx.x];
    const double* __restrict__ T = poses + 12 * (size_t)it.frame;
    for (int q = it.q0 + (int)threadIdx.x; q < it.q1; q += kLinThreads) {
        if (!flags[q]) continue;
        double px, py, pz;
        xform(T, fv.x[q], fv.y[q], fv.z[q], px, py, pz);
        const uint32_t o = offs[q];
        ox[o] = (float)px;
        oy[o] = (float)py;
        oz[o] = (float)pz;
    }
}

hipError_t launch_increment_flags_items(const BlockItem* items, int n_items, const FrameView& fv,
                                        const MapView& mv, const double* poses, int min_count,
                                        uint32_t* flags, hipStream_t s)
{
    if (n_items == 0) return hipSuccess;
    hipLaunchKernelGGL(k_increment_flags_items, dim3(n_items), dim3(kLinThreads), 0, s, items, fv, mv,
                       poses, min_count, flags);
    return hipGetLastError();
}

hipError_t launch_increment_scatter_items(const BlockItem* items, int n_items, const FrameView& fv,
                                          const double* poses, const uint32_t* flags,
                                          const uint32_t* offs, float* ox, float* oy, float* oz,
                                          hipStream_t s)
{
    if (n_items == 0) return hipSuccess;
    hipLaunchKernelGGL(k_increment_scatter_items, dim3(n_items), dim3(kLinThreads), 0, s, items, fv,
                       poses, flags, offs, ox, oy, oz);
    return hipGetLastError();
}

// ================================================== voxel-downsampled insertion
// oracle/icp.c vo_roll_filter_sparse in parallel: key = voxel on the unbounded grid; a stable
// sort groups the new points by voxel in their original order, so the rank of a point inside
// its group is the number of new points of that voxel before it, and it is accepted iff
// map count + rank < min_count.
__device__ __forceinline__ int sparse_coord(float p, float o, float inv_h)
{
    float f = floorf((p - o) * inv_h);
    f = f < -1048576.0f ? -1048576.0f : (f > 1048575.0f ? 1048575.0f : f);
    return (int)f;
}

__global__ __launch_bounds__(256) void k_sparse_keys(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ z, size_t n, MapView mv,
                                                     uint64_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int cx = sparse_coord(x[i], mv.ox, mv.inv_h), cy = sparse_coord(y[i], mv.oy, mv.inv_h),
                  cz = sparse_coord(z[i], mv.oz, mv.inv_h);
        keys[i] = ((uint64_t)(cz + 1048576) << 42) | ((uint64_t)(cy + 1048576) << 21) | (uint64_t)(cx + 1048576);
        idx[i] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_sparse_accept(const uint64_t* __restrict__ ks,
                                                       const uint32_t* __restrict__ is, size_t n, MapView mv,
                                                       int min_count, uint32_t* __restrict__ accept)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t k = ks[i];
        size_t lo = 0, hi = i;  // first position holding key k
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (ks[mid] < k) lo = mid + 1; else hi = mid;
        }
        const int rank = (int)(i - lo);
        int have = min_count;  // enough: a rank >= min_count is rejected whatever the map holds
        if (rank < min_count) {
            const int cx = (int)(k & 0x1FFFFF) - 1048576, cy = (int)((k >> 21) & 0x1FFFFF) - 1048576,
                      cz = (int)((k >> 42) & 0x1FFFFF) - 1048576;
            have = 0;
            if (cx >= 0 && cx < mv.nx && cy >= 0 && cy < mv.ny && cz >= 0 && cz < mv.nz)
                have = voxel_count(mv, cx, cy, cz, min_count);
        }
        accept[is[i]] = (have + rank < min_count) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_compact3(const float* __restrict__ x, const float* __restrict__ y,
                                                  const float* __restrict__ z, size_t n,
                                                  const uint32_t* __restrict__ flags,
                                                  const uint32_t* __restrict__ offs, float* __restrict__ ox,
                                                  float* __restrict__ oy, float* __restrict__ oz)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (!flags[i]) continue;
        const uint32_t o = offs[i];
        ox[o] = x[i];
        oy[o] = y[i];
        oz[o] = z[i];
    }
}

static inline int grid1d(size_t n, int cap)
{
    const size_t g = (n + 255) / 256;
    return (int)(g > (size_t)cap ? (size_t)cap : (g ? g : 1));
}

hipError_t launch_sparse_keys(const float* x, const float* y, const float* z, size_t n, const MapView& mv,
                              uint64_t* keys, uint32_t* idx, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sparse_keys, dim3(grid1d(n, 4096)), dim3(256), 0, s, x, y, z, n, mv, keys, idx);
    return hipGetLastError();
}

hipError_t launch_sparse_accept(const uint64_t* keys_sorted, const uint32_t* idx_sorted, size_t n,
                                const MapView& mv, int min_count, uint32_t* accept, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sparse_accept, dim3(grid1d(n, 4096)), dim3(256), 0, s, keys_sorted, idx_sorted, n, mv,
                       min_count, accept);
    return hipGetLastError();
}

hipError_t launch_compact3(const float* x, const float* y, const float* z, size_t n, const uint32_t* flags,
                           const uint32_t* offs, float* ox, float* oy, float* oz, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_compact3, dim3(grid1d(n, 4096)), dim3(256), 0, s, x, y, z, n, flags, offs, ox, oy, oz);
    return hipGetLastError();
}

}  // namespace velo
