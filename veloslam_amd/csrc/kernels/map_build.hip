// map_build.hip -- voxel-sorted map construction for gfx950 (done once per map
// update, not per ICP iteration): bounds, cell keys, gather into float4, cell
// table, and per-point PCA normals.  Semantics: DESIGN.md "ICP semantics"
// (grid / normals), checked bit-for-bit against oracle/icp.c in tests/.
// The stable key sort itself lives in sortscan.hip (rocPRIM radix sort).
#include "device_math.hpp"
#include "normal_math.hpp"

namespace velo {

// ------------------------------------------------------------------ bounds
__global__ __launch_bounds__(256) void k_minmax(const float* __restrict__ x,
                                                const float* __restrict__ y,
                                                const float* __restrict__ z, size_t n,
                                                unsigned* __restrict__ out6)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const float v[3] = {x[i], y[i], z[i]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = fminf(mn[a], v[a]);
            mx[a] = fmaxf(mx[a], v[a]);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_down(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off, 64));
        }
    }
    // waves -> block through LDS, then ONE set of atomics per block (thousands of waves
    // hammering six addresses cost 0.5 ms on a 1 M-point map)
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_mn[wave][a] = mn[a];
            s_mx[wave][a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float lo = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
        const float hi = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
        atomicMin(&out6[a], enc_f32(lo));
        atomicMax(&out6[3 + a], enc_f32(hi));
    }
}

hipError_t launch_minmax(const float* x, const float* y, const float* z, size_t n,
                         unsigned* d_scratch6, MinMax* out_host, hipStream_t s)
{
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    hipError_t e = hipMemcpyAsync(d_scratch6, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_minmax, dim3(grid), dim3(256), 0, s, x, y, z, n, d_scratch6);
    unsigned h[6];
    e = hipMemcpyAsync(h, d_scratch6, sizeof h, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    for (int a = 0; a < 3; ++a) {
        out_host->mn[a] = dec_f32(h[a]);
        out_host->mx[a] = dec_f32(h[3 + a]);
    }
    return hipGetLastError();
}

// -------------------------------------------------------------------- keys
// fine coordinate of a map point: voxel c = floorf(u), sub-cell s = min(S-1, floorf((u-c)*S))
__device__ __forceinline__ int fine_coord(float p, float o, float inv_h, int S)
{
    const float u = (p - o) * inv_h;
    const float c = floorf(u);
    int sub = (int)floorf((u - c) * (float)S);
    sub = min(max(sub, 0), S - 1);
    return (int)c * S + sub;
}

__global__ __launch_bounds__(256) void k_keys(const float* __restrict__ x,
                                              const float* __restrict__ y,
                                              const float* __restrict__ z, size_t n, float ox,
                                              float oy, float oz, float inv_h, int S, int fx,
                                              int fy, uint32_t* __restrict__ keys,
                                              uint32_t* __restrict__ idx)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const int cx = fine_coord(x[i], ox, inv_h, S);
        const int cy = fine_coord(y[i], oy, inv_h, S);
        const int cz = fine_coord(z[i], oz, inv_h, S);
        keys[i] = (uint32_t)(((size_t)cz * fy + cy) * fx + cx);
        idx[i] = (uint32_t)i;
    }
}

hipError_t launch_keys(const float* x, const float* y, const float* z, size_t n, float ox, float oy,
                       float oz, float inv_h, int S, int fx, int fy, uint32_t* keys, uint32_t* idx,
                       hipStream_t s)
{
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_keys, dim3(grid), dim3(256), 0, s, x, y, z, n, ox, oy, oz, inv_h, S, fx,
                       fy, keys, idx);
    return hipGetLastError();
}

// ------------------------------------------------------------------ gather
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ x,
                                                const float* __restrict__ y,
                                                const float* __restrict__ z,
                                                const uint32_t* __restrict__ perm, size_t n,
                                                float4* __restrict__ pts)
{
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < n;
         s += (size_t)gridDim.x * blockDim.x) {
        const uint32_t i = perm[s];
        pts[s] = make_float4(x[i], y[i], z[i], 0.0f);
    }
}

hipError_t launch_gather(const float* x, const float* y, const float* z, const uint32_t* perm,
                         size_t n, float4* pts, hipStream_t s)
{
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, s, x, y, z, perm, n, pts);
    return hipGetLastError();
}

// -------------------------------------------------------------- cell table
// cell_start[c] = number of sorted keys < c  (lower bound), c in [0, ncell]
// One workgroup per tile of kCsTile consecutive entries: the first key at or beyond every tile boundary comes from a
// small kernel of its own (one thread per boundary), the tile's keys -- a few hundred -- are staged in LDS, and every
// entry is a short binary search there.  (The per-entry binary search over
// the whole key array this replaces ran 23 dependent global loads per entry: 1.5 ms on the stream's 78 M-entry
// table, 0.2 TB/s; a re-anchoring roll pays it.)
constexpr int kCsTile = 4096;
constexpr int kCsCap = 4096;
// tb[t] = number of keys below entry t * kCsTile (t = 0 .. number of tiles): one thread per tile boundary
__global__ __launch_bounds__(256) void k_cs_bounds(const uint32_t* __restrict__ keys, size_t n, size_t n_entries,
                                                   uint32_t* __restrict__ tb, size_t n_bounds)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_bounds) return;
    const size_t c = min(t * (size_t)kCsTile, n_entries);
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if ((size_t)keys[mid] < c) lo = mid + 1; else hi = mid;
    }
    tb[t] = (uint32_t)lo;
}
__global__ __launch_bounds__(256) void k_cell_start(const uint32_t* __restrict__ keys, size_t n,
                                                    size_t ncell, int32_t* __restrict__ cell_start,
                                                    const uint32_t* __restrict__ tb)
{
    __shared__ uint32_t s_keys[kCsCap];
    const size_t n_entries = ncell + 1;
    const size_t ntile = (n_entries + kCsTile - 1) / kCsTile;
    for (size_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const size_t c0 = tile * kCsTile;
        const size_t c1 = c0 + kCsTile < n_entries ? c0 + kCsTile : n_entries;
        const size_t a = tb[tile], b = tb[tile + 1];  // keys [a, b) are the tile's
        const size_t cnt = b - a;
        const bool staged = cnt <= (size_t)kCsCap;
        if (staged)
            for (size_t i = threadIdx.x; i < cnt; i += 256) s_keys[i] = keys[a + i];
        __syncthreads();
        for (size_t c = c0 + threadIdx.x; c < c1; c += 256) {
            size_t lo = 0, hi = cnt;
            if (staged) {
                while (lo < hi) {
                    const size_t mid = (lo + hi) >> 1;
                    if ((size_t)s_keys[mid] < c) lo = mid + 1; else hi = mid;
                }
            } else {
                while (lo < hi) {
                    const size_t mid = (lo + hi) >> 1;
                    if ((size_t)keys[a + mid] < c) lo = mid + 1; else hi = mid;
                }
            }
            cell_start[c] = (int32_t)(a + lo);
        }
        __syncthreads();  // (the next tile's keys overwrite the stage)
    }
}

// ---- sparse table ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count_runs(const uint32_t* __restrict__ keys, size_t n, uint32_t S,
                                                    unsigned long long* __restrict__ total)
{
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        c += (i == 0 || keys[i] / S != keys[i - 1] / S) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, (unsigned long long)c);
}
// Row-piece hash (MapView): the first point of every run of equal PIECE keys (fine key / S: the S fine cells of a voxel
// along a fine row, consecutive in the sorted order) inserts {piece key, start, offsets}: o_c = points of the piece in
// sub-cells < c, by a binary search for the first key >= piece * S + c; the slot is claimed with a compare-and-swap on
// the key word.  A piece of 65 536 points or more does not fit its 16-bit offsets: *overflow is raised.
__global__ __launch_bounds__(256) void k_hash_build(const uint32_t* __restrict__ keys, size_t n,
                                                    int4* __restrict__ hash, uint32_t cap, uint32_t S, uint32_t stride,
                                                    unsigned* __restrict__ overflow)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t pk = keys[i] / S;
        if (i != 0 && keys[i - 1] / S == pk) continue;
        uint32_t off[9];
        off[0] = 0;
        size_t from = i;
        for (uint32_t c = 1; c <= S; ++c) {   // first index of the run whose key is >= pk * S + c (64-bit: pk * S + S may pass 2^32)
            const unsigned long long want = (unsigned long long)pk * S + c;
            size_t lo = from, hi = n;
            while (lo < hi) {
                const size_t mid = (lo + hi) >> 1;
                if ((unsigned long long)keys[mid] >= want) hi = mid; else lo = mid + 1;
            }
            off[c] = (uint32_t)(lo - i);
            from = lo;
        }
        for (uint32_t c = S + 1; c <= 8; ++c) off[c] = off[S];
        if (off[S] >= 65536u) atomicOr(overflow, 1u);
        uint32_t h = (uint32_t)(((unsigned long long)(pk * 0x9E3779B1u) * cap) >> 32);
        for (;;) {
            const unsigned prev = atomicCAS(reinterpret_cast<unsigned*>(&hash[(size_t)h * stride].x), 0xffffffffu, pk);
            if (prev == 0xffffffffu) break;
            h = h + 1 == cap ? 0u : h + 1;
        }
        int4& e = hash[(size_t)h * stride];
        e.y = (int)i;
        e.z = (int)((off[1] & 0xffffu) | (off[2] << 16));
        e.w = (int)((off[3] & 0xffffu) | (off[4] << 16));
        if (stride > 1)
            hash[(size_t)h * stride + 1] = make_int4((int)((off[5] & 0xffffu) | (off[6] << 16)),
                                                     (int)((off[7] & 0xffffu) | (off[8] << 16)), 0, 0);
    }
}
hipError_t launch_count_runs(const uint32_t* sorted_keys, size_t n, int S, unsigned long long* d_count, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess || n == 0) return e;
    const size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(k_count_runs, dim3((int)(g > 8192 ? 8192 : g)), dim3(256), 0, s, sorted_keys, n, (uint32_t)S, d_count);
    return hipGetLastError();
}
hipError_t launch_hash_build(const uint32_t* sorted_keys, size_t n, int4* hash, uint32_t cap, int S, unsigned* d_overflow,
                             hipStream_t s)
{
    const uint32_t stride = S > 4 ? 2u : 1u;
    hipError_t e = hipMemsetAsync(hash, 0xFF, (size_t)cap * stride * sizeof(int4), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_overflow, 0, sizeof(unsigned), s);
    if (e != hipSuccess || n == 0) return e;
    const size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(k_hash_build, dim3((int)(g > 16384 ? 16384 : g)), dim3(256), 0, s, sorted_keys, n, hash,
                       cap, (uint32_t)S, stride, d_overflow);
    return hipGetLastError();
}

size_t cell_start_bounds(size_t ncell) { return (ncell + 1 + kCsTile - 1) / kCsTile + 1; }

hipError_t launch_cell_start(const uint32_t* sorted_keys, size_t n, size_t ncell,
                             int32_t* cell_start, uint32_t* tile_scratch, hipStream_t s)
{
    const size_t nb = cell_start_bounds(ncell);
    hipLaunchKernelGGL(k_cs_bounds, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, sorted_keys, n, ncell + 1,
                       tile_scratch, nb);
    size_t g = nb - 1;
    int grid = (int)(g > 16384 ? 16384 : g);
    hipLaunchKernelGGL(k_cell_start, dim3(grid), dim3(256), 0, s, sorted_keys, n, ncell,
                       cell_start, tile_scratch);
    return hipGetLastError();
}

// ----------------------------------------------------------------- normals
// One thread per map point.  The running k-best list lives in LDS as
// [slot][thread] (stride = blockDim: bank-conflict free), because a
// dynamically indexed per-thread array would otherwise go to scratch memory.
#ifndef VELO_NRM_W
#define VELO_NRM_W 4  // candidate loads in flight per trip of the neighbour search
#endif
#ifndef VELO_NRM_THREADS
#define VELO_NRM_THREADS 128
#endif
constexpr int kNrmThreads = VELO_NRM_THREADS;

// The k smallest (d2, sorted index) among the points of the 27 voxels around q with
// d2 <= r2, ascending, into the LDS lists s_d/s_i ([slot][thread]).  Returns how many.
// Exact ball search: fine rows are visited centre-out from the query's own row, and once the
// list is full its k-th distance is the search radius -- rows, and the cells of a row, that
// lie outside it are never touched.  The visiting order therefore is not ascending in the
// sorted index, so ties are ordered explicitly by (d2, index): the list is the oracle's.
// Margins keep every pruning test conservative (cell membership is decided in float).
// TIE_RAW: equal distances are ordered by the append-order index perm[j] (normals: grid
// independent) instead of the sorted index j (velo_knn's documented order).
// STATS (velo_knn_dev's counting instantiation only): n_cand += candidate points fetched (16 B each),
// n_rows += fine rows looked up in the table (2 x 4 B dense; one 16-byte slot per cell of the row, hashed).
template <bool TIE_RAW, bool STATS = false>
__device__ __forceinline__ int collect_knn(const MapView& mv, const uint32_t* __restrict__ perm,
                                           float qx, float qy, float qz, float r2, int k,
                                           float (*s_d)[kNrmThreads], int (*s_i)[kNrmThreads],
                                           int tid, unsigned* n_cand = nullptr, unsigned* n_rows = nullptr,
                                           unsigned* n_cells = nullptr)
{
    auto before = [&](float d2, int j, float pd, int pi) -> bool {
        if (d2 != pd) return d2 < pd;
        return TIE_RAW ? perm[j] < perm[pi] : j < pi;
    };
    const int cx = cell_coord(qx, mv.ox, mv.inv_h, mv.nx);
    const int cy = cell_coord(qy, mv.oy, mv.inv_h, mv.ny);
    const int cz = cell_coord(qz, mv.oz, mv.inv_h, mv.nz);
    int cnt = 0;
    float bound = r2;  // min(r2, d2 of the last slot once the list is full)
    const int vx0 = max(cx - 1, 0), vx1 = min(cx + 1, mv.nx - 1);
    const int vy0 = max(cy - 1, 0), vy1 = min(cy + 1, mv.ny - 1);
    const int vz0 = max(cz - 1, 0), vz1 = min(cz + 1, mv.nz - 1);
    if (vx0 > vx1 || vy0 > vy1 || vz0 > vz1) return 0;
    const int S = mv.S;
    const float hf = mv.h / (float)S;
    const float inv_hf = mv.inv_h * (float)S;
    const float ux = (qx - mv.ox) * inv_hf, uy = (qy - mv.oy) * inv_hf, uz = (qz - mv.oz) * inv_hf;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    const int x0 = vx0 * S, x1 = (vx1 + 1) * S - 1;  // inclusive fine ranges
    const int y0 = vy0 * S, y1 = (vy1 + 1) * S - 1;
    const int z0 = vz0 * S, z1 = (vz1 + 1) * S - 1;
    const int hy = min(max((int)floorf(fminf(fmaxf(uy, -4.0f), 2.0e9f)), y0), y1);
    const int hz = min(max((int)floorf(fminf(fmaxf(uz, -4.0f), 2.0e9f)), z0), z1);
    for (int dz = 0; dz <= z1 - z0; ++dz) {
        bool any_z = false;
        for (int sz = 0; sz < 2; ++sz) {
            if (dz == 0 && sz) continue;
            const int fz = sz ? hz - dz : hz + dz;
            if (fz < z0 || fz > z1) continue;
            const float gz = fmaxf(fmaxf((float)fz - uz, uz - (float)(fz + 1)) * hf - mg, 0.0f);
            if (gz * gz * 0.99999f > bound) continue;
            any_z = true;
            for (int dy = 0; dy <= y1 - y0; ++dy) {
                bool any_y = false;
                for (int sy = 0; sy < 2; ++sy) {
                    if (dy == 0 && sy) continue;
                    const int fy = sy ? hy - dy : hy + dy;
                    if (fy < y0 || fy > y1) continue;
                    const float gy =
                        fmaxf(fmaxf((float)fy - uy, uy - (float)(fy + 1)) * hf - mg, 0.0f);
                    const float g2 = gz * gz + gy * gy;
                    if (g2 * 0.99999f > bound) continue;
                    any_y = true;
                    // cells of this row that can hold a point within sqrt(bound - g2) in x
                    const float xr = (sqrtf(fmaxf(bound - g2 * 0.99999f, 0.0f)) * 1.00001f + 2.0f * mg) * inv_hf;
                    const int fa = max(x0, (int)floorf(fmaxf(ux - xr, -4.0f)));
                    const int fb = min(x1, (int)floorf(fminf(ux + xr, 2.0e9f)));
                    if (fa > fb) continue;
                    const size_t row = ((size_t)fz * mv.fy + fy) * mv.fx;
                    int j0, j1;
                    if constexpr (STATS) {
                        *n_rows += 1;
                        *n_cells += (unsigned)(fb - fa + 1);
                    }
                    if (!row_range_rt(mv, row, fa, fb, j0, j1)) continue;
                    if constexpr (STATS) *n_cand += (unsigned)(j1 - j0);
                    // four candidate loads in flight per trip (the walk is a latency chain);
                    // a slot past the end repeats the last index and is masked
                    for (int jb = j0; jb < j1; jb += VELO_NRM_W) {
                        float4 cpt[VELO_NRM_W];
#pragma unroll
                        for (int u = 0; u < VELO_NRM_W; ++u) cpt[u] = mv.pts[min(jb + u, j1 - 1)];
#pragma unroll
                        for (int u = 0; u < VELO_NRM_W; ++u) {
                            const int j = jb + u;
                            const float d2 = dist2(cpt[u], qx, qy, qz);
                            if (j >= j1 || !(d2 <= r2)) continue;
                            if (cnt == k && !before(d2, j, s_d[k - 1][tid], s_i[k - 1][tid])) continue;
                            int pos = cnt < k ? cnt : k - 1;
                            while (pos > 0) {
                                const float pd = s_d[pos - 1][tid];
                                const int pi = s_i[pos - 1][tid];
                                if (!before(d2, j, pd, pi)) break;
                                s_d[pos][tid] = pd;
                                s_i[pos][tid] = pi;
                                --pos;
                            }
                            s_d[pos][tid] = d2;
                            s_i[pos][tid] = j;
                            if (cnt < k) ++cnt;
                            if (cnt == k) bound = fminf(r2, s_d[k - 1][tid]);
                        }
                    }
                }
                if (!any_y) break;  // gaps only grow with dy and the bound only shrinks
            }
        }
        if (!any_z) break;
    }
    return cnt;
}

// PCA normal of sorted point s: xyz = {0,0,0} = invalid (fewer than kMinNb neighbours within
// h).  w carries the squared distance of the k-th neighbour used (h^2 when the list is not
// full): a point added or removed further away than that cannot change this normal, which is
// what the incremental update tests before re-estimating (w < 0 marks "no normal yet").
__device__ __forceinline__ float4 point_normal(const MapView& mv, const uint32_t* __restrict__ perm,
                                               int s, int k, float (*s_d)[kNrmThreads],
                                               int (*s_i)[kNrmThreads], int tid)
{
    const float4 q = mv.pts[s];
    // neighbour radius 0.99 h: strictly inside one voxel, see oracle/icp.c point_normal
    const float rn = kNormalRadius * mv.h;
    const float r2 = rn * rn;
    const int cnt = collect_knn<true>(mv, perm, q.x, q.y, q.z, r2, k, s_d, s_i, tid);
    const float rk2 = cnt == k ? s_d[k - 1][tid] : r2;
    return pca_normal(cnt, rk2, [&](int i) { return mv.pts[s_i[i][tid]]; });
}

__device__ __forceinline__ bool is_zero3(const float4& v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f; }

// KMAX = capacity of the per-thread neighbour list in LDS (8, 16 or 32 slots: 8 bytes x KMAX x
// 128 threads per workgroup); the launcher picks the smallest that holds k, which doubles or
// quadruples the workgroups a CU can hold for the usual k = 16 / k = 8.
template <int KMAX>
__global__ __launch_bounds__(kNrmThreads) void k_normals(MapView mv,
                                                         const uint32_t* __restrict__ perm, int k,
                                                         float4* __restrict__ nrm,
                                                         unsigned long long* __restrict__ invalid)
{
    __shared__ float s_d[KMAX][kNrmThreads];
    __shared__ int s_i[KMAX][kNrmThreads];
    const int tid = threadIdx.x;
    const int s = blockIdx.x * kNrmThreads + tid;
    if (s >= mv.n) return;
    const float4 nv = point_normal(mv, perm, s, k, s_d, s_i, tid);
    nrm[s] = nv;
    if (is_zero3(nv)) atomicAdd(invalid, 1ull);
}

// Is a changed (added / removed) point possibly within sqrt(rk2) of p?  chg = sorted fine keys
// of the changed points; one binary search per fine row that the ball touches.  Conservative.
__device__ __forceinline__ bool near_changed(const MapView& mv, const float4& p, float rk2,
                                             const uint32_t* __restrict__ chg, uint32_t m)
{
    const float inv_hf = mv.inv_h * (float)mv.S;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    const float r = (sqrtf(rk2) * 1.00001f + 2.0f * mg) * inv_hf;  // fine-cell units
    const float ux = (p.x - mv.ox) * inv_hf, uy = (p.y - mv.oy) * inv_hf, uz = (p.z - mv.oz) * inv_hf;
    const int x0 = max((int)floorf(ux - r), 0), x1 = min((int)floorf(ux + r), mv.fx - 1);
    const int y0 = max((int)floorf(uy - r), 0), y1 = min((int)floorf(uy + r), mv.fy - 1);
    const int z0 = max((int)floorf(uz - r), 0), z1 = min((int)floorf(uz + r), mv.fz - 1);
    if (x0 > x1) return false;
    for (int fz = z0; fz <= z1; ++fz)
        for (int fy = y0; fy <= y1; ++fy) {
            const uint32_t klo = (uint32_t)(((size_t)fz * mv.fy + fy) * mv.fx + x0);
            const uint32_t khi = klo + (uint32_t)(x1 - x0);
            uint32_t lo = 0, hi = m;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (chg[mid] < klo)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo < m && chg[lo] <= khi) return true;
        }
    return false;
}

// Incremental update: re-estimate the normals of the listed sorted points only.  nrm[] holds
// the previous normal of every surviving point (w = squared reach of its neighbour list) and
// w < 0 for points that are new.  A listed point whose reach no changed point can touch keeps
// its normal untouched (chg == nullptr: no such test).  The invalid-normal count is
// maintained by difference (two's complement add).
#ifndef VELO_NRM_SUBSET_WAVES
#define VELO_NRM_SUBSET_WAVES 4  // wavefronts per SIMD the register allocation aims at (3 = 133 VGPRs as the compiler chooses freely)
#endif
template <int KMAX>
__global__ __launch_bounds__(kNrmThreads) __attribute__((amdgpu_waves_per_eu(KMAX <= 16 ? VELO_NRM_SUBSET_WAVES : 2, KMAX <= 16 ? VELO_NRM_SUBSET_WAVES : 3))) void k_normals_subset(
    MapView mv, const uint32_t* __restrict__ perm, int k, const int32_t* __restrict__ work,
    int n_work, const unsigned* __restrict__ n_work_dev, const uint32_t* __restrict__ chg, uint32_t n_chg,
    float4* __restrict__ nrm, unsigned long long* __restrict__ invalid, unsigned* __restrict__ n_done)
{
    __shared__ float s_d[KMAX][kNrmThreads];
    __shared__ int s_i[KMAX][kNrmThreads];
    const int tid = threadIdx.x;
    // (n_work_dev: the length of the work list is still on the device -- a roll enqueued without a host wait.  The
    //  grid is then a bounded one and strides over the list: a grid sized for the upper bound, every point of the
    //  map, was 91 000 workgroups of which 90 000 left at once -- and held the dispatcher for 0.8 ms while a
    //  registration on the other stream waited for slots, profiles/r05)
    const int nw = n_work_dev ? (int)min(*n_work_dev, (unsigned)n_work) : n_work;
    for (int w = blockIdx.x * kNrmThreads + tid; w < nw; w += gridDim.x * kNrmThreads) {
        const int s = work[w];
        const float4 old = nrm[s];
        if (chg && old.w >= 0.0f && !near_changed(mv, mv.pts[s], old.w, chg, n_chg)) continue;
        const float4 nv = point_normal(mv, perm, s, k, s_d, s_i, tid);
        nrm[s] = nv;
        const int was = (old.w >= 0.0f && is_zero3(old)) ? 1 : 0;
        const int now = is_zero3(nv) ? 1 : 0;
        if (now != was) atomicAdd(invalid, (unsigned long long)(long long)(now - was));
        if (n_done) atomicAdd(n_done, 1u);
    }
}

// sorted fine keys of the points an eviction removes (keep == 0), compacted in order
__global__ __launch_bounds__(256) void k_removed_keys(const uint32_t* __restrict__ keys,
                                                      const uint32_t* __restrict__ keep,
                                                      const uint32_t* __restrict__ offs, uint32_t n,
                                                      uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || keep[i]) return;
    out[i - offs[i]] = keys[i];  // offs = kept points before i
}

// a10 with k > 1: the k nearest map points of every (transformed) query within d_max,
// ascending (d2, sorted index); rows of idx/d2 are padded with -1 / +inf.
// [0] queries, [1] candidate points fetched, [2] fine rows looked up, [3] fine cells those rows span
__device__ unsigned long long g_knn_lane_stats[4];

template <int KMAX, bool STATS = false>
__global__ __launch_bounds__(kNrmThreads) void k_knn(MapView mv, const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ z, int n,
                                                     Pose12 T, float dmax2,
                                                     int k, int32_t* __restrict__ idx,
                                                     float* __restrict__ d2o,
                                                     int32_t* __restrict__ count)
{
    __shared__ float s_d[KMAX][kNrmThreads];
    __shared__ int s_i[KMAX][kNrmThreads];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kNrmThreads + tid;
    if (i >= n) return;
    double px, py, pz;
    xform(T.t, x[i], y[i], z[i], px, py, pz);
    unsigned nc = 0, nr = 0, ncell = 0;
    const int cnt = collect_knn<false, STATS>(mv, nullptr, (float)px, (float)py, (float)pz, dmax2, k, s_d,
                                              s_i, tid, &nc, &nr, &ncell);
    for (int m = 0; m < k; ++m) {
        idx[(size_t)i * k + m] = m < cnt ? s_i[m][tid] : -1;
        d2o[(size_t)i * k + m] = m < cnt ? s_d[m][tid] : INFINITY;
    }
    if (count) count[i] = cnt;
    if constexpr (STATS) {
        atomicAdd(&g_knn_lane_stats[0], 1ull);
        atomicAdd(&g_knn_lane_stats[1], (unsigned long long)nc);
        atomicAdd(&g_knn_lane_stats[2], (unsigned long long)nr);
        atomicAdd(&g_knn_lane_stats[3], (unsigned long long)ncell);
    }
}

// which k-NN kernel.  The cooperative one (one wavefront per query) spends a fixed ~600 vector instructions per
// query on its row geometry and its first sort whatever the map holds; the per-lane one runs a chain of dependent
// round trips per candidate row.  Measured on the scene at k = 32 (profiles/r05/knn_density_ab.txt, ms per
// 115 200-query frame, per-lane / cooperative): 50 k points 0.08 / 0.33, 200 k 0.31 / 0.43, 1 M 0.85 / 0.51,
// 10 M 1.31 / 0.34, 30 M 1.61 / 0.26, 100 M 3.4 / 0.19 -- and 100 M at h = 0.5 m 1.75 / 0.40, at h = 0.25 m
// (hashed) 1.66 / 0.68: the crossover follows the points per VOXEL (0.6 -> 2.8 between 200 k and 1 M), not the points
// per fine cell round 4 used.
bool knn_use_wave(const MapView& mv, int mode)
{
    if (mode == 1) return false;
    if (mode == 2) return true;
    return (double)mv.n >= 1.5 * (double)mv.nx * mv.ny * mv.nz;
}
// ... and which normals kernel for a full build: there every point of the map is a query, neighbours in the sorted
// order share their rows in L1 / L2 and the per-lane kernel holds up longer (full build incl. sorts, per-lane /
// cooperative: 1 M 12 / 16 ms, 10 M 25 / 35 ms, 30 M 70 / 69 ms, 100 M 1 600 / 200 ms)
bool normals_use_wave(const MapView& mv, int mode)
{
    if (mode == 1) return false;
    if (mode == 2) return true;
    const double fine_cells = (double)mv.fx * mv.fy * mv.fz;
    return (double)mv.n >= 0.25 * fine_cells;  // (100 M points on 186 M fine cells: 0.54; 30 M: 0.16)
}

hipError_t launch_knn(const MapView& mv, const float* x, const float* y, const float* z, size_t n,
                      const Pose12& T, float dmax2, int k, int32_t* idx, float* d2, int32_t* count,
                      hipStream_t s, unsigned long long* stats_out, int mode)
{
    if (n == 0) return hipSuccess;
    if (knn_use_wave(mv, mode))  // kernels/knn_wave.hip
        return launch_knn_wave(mv, x, y, z, n, T, dmax2, k, idx, d2, count, s, stats_out);
    const int grid = (int)((n + kNrmThreads - 1) / kNrmThreads);
    if (stats_out) {  // counting instantiation (widest list: the LDS footprint is not what is measured here)
        const unsigned long long z4[4] = {0, 0, 0, 0};
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_knn_lane_stats), z4, sizeof z4);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_knn<VELO_MAX_KNORMALS, true>), dim3(grid), dim3(kNrmThreads), 0, s, mv, x, y, z,
                           (int)n, T, dmax2, k, idx, d2, count);
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        return hipMemcpyFromSymbol(stats_out, HIP_SYMBOL(g_knn_lane_stats), sizeof z4);
    }
    if (k <= 8)
        hipLaunchKernelGGL(k_knn<8>, dim3(grid), dim3(kNrmThreads), 0, s, mv, x, y, z, (int)n, T, dmax2,
                           k, idx, d2, count);
    else if (k <= 16)
        hipLaunchKernelGGL(k_knn<16>, dim3(grid), dim3(kNrmThreads), 0, s, mv, x, y, z, (int)n, T, dmax2,
                           k, idx, d2, count);
    else
        hipLaunchKernelGGL(k_knn<VELO_MAX_KNORMALS>, dim3(grid), dim3(kNrmThreads), 0, s, mv, x, y, z,
                           (int)n, T, dmax2, k, idx, d2, count);
    return hipGetLastError();
}

hipError_t launch_normals(const MapView& mv, const uint32_t* perm, int k, float4* nrm,
                          unsigned long long* d_invalid, hipStream_t s, int mode)
{
    // dense maps: one wavefront per point for the search, one lane per point for the PCA (kernels/knn_wave.hip)
    if (normals_use_wave(mv, mode)) return launch_normals_wave(mv, perm, k, nrm, d_invalid, s);
    hipError_t e = hipMemsetAsync(d_invalid, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const int grid = (mv.n + kNrmThreads - 1) / kNrmThreads;
    if (k <= 8)
        hipLaunchKernelGGL(k_normals<8>, dim3(grid), dim3(kNrmThreads), 0, s, mv, perm, k, nrm, d_invalid);
    else if (k <= 16)
        hipLaunchKernelGGL(k_normals<16>, dim3(grid), dim3(kNrmThreads), 0, s, mv, perm, k, nrm, d_invalid);
    else
        hipLaunchKernelGGL(k_normals<VELO_MAX_KNORMALS>, dim3(grid), dim3(kNrmThreads), 0, s, mv, perm, k,
                           nrm, d_invalid);
    return hipGetLastError();
}


// ============================================================ incremental map update (f3)
// The sorted arrays of a map are updated in place of a rebuild when the grid (origin, dims)
// keeps: a stable merge of the new points' sorted keys into the old order, a shifted cell
// table, and normals recomputed only where a changed point is within one voxel.  The result
// is bit-identical to a fresh build on the same grid (oracle: vo_roll / vo_map_build_grid).

__global__ __launch_bounds__(256) void k_keys4(const float4* __restrict__ pts, size_t n, float ox,
                                               float oy, float oz, float inv_h, int S, int fx,
                                               int fy, uint32_t* __restrict__ keys)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const float4 p = pts[i];
        const int cx = fine_coord(p.x, ox, inv_h, S);
        const int cy = fine_coord(p.y, oy, inv_h, S);
        const int cz = fine_coord(p.z, oz, inv_h, S);
        keys[i] = (uint32_t)(((size_t)cz * fy + cy) * fx + cx);
    }
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t n,
                                                    uint32_t v)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t* __restrict__ a, uint32_t n,
                                                    uint32_t v)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] <= v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// old sorted element i moves right by the number of new keys strictly below its key
// (old points precede new ones inside a cell: the stable order of the concatenated list)
__global__ __launch_bounds__(256) void k_merge_old(
    const float4* __restrict__ pts, const float4* __restrict__ nrm,
    const uint32_t* __restrict__ perm, const uint32_t* __restrict__ keys, uint32_t n,
    const uint32_t* __restrict__ nk, uint32_t m, float4* __restrict__ pts2,
    float4* __restrict__ nrm2, uint32_t* __restrict__ perm2, uint32_t* __restrict__ keys2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t key = keys[i];
    const uint32_t d = i + lower_bound_u32(nk, m, key);
    pts2[d] = pts[i];
    nrm2[d] = nrm[i];
    perm2[d] = perm[i];
    keys2[d] = key;
}

__global__ __launch_bounds__(256) void k_merge_new(
    const float* __restrict__ rx, const float* __restrict__ ry, const float* __restrict__ rz,
    uint32_t raw_base, const uint32_t* __restrict__ nk, const uint32_t* __restrict__ nidx,
    uint32_t m, const uint32_t* __restrict__ keys_old, uint32_t n, float4* __restrict__ pts2,
    float4* __restrict__ nrm2, uint32_t* __restrict__ perm2, uint32_t* __restrict__ keys2)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t key = nk[j];
    const uint32_t d = upper_bound_u32(keys_old, n, key) + j;
    const uint32_t r = raw_base + nidx[j];
    pts2[d] = make_float4(rx[r], ry[r], rz[r], 0.0f);
    nrm2[d] = make_float4(0.f, 0.f, 0.f, -1.0f);  // w < 0: "no normal yet"
    perm2[d] = r;
    keys2[d] = key;
}

// cell_start[c] += number of new keys < c.  The new keys of a rolling update land in a few per
// cent of the table (the strip that entered), so almost every 1 024-entry tile gets ONE shift for
// all its entries: two searches per tile (first and last entry) decide that, and only a tile the
// new keys fall into searches per entry -- inside the tile's own key range.  (Round 2's form did an
// 18-step binary search per four entries over the whole table: 418 us on a 78 M-entry table, against
// the ~130 us its 2 x 311 MB take to stream.)
constexpr int kTableTile = 1024;  // entries per workgroup step: 256 threads x 4
// (src == dst: in place; src != dst: the shifted table is written to dst and src stays as it was --
// a registration still running on the other stream keeps reading src)
// (round 5: the two searches per tile -- 20 dependent loads in front of 8 KB of streaming, 362 us on the 78 M-entry
//  table where the bytes take 150 -- are made for all tiles at once by k_tile_bounds, one thread per tile boundary;
//  a tile then starts with one 8-byte load.  tb[t] = number of new keys below entry t * kTableTile.)
__global__ __launch_bounds__(256) void k_tile_bounds(const uint32_t* __restrict__ nk, uint32_t m, size_t n_entries,
                                                     uint32_t* __restrict__ tb, size_t n_bounds)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_bounds) return;
    const size_t c = min(t * (size_t)kTableTile, n_entries);
    tb[t] = lower_bound_u32(nk, m, (uint32_t)min(c, (size_t)0xffffffffu));
}
__global__ __launch_bounds__(256) void k_table_shift(const int32_t* src, int32_t* dst,
                                                     size_t n_entries,
                                                     const uint32_t* __restrict__ nk, uint32_t m,
                                                     const uint32_t* __restrict__ tb)
{
    const bool oop = src != dst;
    const size_t n_tiles = (n_entries + kTableTile - 1) / kTableTile;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t t0 = tile * kTableTile, t1 = min(t0 + (size_t)kTableTile, n_entries);
        // new keys below the tile's first entry / below the NEXT tile's first entry (an upper limit for this tile's)
        const uint32_t jlo = tb[tile], jhi = tb[tile + 1];
        const size_t c0 = t0 + (size_t)threadIdx.x * 4;
        if (c0 >= t1) continue;
        if (jhi == 0) {  // every entry of the tile lies at or below the first new key: unchanged
            if (oop) {
                if (c0 + 3 < t1) *reinterpret_cast<int4*>(dst + c0) = *reinterpret_cast<const int4*>(src + c0);
                else for (size_t c = c0; c < t1; ++c) dst[c] = src[c];
            }
            continue;
        }
        uint32_t j = jlo;
        if (jhi != jlo) {  // new keys inside the tile: this quad's own count, searched in [jlo, jhi)
            uint32_t lo = jlo, hi = jhi;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((size_t)nk[mid] < c0) lo = mid + 1; else hi = mid;
            }
            j = lo;
        }
        if (c0 + 3 < t1) {
            int4 v = *reinterpret_cast<const int4*>(src + c0);  // rows are 16-byte aligned
            v.x += (int)j;
            while (j < jhi && (size_t)nk[j] < c0 + 1) ++j;
            v.y += (int)j;
            while (j < jhi && (size_t)nk[j] < c0 + 2) ++j;
            v.z += (int)j;
            while (j < jhi && (size_t)nk[j] < c0 + 3) ++j;
            v.w += (int)j;
            *reinterpret_cast<int4*>(dst + c0) = v;
        } else {
            for (size_t c = c0; c < t1; ++c) {
                while (j < jhi && (size_t)nk[j] < c) ++j;
                dst[c] = src[c] + (int)j;
            }
        }
    }
}

// mark the 27 voxels around the voxel of every listed fine key (sel == nullptr: all of
// keys[0..m); else only those with sel[i] == 0, i.e. the removed points of an eviction)
__global__ __launch_bounds__(256) void k_mark_dirty(const uint32_t* __restrict__ keys, uint32_t m,
                                                    const uint32_t* __restrict__ sel, int S,
                                                    int fx, int fy, int nx, int ny, int nz,
                                                    uint8_t* __restrict__ dirty)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    if (sel && sel[i]) return;
    const uint32_t key = keys[i];
    const int Fx = (int)(key % (uint32_t)fx);
    const uint32_t t = key / (uint32_t)fx;
    const int Fy = (int)(t % (uint32_t)fy), Fz = (int)(t / (uint32_t)fy);
    const int vx = Fx / S, vy = Fy / S, vz = Fz / S;
    for (int dz = -1; dz <= 1; ++dz) {
        const int z = vz + dz;
        if (z < 0 || z >= nz) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = vy + dy;
            if (y < 0 || y >= ny) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = vx + dx;
                if (x < 0 || x >= nx) continue;
                dirty[((size_t)z * ny + y) * nx + x] = 1;
            }
        }
    }
}

// work list of the sorted points that live in a dirty voxel (order irrelevant) -- and, where the changed points are
// given (chg: their sorted fine keys), only those of them a changed point can actually reach: a surviving normal
// carries the squared reach of its neighbour list in w, and a point is listed only if a changed point's fine cell
// intersects that ball (near_changed; new points, w < 0, always).  Round 5: the test used to sit at the top of
// k_normals_subset, where one lane that passed it kept its 63 neighbours waiting through a whole k-NN + PCA -- after an
// eviction 8 k of 100 k listed points pass, scattered over nearly every wavefront: 0.66 ms per call.  Filtered HERE,
// the normals kernel gets a list of points that all have work to do.
__global__ __launch_bounds__(256) void k_select_dirty(const uint32_t* __restrict__ keys,
                                                      uint32_t n, MapView mv,
                                                      const uint8_t* __restrict__ dirty,
                                                      const float4* __restrict__ nrm,
                                                      const uint32_t* __restrict__ chg, uint32_t n_chg,
                                                      int32_t* __restrict__ work,
                                                      unsigned* __restrict__ count)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (i < n) {
        const uint32_t key = keys[i];
        const int Fx = (int)(key % (uint32_t)mv.fx);
        const uint32_t t = key / (uint32_t)mv.fx;
        const int Fy = (int)(t % (uint32_t)mv.fy), Fz = (int)(t / (uint32_t)mv.fy);
        hit = dirty[((size_t)(Fz / mv.S) * mv.ny + (Fy / mv.S)) * mv.nx + (Fx / mv.S)] != 0;
        if (hit && chg) {
            const float w = nrm[i].w;
            if (w >= 0.0f && !near_changed(mv, mv.pts[i], w, chg, n_chg)) hit = false;
        }
    }
    const unsigned long long mask = __ballot(hit);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == __ffsll((long long)mask) - 1) base = atomicAdd(count, (unsigned)__popcll(mask));
    base = __shfl(base, __ffsll((long long)mask) - 1, 64);
    if (hit) work[base + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)i;
}

// ---- eviction: keep the points inside the closed box [lo, hi]
// keep region: closed box, optionally intersected with a vertical cylinder (ground-plane
// distance to (cx, cy) <= radius) -- same float expression as the oracle's roll_keep
__device__ __forceinline__ bool keep_pt(const KeepRegion& g, float px, float py, float pz)
{
    if (!(px >= g.lo[0] && px <= g.hi[0] && py >= g.lo[1] && py <= g.hi[1] && pz >= g.lo[2] &&
          pz <= g.hi[2]))
        return false;
    if (!g.use_radius) return true;
    const float dx = px - g.cx, dy = py - g.cy;
    return fmaf(dy, dy, dx * dx) <= g.r2;
}
__global__ __launch_bounds__(256) void k_keep4(const float4* __restrict__ pts, uint32_t n,
                                               KeepRegion g, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    flags[i] = keep_pt(g, p.x, p.y, p.z) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_keep3(const float* __restrict__ x,
                                               const float* __restrict__ y,
                                               const float* __restrict__ z, uint32_t n,
                                               KeepRegion g, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = keep_pt(g, x[i], y[i], z[i]) ? 1u : 0u;
}
// keep flags in append order AND the bounding box of the kept points (what decides whether the grid
// has to be re-anchored): one pass instead of a flag pass + a min/max pass over the compacted list
__global__ __launch_bounds__(256) void k_keep3_minmax(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ z, uint32_t n, KeepRegion g,
                                                      uint32_t* __restrict__ flags, unsigned* __restrict__ out6)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float v[3] = {x[i], y[i], z[i]};
        const bool keep = keep_pt(g, v[0], v[1], v[2]);
        flags[i] = keep ? 1u : 0u;
        if (keep) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                mn[a] = fminf(mn[a], v[a]);
                mx[a] = fmaxf(mx[a], v[a]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_down(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_down(mx[a], off, 64));
        }
    }
    __shared__ float s_mn[4][3], s_mx[4][3];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_mn[wave][a] = mn[a];
            s_mx[wave][a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float lo = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
        const float hi = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
        atomicMin(&out6[a], enc_f32(lo));
        atomicMax(&out6[3 + a], enc_f32(hi));
    }
}
__global__ __launch_bounds__(256) void k_compact_sorted(
    const float4* __restrict__ pts, const float4* __restrict__ nrm,
    const uint32_t* __restrict__ perm, const uint32_t* __restrict__ keys, uint32_t n,
    const uint32_t* __restrict__ flags, const uint32_t* __restrict__ offs,
    const uint32_t* __restrict__ raw_offs, float4* __restrict__ pts2, float4* __restrict__ nrm2,
    uint32_t* __restrict__ perm2, uint32_t* __restrict__ keys2, unsigned long long* __restrict__ invalid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const bool keep = in && flags[i];
    float4 nv = make_float4(1.f, 0.f, 0.f, 0.f);
    if (in) nv = nrm[i];
    // the running count of invalid normals follows the points that leave (the dirty voxels are
    // re-estimated and re-counted afterwards): no pass over the whole map to count again
    if (invalid) {
        const unsigned long long gone = __ballot(in && !keep && nv.w >= 0.0f && is_zero3(nv));
        if (gone && (threadIdx.x & 63) == 0) atomicAdd(invalid, 0ull - (unsigned long long)__popcll(gone));
    }
    if (!keep) return;
    const uint32_t d = offs[i];
    pts2[d] = pts[i];
    nrm2[d] = nv;
    perm2[d] = raw_offs[perm[i]];
    keys2[d] = keys[i];
}
__global__ __launch_bounds__(256) void k_compact_raw(const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ z, uint32_t n,
                                                     const uint32_t* __restrict__ flags,
                                                     const uint32_t* __restrict__ offs,
                                                     float* __restrict__ x2, float* __restrict__ y2,
                                                     float* __restrict__ z2)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const uint32_t d = offs[i];
    x2[d] = x[i];
    y2[d] = y[i];
    z2[d] = z[i];
}
// cell_start[c] (a position in the old order) -> number of kept points before it.  An eviction
// removes points in a few per cent of the table's key range; everywhere else "removed before this
// position" is one number for a whole 1 024-entry tile (positions ascend with c), so the tile
// streams (v - removed) instead of gathering offs[v] per entry.
// (tr[t] = points removed before the position entry t * kTableTile holds, made for all tiles at once by
//  k_tile_removed: as k_tile_bounds)
__global__ __launch_bounds__(256) void k_tile_removed(const int32_t* __restrict__ src, size_t n_entries,
                                                      const uint32_t* __restrict__ offs, uint32_t n, uint32_t kept,
                                                      uint32_t* __restrict__ tr, size_t n_bounds)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_bounds) return;
    const uint32_t v = (uint32_t)src[min(t * (size_t)kTableTile, n_entries - 1)];
    tr[t] = v < n ? v - offs[v] : n - kept;  // points removed before position v
}
__global__ __launch_bounds__(256) void k_table_remap(const int32_t* src, int32_t* dst,
                                                     size_t n_entries,
                                                     const uint32_t* __restrict__ offs, uint32_t n,
                                                     uint32_t kept, const uint32_t* __restrict__ tr)
{
    const bool oop = src != dst;  // (as k_table_shift)
    const size_t n_tiles = (n_entries + kTableTile - 1) / kTableTile;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t t0 = tile * kTableTile, t1 = min(t0 + (size_t)kTableTile, n_entries);
        // removed before the tile's first entry / before the next tile's first (positions ascend: the tile's lie between)
        const uint32_t r0 = tr[tile], r1 = tr[tile + 1];
        const size_t c0 = t0 + (size_t)threadIdx.x * 4;
        if (c0 >= t1) continue;
        if (r0 == r1) {
            if (r0 == 0 && !oop) continue;  // nothing removed below this tile: entries unchanged
            if (c0 + 3 < t1) {
                int4 v = *reinterpret_cast<const int4*>(src + c0);
                v.x -= (int)r0, v.y -= (int)r0, v.z -= (int)r0, v.w -= (int)r0;
                *reinterpret_cast<int4*>(dst + c0) = v;
            } else {
                for (size_t c = c0; c < t1; ++c) dst[c] = src[c] - (int)r0;
            }
        } else {
            for (size_t c = c0; c < min(c0 + 4, t1); ++c) {
                const uint32_t v = (uint32_t)src[c];
                dst[c] = (int32_t)(v < n ? offs[v] : kept);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_count_invalid(const float4* __restrict__ nrm, uint32_t n,
                                                       unsigned long long* __restrict__ invalid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool z = i < n && is_zero3(nrm[i]);
    const unsigned long long mask = __ballot(z);
    if (mask && (threadIdx.x & 63) == 0) atomicAdd(invalid, (unsigned long long)__popcll(mask));
}

static inline int grid_for(size_t n, int threads, int cap)
{
    size_t g = (n + threads - 1) / threads;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? (size_t)cap : g);
}

hipError_t launch_keys4(const float4* pts, size_t n, const MapView& g, uint32_t* keys, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_keys4, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, pts, n, g.ox, g.oy,
                       g.oz, g.inv_h, g.S, g.fx, g.fy, keys);
    return hipGetLastError();
}

hipError_t launch_merge(const float4* pts, const float4* nrm, const uint32_t* perm,
                        const uint32_t* keys, uint32_t n, const float* rx, const float* ry,
                        const float* rz, uint32_t raw_base, const uint32_t* nk,
                        const uint32_t* nidx, uint32_t m, float4* pts2, float4* nrm2,
                        uint32_t* perm2, uint32_t* keys2, hipStream_t s)
{
    if (n)
        hipLaunchKernelGGL(k_merge_old, dim3((n + 255) / 256), dim3(256), 0, s, pts, nrm, perm, keys,
                           n, nk, m, pts2, nrm2, perm2, keys2);
    if (m)
        hipLaunchKernelGGL(k_merge_new, dim3((m + 255) / 256), dim3(256), 0, s, rx, ry, rz, raw_base,
                           nk, nidx, m, keys, n, pts2, nrm2, perm2, keys2);
    return hipGetLastError();
}

size_t table_tile_bounds(size_t n_entries) { return (n_entries + kTableTile - 1) / kTableTile + 1; }

hipError_t launch_table_shift(const int32_t* src, int32_t* dst, size_t n_entries, const uint32_t* nk, uint32_t m,
                              uint32_t* tile_scratch, hipStream_t s)
{
    const size_t nb = table_tile_bounds(n_entries);
    hipLaunchKernelGGL(k_tile_bounds, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, nk, m, n_entries, tile_scratch, nb);
    hipLaunchKernelGGL(k_table_shift, dim3(grid_for((n_entries + kTableTile - 1) / kTableTile, 1, 16384)), dim3(256), 0,
                       s, src, dst, n_entries, nk, m, tile_scratch);
    return hipGetLastError();
}

hipError_t launch_mark_dirty(const uint32_t* keys, uint32_t m, const uint32_t* sel, const MapView& g,
                             uint8_t* dirty, hipStream_t s)
{
    if (m == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mark_dirty, dim3((m + 255) / 256), dim3(256), 0, s, keys, m, sel, g.S, g.fx,
                       g.fy, g.nx, g.ny, g.nz, dirty);
    return hipGetLastError();
}

hipError_t launch_select_dirty(const uint32_t* keys, uint32_t n, const MapView& g, const uint8_t* dirty,
                               const float4* nrm, const uint32_t* chg_keys, uint32_t n_chg, int32_t* work,
                               unsigned* count, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_select_dirty, dim3((n + 255) / 256), dim3(256), 0, s, keys, n, g, dirty, nrm, chg_keys, n_chg,
                       work, count);
    return hipGetLastError();
}

// which kernel re-estimates a work list.  MEASURED (round 6, profiles/r06/ab_nrm_subset.txt): the cooperative kernel
// finishes a list sooner when it runs alone, but a stream's updates run BESIDE a registration, and there its ~1 000 vector
// instructions per point compete for the issue slots the registration needs, where the per-lane kernel mostly waits
// for memory: mapping stream 788 frames/s against 858, localisation stream 1 400 against 1 470.  The per-lane kernel stays
// the default; mode 2 (cfg.force_kernel) or VELO_NRM_SUBSET_WAVE=1 select the cooperative one (tests hold both to the
// oracle; same bits).
bool normals_subset_use_wave(const MapView& mv, int n_work, int mode)
{
    (void)mv;
    (void)n_work;
    if (mode == 1) return false;
    if (mode == 2) return true;
    static const bool wave = getenv("VELO_NRM_SUBSET_WAVE") != nullptr;
    return wave;
}

hipError_t launch_normals_subset(const MapView& mv, const uint32_t* perm, int k,
                                 const int32_t* work, int n_work, const uint32_t* chg_keys,
                                 uint32_t n_chg, float4* nrm, unsigned long long* d_invalid,
                                 unsigned* d_done, hipStream_t s, const unsigned* n_work_dev, int mode)
{
    if (n_work <= 0) return hipSuccess;
    // Round 6: the cooperative form (kernels/knn_wave.hip k_normals_wave_subset: a wavefront per listed point for the
    // search, a lane per point for the PCA), on request -- see normals_subset_use_wave.  mode: cfg.force_kernel.
    if (!chg_keys && k >= 1 && normals_subset_use_wave(mv, n_work, mode))
        return launch_normals_wave_subset(mv, perm, k, work, n_work, nrm, d_invalid, d_done, s, n_work_dev);
    // (length on the device: a bounded grid that strides over the list -- 4 096 workgroups cover 524 288 points in one
    //  pass, more than a roll of the stream marks; the roll's stream keeps a quarter of the CUs free, capi.cpp)
    const int blocks = (n_work + kNrmThreads - 1) / kNrmThreads;
    const dim3 g(n_work_dev ? min(blocks, 4096) : blocks), b(kNrmThreads);
    if (k <= 8)
        hipLaunchKernelGGL(k_normals_subset<8>, g, b, 0, s, mv, perm, k, work, n_work, n_work_dev, chg_keys, n_chg,
                           nrm, d_invalid, d_done);
    else if (k <= 16)
        hipLaunchKernelGGL(k_normals_subset<16>, g, b, 0, s, mv, perm, k, work, n_work, n_work_dev, chg_keys, n_chg,
                           nrm, d_invalid, d_done);
    else
        hipLaunchKernelGGL(k_normals_subset<VELO_MAX_KNORMALS>, g, b, 0, s, mv, perm, k, work, n_work, n_work_dev,
                           chg_keys, n_chg, nrm, d_invalid, d_done);
    return hipGetLastError();
}

hipError_t launch_removed_keys(const uint32_t* keys, const uint32_t* keep, const uint32_t* offs,
                               uint32_t n, uint32_t* out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_removed_keys, dim3((n + 255) / 256), dim3(256), 0, s, keys, keep, offs, n,
                       out);
    return hipGetLastError();
}

hipError_t launch_count_invalid(const float4* nrm, uint32_t n, unsigned long long* d_invalid,
                                hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_invalid, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_count_invalid, dim3((n + 255) / 256), dim3(256), 0, s, nrm, n, d_invalid);
    return hipGetLastError();
}

hipError_t launch_keep_flags(const float4* pts, const float* x, const float* y, const float* z,
                             uint32_t n, const KeepRegion& g, uint32_t* flags, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    if (pts)
        hipLaunchKernelGGL(k_keep4, dim3((n + 255) / 256), dim3(256), 0, s, pts, n, g, flags);
    else
        hipLaunchKernelGGL(k_keep3, dim3((n + 255) / 256), dim3(256), 0, s, x, y, z, n, g, flags);
    return hipGetLastError();
}

// flags of the append-order list + the bounding box of what is kept (d_scratch6: enc_f32 min x3,
// max x3; initialised here, read back by the caller together with its other scalars)
hipError_t launch_keep_flags_minmax(const float* x, const float* y, const float* z, uint32_t n,
                                    const KeepRegion& g, uint32_t* flags, unsigned* d_scratch6, hipStream_t s)
{
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    hipError_t e = hipMemcpyAsync(d_scratch6, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_keep3_minmax, dim3(grid), dim3(256), 0, s, x, y, z, n, g, flags, d_scratch6);
    return hipGetLastError();
}

hipError_t launch_compact_sorted(const float4* pts, const float4* nrm, const uint32_t* perm,
                                 const uint32_t* keys, uint32_t n, const uint32_t* flags,
                                 const uint32_t* offs, const uint32_t* raw_offs, float4* pts2,
                                 float4* nrm2, uint32_t* perm2, uint32_t* keys2,
                                 unsigned long long* d_invalid, hipStream_t s)
{
    hipLaunchKernelGGL(k_compact_sorted, dim3((n + 255) / 256), dim3(256), 0, s, pts, nrm, perm,
                       keys, n, flags, offs, raw_offs, pts2, nrm2, perm2, keys2, d_invalid);
    return hipGetLastError();
}

hipError_t launch_compact_raw(const float* x, const float* y, const float* z, uint32_t n,
                              const uint32_t* flags, const uint32_t* offs, float* x2, float* y2,
                              float* z2, hipStream_t s)
{
    hipLaunchKernelGGL(k_compact_raw, dim3((n + 255) / 256), dim3(256), 0, s, x, y, z, n, flags,
                       offs, x2, y2, z2);
    return hipGetLastError();
}

hipError_t launch_table_remap(const int32_t* src, int32_t* dst, size_t n_entries, const uint32_t* offs, uint32_t n,
                              uint32_t kept, uint32_t* tile_scratch, hipStream_t s)
{
    const size_t nb = table_tile_bounds(n_entries);
    hipLaunchKernelGGL(k_tile_removed, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, src, n_entries, offs, n, kept,
                       tile_scratch, nb);
    hipLaunchKernelGGL(k_table_remap, dim3(grid_for((n_entries + kTableTile - 1) / kTableTile, 1, 16384)), dim3(256), 0, s,
                       src, dst, n_entries, offs, n, kept, tile_scratch);
    return hipGetLastError();
}

// ---- re-anchoring with carried normals: a normal depends on the point list only (ties are
// broken by append-order index), so when the grid moves the normals are permuted, not
// re-estimated; only points near added / removed points go through the PCA again.
__global__ __launch_bounds__(256) void k_scatter_nrm_raw(const float4* __restrict__ nrm,
                                                         const uint32_t* __restrict__ perm,
                                                         uint32_t n, const uint32_t* __restrict__ keep,
                                                         const uint32_t* __restrict__ raw_offs,
                                                         float4* __restrict__ nrm_raw)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n || (keep && !keep[s])) return;
    const uint32_t r = perm[s];
    nrm_raw[raw_offs ? raw_offs[r] : r] = nrm[s];
}
__global__ __launch_bounds__(256) void k_fill_fresh(float4* __restrict__ nrm_raw, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) nrm_raw[i] = make_float4(0.f, 0.f, 0.f, -1.0f);
}
__global__ __launch_bounds__(256) void k_gather_nrm(const float4* __restrict__ nrm_raw,
                                                    const uint32_t* __restrict__ perm, uint32_t n,
                                                    float4* __restrict__ nrm)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) nrm[s] = nrm_raw[perm[s]];
}
// mark the 27 voxels around points chosen by position: mode 0 = nrm[i].w < 0 (fresh points
// of the new order), mode 1 = keep[i] == 0 (removed points of the old order; they may lie
// outside the new grid)
__global__ __launch_bounds__(256) void k_mark_dirty_pts(const float4* __restrict__ pts,
                                                        const float4* __restrict__ nrm,
                                                        const uint32_t* __restrict__ keep,
                                                        uint32_t n, MapView g,
                                                        uint8_t* __restrict__ dirty)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (nrm ? nrm[i].w >= 0.0f : keep[i] != 0u) return;
    const float4 p = pts[i];
    const int vx = cell_coord(p.x, g.ox, g.inv_h, g.nx);
    const int vy = cell_coord(p.y, g.oy, g.inv_h, g.ny);
    const int vz = cell_coord(p.z, g.oz, g.inv_h, g.nz);
    for (int dz = -1; dz <= 1; ++dz) {
        const int z = vz + dz;
        if (z < 0 || z >= g.nz) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = vy + dy;
            if (y < 0 || y >= g.ny) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = vx + dx;
                if (x < 0 || x >= g.nx) continue;
                dirty[((size_t)z * g.ny + y) * g.nx + x] = 1;
            }
        }
    }
}

hipError_t launch_scatter_nrm_raw(const float4* nrm, const uint32_t* perm, uint32_t n,
                                  const uint32_t* keep, const uint32_t* raw_offs, float4* nrm_raw,
                                  hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_nrm_raw, dim3((n + 255) / 256), dim3(256), 0, s, nrm, perm, n, keep,
                       raw_offs, nrm_raw);
    return hipGetLastError();
}
hipError_t launch_fill_fresh(float4* nrm_raw, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_fresh, dim3((n + 255) / 256), dim3(256), 0, s, nrm_raw, n);
    return hipGetLastError();
}
hipError_t launch_gather_nrm(const float4* nrm_raw, const uint32_t* perm, uint32_t n, float4* nrm,
                             hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_nrm, dim3((n + 255) / 256), dim3(256), 0, s, nrm_raw, perm, n, nrm);
    return hipGetLastError();
}
hipError_t launch_mark_dirty_pts(const float4* pts, const float4* nrm, const uint32_t* keep,
                                 uint32_t n, const MapView& g, uint8_t* dirty, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mark_dirty_pts, dim3((n + 255) / 256), dim3(256), 0, s, pts, nrm, keep, n,
                       g, dirty);
    return hipGetLastError();
}

// ---- dilated voxel occupancy: near[v] == 0 <=> the 27 voxels around v hold no point, i.e.
// the candidate set of any query in v is empty (exact; lets the search of far-range frame
// points over a cropped map stop after one byte load)
__global__ __launch_bounds__(256) void k_vox_occ(MapView mv, uint8_t* __restrict__ occ)
{
    const size_t nvox = (size_t)mv.nx * mv.ny * mv.nz;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox;
         v += (size_t)gridDim.x * blockDim.x) {
        const int vx = (int)(v % (size_t)mv.nx);
        const size_t t = v / (size_t)mv.nx;
        const int vy = (int)(t % (size_t)mv.ny), vz = (int)(t / (size_t)mv.ny);
        const int S = mv.S;
        // the voxel's S*S fine rows are S runs of S consecutive rows: first and last index of
        // each row piece; any non-empty piece makes the voxel occupied
        bool any = false;
        for (int sz = 0; sz < S && !any; ++sz)
            for (int sy = 0; sy < S; ++sy) {
                const size_t row = ((size_t)(vz * S + sz) * mv.fy + (size_t)(vy * S + sy)) * mv.fx;
                if (mv.cell_start[row + (size_t)(vx + 1) * S] > mv.cell_start[row + (size_t)vx * S]) {
                    any = true;
                    break;
                }
            }
        occ[v] = any ? 1 : 0;
    }
}
__global__ __launch_bounds__(256) void k_vox_near(const uint8_t* __restrict__ occ, int nx, int ny,
                                                  int nz, uint8_t* __restrict__ near)
{
    const size_t nvox = (size_t)nx * ny * nz;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvox;
         v += (size_t)gridDim.x * blockDim.x) {
        const int vx = (int)(v % (size_t)nx);
        const size_t t = v / (size_t)nx;
        const int vy = (int)(t % (size_t)ny), vz = (int)(t / (size_t)ny);
        unsigned any = 0;
        for (int z = max(vz - 1, 0); z <= min(vz + 1, nz - 1); ++z)
            for (int y = max(vy - 1, 0); y <= min(vy + 1, ny - 1); ++y)
                for (int x = max(vx - 1, 0); x <= min(vx + 1, nx - 1); ++x)
                    any |= occ[((size_t)z * ny + y) * nx + x];
        near[v] = any ? 1 : 0;
    }
}
// ---- density probe for the automatic sub-division (oracle/icp.c vo_auto_subdiv): number of
// distinct voxels (anchored on the component-wise minimum) that hold a point
__global__ __launch_bounds__(256) void k_mark_voxels(const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ z, size_t n, float ox,
                                                     float oy, float oz, float inv_h, size_t dx,
                                                     size_t dy, uint8_t* __restrict__ occ)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t cx = (size_t)floorf((x[i] - ox) * inv_h), cy = (size_t)floorf((y[i] - oy) * inv_h),
                     cz = (size_t)floorf((z[i] - oz) * inv_h);
        occ[(cz * dy + cy) * dx + cx] = 1;
    }
}
__global__ __launch_bounds__(256) void k_count_bytes(const uint8_t* __restrict__ occ, size_t n,
                                                     unsigned long long* __restrict__ total)
{
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        c += occ[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, (unsigned long long)c);
}
hipError_t launch_count_occupied_voxels(const float* x, const float* y, const float* z, size_t n,
                                        const float mn[3], float inv_h, const size_t dims[3],
                                        uint8_t* occ, unsigned long long* d_count, hipStream_t s)
{
    const size_t nv = dims[0] * dims[1] * dims[2];
    hipError_t e = hipMemsetAsync(occ, 0, nv, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_mark_voxels, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, x, y, z, n, mn[0],
                       mn[1], mn[2], inv_h, dims[0], dims[1], occ);
    hipLaunchKernelGGL(k_count_bytes, dim3(grid_for(nv, 256, 4096)), dim3(256), 0, s, occ, nv, d_count);
    return hipGetLastError();
}

// occupancy from the sorted fine keys (one 4-byte read per point) instead of probing every
// voxel's S*S row pieces in the table: the table walk costs 36 pairs of loads per voxel at S = 6
__global__ __launch_bounds__(256) void k_vox_occ_keys(const uint32_t* __restrict__ keys, uint32_t n,
                                                      int fx, int fy, int S, int nx, int ny,
                                                      uint8_t* __restrict__ occ)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t k = keys[i];
        const uint32_t Fx = k % (uint32_t)fx, t = k / (uint32_t)fx;
        const uint32_t Fy = t % (uint32_t)fy, Fz = t / (uint32_t)fy;
        occ[((size_t)(Fz / S) * ny + Fy / S) * nx + Fx / S] = 1;
    }
}
hipError_t launch_vox_near(const MapView& mv, const uint32_t* keys_sorted, uint8_t* occ, uint8_t* near,
                           hipStream_t s)
{
    const size_t nvox = (size_t)mv.nx * mv.ny * mv.nz;
    const int grid = grid_for(nvox, 256, 16384);
    if (keys_sorted) {
        hipError_t e = hipMemsetAsync(occ, 0, nvox, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_vox_occ_keys, dim3(grid_for((size_t)mv.n, 256, 8192)), dim3(256), 0, s,
                           keys_sorted, (uint32_t)mv.n, mv.fx, mv.fy, mv.S, mv.nx, mv.ny, occ);
    } else {
        hipLaunchKernelGGL(k_vox_occ, dim3(grid), dim3(256), 0, s, mv, occ);
    }
    hipLaunchKernelGGL(k_vox_near, dim3(grid), dim3(256), 0, s, occ, mv.nx, mv.ny, mv.nz, near);
    return hipGetLastError();
}

}  // namespace velo
