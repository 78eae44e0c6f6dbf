// map_build.hip -- voxel-sorted map construction for gfx950 (done once per map
// update, not per ICP iteration): bounds, cell keys, gather into float4, cell
// table, and per-point PCA normals.  Semantics: DESIGN.md "ICP semantics"
// (grid / normals), checked bit-for-bit against oracle/icp.c in tests/.
// The stable key sort itself lives in sortscan.hip (rocPRIM radix sort).
#include "device_math.hpp"

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
__global__ __launch_bounds__(256) void k_cell_start(const uint32_t* __restrict__ keys, size_t n,
                                                    size_t ncell, int32_t* __restrict__ cell_start)
{
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c <= ncell;
         c += (size_t)gridDim.x * blockDim.x) {
        size_t lo = 0, hi = n;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if ((size_t)keys[mid] < c)
                lo = mid + 1;
            else
                hi = mid;
        }
        cell_start[c] = (int32_t)lo;
    }
}

hipError_t launch_cell_start(const uint32_t* sorted_keys, size_t n, size_t ncell,
                             int32_t* cell_start, hipStream_t s)
{
    size_t g = (ncell + 1 + 255) / 256;
    int grid = (int)(g > 8192 ? 8192 : g);
    hipLaunchKernelGGL(k_cell_start, dim3(grid), dim3(256), 0, s, sorted_keys, n, ncell,
                       cell_start);
    return hipGetLastError();
}

// ----------------------------------------------------------------- normals
// One thread per map point.  The running k-best list lives in LDS as
// [slot][thread] (stride = blockDim: bank-conflict free), because a
// dynamically indexed per-thread array would otherwise go to scratch memory.
constexpr int kNrmThreads = 128;
constexpr int kMinNb = 5;

__device__ __forceinline__ void jacobi_rot(double A[3][3], double V[3][3], int p, int q)
{
    if (A[p][q] == 0.0) return;
    const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
    double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
    if (theta < 0.0) t = -t;
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    const int r = 3 - p - q;
    const double app = A[p][p], aqq = A[q][q], apq = A[p][q];
    const double arp = A[r][p], arq = A[r][q];
    A[p][p] = app - t * apq;
    A[q][q] = aqq + t * apq;
    A[p][q] = A[q][p] = 0.0;
    A[r][p] = A[p][r] = c * arp - s * arq;
    A[r][q] = A[q][r] = s * arp + c * arq;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - s * vkq;
        V[k][q] = s * vkp + c * vkq;
    }
}

// The k smallest (d2, sorted index) among the points of the 27 voxels around q with
// d2 <= r2, ascending, into the LDS lists s_d/s_i ([slot][thread]).  Returns how many.
// Fine rows further than sqrt(r2) from q (in y,z) cannot contribute and are skipped; the
// margin keeps that conservative, so the list is the oracle's.
__device__ __forceinline__ int collect_knn(const MapView& mv, float qx, float qy, float qz,
                                           float r2, int k, float (*s_d)[kNrmThreads],
                                           int (*s_i)[kNrmThreads], int tid)
{
    const int cx = cell_coord(qx, mv.ox, mv.inv_h, mv.nx);
    const int cy = cell_coord(qy, mv.oy, mv.inv_h, mv.ny);
    const int cz = cell_coord(qz, mv.oz, mv.inv_h, mv.nz);
    int cnt = 0;
    float worst = INFINITY;  // d2 of the last slot once the list is full
    const int vx0 = max(cx - 1, 0), vx1 = min(cx + 1, mv.nx - 1);
    const int vy0 = max(cy - 1, 0), vy1 = min(cy + 1, mv.ny - 1);
    const int vz0 = max(cz - 1, 0), vz1 = min(cz + 1, mv.nz - 1);
    if (vx0 > vx1 || vy0 > vy1 || vz0 > vz1) return 0;
    const int S = mv.S;
    const float hf = mv.h / (float)S;
    const float uy = (qy - mv.oy) * mv.inv_h * (float)S, uz = (qz - mv.oz) * mv.inv_h * (float)S;
    const float mg = 1e-6f * (float)max(max(mv.nx, mv.ny), mv.nz) * mv.h + 1e-6f;
    for (int fz = vz0 * S; fz < (vz1 + 1) * S; ++fz) {
        const float gz = fmaxf(fmaxf((float)fz - uz, uz - (float)(fz + 1)) * hf - mg, 0.0f);
        if (gz * gz * 0.99999f > r2) continue;
        for (int fy = vy0 * S; fy < (vy1 + 1) * S; ++fy) {
            const float gy = fmaxf(fmaxf((float)fy - uy, uy - (float)(fy + 1)) * hf - mg, 0.0f);
            if ((gz * gz + gy * gy) * 0.99999f > r2) continue;
            const size_t row = ((size_t)fz * mv.fy + fy) * mv.fx;
            const int j0 = mv.cell_start[row + (size_t)vx0 * S];
            const int j1 = mv.cell_start[row + (size_t)(vx1 + 1) * S];
            for (int j = j0; j < j1; ++j) {
                const float d2 = dist2(mv.pts[j], qx, qy, qz);
                if (!(d2 <= r2)) continue;
                if (cnt == k && !(d2 < worst)) continue;
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && d2 < s_d[pos - 1][tid]) {
                    s_d[pos][tid] = s_d[pos - 1][tid];
                    s_i[pos][tid] = s_i[pos - 1][tid];
                    --pos;
                }
                s_d[pos][tid] = d2;
                s_i[pos][tid] = j;
                if (cnt < k) ++cnt;
                if (cnt == k) worst = s_d[k - 1][tid];
            }
        }
    }
    return cnt;
}

__global__ __launch_bounds__(kNrmThreads) void k_normals(MapView mv, int k,
                                                         float4* __restrict__ nrm,
                                                         unsigned long long* __restrict__ invalid)
{
    __shared__ float s_d[VELO_MAX_KNORMALS][kNrmThreads];
    __shared__ int s_i[VELO_MAX_KNORMALS][kNrmThreads];
    const int tid = threadIdx.x;
    const int s = blockIdx.x * kNrmThreads + tid;
    if (s >= mv.n) return;
    const float4 q = mv.pts[s];
    const int cnt = collect_knn(mv, q.x, q.y, q.z, mv.h * mv.h, k, s_d, s_i, tid);
    if (cnt < kMinNb) {
        nrm[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        atomicAdd(invalid, 1ull);
        return;
    }
    double mx = 0, my = 0, mz = 0;
    for (int i = 0; i < cnt; ++i) {
        const float4 p = mv.pts[s_i[i][tid]];
        mx += (double)p.x;
        my += (double)p.y;
        mz += (double)p.z;
    }
    const double invn = 1.0 / (double)cnt;
    mx *= invn;
    my *= invn;
    mz *= invn;
    double C0 = 0, C1 = 0, C2 = 0, C3 = 0, C4 = 0, C5 = 0;
    for (int i = 0; i < cnt; ++i) {
        const float4 p = mv.pts[s_i[i][tid]];
        const double dx = (double)p.x - mx, dy = (double)p.y - my, dz = (double)p.z - mz;
        C0 = fma(dx, dx, C0);
        C1 = fma(dx, dy, C1);
        C2 = fma(dx, dz, C2);
        C3 = fma(dy, dy, C3);
        C4 = fma(dy, dz, C4);
        C5 = fma(dz, dz, C5);
    }
    double A[3][3] = {{C0, C1, C2}, {C1, C3, C4}, {C2, C4, C5}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 8; ++sweep) {
        jacobi_rot(A, V, 0, 1);
        jacobi_rot(A, V, 0, 2);
        jacobi_rot(A, V, 1, 2);
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    double vx = V[0][m], vy = V[1][m], vz = V[2][m];
    const double inv = 1.0 / sqrt(vx * vx + vy * vy + vz * vz);
    vx *= inv;
    vy *= inv;
    vz *= inv;
    const bool flip = (vz < 0.0) || (vz == 0.0 && (vy < 0.0 || (vy == 0.0 && vx < 0.0)));
    if (flip) {
        vx = -vx;
        vy = -vy;
        vz = -vz;
    }
    nrm[s] = make_float4((float)vx, (float)vy, (float)vz, 0.0f);
}

// a10 with k > 1: the k nearest map points of every (transformed) query within d_max,
// ascending (d2, sorted index); rows of idx/d2 are padded with -1 / +inf.
__global__ __launch_bounds__(kNrmThreads) void k_knn(MapView mv, const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ z, int n,
                                                     const double* __restrict__ T, float dmax2,
                                                     int k, int32_t* __restrict__ idx,
                                                     float* __restrict__ d2o,
                                                     int32_t* __restrict__ count)
{
    __shared__ float s_d[VELO_MAX_KNORMALS][kNrmThreads];
    __shared__ int s_i[VELO_MAX_KNORMALS][kNrmThreads];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * kNrmThreads + tid;
    if (i >= n) return;
    double px, py, pz;
    xform(T, x[i], y[i], z[i], px, py, pz);
    const int cnt = collect_knn(mv, (float)px, (float)py, (float)pz, dmax2, k, s_d, s_i, tid);
    for (int m = 0; m < k; ++m) {
        idx[(size_t)i * k + m] = m < cnt ? s_i[m][tid] : -1;
        d2o[(size_t)i * k + m] = m < cnt ? s_d[m][tid] : INFINITY;
    }
    if (count) count[i] = cnt;
}

hipError_t launch_knn(const MapView& mv, const float* x, const float* y, const float* z, size_t n,
                      const double* T, float dmax2, int k, int32_t* idx, float* d2, int32_t* count,
                      hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const int grid = (int)((n + kNrmThreads - 1) / kNrmThreads);
    hipLaunchKernelGGL(k_knn, dim3(grid), dim3(kNrmThreads), 0, s, mv, x, y, z, (int)n, T, dmax2, k,
                       idx, d2, count);
    return hipGetLastError();
}

hipError_t launch_normals(const MapView& mv, int k, float4* nrm, unsigned long long* d_invalid,
                          hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_invalid, 0, sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const int grid = (mv.n + kNrmThreads - 1) / kNrmThreads;
    hipLaunchKernelGGL(k_normals, dim3(grid), dim3(kNrmThreads), 0, s, mv, k, nrm, d_invalid);
    return hipGetLastError();
}

}  // namespace velo
