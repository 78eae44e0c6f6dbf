// lds_staged_probe.hip -- a MEASUREMENT, not a product path: the north star's "LDS-staged voxel neighbourhoods" built
// the way it would have to be built for the headline batch (cross-frame cell-sorted, LDS-staged
// first iteration), so that its cost stands next to the kernel that ships (k_search_a: certificate test + stage A of
// every query, 284 us of the 339 us unhinted launch on the same inputs).
//
//   1. k_qkeys     every query of all F frames transformed by its frame's pose (fp64, as the product), voxel key
//   2. sort        rocPRIM radix sort of (voxel key, query id) over ALL frames (7.37 M pairs at F = 64)
//   3. k_staged    one wavefront per 64 consecutive sorted queries; for every distinct voxel among them (two on
//                  average) the points of the 27 voxels around it are staged in LDS (9 contiguous ranges of the
//                  voxel-sorted map) and every lane of that voxel scans ALL of them: exact nearest neighbour of the
//                  27-voxel candidate set (the specification's candidate set; lowest index among equal distances)
//
// Inputs: a binary file written by tools/lds_staged_probe.py from bench.py's own headline inputs (map, compensated
// frames, initial poses).  Results are checked against a plain per-thread scan of the same 27 voxels on a sample.
//
//   build:  hipcc --offload-arch=gfx950 -O3 tools/lds_staged_probe.hip -o tools/lds_staged_probe
//   run:    tools/lds_staged_probe inputs.bin [repeats]      -> one JSON line
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e__ = (x);                                                                   \
        if (e__ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                            \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

struct Grid {
    float ox, oy, oz, inv_h;
    int nx, ny, nz;
};

__device__ __forceinline__ int vcoord(float p, float o, float inv_h) { return (int)floorf((p - o) * inv_h); }

__global__ void k_mapkeys(const float* x, const float* y, const float* z, uint32_t n, Grid g, uint32_t* keys, uint32_t* idx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cx = vcoord(x[i], g.ox, g.inv_h), cy = vcoord(y[i], g.oy, g.inv_h), cz = vcoord(z[i], g.oz, g.inv_h);
    keys[i] = (uint32_t)((cz * g.ny + cy) * g.nx + cx);
    idx[i] = i;
}
__global__ void k_gather4(const float* x, const float* y, const float* z, const uint32_t* perm, uint32_t n, float4* pts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = perm[i];
    pts[i] = make_float4(x[r], y[r], z[r], 0.f);
}
__global__ void k_voxstart(const uint32_t* keys, uint32_t n, uint32_t nvox, uint32_t* start)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nvox) return;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys[mid] < c) lo = mid + 1; else hi = mid;
    }
    start[c] = lo;
}

// ---- timed 1: transformed query + voxel key (queries outside the grid: key = nvox, sorted to the end, no candidates)
__global__ void k_qkeys(const float* x, const float* y, const float* z, const uint32_t* qframe, uint32_t n,
                        const double* poses, Grid g, uint32_t nvox, uint32_t* keys, uint32_t* ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* T = poses + 12 * (size_t)qframe[i];
    const double sx = x[i], sy = y[i], sz = z[i];
    const float qx = (float)fma(T[0], sx, fma(T[1], sy, fma(T[2], sz, T[3])));
    const float qy = (float)fma(T[4], sx, fma(T[5], sy, fma(T[6], sz, T[7])));
    const float qz = (float)fma(T[8], sx, fma(T[9], sy, fma(T[10], sz, T[11])));
    const int cx = vcoord(qx, g.ox, g.inv_h), cy = vcoord(qy, g.oy, g.inv_h), cz = vcoord(qz, g.oz, g.inv_h);
    const bool in = cx >= 0 && cx < g.nx && cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz;
    keys[i] = in ? (uint32_t)((cz * g.ny + cy) * g.nx + cx) : nvox;
    ids[i] = i;
}

// ---- timed 3: the staged search
constexpr int kCap = 1024;  // staged points per wavefront (16 KB): larger neighbourhoods go through in tiles
__global__ __launch_bounds__(64) void k_staged(const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ sids,
                                               uint32_t n, const float* __restrict__ x, const float* __restrict__ y,
                                               const float* __restrict__ z, const uint32_t* __restrict__ qframe,
                                               const double* __restrict__ poses, Grid g, uint32_t nvox,
                                               const uint32_t* __restrict__ vstart, const float4* __restrict__ pts,
                                               float dmax2, int* __restrict__ out_j, float* __restrict__ out_d2)
{
    __shared__ float4 s_p[kCap];
    const int lane = threadIdx.x;
    const uint32_t i = blockIdx.x * 64u + (uint32_t)lane;
    const bool live = i < n;
    const uint32_t key = live ? skeys[i] : nvox;
    const uint32_t id = live ? sids[i] : 0u;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        const double* T = poses + 12 * (size_t)qframe[id];
        const double sx = x[id], sy = y[id], sz = z[id];
        qx = (float)fma(T[0], sx, fma(T[1], sy, fma(T[2], sz, T[3])));
        qy = (float)fma(T[4], sx, fma(T[5], sy, fma(T[6], sz, T[7])));
        qz = (float)fma(T[8], sx, fma(T[9], sy, fma(T[10], sz, T[11])));
    }
    float best = INFINITY;
    int bj = -1;
    unsigned long long todo = __ballot(live && key < nvox);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t v = (uint32_t)__shfl((int)key, src, 64);
        const bool mine = live && key == v;
        todo &= ~__ballot(mine);
        const int cx = (int)(v % (uint32_t)g.nx), cy = (int)((v / (uint32_t)g.nx) % (uint32_t)g.ny), cz = (int)(v / ((uint32_t)g.nx * g.ny));
        // the 9 rows of three x-adjacent voxels: each one contiguous range of the voxel-sorted map
        for (int r = 0; r < 9; ++r) {
            const int yy = cy + r % 3 - 1, zz = cz + r / 3 - 1;
            if (yy < 0 || yy >= g.ny || zz < 0 || zz >= g.nz) continue;
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
            const uint32_t k0 = (uint32_t)((zz * g.ny + yy) * g.nx + x0), k1 = (uint32_t)((zz * g.ny + yy) * g.nx + x1);
            const uint32_t a = vstart[k0], b = vstart[k1 + 1];
            for (uint32_t t0 = a; t0 < b; t0 += kCap) {
                const uint32_t cnt = min(b - t0, (uint32_t)kCap);
                __syncthreads();
                for (uint32_t j = lane; j < cnt; j += 64) s_p[j] = pts[t0 + j];
                __syncthreads();
                if (mine) {
#pragma unroll 4
                    for (uint32_t j = 0; j < cnt; ++j) {
                        const float4 p = s_p[j];
                        const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
                        const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        if (d2 < best) {  // (ascending index: '<' keeps the lowest index among equal distances)
                            best = d2;
                            bj = (int)(t0 + j);
                        }
                    }
                }
            }
        }
    }
    if (live) {
        const bool ok = bj >= 0 && best <= dmax2;
        out_j[id] = ok ? bj : -1;
        out_d2[id] = ok ? best : INFINITY;
    }
}

// ---- the check: the same 27 voxels scanned by one thread per sampled query, from global memory
__global__ void k_check(const uint32_t* sample, uint32_t ns, const float* x, const float* y, const float* z,
                        const uint32_t* qframe, const double* poses, Grid g, const uint32_t* vstart, const float4* pts,
                        float dmax2, const int* out_j, const float* out_d2, unsigned* bad)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    const uint32_t id = sample[s];
    const double* T = poses + 12 * (size_t)qframe[id];
    const double sx = x[id], sy = y[id], sz = z[id];
    const float qx = (float)fma(T[0], sx, fma(T[1], sy, fma(T[2], sz, T[3])));
    const float qy = (float)fma(T[4], sx, fma(T[5], sy, fma(T[6], sz, T[7])));
    const float qz = (float)fma(T[8], sx, fma(T[9], sy, fma(T[10], sz, T[11])));
    const int cx = vcoord(qx, g.ox, g.inv_h), cy = vcoord(qy, g.oy, g.inv_h), cz = vcoord(qz, g.oz, g.inv_h);
    float best = INFINITY;
    int bj = -1;
    if (cx >= 0 && cx < g.nx && cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz)
        for (int zz = max(cz - 1, 0); zz <= min(cz + 1, g.nz - 1); ++zz)
            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, g.ny - 1); ++yy) {
                const uint32_t k0 = (uint32_t)((zz * g.ny + yy) * g.nx + max(cx - 1, 0)), k1 = (uint32_t)((zz * g.ny + yy) * g.nx + min(cx + 1, g.nx - 1));
                for (uint32_t j = vstart[k0]; j < vstart[k1 + 1]; ++j) {
                    const float4 p = pts[j];
                    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
                    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if (d2 < best) {
                        best = d2;
                        bj = (int)j;
                    }
                }
            }
    const bool ok = bj >= 0 && best <= dmax2;
    const int ej = ok ? bj : -1;
    if (out_j[id] != ej || (ok && out_d2[id] != best)) atomicAdd(bad, 1u);
}

template <typename T>
static T* dev(const std::vector<T>& h)
{
    T* p = nullptr;
    CK(hipMalloc((void**)&p, std::max<size_t>(h.size(), 1) * sizeof(T)));
    if (!h.empty()) CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}
template <typename T>
static T* devn(size_t n)
{
    T* p = nullptr;
    CK(hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)));
    return p;
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: lds_staged_probe inputs.bin [repeats]\n");
        return 2;
    }
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    uint64_t hdr[3];  // n_map, n_frames, n_q
    float hv;
    if (fread(hdr, 8, 3, f) != 3 || fread(&hv, 4, 1, f) != 1) return 3;
    const size_t nm = hdr[0], F = hdr[1], nq = hdr[2];
    std::vector<float> mx(nm), my(nm), mz(nm), qx(nq), qy(nq), qz(nq);
    std::vector<int64_t> fs(F + 1);
    std::vector<double> poses(12 * F);
    if (fread(mx.data(), 4, nm, f) != nm || fread(my.data(), 4, nm, f) != nm || fread(mz.data(), 4, nm, f) != nm) return 3;
    if (fread(fs.data(), 8, F + 1, f) != F + 1) return 3;
    if (fread(qx.data(), 4, nq, f) != nq || fread(qy.data(), 4, nq, f) != nq || fread(qz.data(), 4, nq, f) != nq) return 3;
    if (fread(poses.data(), 8, 12 * F, f) != 12 * F) return 3;
    fclose(f);
    // grid: anchored on the map's minimum, as velo_map_reset without margins
    Grid g;
    float mn[3] = {mx[0], my[0], mz[0]}, mxx[3] = {mx[0], my[0], mz[0]};
    for (size_t i = 0; i < nm; ++i) {
        mn[0] = std::min(mn[0], mx[i]), mn[1] = std::min(mn[1], my[i]), mn[2] = std::min(mn[2], mz[i]);
        mxx[0] = std::max(mxx[0], mx[i]), mxx[1] = std::max(mxx[1], my[i]), mxx[2] = std::max(mxx[2], mz[i]);
    }
    g.ox = mn[0], g.oy = mn[1], g.oz = mn[2], g.inv_h = 1.0f / hv;
    g.nx = (int)floorf((mxx[0] - mn[0]) * g.inv_h) + 1, g.ny = (int)floorf((mxx[1] - mn[1]) * g.inv_h) + 1,
    g.nz = (int)floorf((mxx[2] - mn[2]) * g.inv_h) + 1;
    const uint32_t nvox = (uint32_t)g.nx * g.ny * g.nz;
    std::vector<uint32_t> qframe(nq);
    for (size_t fr = 0; fr < F; ++fr)
        for (int64_t i = fs[fr]; i < fs[fr + 1]; ++i) qframe[(size_t)i] = (uint32_t)fr;
    float *dmx = dev(mx), *dmy = dev(my), *dmz = dev(mz), *dqx = dev(qx), *dqy = dev(qy), *dqz = dev(qz);
    uint32_t* dqf = dev(qframe);
    double* dposes = dev(poses);
    // ---- setup (untimed): the map sorted by voxel, per-voxel starts
    uint32_t *mk = devn<uint32_t>(nm), *mk2 = devn<uint32_t>(nm), *mi = devn<uint32_t>(nm), *mi2 = devn<uint32_t>(nm);
    hipLaunchKernelGGL(k_mapkeys, dim3((nm + 255) / 256), dim3(256), 0, 0, dmx, dmy, dmz, (uint32_t)nm, g, mk, mi);
    int vbits = 1;
    while ((1u << vbits) <= nvox) ++vbits;
    size_t tb = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tb, mk, mk2, mi, mi2, nm, 0u, (unsigned)vbits, 0));
    void* tmp = nullptr;
    CK(hipMalloc(&tmp, tb));
    CK(rocprim::radix_sort_pairs(tmp, tb, mk, mk2, mi, mi2, nm, 0u, (unsigned)vbits, 0));
    float4* pts = devn<float4>(nm);
    hipLaunchKernelGGL(k_gather4, dim3((nm + 255) / 256), dim3(256), 0, 0, dmx, dmy, dmz, mi2, (uint32_t)nm, pts);
    uint32_t* vstart = devn<uint32_t>((size_t)nvox + 2);
    hipLaunchKernelGGL(k_voxstart, dim3((nvox + 1 + 255) / 256), dim3(256), 0, 0, mk2, (uint32_t)nm, nvox, vstart);
    CK(hipDeviceSynchronize());
    // ---- timed
    uint32_t *qk = devn<uint32_t>(nq), *qk2 = devn<uint32_t>(nq), *qi = devn<uint32_t>(nq), *qi2 = devn<uint32_t>(nq);
    int* out_j = devn<int>(nq);
    float* out_d2 = devn<float>(nq);
    int qbits = 1;
    while ((1u << qbits) <= nvox) ++qbits;  // (key nvox = outside the grid)
    size_t tq = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tq, qk, qk2, qi, qi2, nq, 0u, (unsigned)qbits, 0));
    void* tmpq = nullptr;
    CK(hipMalloc(&tmpq, tq));
    hipEvent_t ev[4];
    for (auto& e : ev) CK(hipEventCreate(&e));
    double sum[3] = {0, 0, 0}, best3[3] = {1e30, 1e30, 1e30};
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(ev[0], 0));
        hipLaunchKernelGGL(k_qkeys, dim3((nq + 255) / 256), dim3(256), 0, 0, dqx, dqy, dqz, dqf, (uint32_t)nq, dposes, g, nvox, qk, qi);
        CK(hipEventRecord(ev[1], 0));
        CK(rocprim::radix_sort_pairs(tmpq, tq, qk, qk2, qi, qi2, nq, 0u, (unsigned)qbits, 0));
        CK(hipEventRecord(ev[2], 0));
        hipLaunchKernelGGL(k_staged, dim3((nq + 63) / 64), dim3(64), 0, 0, qk2, qi2, (uint32_t)nq, dqx, dqy, dqz, dqf, dposes, g,
                           nvox, vstart, pts, 1.0f, out_j, out_d2);
        CK(hipEventRecord(ev[3], 0));
        CK(hipEventSynchronize(ev[3]));
        if (r < 2) continue;  // warm-up
        for (int k = 0; k < 3; ++k) {
            float ms = 0;
            CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            sum[k] += ms * 1e3;
            best3[k] = std::min(best3[k], (double)ms * 1e3);
        }
    }
    // ---- check a sample
    std::vector<uint32_t> sample;
    for (size_t i = 0; i < nq; i += std::max<size_t>(nq / 65536, 1)) sample.push_back((uint32_t)i);
    uint32_t* dsample = dev(sample);
    unsigned* dbad = devn<unsigned>(1);
    CK(hipMemset(dbad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3((sample.size() + 255) / 256), dim3(256), 0, 0, dsample, (uint32_t)sample.size(), dqx, dqy, dqz, dqf,
                       dposes, g, vstart, pts, 1.0f, out_j, out_d2, dbad);
    unsigned bad = 0;
    CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
    std::vector<int> hj(nq);
    CK(hipMemcpy(hj.data(), out_j, nq * sizeof(int), hipMemcpyDeviceToHost));
    size_t matched = 0;
    for (int v : hj) matched += v >= 0;
    printf("{\"queries\": %zu, \"frames\": %zu, \"map_points\": %zu, \"voxels\": %u, \"key_bits\": %d, \"repeats\": %d, "
           "\"us_mean\": {\"query_keys\": %.1f, \"sort\": %.1f, \"staged_search\": %.1f, \"total\": %.1f}, "
           "\"us_min\": {\"query_keys\": %.1f, \"sort\": %.1f, \"staged_search\": %.1f, \"total\": %.1f}, "
           "\"matched\": %zu, \"sample_checked\": %zu, \"sample_mismatches\": %u}\n",
           nq, F, nm, nvox, qbits, reps, sum[0] / reps, sum[1] / reps, sum[2] / reps, (sum[0] + sum[1] + sum[2]) / reps, best3[0],
           best3[1], best3[2], best3[0] + best3[1] + best3[2], matched, sample.size(), bad);
    return bad ? 1 : 0;
}
