// pmc_calib.hip -- micro-kernels of KNOWN byte counts, to calibrate what rocprofv3's memory-side
// counters (FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ*, TCC_MISS) report on gfx950 for the access
// shapes this library uses (MI355X_MICROARCH.md, HBM: "Other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern before trusting an absolute").
//
//   build:  hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib
//   run:    rocprofv3 --kernel-trace --pmc <counters> ... -- ./tools/pmc_calib   (tools/pmc_calib.sh)
//   stdout: one JSON object: kernel name -> useful bytes read / written per launch
//
// Every kernel touches a region far beyond the 256 MiB Infinity Cache exactly once per launch, so
// "useful bytes" is also the compulsory HBM traffic for the streams and the whole-row gathers; for
// the 16-byte and 4-byte random gathers the hardware fetches more than it is asked for, and the
// counters -- calibrated on the other classes -- say how much.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e__ = (x);                                                                   \
        if (e__ != hipSuccess) {                                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                            \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// ---- streams: every lane reads W bytes, consecutive lanes consecutive addresses
template <typename T>
__global__ __launch_bounds__(256) void calib_stream(const T* __restrict__ in, size_t n, float* sink, unsigned magic)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T v = in[i];
    const unsigned char* b = (const unsigned char*)&v;
    unsigned s = 0;
    for (unsigned k = 0; k < sizeof(T); ++k) s += b[k];
    if (s == magic) sink[0] = 1.0f;  // never true for the value passed (the compiler cannot know): keeps the load alive
}
// ---- K1's shape: three float4 streams + one ushort4 stream in, three float4 streams out
__global__ __launch_bounds__(256) void calib_k1_shape(const float4* __restrict__ x, const float4* __restrict__ y,
                                                      const float4* __restrict__ z, const ushort4* __restrict__ p,
                                                      size_t nq, float4* ox, float4* oy, float4* oz)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    float4 a = x[i], b = y[i], c = z[i];
    ushort4 k = p[i];
    a.x += k.x; b.y += k.y; c.z += k.z; a.w += k.w;
    ox[i] = a; oy[i] = b; oz[i] = c;
}
// ---- whole rows of ROW bytes at random row indices: ROW/16 lanes share a row, all bytes useful
template <int ROW>
__global__ __launch_bounds__(256) void calib_rows(const float4* __restrict__ in, uint32_t n_rows, size_t n_lanes,
                                                  float* sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lanes) return;
    constexpr int L = ROW / 16;                       // lanes per row
    const uint32_t row = mix((uint32_t)(i / L) * 2654435761u + 12345u) % n_rows;
    float4 v = in[(size_t)row * L + (i % L)];
    if (v.x + v.y + v.z + v.w == 1.2345e30f) sink[0] = 1.0f;
}
// ---- one 16-byte / 4-byte element per lane at a random index (the candidate / hinted-point
// gathers of k_linearize and the 4-byte gathers of the map build)
__global__ __launch_bounds__(256) void calib_gather16(const float4* __restrict__ in, uint32_t n_el, size_t n_lanes,
                                                      float* sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lanes) return;
    float4 v = in[mix((uint32_t)i * 2654435761u + 777u) % n_el];
    if (v.x + v.y + v.z + v.w == 1.2345e30f) sink[0] = 1.0f;
}
__global__ __launch_bounds__(256) void calib_gather4(const float* __restrict__ in, uint32_t n_el, size_t n_lanes,
                                                     float* sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lanes) return;
    float v = in[mix((uint32_t)i * 2654435761u + 999u) % n_el];
    if (v == 1.2345e30f) sink[0] = 1.0f;
}
// ---- sorted-query gathers: consecutive lanes hit NEARBY elements (within a window), the shape of
// a converged k_linearize launch (neighbouring returns of a LiDAR beam match neighbouring map points)
__global__ __launch_bounds__(256) void calib_gather16_local(const float4* __restrict__ in, uint32_t n_el,
                                                            size_t n_lanes, uint32_t window, float* sink)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_lanes) return;
    // a wavefront's 64 lanes land inside one `window`-element neighbourhood at a random place
    const uint32_t base = mix((uint32_t)(i / 64) * 2654435761u + 31u) % (n_el - window);
    float4 v = in[base + mix((uint32_t)i) % window];
    if (v.x + v.y + v.z + v.w == 1.2345e30f) sink[0] = 1.0f;
}
// ---- stores
template <typename T>
__global__ __launch_bounds__(256) void calib_write(T* out, size_t n, T v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 3;
    const size_t GiB = 1ull << 30;
    const size_t buf_bytes = 6 * GiB;
    char* buf = nullptr;
    float* sink = nullptr;
    CK(hipSetDevice(0));
    CK(hipMalloc((void**)&buf, buf_bytes));
    CK(hipMalloc((void**)&sink, 256));
    CK(hipMemset(buf, 1, buf_bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{");
    bool first = true;
    auto report = [&](const char* name, double rd, double wr, float ms) {
        printf("%s\n \"%s\": {\"read_bytes\": %.0f, \"write_bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}", first ? "" : ",",
               name, rd, wr, ms, (rd + wr) / (ms * 1e-3) / 1e9);
        first = false;
    };
#define TIMED(name, rd, wr, launch)                                     \
    for (int r = 0; r < reps; ++r) {                                    \
        CK(hipEventRecord(e0, 0));                                      \
        launch;                                                         \
        CK(hipGetLastError());                                          \
        CK(hipEventRecord(e1, 0));                                      \
        CK(hipEventSynchronize(e1));                                    \
        float ms = 0;                                                   \
        CK(hipEventElapsedTime(&ms, e0, e1));                           \
        if (r == reps - 1) report(name, rd, wr, ms);                    \
    }
    // streams over 1 GiB each (4x the Infinity Cache), each at its own offset
    {
        const size_t n = GiB / 4;
        TIMED("calib_stream<float>", (double)GiB, 0.0,
              hipLaunchKernelGGL(calib_stream<float>, dim3((n + 255) / 256), dim3(256), 0, 0, (const float*)buf, n, sink, 0xFFFFFFFFu));
    }
    {
        const size_t n = GiB / 8;
        TIMED("calib_stream<float2>", (double)GiB, 0.0,
              hipLaunchKernelGGL(calib_stream<float2>, dim3((n + 255) / 256), dim3(256), 0, 0, (const float2*)(buf + GiB), n, sink, 0xFFFFFFFFu));
    }
    {
        const size_t n = GiB / 16;
        TIMED("calib_stream<float4>", (double)GiB, 0.0,
              hipLaunchKernelGGL(calib_stream<float4>, dim3((n + 255) / 256), dim3(256), 0, 0, (const float4*)(buf + 2 * GiB), n, sink, 0xFFFFFFFFu));
    }
    {
        const size_t n = GiB / 2 / 2;  // 0.5 GiB of ushort
        TIMED("calib_stream<unsigned short>", (double)GiB / 2, 0.0,
              hipLaunchKernelGGL(calib_stream<unsigned short>, dim3((n + 255) / 256), dim3(256), 0, 0,
                                 (const unsigned short*)(buf + 3 * GiB), n, sink, 0xFFFFFFFFu));
    }
    {   // K1's shape: nq quads; reads 3 x 16 + 8 B, writes 3 x 16 B per quad
        const size_t nq = 16u << 20;  // 16 Mi quads = 64 Mi points: 0.94 GiB in, 0.81 GiB out
        const float4* x = (const float4*)buf;
        const float4* y = x + nq;
        const float4* z = y + nq;
        const ushort4* p = (const ushort4*)(z + nq);
        float4* ox = (float4*)(buf + 3 * GiB);
        TIMED("calib_k1_shape", 56.0 * nq, 48.0 * nq,
              hipLaunchKernelGGL(calib_k1_shape, dim3((nq + 255) / 256), dim3(256), 0, 0, x, y, z, p, nq, ox, ox + nq, ox + 2 * nq));
    }
    // gathers from a 4 GiB table; 1 GiB useful per launch
    {
        const size_t lanes = GiB / 16;
        TIMED("calib_rows<128>", (double)GiB, 0.0,
              hipLaunchKernelGGL(calib_rows<128>, dim3((lanes + 255) / 256), dim3(256), 0, 0, (const float4*)buf,
                                 (uint32_t)(4 * GiB / 128), lanes, sink));
        TIMED("calib_rows<64>", (double)GiB, 0.0,
              hipLaunchKernelGGL(calib_rows<64>, dim3((lanes + 255) / 256), dim3(256), 0, 0, (const float4*)buf,
                                 (uint32_t)(4 * GiB / 64), lanes, sink));
        TIMED("calib_rows<32>", (double)GiB, 0.0,
              hipLaunchKernelGGL(calib_rows<32>, dim3((lanes + 255) / 256), dim3(256), 0, 0, (const float4*)buf,
                                 (uint32_t)(4 * GiB / 32), lanes, sink));
        TIMED("calib_gather16", (double)GiB, 0.0,
              hipLaunchKernelGGL(calib_gather16, dim3((lanes + 255) / 256), dim3(256), 0, 0, (const float4*)buf,
                                 (uint32_t)(4 * GiB / 16), lanes, sink));
        TIMED("calib_gather16_local", (double)GiB, 0.0,
              hipLaunchKernelGGL(calib_gather16_local, dim3((lanes + 255) / 256), dim3(256), 0, 0, (const float4*)buf,
                                 (uint32_t)(4 * GiB / 16), lanes, 512u, sink));
    }
    {
        const size_t lanes = GiB / 16;  // 64 Mi gathers of 4 B = 0.25 GiB useful
        TIMED("calib_gather4", 4.0 * lanes, 0.0,
              hipLaunchKernelGGL(calib_gather4, dim3((lanes + 255) / 256), dim3(256), 0, 0, (const float*)buf,
                                 (uint32_t)(4 * GiB / 4 - 1), lanes, sink));
    }
    {
        const size_t n4 = GiB / 4, n16 = GiB / 16;
        TIMED("calib_write<float>", 0.0, (double)GiB,
              hipLaunchKernelGGL(calib_write<float>, dim3((n4 + 255) / 256), dim3(256), 0, 0, (float*)(buf + 4 * GiB), n4, 2.0f));
        TIMED("calib_write<float4>", 0.0, (double)GiB,
              hipLaunchKernelGGL(calib_write<float4>, dim3((n16 + 255) / 256), dim3(256), 0, 0, (float4*)(buf + 5 * GiB), n16,
                                 make_float4(1, 2, 3, 4)));
    }
    printf("\n}\n");
    CK(hipFree(buf));
    CK(hipFree(sink));
    return 0;
}
