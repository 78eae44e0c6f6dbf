// cu_mask_probe.hip -- where do the workgroups of a stream made by hipExtStreamCreateWithCUMask run?  (gfx950: 8 XCDs x
// 32 CUs.)  Prints, per XCD, the set of CU ids (HW_ID: CU_ID + 16 * SH_ID, per SE) seen by a 20 000-workgroup kernel
// on a plain stream and on a stream with the mask velo_map_roll_begin uses (the first three quarters of the bits), and
// where single bits of the mask land: bit i = XCD i % 8, shader engine (i / 8) % 4, CU (i / 32) of that engine.
//   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void k_where(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
        for (volatile int i = 0; i < 2000; ++i) {}
    }
}
static void run(hipStream_t s, const char* tag)
{
    const int n = 20000;
    unsigned* d;
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k_where, dim3(n), dim3(256), 0, s, d);
    std::vector<unsigned> h(n);
    hipMemcpyAsync(h.data(), d, n * 4, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    std::map<unsigned, std::set<unsigned>> per;
    for (unsigned v : h) {
        const unsigned xcc = v >> 16, hw = v & 0xffff;
        // HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
        per[xcc].insert(((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 15));
    }
    size_t total = 0;
    for (auto& kv : per) total += kv.second.size();
    std::printf("%s: %zu distinct (se, sh, cu) over %zu XCDs:", tag, total, per.size());
    for (auto& kv : per) std::printf(" xcd%u=%zu", kv.first, kv.second.size());
    std::printf("\n");
    if (per.size() && per.begin()->second.size() <= 16) {   // small sets: which (se, cu) they are, XCD 0
        std::printf("   xcd%u:", per.begin()->first);
        for (unsigned v : per.begin()->second) std::printf(" se%u.cu%u", v >> 8, v & 31);
        std::printf("\n");
    }
    hipFree(d);
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    hipStream_t plain, masked;
    hipStreamCreateWithFlags(&plain, hipStreamNonBlocking);
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int i = 0; i < ncu - ncu / 4; ++i) mask[i / 32] |= 1u << (i % 32);
    const hipError_t e = hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask.size(), mask.data());
    std::printf("CUs %d, mask words %zu, create: %s\n", ncu, mask.size(), hipGetErrorString(e));
    run(plain, "plain ");
    if (e == hipSuccess) run(masked, "masked");
    // a mask with only the first quarter of the bits set: which CUs are those?
    std::vector<uint32_t> m2((ncu + 31) / 32, 0u);
    for (int i = 0; i < ncu / 4; ++i) m2[i / 32] |= 1u << (i % 32);
    hipStream_t q;
    if (hipExtStreamCreateWithCUMask(&q, (uint32_t)m2.size(), m2.data()) == hipSuccess) run(q, "first quarter of the bits");
    // single bits: where does bit i of the mask land?  (bits 0, 1, 2, 8, 9, 16, 32, 33, 64, 128)
    for (int bit : {0, 1, 2, 3, 4, 8, 9, 16, 24, 32, 33, 40, 64, 128}) {
        std::vector<uint32_t> m3((ncu + 31) / 32, 0u);
        m3[bit / 32] |= 1u << (bit % 32);
        hipStream_t q3;
        if (hipExtStreamCreateWithCUMask(&q3, (uint32_t)m3.size(), m3.data()) == hipSuccess) {
            char tag[32];
            std::snprintf(tag, sizeof tag, "bit %d", bit);
            run(q3, tag);
        }
    }
    return 0;
}
