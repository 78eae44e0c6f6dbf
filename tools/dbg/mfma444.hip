// Layout probe for v_mfma_f64_4x4x4_4b_f64 on gfx950: one-hot A and B lanes, which D lane lights up.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int la, int lb, double* out)
{
    const int l = threadIdx.x;
    const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
int main()
{
    double* d;
    hipMalloc(&d, 64 * sizeof(double));
    double h[64];
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, la, lb, d);
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l)
                if (h[l] != 0.0) printf(" (B%d->D%d)", lb, l);
        }
        printf("\n");
    }
    return 0;
}
