#include <hip/hip_runtime.h>
#include <cstdio>
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
__global__ void k(const double* C, double* out)
{
    const double C0 = C[0], C1 = C[1], C2 = C[2], C3 = C[3], C4 = C[4], C5 = C[5];
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
    vx *= inv; vy *= inv; vz *= inv;
    out[4] = vx; out[5] = vy; out[6] = vz;
    const bool flip = (vz < 0.0) || (vz == 0.0 && (vy < 0.0 || (vy == 0.0 && vx < 0.0)));
    if (flip) { vx = -vx; vy = -vy; vz = -vz; }
    out[0] = vx; out[1] = vy; out[2] = vz; out[3] = flip ? 1.0 : 0.0;
    out[7] = A[0][0]; out[8] = A[1][1]; out[9] = A[2][2]; out[10] = m;
}
int main()
{
    const double C[6] = {0x1.6db6db6db6db8p+0, 0x1.b6db692492492p-2, 0x0p+0, 0x1.b6db6892492b6p+1, 0x0p+0, 0x1p+1};
    double *dC, *dO, O[11];
    hipMalloc(&dC, sizeof C); hipMalloc(&dO, sizeof O);
    hipMemcpy(dC, C, sizeof C, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, dC, dO);
    hipMemcpy(O, dO, sizeof O, hipMemcpyDeviceToHost);
    printf("out %a %a %a flip %g | pre %a %a %a | A %a %a %a m %g\n", O[0], O[1], O[2], O[3], O[4], O[5], O[6], O[7], O[8], O[9], O[10]);
    return 0;
}
