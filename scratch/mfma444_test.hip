// Layout probe for v_mfma_f64_4x4x4f64 used as a 16-lane all-reduce (two MFMAs against a matrix of ones).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(double* out, double* out1) {
    const int lane = threadIdx.x;
    const double x = (double)(1 << (lane & 15)) + 65536.0 * (double)(lane >> 4) ;
    const double s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, 1.0, 0.0, 0, 0, 0);
    const double s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(s1, 1.0, 0.0, 0, 0, 0);
    out1[lane] = s1;
    out[lane] = s2;
}
int main() {
    double *d, *d1, h[64], h1[64];
    hipMalloc(&d, 64 * 8); hipMalloc(&d1, 64 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d1);
    hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(h1, d1, 64 * 8, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        const double want = 65535.0 + 16 * 65536.0 * (l >> 4);
        if (h[l] != want) ok = 0;
        printf("lane %2d  s1 %.0f  s2 %.0f  want %.0f\n", l, h1[l], h[l], want);
    }
    printf(ok ? "ALLREDUCE_OK\n" : "ALLREDUCE_MISMATCH\n");
    return 0;
}
