// accuracy of v_rsq_f64 and of one / two Newton steps on it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double d = x[i];
    double y = __builtin_amdgcn_rsq(d);
    y0[i] = y;
    double h = -0.5 * d;
    y = y * fma(h * y, y, 1.5); y1[i] = y;
    y = y * fma(h * y, y, 1.5); y2[i] = y;
}
int main() {
    const int n = 1 << 20;
    double* hx = new double[n]; double *dx, *d0, *d1, *d2;
    for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 200) - 100);
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    double *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / sqrtl((long double)hx[i]);
        e0 = fmax(e0, fabs((double)((h0[i] - t) / t))); e1 = fmax(e1, fabs((double)((h1[i] - t) / t))); e2 = fmax(e2, fabs((double)((h2[i] - t) / t)));
    }
    printf("max relative error: v_rsq_f64 %.3e, + one Newton step %.3e, + two %.3e\n", e0, e1, e2);
    return 0;
}
