// Peak probe: back-to-back v_mfma_f64_16x16x4_f64 (and 4x4x4, and v_fma_f64) on ONE compute unit, W waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(1024) k(long long* cyc, double* sink, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-9, y = 1.0 + threadIdx.x * 1e-12;
    double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3, f4 = x + 4, f5 = x + 5, f6 = x + 6, f7 = x + 7;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        } else if (MODE == 1) {
            f0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, f1, 0, 0, 0);
            f2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, f2, 0, 0, 0);
            f3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, f3, 0, 0, 0);
        } else {
            f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x);
            f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x); f7 = fma(f7, y, x);
        }
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) *cyc = t1 - t0;
    sink[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
int main() {
    long long* dc; double* ds; hipMalloc(&dc, 8); hipMalloc(&ds, 8 * 1024);
    const int iters = 2000;
    for (int threads : {256, 512, 1024}) {
        long long c;
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, dc, ds, iters); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("mfma16x16x4  threads %4d: %.1f clk per MFMA per SIMD, %.1f flop/clk/CU\n", threads, (double)c / (iters * 4.0 * (threads / 256)), 2048.0 * iters * 4 * (threads / 64) / c);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, dc, ds, iters); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("mfma4x4x4    threads %4d: %.1f clk per MFMA per SIMD, %.1f flop/clk/CU\n", threads, (double)c / (iters * 4.0 * (threads / 256)), 512.0 * iters * 4 * (threads / 64) / c);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, dc, ds, iters); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        printf("v_fma_f64    threads %4d: %.1f clk per FMA per SIMD, %.1f flop/clk/CU\n", threads, (double)c / (iters * 8.0 * (threads / 256)), 128.0 * iters * 8 * (threads / 64) / c);
    }
    return 0;
}
