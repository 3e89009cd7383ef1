// Micro-benchmark (gfx950): issue rate of v_fmac_f64 with a DPP row_newbcast source against plain v_fma_f64, for 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/dpp_rate scratch/dpp_rate.hip && scratch/dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int J>
__device__ __forceinline__ void fmac_bcast(double& acc, double src, double mul) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(J));
}
template <int MODE>
__global__ void k(double* out, long long* clk, int iters) {
    double a[16];
    const double src = out[threadIdx.x & 63], mul = 1.0 + 1e-9 * threadIdx.x;
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = j;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define U(j) fmac_bcast<j>(a[j], src, mul);
            U(0) U(1) U(2) U(3) U(4) U(5) U(6) U(7) U(8) U(9) U(10) U(11) U(12) U(13) U(14) U(15)
#undef U
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[j]) : "v"(src), "v"(mul));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a[j];
    out[64 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    double* out; long long* clk;
    hipMalloc(&out, 8 * 4096); hipMalloc(&clk, 64);
    hipMemset(out, 0, 8 * 4096);
    const int iters = 4096;
    for (int mode = 0; mode < 2; ++mode)
        for (int nt : {64, 256, 512, 1024}) {
            long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(nt), 0, 0, out, clk, iters);
                else hipLaunchKernelGGL(k<1>, dim3(1), dim3(nt), 0, 0, out, clk, iters);
                hipDeviceSynchronize();
                hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
            }
            printf("%s threads %4d (%d waves/SIMD): %.2f ticks per instruction per wave-slot (16 independent accumulators)\n", mode ? "v_fma_f64      " : "v_fmac_f64_dpp ",
                   nt, nt <= 256 ? 1 : nt / 256, (double)h / (16.0 * iters));
        }
    return 0;
}
