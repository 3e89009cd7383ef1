// latency microbenchmarks for the diagonal-block chain (one wave): s_memtime around unrolled dependent sequences
#include <hip/hip_runtime.h>
#include <cstdio>
template <int J>
__device__ __forceinline__ void fmac_bcast(double& a, double w, double s_) {
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(w), "v"(s_), "n"(J));
}
template <int J>
__device__ __forceinline__ void fmac_bcast_nonop(double& a, double w, double s_) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(w), "v"(s_), "n"(J));
}
__device__ __forceinline__ double rl(double v, int idx) {
    union { double d; int i[2]; } u, r; u.d = v;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], idx); r.i[1] = __builtin_amdgcn_readlane(u.i[1], idx); return r.d;
}
__device__ __forceinline__ double rsq2(double d) {
    double y = __builtin_amdgcn_rsq(d);
    double h = -0.5 * d;
    y = y * fma(h * y, y, 1.5);
    y = y * fma(h * y, y, 1.5);
    return y;
}
__global__ void spin(long long* o, long long n) { long long t = 0; for (long long i = 0; i < n; ++i) { t += __builtin_amdgcn_s_memtime() & 1; } o[0] = t; }
__global__ void k(double* out, long long* clk, double seed) {
    const int lane = threadIdx.x & 63;
    double a = seed + lane, b = 1.0 + 1e-9 * lane, c = 0.5;
    long long t0, t1;
    // (0) 64 dependent v_fma_f64
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) { a = fma(a, b, c); asm volatile("" : "+v"(a)); }
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) clk[0] = t1 - t0;
    // (1) 64 independent-ish fmas (4 chains)
    double a0 = a, a1 = a + 1, a2 = a + 2, a3 = a + 3;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c); asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) clk[1] = t1 - t0;
    a = a0 + a1 + a2 + a3;
    // (2) 64 DPP fmacs, independent accumulators (16 regs x 4), with s_nop
    double e[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i] = a + i;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        fmac_bcast<0>(e[0], a, b); fmac_bcast<1>(e[1], a, b); fmac_bcast<2>(e[2], a, b); fmac_bcast<3>(e[3], a, b);
        fmac_bcast<4>(e[4], a, b); fmac_bcast<5>(e[5], a, b); fmac_bcast<6>(e[6], a, b); fmac_bcast<7>(e[7], a, b);
        fmac_bcast<8>(e[8], a, b); fmac_bcast<9>(e[9], a, b); fmac_bcast<10>(e[10], a, b); fmac_bcast<11>(e[11], a, b);
        fmac_bcast<12>(e[12], a, b); fmac_bcast<13>(e[13], a, b); fmac_bcast<14>(e[14], a, b); fmac_bcast<15>(e[15], a, b);
    }
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) clk[2] = t1 - t0;
    // (3) the same without s_nop
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        fmac_bcast_nonop<0>(e[0], a, b); fmac_bcast_nonop<1>(e[1], a, b); fmac_bcast_nonop<2>(e[2], a, b); fmac_bcast_nonop<3>(e[3], a, b);
        fmac_bcast_nonop<4>(e[4], a, b); fmac_bcast_nonop<5>(e[5], a, b); fmac_bcast_nonop<6>(e[6], a, b); fmac_bcast_nonop<7>(e[7], a, b);
        fmac_bcast_nonop<8>(e[8], a, b); fmac_bcast_nonop<9>(e[9], a, b); fmac_bcast_nonop<10>(e[10], a, b); fmac_bcast_nonop<11>(e[11], a, b);
        fmac_bcast_nonop<12>(e[12], a, b); fmac_bcast_nonop<13>(e[13], a, b); fmac_bcast_nonop<14>(e[14], a, b); fmac_bcast_nonop<15>(e[15], a, b);
    }
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) clk[3] = t1 - t0;
    // (4) 16 x (readlane -> rsq2 -> mul) dependent
    double x = fabs(e[3]) + 2.0;
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) { const double d = rl(x, i); const double r_ = rsq2(d); x = x * r_ + 1.5; }
    asm volatile("" : "+v"(x));
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) clk[4] = t1 - t0;
    // (5) 64 plain fma with an SGPR operand obtained by readlane (2 readlanes + fma), independent accumulators
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const double s_ = rl(x, i + 16 * r); e[i] = fma(a, s_, e[i]); }
    }
    asm volatile("" : "+v"(e[0]));
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) clk[5] = t1 - t0;
    // (6) 16 dependent DPP fmacs on ONE accumulator
    t0 = __builtin_amdgcn_s_memtime();
    fmac_bcast<0>(e[0], e[0], b); fmac_bcast<1>(e[0], e[0], b); fmac_bcast<2>(e[0], e[0], b); fmac_bcast<3>(e[0], e[0], b);
    fmac_bcast<4>(e[0], e[0], b); fmac_bcast<5>(e[0], e[0], b); fmac_bcast<6>(e[0], e[0], b); fmac_bcast<7>(e[0], e[0], b);
    fmac_bcast<8>(e[0], e[0], b); fmac_bcast<9>(e[0], e[0], b); fmac_bcast<10>(e[0], e[0], b); fmac_bcast<11>(e[0], e[0], b);
    fmac_bcast<12>(e[0], e[0], b); fmac_bcast<13>(e[0], e[0], b); fmac_bcast<14>(e[0], e[0], b); fmac_bcast<15>(e[0], e[0], b);
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) clk[6] = t1 - t0;
    // (7) calibration: 64 x s_nop 15 = 1024 clk
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 64; ++i) asm volatile("s_nop 15");
    t1 = __builtin_amdgcn_s_memtime(); if (lane == 0) clk[7] = t1 - t0;
    double s = x;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += e[i];
    out[threadIdx.x] = s;
}
int main() {
    double* out; long long* clk;
    hipMalloc(&out, 64 * 8); hipMalloc(&clk, 8 * 8);
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, clk, 200000LL);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, clk, 1.0);
        long long h[8]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        printf("64 dep fma %lld | 64 fma 4 chains %lld | 64 dpp fmac (nop) %lld | 64 dpp fmac %lld | 16 x readlane+rsq2+mul %lld | 64 readlane+fma %lld | 16 dep dpp %lld | 64 x s_nop 15 (1024 clk) %lld\n",
               h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    return 0;
}
