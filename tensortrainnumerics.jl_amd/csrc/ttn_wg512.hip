// ttn_wg512.hip — the 512-THREAD build of the bond-step machinery (second translation unit of libttn_hip.so).
//
// The same sources as the 1024-thread kernels of ttn_api.hip (ttn_dense_kernels.h, ttn_eig_kernels.h), compiled with TTN_WG = 512
// inside their own namespace: 8 waves per workgroup, the LDS image halved (TTN_LDS_IMG = 64 x 128 doubles, < 80 KB in all) and at
// most 128 VGPRs per lane, so that TWO workgroups — two trains — are resident on every CU.  A bond step is a chain of short
// dependent phases (reflectors, pivots, bisection rounds, one barrier after another): one workgroup per CU leaves two thirds of
// the VALU issue slots idle (SQ_ACTIVE_INST_VALU 34 %, profiles/r01o_pmc.json); the second resident train fills them.
// ttn_api.hip launches this build for batches that have more trains than the chip has CUs (throughput), the 1024-thread build
// otherwise (latency of a single train).  Only the launchers below cross the translation-unit boundary.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include <string.h>

#define TTN_WG 512
namespace ttn_wg512 {
#include "ttn_common.h"
#include "ttn_dense_kernels.h"
#include "ttn_eig_kernels.h"
}  // namespace ttn_wg512

// The argument structs are laid out identically in both builds (same headers): the caller fills its own CompressArgs and hands
// over the bytes.
extern "C" {

// dynamic LDS above the default 64 KiB must be granted per kernel (ttn_init calls this once per process)
int ttn_wg512_init(void) {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ttn_wg512::k_compress), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COMPRESS_LDS_BYTES)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ttn_wg512::k_selftest_eig128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COMPRESS_LDS_BYTES)) != hipSuccess) return (int)e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ttn_wg512::k_selftest_gemm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COMPRESS_LDS_BYTES)) != hipSuccess) return (int)e;
    return 0;
}

size_t ttn_wg512_compress_args_bytes(void) { return sizeof(ttn_wg512::CompressArgs); }
size_t ttn_wg512_lds_bytes(void) { return COMPRESS_LDS_BYTES; }

int ttn_wg512_launch_compress(const void* args, size_t nbytes, int grid, hipStream_t stream) {
    if (nbytes != sizeof(ttn_wg512::CompressArgs)) return (int)hipErrorInvalidValue;
    ttn_wg512::CompressArgs P;
    memcpy(&P, args, sizeof(P));
    hipLaunchKernelGGL(ttn_wg512::k_compress, dim3(grid), dim3(TTN_WG), COMPRESS_LDS_BYTES, stream, P);
    return (int)hipGetLastError();
}

// self-tests of the building blocks in this build (tests/test_gpu_kernels.py): device pointers, caller's stream
int ttn_wg512_selftest_eig(const double* G, double* Vst, int n, int r, int nev, double* sig, double* Xout, long long* clk, hipStream_t stream) {
    hipLaunchKernelGGL(ttn_wg512::k_selftest_eig128, dim3(1), dim3(TTN_WG), COMPRESS_LDS_BYTES, stream, G, Vst, n, r, nev, sig, Xout, clk);
    return (int)hipGetLastError();
}
int ttn_wg512_selftest_gemm(int m, int n, int k, double* A, double* B, double* C, double alpha, double beta, int ta, int tb, hipStream_t stream) {
    hipLaunchKernelGGL(ttn_wg512::k_selftest_gemm, dim3(1), dim3(TTN_WG), COMPRESS_LDS_BYTES, stream, m, n, k, A, B, C, alpha, beta, ta, tb);
    return (int)hipGetLastError();
}

int ttn_wg512_bench_gemm(int m, int n, int k, double* A, double* B, double* C, int ta, int tb, int reps, long long* cycles, int grid, hipStream_t stream) {
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute(reinterpret_cast<const void*>(ttn_wg512::k_bench_gemm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)COMPRESS_LDS_BYTES); attr = true; }
    hipLaunchKernelGGL(ttn_wg512::k_bench_gemm, dim3(grid), dim3(TTN_WG), COMPRESS_LDS_BYTES, stream, m, n, k, A, B, C, ta, tb, reps, cycles);
    return (int)hipGetLastError();
}

}  // extern "C"
