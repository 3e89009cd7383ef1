// Cost of a device-scope barrier among G co-scheduled workgroups (atomic arrive + spin on a generation word in global memory):
// the synchronisation a cooperative multi-workgroup bond step would need between its split GEMMs.  hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k(int G, int reps, unsigned int* cnt, volatile unsigned int* gen, long long* out) {
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned int g = *gen;
            if (atomicAdd(cnt, 1u) == (unsigned)G - 1) { *cnt = 0; __threadfence(); atomicAdd((unsigned int*)gen, 1u); }
            else while (*gen == g) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (t1 - t0) / reps;
}
int main() {
    unsigned int *cnt, *gen; long long* out;
    hipMalloc(&cnt, 4); hipMalloc(&gen, 4); hipMalloc(&out, 8);
    for (int G : {2, 4, 8, 16}) {
        hipMemset(cnt, 0, 4); hipMemset(gen, 0, 4);
        hipLaunchKernelGGL(k, dim3(G), dim3(1024), 0, 0, G, 2000, cnt, gen, out);
        long long h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("device-scope barrier among %2d workgroups: %lld clk per barrier\n", G, h);
    }
    return 0;
}
