// ttn_common.h — shared host/device declarations of libttn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef TTN_WG
#define TTN_WG 1024            // threads of the dense (one-workgroup-per-train) kernels (512 is supported for experiments)
#endif
#define TTN_NWAVES (TTN_WG / 64)
// Register budget of the dense kernels: 128 VGPRs per lane in both builds.  1024 threads = 4 waves per SIMD already imply it; the
// 512-thread build asks for 4 waves per SIMD explicitly (second launch-bounds argument = minimum waves per execution unit), i.e.
// two resident workgroups per CU — without it the compiler would take up to 256 registers and one workgroup would own the CU.
#if TTN_WG == 512 && defined(TTN_WG512_ONE_PER_CU)
#define TTN_KERNEL_BOUNDS __launch_bounds__(512)        // experiment: up to 256 VGPRs, no spills, one workgroup per CU
#elif TTN_WG == 512
#define TTN_KERNEL_BOUNDS __launch_bounds__(512, 4)
#else
#define TTN_KERNEL_BOUNDS __launch_bounds__(TTN_WG)
#endif
// Wave priorities of the dense kernels (s_setprio; see ttn_dense_kernels.h): matrix-product routines 0, everything else
// TTN_PRIO_BASE, the symmetric eigensolver TTN_EIG_PRIO, the one wave that carries the tridiagonalisation's serial chain
// TTN_TRIDIAG_PRIO.  Only the order matters (3/3/2 and 3/3/1 measure the same).  -DTTN_NO_PRIO builds without them: 556 k instead of
// 583 k cores/s on the benchmark (same box), no difference for one train alone on a CU.
#ifndef TTN_NO_PRIO
#ifndef TTN_PRIO_BASE
#define TTN_PRIO_BASE 1
#endif
#ifndef TTN_EIG_PRIO
#define TTN_EIG_PRIO 2
#endif
#ifndef TTN_TRIDIAG_PRIO
#define TTN_TRIDIAG_PRIO 3
#endif
#endif
#define TTN_MAX_D 64           // max chain length handled by the on-stack tables of the host API
#define TTN_SV_NONE (-1)

// Device view of one batch of TT vectors (see include/ttn.h: ttn_tt).
struct TTDev {
    double*        data;      // arena: train b, core k at data + b*stride + off[k]
    long long      stride;    // doubles per train
    const long long* off;     // [d+1] device, slot offsets from CAPACITY ranks
    long long*     rks;       // [batch][d+1] device, current ranks
    const int*     dims;      // [d] device
    const long long* cap;     // [d+1] device, capacity ranks
    int            d;
    int            batch;
};

// Device view of one TT operator.
struct TTODev {
    const double*  data;
    const long long* off;     // [d+1] device
    const long long* rks;     // [d+1] device
    const int*     dims;      // [d] device
    int            d;
};

// A 2-level strided index: idx(i) = q ? (i % q)*lo + (i / q)*hi : i*lo.
// Lets a TT core (n, rl, rr) be read as the matrices the reference forms with
// permutedims+reshape (src/tt_tools.jl:746-752) without any copy.
struct Idx {
    int q;
    long long lo, hi;
};
struct View {
    double* p;
    Idx r, c;
};

__host__ __device__ inline long long ix(const Idx& d, int i) {
    return d.q ? (long long)(i % d.q) * d.lo + (long long)(i / d.q) * d.hi : (long long)i * d.lo;
}
__host__ __device__ inline Idx plain(long long stride) { return Idx{0, stride, 0}; }
__host__ __device__ inline View tview(View v) { return View{v.p, v.c, v.r}; }
__host__ __device__ inline View mkview(double* p, Idx r, Idx c) { return View{p, r, c}; }
__host__ __device__ inline long long minstride(const Idx& d) { return d.q ? (d.lo < d.hi ? d.lo : d.hi) : d.lo; }

// Per-train failure code of a handle (include/ttn.h: ttn_compress_status): the FIRST condition a train meets is kept — only the
// workgroup that owns train b writes status[b], so a plain test is enough.
__device__ inline void ttn_set_status(int* st, int code) { if (*st == 0) *st = code; }
