// ttn_dot_kernels.h — dot(A, B) (src/tt_operations.jl:239-250): the transfer-matrix recurrence with the r^A x r^B state RESIDENT IN
// LDS, both cores of a site read straight from global memory into MFMA operand registers, and the intermediate T never leaving the
// registers of the wave that computed it.
//
//   reference, left to right:   M'[a, b] = sum_{z, al, be} A_k[z, al, a] M[al, be] B_k[z, be, b]
//   here, RIGHT TO LEFT:        T_z[a, be]  = sum_b M[a, b] * B_k[z, be, b]            (one product per physical index z, shared A operand)
//                               M'[al, be]  = sum_{z, a} A_k[z, al, a] * T_z[a, be]
// — the same number (the chain of transfer matrices is a scalar; only the order of the rounding errors differs, 1e-16 relative).  The
// direction decides which rank index of a core the MFMA fragments walk: left to right the 16 lanes of a fragment row run over the
// RIGHT rank index, 1 KB apart in memory, and every wave-level load touches 16 half-used cache lines; right to left they run over the
// LEFT index — contiguous — and a load touches 8 full lines (the first version of this kernel went left to right and was bound by the
// line-request rate of the vector L1: 34 k clk per rank-64 site whatever the schedule).
//
// Round 2 ran this as two calls of the general workgroup GEMM per site (descriptor, offset tables, both operands staged through
// LDS, C written to and re-read from global memory, five barriers per call): 52 k clk per rank-64 site against 16.4 k of matrix-pipe
// time — 19.9 % of the fp64 peak.  For QTT trains (n = 2) whose ranks fit one LDS image (<= 64) a site now costs ONE barrier:
//   * M lives in LDS (k-major with a swizzled row pitch, DOT_AT below: neither the fragment reads nor the atomic adds meet a bank
//     conflict) from the first site to the last;
//   * a core is stored (z, left, right) with z fastest, so ONE 16-byte load per lane yields the fragment entries of z = 0 AND z = 1:
//     B_k[., be, b] feeds the two accumulators T_0, T_1 of a wave's (al, b) tile.  No staging pass, no offset tables, no integer
//     division anywhere in the loops;
//   * the fp64 MFMA accumulator layout D[row = (lane >> 4) + 4 reg][col = lane & 15] IS the B-operand layout of the k-steps
//     k = 4 reg + (lane >> 4): the wave that owns tile (al-block tr, b-block tc) of T holds, as they are, the B fragments of the second
//     product for the 16 values of al of its block.  It therefore computes the PARTIAL sums over its al-block of all four output tiles
//     (a-block ta = 0..3, b-block tc) — A_k[., al, a] again as 16-byte fragment loads, 32 MFMAs as before — and adds them into the
//     next state with LDS atomics (ds_add_f64; the four waves tr = 0..3 of a column block meet there).  T is never written anywhere,
//     there is no barrier between the two products, and the LDS traffic of a site drops from 960 to 576 wave-level operations.
//   * three state buffers rotate: site k reads X, accumulates into Y (all zero when the site starts) and zeroes Z, which becomes the
//     accumulation target of site k + 1 — one barrier separates the zeroing from the adds, the adds from the reads.
// Sites outside that shape class (n != 2, a rank above 64) take the general GEMM path with M in global memory, as before; the
// state moves between the two homes when consecutive sites differ.  The sum order of the atomics is not fixed: results are
// reproducible to rounding (1e-16 relative), not bitwise.
#pragma once
#include "ttn_common.h"
#include "ttn_dense_kernels.h"

#define DOT_RMAX 64                      // largest rank of the LDS-resident form
// Element (al, be) of a state image sits at DOT_AT(be, al) = 80 be + 4 (be >> 1) + al (k-major with a swizzled row pitch).  Two LDS access
// patterns meet in an image and both must be free of bank conflicts (64 lanes x 8 bytes = four passes of 32 doubles at best):
//   * fragment reads of the first product: four rows be = 4 t + lk, 16 consecutive al each — the two rows of a half-wave must differ
//     by 16 doubles modulo 32: 80 = 16 mod 32, and the + 4 (be >> 1) term is the same for both rows of a half-wave;
//   * the atomic adds of the second product: 16 consecutive rows be (= b), four consecutive al each — with a plain pitch of 80 the
//     rows b = 0, 2, 4, ... all start on bank 0 (8-way conflict: the first version of this kernel lost 14 k clk per site to it);
//     the swizzle puts rows 0, 2, 4, 6 / 1, 3, 5, 7 at 0, 4, 8, 12 / 16, 20, 24, 28 modulo 32: eight rows tile the 32 double-banks.
#define DOT_AT(be, al) (80 * (be) + 4 * ((be) >> 1) + (al))
#define DOT_MS_DOUBLES 5248              // >= DOT_AT(63, 63) + 1 = 5228, even
#define DOT_LDS_DOUBLES ((3 * DOT_MS_DOUBLES) > GEMM_LDS_TOTAL ? (3 * DOT_MS_DOUBLES) : GEMM_LDS_TOTAL)
#define DOT_MAX_D 480                    // the per-site table (6 ints per site) must fit the LDS behind the images
#define DOT_LDS_BYTES(d) (sizeof(double) * DOT_LDS_DOUBLES + sizeof(int) * 6 * ((d) + 2))

typedef double dot_f64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) dot_f64x2 gmem_f64x2;

// LDS-only barrier: fragment loads of the NEXT site stay in flight across it (__syncthreads would wait for them: vmcnt(0))
__device__ __forceinline__ void dot_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// fragment pair (z = 0, 1) of core element (., p, q): 16-byte load at 2 * (p + ldp * q); zero outside p < np, q < nq
__device__ __forceinline__ dot_f64x2 dot_load2(const double* core, int p, int q, int np, int nq, int ldp) {
    const bool ok = (p < np) && (q < nq);
    const int off = ok ? 2 * (p + ldp * q) : 0;
    dot_f64x2 v = *(gmem_f64x2*)(core + off);
    if (!ok) { v.x = 0.0; v.y = 0.0; }
    return v;
}

// One site (right to left) on a workgroup of 16 waves; wave w owns tile (tr = w & 3, tc = w >> 2) of T.
//   ra, rb  : RIGHT ranks of A_k, B_k (the state M[a, b] coming in);   ra2, rb2: their LEFT ranks (the state M'[al, be] going out)
//   Mcur : M[a, b] at Mcur[DOT_AT(b, a)], zero outside (ra, rb)
//   Mnxt : all zero on entry; receives M'[al, be] at Mnxt[DOT_AT(be, al)]
//   Mzero: zeroed here (the accumulation target of the NEXT site)
// FULL: ra = ra2 = rb = rb2 = 64 (the interior sites of a rank-64 train) — no masks, constant trip counts, addresses base + immediate.
#define DOT_STAMP(i) if (FULL && stamps) ts_[i] = (long long)__builtin_amdgcn_s_memtime() - tstart;
template <bool FULL>
__device__ __forceinline__ void dot_site(const double* Ak, const double* Bk, int ra, int ra2, int rb, int rb2, const lds_f64* Mcur, lds_f64* Mnxt,
                                         lds_f64* Mzero, long long* stamps = nullptr /* diagnostics: 8 waves x 8 accumulated phase clocks */) {
    const long long tstart = (FULL && stamps) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    long long ts_[6] = {0, 0, 0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int tr = wave & 3, tc = wave >> 2;
    if (FULL) { ra = ra2 = rb = rb2 = DOT_RMAX; }
    // the spare image: 5120 doubles, 16 bytes per lane and pass
    {
        typedef __attribute__((address_space(3))) dot_f64x2 lds_f64x2;
        lds_f64x2* z2 = (lds_f64x2*)Mzero;
        const dot_f64x2 zero2 = {0.0, 0.0};
        for (int e = threadIdx.x; e < DOT_MS_DOUBLES / 2; e += TTN_WG) z2[e] = zero2;
    }
    if (16 * tr < ra && 16 * tc < rb2) {                                  // wave-uniform
        // ---- T_z[a, be] = sum_b M[a, b] B_k[z, be, b]: rows a = 16 tr + ., columns be = 16 tc + . ----
        mfma_acc_t t0 = (mfma_acc_t){0.0, 0.0, 0.0, 0.0}, t1 = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
        const int bq = 16 * tc + li;
        dot_f64x2 av[4][4];                                               // FULL: every A fragment of the second product, requested up front
        if (FULL) {
#pragma unroll
            for (int ta = 0; ta < 4; ++ta) {
                gmem_f64x2* ap = (gmem_f64x2*)(Ak + 2 * (16 * ta + li + DOT_RMAX * (16 * tr + lk)));     // A_k[., al = 16 ta + li, a = 16 tr + 4 r + lk]
#pragma unroll
                for (int r = 0; r < 4; ++r) av[ta][r] = ap[4 * DOT_RMAX * r];
            }
            gmem_f64x2* bp = (gmem_f64x2*)(Bk + 2 * (bq + DOT_RMAX * lk));                 // B_k[., be = bq, b = 4 t + lk]: + 512 t doubles per k-step
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const dot_f64x2 bv = bp[4 * DOT_RMAX * t];
                const double a = Mcur[DOT_AT(4 * t + lk, 16 * tr + li)];
                t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv.x, t0, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv.y, t1, 0, 0, 0);
            }
        } else {
            const int nt = (rb + 3) >> 2;                                 // k-steps of four b
            for (int t = 0; t < nt; ++t) {
                const dot_f64x2 bv = dot_load2(Bk, bq, 4 * t + lk, rb2, rb, rb2);
                const double a = Mcur[DOT_AT(4 * t + lk, 16 * tr + li)];
                t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv.x, t0, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv.y, t1, 0, 0, 0);
            }
        }
        DOT_STAMP(0)
        // ---- partial M'[al, be] over a in this wave's block: sum_{z, r} A_k[z, al, 16 tr + 4 r + lk] * T_z[16 tr + 4 r + lk, be] — the
        //      accumulator register r of T_z is the B fragment of k-step r as it is ----
        const int nr = FULL ? 4 : min(4, (ra - 16 * tr + 3) >> 2);        // k-steps of this block that hold rows a < ra
        const int nta = FULL ? 4 : (ra2 + 15) >> 4;
#pragma unroll
        for (int ta = 0; ta < 4; ++ta) {
            if (ta < nta) {                                               // wave-uniform
                mfma_acc_t m = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
                const int aq = 16 * ta + li;
                if (FULL) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        m = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ta][r].x, t0[r], m, 0, 0, 0);
                        m = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ta][r].y, t1[r], m, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r < nr) {
                            const dot_f64x2 av = dot_load2(Ak, aq, 16 * tr + 4 * r + lk, ra2, ra, ra2);
                            m = __builtin_amdgcn_mfma_f64_16x16x4f64(av.x, t0[r], m, 0, 0, 0);
                            m = __builtin_amdgcn_mfma_f64_16x16x4f64(av.y, t1[r], m, 0, 0, 0);
                        }
                    }
                }
                if (ta == 3) { DOT_STAMP(1) }
                // M'[al = 16 ta + lk + 4 reg, be = 16 tc + li] += m[reg]     (entries beyond (ra2, rb2) are exact zeros: masked fragments)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    __hip_atomic_fetch_add(Mnxt + DOT_AT(16 * tc + li, 16 * ta + lk + 4 * reg), m[reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    DOT_STAMP(2)
    dot_lds_barrier();
    DOT_STAMP(3)
    if (FULL && stamps && (threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        const int slot = w < 4 ? w : (w == 4 ? 4 : (w == 8 ? 5 : (w == 12 ? 6 : (w == 15 ? 7 : -1))));
        if (slot >= 0) for (int i = 0; i < 6; ++i) stamps[8 * slot + i] += ts_[i];
    }
}

__device__ __noinline__ void dot_site_masked(const double* Ak, const double* Bk, int ra, int ra2, int rb, int rb2, const lds_f64* Mcur, lds_f64* Mnxt,
                                             lds_f64* Mzero) {
    Ak = unip(Ak); Bk = unip(Bk); ra = uni32(ra); ra2 = uni32(ra2); rb = uni32(rb); rb2 = uni32(rb2);
    dot_site<false>(Ak, Bk, ra, ra2, rb, rb2, Mcur, Mnxt, Mzero);
}

// A site outside the LDS-resident shape class: two calls of the general workgroup GEMM, M (ra x rb, column-major) and T in global memory
__device__ __noinline__ void dot_site_generic(double* Ak, double* Bk, double* Mc, double* Mn, double* Tb, int n, int ra, int ra2, int rb, int rb2, double* lds) {
    Ak = unip(Ak); Bk = unip(Bk); Mc = unip(Mc); Mn = unip(Mn); Tb = unip(Tb); lds = unip(lds);
    n = uni32(n); ra = uni32(ra); ra2 = uni32(ra2); rb = uni32(rb); rb2 = uni32(rb2);
    const View Mv = mkview(Mc, plain(1), plain(ra));
    const View Bv = mkview(Bk, plain(n), Idx{n, 1, (long long)n * rb});           // B as [be, (z + n*b)]
    const View Tv = mkview(Tb, plain(n), Idx{n, 1, (long long)n * ra});           // T as [al, (z + n*b)] stored at z + n*al + n*ra*b
    wg_gemm(ra, n * rb2, rb, Mv, Bv, Tv, 1.0, 0.0, lds);
    const View Atv = mkview(Ak, plain((long long)n * ra), plain(1));              // A^T as [a, (z + n*al)]
    const View T2v = mkview(Tb, plain(1), plain((long long)n * ra));
    const View Mnv = mkview(Mn, plain(1), plain(ra2));
    wg_gemm(ra2, rb2, n * ra, Atv, T2v, Mnv, 1.0, 0.0, lds);
}

// per-site table in LDS (behind the images): ranks and core offsets of every site are read from global memory ONCE, at the start —
// read site by site they put a dependent global round trip (~1 k clk) at the head of every site of a train
__global__ void __launch_bounds__(TTN_WG) k_dot_fused(DotArgs P) {
    extern __shared__ double lds[];
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    const TTDev& A = P.a; const TTDev& B = P.b;
    const int d = A.d;
    double* scr = P.scratch + (long long)t * P.scratch_stride;
    double* M0 = scr;
    double* M1 = M0 + (long long)P.ramax * P.rbmax;
    double* Tb = M1 + (long long)P.ramax * P.rbmax;
    lds_f64* img = (lds_f64*)lds;                                         // three state images, rotating
    lds_i32* tab = (lds_i32*)((lds_f64*)lds + DOT_LDS_DOUBLES);           // [0] ra_k, [1] rb_k, [2] n_k, [3] offA_k (doubles), [4] offB_k, stride 6
    for (int k = tid; k <= d; k += TTN_WG) {
        tab[6 * k + 0] = (int)A.rks[(long long)t * (d + 1) + k];
        tab[6 * k + 1] = (int)B.rks[(long long)t * (d + 1) + k];
        tab[6 * k + 2] = k < d ? A.dims[k] : 0;
        tab[6 * k + 3] = k < d ? (int)A.off[k] : 0;                        // a train is far below 2^31 doubles (checked by the host)
        tab[6 * k + 4] = k < d ? (int)B.off[k] : 0;
    }
    __syncthreads();
    // the whole train runs in ONE direction: right to left with the state in LDS when every site fits, else left to right on the GEMMs
    bool all_fit = true;
    for (int k = 0; k < d; ++k)
        all_fit = all_fit && uni32(tab[6 * k + 2]) == 2 && uni32(tab[6 * k]) <= DOT_RMAX && uni32(tab[6 * k + 1]) <= DOT_RMAX;
    all_fit = all_fit && uni32(tab[6 * d]) <= DOT_RMAX && uni32(tab[6 * d + 1]) <= DOT_RMAX;
    double* Abase = A.data + (long long)t * A.stride;
    double* Bbase = B.data + (long long)t * B.stride;
    if (all_fit) {
        const int ra_last = uni32(tab[6 * d]), rb_last = uni32(tab[6 * d + 1]);              // 1 x 1 in practice (not enforced by the reference)
        for (int e = tid; e < 2 * DOT_MS_DOUBLES; e += TTN_WG) img[e] = 0.0;
        __syncthreads();
        // dot() starts from out = e_1 e_1^T and returns out[1, 1] of the last state (tt_operations.jl:241-249): read right to left, the
        // start state is e_1 e_1^T at the right end and the result is entry [1, 1] of the state that leaves site 1 (boundary ranks are 1
        // in every train the reference builds: the 1 x 1 matrix [1])
        for (int e = tid; e < ra_last * rb_last; e += TTN_WG) img[DOT_AT(e / ra_last, e % ra_last)] = (e == 0) ? 1.0 : 0.0;
        __syncthreads();
        int cur = 0;                                                     // image that holds M; (cur + 1) % 3 is all zero, (cur + 2) % 3 is free
        for (int k = d - 1; k >= 0; --k) {
            const int ra2 = uni32(tab[6 * k]), rb2 = uni32(tab[6 * k + 1]);               // left ranks: the outgoing state
            const int ra = uni32(tab[6 * k + 6]), rb = uni32(tab[6 * k + 7]);             // right ranks: the incoming state
            double* Ak = Abase + uni32(tab[6 * k + 3]);
            double* Bk = Bbase + uni32(tab[6 * k + 4]);
            const int nx = cur == 2 ? 0 : cur + 1, sp = nx == 2 ? 0 : nx + 1;
            if (ra == DOT_RMAX && ra2 == DOT_RMAX && rb == DOT_RMAX && rb2 == DOT_RMAX)
                dot_site<true>(Ak, Bk, ra, ra2, rb, rb2, img + cur * DOT_MS_DOUBLES, img + nx * DOT_MS_DOUBLES, img + sp * DOT_MS_DOUBLES,
                               P.prof ? P.prof + 136LL * gridDim.x + 64LL * t : nullptr);
            else
                dot_site_masked(Ak, Bk, ra, ra2, rb, rb2, img + cur * DOT_MS_DOUBLES, img + nx * DOT_MS_DOUBLES, img + sp * DOT_MS_DOUBLES);
            cur = nx;
            if (P.prof && tid == 0 && k < 120) P.prof[16LL * gridDim.x + 120LL * t + (d - 1 - k)] = (long long)__builtin_amdgcn_s_memtime();
        }
        if (tid == 0) P.out[t] = (double)img[cur * DOT_MS_DOUBLES];
        return;
    }
    double* Mc = M0; double* Mn = M1;
    if (tid == 0) M0[0] = 1.0;
    __syncthreads();
    for (int k = 0; k < d; ++k) {
        const int ra = uni32(tab[6 * k]), rb = uni32(tab[6 * k + 1]), n = uni32(tab[6 * k + 2]);
        const int ra2 = uni32(tab[6 * k + 6]), rb2 = uni32(tab[6 * k + 7]);
        dot_site_generic(Abase + uni32(tab[6 * k + 3]), Bbase + uni32(tab[6 * k + 4]), Mc, Mn, Tb, n, ra, ra2, rb, rb2, lds);
        double* tmp = Mc; Mc = Mn; Mn = tmp;
    }
    if (tid == 0) P.out[t] = Mc[0];
}
