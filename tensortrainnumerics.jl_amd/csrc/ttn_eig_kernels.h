// ttn_eig_kernels.h — symmetric eigen-decomposition of a 128 x 128 Gram matrix by ONE 1024-thread workgroup, the fast replacement
// of "Cholesky + one-sided Jacobi on L" in the Gram route of the bond step:
//   1. Householder tridiagonalisation, the matrix REGISTER-resident (thread = row i, 16-column chunk c): per column 16 FMAs for
//      the symmetric matrix-vector product and 32 for the rank-2 update, vectors broadcast from LDS, 4 barriers;
//   2. eigenvalues by multisection on Sturm counts (T lanes per eigenvalue, log2(T+1) bits per round, fast reciprocal:
//      the count is that of a matrix whose off-diagonal squares are perturbed by 2^-50 relative);
//   3. eigenvectors of the kept eigenvalues by one twisted factorisation each (Fernando 1997; LAPACK dlar1v without the
//      representation tree): D+ and D- pivots in LDS, one lane per vector, no pivoting, no reorthogonalisation —
//      scratch/eig_feasibility2.py measures |U'U - I| ~ 1e-13 on merged matrices of the benchmark; the caller's a-posteriori
//      check (FAST_CHECK_TOL) and its Householder + Jacobi fallback cover the rest;
//   4. back-transformation by the stored reflectors (lane = column, wave = 8 rows).
#pragma once
#include "ttn_dense_kernels.h"

#define EIG_N 128
// a += (lane J of the caller's row of 16 lanes of w) * s : the DPP form of the fp64 multiply-add broadcasts a lane of each row for
// free (row_newbcast is the one DPP control 64-bit VALU ops take on gfx90a+), which replaces an LDS broadcast read per operand
// HAZARD NOTE: gfx9 needs 2 wait states between a VALU write of a VGPR and a DPP read of it, and the compiler does not see the DPP
// inside this asm.  Every register passed as `w` below is therefore loaded (LDS / memory) and never touched by the VALU before
// its use, with an s_nop after the loads; a build whose register allocator inserts a copy in between would produce inaccurate
// eigenpairs, which the routes' a-posteriori check (FAST_CHECK_TOL) turns into a fallback to Householder + Jacobi, not into wrong
// results.  Putting `s_nop 1` into the asm itself is the belt-and-braces variant (measured: -3 % on the headline).
template <int J>
__device__ __forceinline__ void fmac_bcast(double& a, double w, double s_) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(w), "v"(s_), "n"(J));
}
#define EIG_BCAST16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
// LDS map (doubles) inside the 128*128 + 512 Jacobi/GEMM image `L`:
//   [0, 16384)            phase 1: vL, wL, xcol (3 x 128), part (8 x 128)   | phase 3: D+ then D- ([row][lane], 2 x 8192)
//                         phase 4: partial dot products (2 x 16 x 64), then the output image X (ld 128)
//   [16384, 16896)        dg, e, lam, beta (4 x 128)
#define EIG_TAIL TTN_LDS_IMG

// ---- 1. tridiagonalisation: dg[0..127], e[0..126] (e[k] couples k, k+1), reflectors v_k (rows of Vst, zeros up to k) and beta_k ----
// Per column k:  v_k, beta_k (reflector of column k below the diagonal);  p = beta A v;  w = p - (beta/2)(p'v) v;  A -= v w' + w v'.
// The reflector and w are "scalar-ish" code run by wave 0 alone.  Look-ahead: the owners of row k+1 publish it BEFORE the update
// of step k, so wave 0 can form column k+1 of the updated matrix (x - v w_{k+1} - w v_{k+1}) and the NEXT reflector while the
// other waves are still applying update k — 3 barriers per column and the reflector off the critical path.
// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding GLOBAL stores (the
// reflector rows written to Vst for the back-transformation), ~1 k clk per column of the tridiagonalisation for nothing.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// 64-lane sum, result in every lane: two v_mfma_f64_4x4x4 against ones reduce the 16 lanes that share a quad position (grp_sum,
// ttn_dense_kernels.h), two DPP row rotations add the four quad positions.  Whole-wave control flow only.
__device__ __forceinline__ double wave64_sum_mfma(double v) {
    v = grp_sum(v);
    v += dpp_mov_f64<0x124>(v);      // row_ror:4
    v += dpp_mov_f64<0x128>(v);      // row_ror:8
    return v;
}
// lane `idx` (wave-uniform) of a double
__device__ __forceinline__ double readlane_f64(double v, int idx) {
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], idx);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], idx);
    return r.d;
}
__device__ __forceinline__ void tridiag_reflector(int k, double x0, double x1, double xk1, int lane, lds_f64* vbuf, double* Vst,
                                                  lds_f64* dg, lds_f64* e, lds_f64* beta, double dgk) {
    // lanes hold entries lane, lane + 64 of column k (zero at and above the diagonal position k)
    const double s2 = wave64_sum_mfma(fma(x0, x0, x1 * x1));
    // sqrt and reciprocal from the hardware seeds + Newton steps (full fp64 accuracy; this chain is on the critical path)
    const double alpha = (s2 > 0.0) ? -copysign(s2 * fast_rsqrt2(s2), xk1) : 0.0;
    const double den = s2 - alpha * xk1;                           // = v'v / 2
    double bta = 0.0;
    if (den > 0.0) { bta = fast_rcp(den); bta = fma(fma(-den, bta, 1.0), bta, bta); }
    const double v0 = x0 - ((lane == k + 1) ? alpha : 0.0), v1 = x1 - ((lane + 64 == k + 1) ? alpha : 0.0);
    vbuf[lane] = v0; vbuf[lane + 64] = v1;
    // global_store, not flat_store: a flat store also counts on lgkmcnt, and the LDS barrier right after would wait for it (~800 clk)
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble* vg = (gdouble*)(unsigned long long)(Vst + k * 128);
    vg[lane] = v0; vg[lane + 64] = v1;
    if (lane == 0) { dg[k] = dgk; e[k] = alpha; beta[k] = bta; }
}

template <int N>
__device__ __noinline__ void wg_tridiag(const double* Gg, int ldg, double* Vst, double* lds) {
    Gg = unip(Gg); Vst = unip(Vst); lds = unip(lds); ldg = uni32(ldg);
    constexpr bool TWO = (N == 128);                              // wave 0 holds one (N = 64) or two entries of a vector per lane
    // thread = (row i, chunk c of CW columns), tid < N * NC: 16 columns per thread, or 32 when the workgroup has fewer than N*N/16
    // threads (N = 128 in the 512-thread build: 64 VGPRs of matrix per lane)
    constexpr int CW = (N * N / 16 > TTN_WG) ? 32 : 16;
    constexpr int NH = CW / 16;                                   // 16-lane broadcast groups per chunk
    constexpr int NC = N / CW;
    static_assert(N * NC <= TTN_WG, "one thread per (row, column chunk)");
    lds_f64* L = (lds_f64*)lds;
    lds_f64 *vbuf0 = L, *vbuf1 = L + 128, *wL = L + 256, *xnext = L + 384, *part = L + 512;      // part: 8 x 128
    lds_f64 *dg = L + EIG_TAIL, *e = dg + 128, *beta = dg + 384;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tid & (N - 1), c = tid / N;
    const bool actv = tid < N * NC;
    double a[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) a[j] = actv ? Gg[i + (long long)ldg * (CW * c + j)] : 0.0;
#ifndef TTN_TRIDIAG_V1
    // ---- schedule (round 2): the reflector is formed in TWO halves around a barrier, and the matrix-vector product runs on the
    // known column x~ instead of v.  v_k = x~_k - alpha_k e_{k+1} (x~_k = column k of the current matrix below the diagonal), so
    //   A v_k = A x~_k - alpha_k A[:, k+1],   A[:, k+1] = row k+1 = the look-ahead buffer —
    // the product needs x~_k only, which wave 0 has ~100 clk after w_{k-1}, not the scalars alpha, beta that take it another ~900.
    // Per column k, three LDS barriers as before:
    //   S1  all: y = A x~_k (partial sums)            | wave 0 first: beta_k, v_k -> LDS / global, (d_k, e_k, beta_k)
    //   S2  wave 0: p = beta (y - alpha row_{k+1}), w_k = p - (beta/2)(p'v) v
    //   S3  others: A -= v w' + w v', row k+2 out      | wave 0: x~_{k+1} -> LDS, ||x~||^2, alpha_{k+1}   (first half of reflector k+1)
    // i.e. the product overlaps the second half of the reflector and the update its first half: 2.05 k clk of dependent work per
    // column instead of 2.4 k.
    lds_f64* xt = L + 1664;                                       // x~ of the current column (128)
    if (actv && i == 0) {
#pragma unroll
        for (int j = 0; j < CW; ++j) wL[CW * c + j] = a[j];      // row 0 (wL is free here)
    }
    if (actv && i == 1) {
#pragma unroll
        for (int j = 0; j < CW; ++j) xnext[CW * c + j] = a[j];
    }
    __syncthreads();
    // wave 0's state of the reflector in flight: x~ (two entries per lane), alpha, v'v / 2, the diagonal entry
    double rx0 = 0.0, rx1 = 0.0, ralpha = 0.0, rden = 0.0, rdg = 0.0;
    if (wave == 0) {
        rx0 = (lane > 0) ? wL[lane] : 0.0; rx1 = TWO ? wL[lane + 64] : 0.0;
        const double xk1 = wL[1];
        rdg = wL[0];
        xt[lane] = rx0; xt[lane + 64] = rx1;
        const double s2 = wave64_sum_mfma(fma(rx0, rx0, rx1 * rx1));
        ralpha = (s2 > 0.0) ? -copysign(s2 * fast_rsqrt2(s2), xk1) : 0.0;
        rden = s2 - ralpha * xk1;
    }
    for (int k = 0; k < N - 2; ++k) {
        lds_f64* vL = (k & 1) ? vbuf1 : vbuf0;
        lds_barrier();                                             // x~_k, row k+1 (xnext) are visible; update k-1 is done
        const bool live = actv && (CW * (c + 1) > k + 1) && ((TWO ? 64 * (wave & 1) : 0) + 63 > k);
        double v0 = 0.0, v1 = 0.0, w0 = 0.0, w1 = 0.0, bta = 0.0;
        // ---- S1 ----
        if (wave == 0) {
#ifdef TTN_TRIDIAG_PRIO
            __builtin_amdgcn_s_setprio(TTN_TRIDIAG_PRIO);          // the serial chain of the column ahead of everybody's parallel work
#endif
            if (rden > 0.0) { bta = fast_rcp(rden); bta = fma(fma(-rden, bta, 1.0), bta, bta); }
            v0 = rx0 - ((lane == k + 1) ? ralpha : 0.0); v1 = rx1 - ((lane + 64 == k + 1) ? ralpha : 0.0);
            vL[lane] = v0; vL[lane + 64] = v1;
            typedef __attribute__((address_space(1))) double gdouble;
            gdouble* vg = (gdouble*)(unsigned long long)(Vst + k * 128);
            vg[lane] = v0; vg[lane + 64] = v1;
            if (lane == 0) { dg[k] = rdg; e[k] = ralpha; beta[k] = bta; }
        }
        if (live) {
            double xreg[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) xreg[h] = xt[CW * c + 16 * h + (lane & 15)];
            double pp = 0.0;
            asm volatile("s_nop 1");
#define EIG_MV(j) fmac_bcast<j>(pp, xreg[0], a[j]);
            EIG_BCAST16(EIG_MV)
#undef EIG_MV
            if constexpr (NH == 2) {
#define EIG_MV(j) fmac_bcast<j>(pp, xreg[1], a[16 + j]);
                EIG_BCAST16(EIG_MV)
#undef EIG_MV
            }
            part[c * 128 + i] = pp;
        } else if (actv) part[c * 128 + i] = 0.0;
        lds_barrier();
        // ---- S2 ----
        lds_f64* xn_r = (k & 1) ? xnext + 1152 : xnext;          // row k+1 before update k (= column k+1 of the matrix the product used)
        lds_f64* xn_w = (k & 1) ? xnext : xnext + 1152;          // written this step (row k+2 after update k)
        if (wave == 0) {
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) { p0 += part[cc * 128 + lane]; if (TWO) p1 += part[cc * 128 + lane + 64]; }
            p0 = bta * fma(-ralpha, xn_r[lane], p0); p1 = TWO ? bta * fma(-ralpha, xn_r[lane + 64], p1) : 0.0;
            const double Kc = 0.5 * bta * wave64_sum_mfma(fma(p0, v0, p1 * v1));
            w0 = fma(-Kc, v0, p0); w1 = fma(-Kc, v1, p1);
            wL[lane] = w0; if (TWO) wL[lane + 64] = w1;
        }
        lds_barrier();
        // ---- S3 ----
        if (wave == 0 && k + 1 < N - 2) {
            // column k+1 of the updated matrix: x - v w_{k+1} - w v_{k+1}; first half of reflector k+1
            const double wk1 = wL[k + 1], vk1 = vL[k + 1];
            const double c0 = fma(-v0, wk1, fma(-w0, vk1, xn_r[lane]));
            const double c1 = TWO ? fma(-v1, wk1, fma(-w1, vk1, xn_r[lane + 64])) : 0.0;
            const int kk = k + 1;
            rdg = (kk < 64) ? readlane_f64(c0, kk) : readlane_f64(c1, kk - 64);
            const double xk1 = (kk + 1 < 64) ? readlane_f64(c0, kk + 1) : readlane_f64(c1, kk + 1 - 64);
            rx0 = (lane > kk) ? c0 : 0.0; rx1 = (TWO && lane + 64 > kk) ? c1 : 0.0;
            xt[lane] = rx0; xt[lane + 64] = rx1;
            const double s2 = wave64_sum_mfma(fma(rx0, rx0, rx1 * rx1));
            ralpha = (s2 > 0.0) ? -copysign(s2 * fast_rsqrt2(s2), xk1) : 0.0;
            rden = s2 - ralpha * xk1;
        }
        if (wave == 0) {
#ifdef TTN_TRIDIAG_PRIO
#ifdef TTN_EIG_PRIO
            __builtin_amdgcn_s_setprio(TTN_EIG_PRIO);
#else
            __builtin_amdgcn_s_setprio(0);
#endif
#endif
        }
        if (live) {
            // A -= v w' + w v'
            double vreg[NH], wreg[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) { vreg[h] = vL[CW * c + 16 * h + (lane & 15)]; wreg[h] = wL[CW * c + 16 * h + (lane & 15)]; }
            const double nvi = -vL[i], nwi = -wL[i];
            asm volatile("s_nop 1");
#define EIG_UP(j) fmac_bcast<j>(a[j], wreg[0], nvi); fmac_bcast<j>(a[j], vreg[0], nwi);
            EIG_BCAST16(EIG_UP)
#undef EIG_UP
            if constexpr (NH == 2) {
#define EIG_UP(j) fmac_bcast<j>(a[16 + j], wreg[1], nvi); fmac_bcast<j>(a[16 + j], vreg[1], nwi);
                EIG_BCAST16(EIG_UP)
#undef EIG_UP
            }
            if (i == k + 2) {
#pragma unroll
                for (int j = 0; j < CW; ++j) xn_w[CW * c + j] = a[j];
            }
        }
    }
#else
    // prologue: rows 0 and 1 of the matrix; reflector 0
    if (actv && i == 0) {
#pragma unroll
        for (int j = 0; j < CW; ++j) wL[CW * c + j] = a[j];      // row 0 (wL is free here)
    }
    if (actv && i == 1) {
#pragma unroll
        for (int j = 0; j < CW; ++j) xnext[CW * c + j] = a[j];
    }
    __syncthreads();
    if (wave == 0) {
        const double x0 = (lane > 0) ? wL[lane] : 0.0, x1 = TWO ? wL[lane + 64] : 0.0;
        tridiag_reflector(0, x0, x1, wL[1], lane, vbuf0, Vst, dg, e, beta, wL[0]);
    }
    for (int k = 0; k < N - 2; ++k) {
        lds_f64* vL = (k & 1) ? vbuf1 : vbuf0;
        lds_f64* vN = (k & 1) ? vbuf0 : vbuf1;
        lds_barrier();                                             // v_k, row k+1 (xnext) are visible
        // p = A v (partial over the thread's 16 columns): lane l of a row of 16 lanes holds v[16c + l], the DPP multiply-add
        // broadcasts it — 2 LDS reads per thread and step instead of 48
        // A wave owns 64 rows of one 16-column chunk.  Once all its columns are <= k (v is zero there and the entries are never read
        // again) or all its rows are <= k (finished rows: their w only feeds dead entries), it has nothing left to do but to keep its
        // partial sums at zero: on average 7 of the 16 waves (N = 128) still work, and the phase is bound by the fp64 multiply-adds.
        const bool live = actv && (CW * (c + 1) > k + 1) && ((TWO ? 64 * (wave & 1) : 0) + 63 > k);
        double vreg[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) vreg[h] = 0.0;
        if (live) {
#pragma unroll
            for (int h = 0; h < NH; ++h) vreg[h] = vL[CW * c + 16 * h + (lane & 15)];
            double pp = 0.0;
            asm volatile("s_nop 1");
#define EIG_MV(j) fmac_bcast<j>(pp, vreg[0], a[j]);
            EIG_BCAST16(EIG_MV)
#undef EIG_MV
            if constexpr (NH == 2) {
#define EIG_MV(j) fmac_bcast<j>(pp, vreg[1], a[16 + j]);
                EIG_BCAST16(EIG_MV)
#undef EIG_MV
            }
            part[c * 128 + i] = pp;
        } else if (actv) part[c * 128 + i] = 0.0;
        lds_barrier();
        double v0 = 0.0, v1 = 0.0, w0 = 0.0, w1 = 0.0;
        if (wave == 0) {
#ifdef TTN_TRIDIAG_PRIO
            __builtin_amdgcn_s_setprio(TTN_TRIDIAG_PRIO);          // the serial chain of the column ahead of everybody's parallel work
#endif
            const double bta = beta[k];
            v0 = vL[lane]; v1 = TWO ? vL[lane + 64] : 0.0;
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) { p0 += part[cc * 128 + lane]; if (TWO) p1 += part[cc * 128 + lane + 64]; }
            p0 *= bta; p1 *= bta;
            const double Kc = 0.5 * bta * wave64_sum_mfma(fma(p0, v0, p1 * v1));
            w0 = fma(-Kc, v0, p0); w1 = fma(-Kc, v1, p1);
            wL[lane] = w0; if (TWO) wL[lane + 64] = w1;
        }
        lds_barrier();
        // row k+2 of the UPDATED matrix goes to the look-ahead buffer the next step reads (two buffers alternate)
        lds_f64* xn_r = (k & 1) ? xnext + 1152 : xnext;          // read this step (row k+1 before update k)
        lds_f64* xn_w = (k & 1) ? xnext : xnext + 1152;          // written this step (row k+2 after update k)
        // wave 0 first forms the next reflector (the other waves are busy with their share of the update meanwhile), then its own share
        if (wave == 0 && k + 1 < N - 2) {
            // column k+1 of the updated matrix: x - v w_{k+1} - w v_{k+1}, then reflector k+1 (entries <= k+1 are not part of it)
            const double wk1 = wL[k + 1], vk1 = vL[k + 1];
            const double c0 = fma(-v0, wk1, fma(-w0, vk1, xn_r[lane]));
            const double c1 = TWO ? fma(-v1, wk1, fma(-w1, vk1, xn_r[lane + 64])) : 0.0;
            // entries k+1 (the new diagonal) and k+2 by lane reads
            const int kk = k + 1;
            const double dgk = (kk < 64) ? readlane_f64(c0, kk) : readlane_f64(c1, kk - 64);
            const double xk1 = (kk + 1 < 64) ? readlane_f64(c0, kk + 1) : readlane_f64(c1, kk + 1 - 64);
            const double x0 = (lane > kk) ? c0 : 0.0, x1 = (TWO && lane + 64 > kk) ? c1 : 0.0;
            tridiag_reflector(kk, x0, x1, xk1, lane, vN, Vst, dg, e, beta, dgk);
#ifdef TTN_TRIDIAG_PRIO
#ifdef TTN_EIG_PRIO
            __builtin_amdgcn_s_setprio(TTN_EIG_PRIO);
#else
            __builtin_amdgcn_s_setprio(0);
#endif
#endif
        }
        if (live) {
            // A -= v w' + w v'
            double wreg[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) wreg[h] = wL[CW * c + 16 * h + (lane & 15)];
            const double nvi = -vL[i], nwi = -wL[i];
            asm volatile("s_nop 1");
#define EIG_UP(j) fmac_bcast<j>(a[j], wreg[0], nvi); fmac_bcast<j>(a[j], vreg[0], nwi);
            EIG_BCAST16(EIG_UP)
#undef EIG_UP
            if constexpr (NH == 2) {
#define EIG_UP(j) fmac_bcast<j>(a[16 + j], wreg[1], nvi); fmac_bcast<j>(a[16 + j], vreg[1], nwi);
                EIG_BCAST16(EIG_UP)
#undef EIG_UP
            }
            if (i == k + 2) {
#pragma unroll
                for (int j = 0; j < CW; ++j) xn_w[CW * c + j] = a[j];
            }
        }
    }
#endif
    // the last 2 x 2 block
    __syncthreads();
    if (actv && i == N - 2 && c == NC - 1) { dg[N - 2] = a[CW - 2]; e[N - 2] = a[CW - 1]; }
    if (actv && i == N - 1 && c == NC - 1) { dg[N - 1] = a[CW - 1]; }
    __syncthreads();
}

// Number of eigenvalues of the tridiagonal below x = number of sign changes in the sequence of leading principal minors
//   p_0 = 1, p_1 = d_0 - x, p_{i+1} = (d_i - x) p_i - e_{i-1}^2 p_{i-1}
// (product form of the Sturm sequence: no reciprocal in the dependent chain).  The matrix is held in REGISTERS, distributed over
// the 16 lanes of every row of lanes (lane l holds d_{16g+l}, e^2_{16g+l-1} for g = 0..7), and each step takes its two
// coefficients through the DPP row broadcast of the fp64 multiply-add — no LDS traffic at all in the 128-step loop (broadcast LDS
// reads cost ~16 clk of the LDS pipe per wave and were the bound of this phase).  The pair (p_i, p_{i-1}) is rescaled by a power
// of two when it leaves [2^-400, 2^400] (checked every 16 steps); an exact zero minor takes the sign opposite to its predecessor.
// ereg holds -e^2 (the sign folded in).  Sign changes are counted from a shift register of sign bits (one v_alignbit per step);
// `zero` reports whether some minor was exactly zero (the caller then repeats the evaluation with sturm_count_guarded).
template <int N>
__device__ __forceinline__ int sturm_count(const double (&dreg)[N / 16], const double (&ereg)[N / 16], double x, bool& zero) {
    double pc = 1.0, pp = 0.0;                                    // p_i, p_{i-1}
    const double nx = -x, one = 1.0;
    int cnt = 0;
    bool z = false;
    unsigned int prev = 0;                                        // sign bit of the last minor of the previous group (p_0 = 1 > 0)
#pragma unroll
    for (int g = 0; g < N / 16; ++g) {
        unsigned int bits = 0;
#define EIG_ST(j) {                                                                                               \
            double dmx = nx;                                                                                      \
            fmac_bcast<j>(dmx, dreg[g], one);                     /* d_i - x in one rounding */                    \
            double pn = dmx * pc;                                                                                 \
            fmac_bcast<j>(pn, ereg[g], pp);                       /* ... - e_{i-1}^2 p_{i-1}: one instruction fewer than forming the product apart */ \
            z |= (pn == 0.0);                                                                                     \
            bits = __builtin_amdgcn_alignbit(bits, (unsigned int)(__double_as_longlong(pn) >> 32), 31);           \
            pp = pc; pc = pn; }
        EIG_BCAST16(EIG_ST)
#undef EIG_ST
        // bits: bit 15 = sign of the first minor of the group ... bit 0 = the last; changes between neighbours, and against `prev`
        const unsigned int seq = (prev << 16) | (bits & 0xffffu);
        cnt += __popc((seq ^ (seq >> 1)) & 0xffffu);
        prev = bits & 1u;
        const double ap = fabs(pc);
        if (ap < 0x1p-400 || ap > 0x1p400) {
            const double sc = (ap < 1.0) ? 0x1p500 : 0x1p-500;
            pc *= sc; pp *= sc;
        }
    }
    zero = z;
    return cnt;
}
// the same count with the zero-minor rule applied step by step (slow path, taken only when `zero` was reported)
__device__ __noinline__ int sturm_count_guarded(const lds_f64* de, double x, int n) {
    double pc = 1.0, pp = 0.0;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        double pn = fma(de[2 * i] - x, pc, de[2 * i + 1] * pp);
        if (pn == 0.0) pn = copysign(DBL_MIN, -pc);
        cnt += (int)(((unsigned long long)(__double_as_longlong(pn) ^ __double_as_longlong(pc))) >> 63);
        pp = pc; pc = pn;
        const double ap = fabs(pc);
        if (ap < 0x1p-400 || ap > 0x1p400) { const double sc = (ap < 1.0) ? 0x1p500 : 0x1p-500; pc *= sc; pp *= sc; }
    }
    return cnt;
}

// ---- 2. eigenvalues lam[0..nev-1] in DESCENDING order (the nev largest), multisection with TL lanes per eigenvalue ----
template <int N, int TL>
__device__ void wg_bisect(int nev, double* lds) {
    lds_f64* L = (lds_f64*)lds;
    lds_f64 *dg = L + EIG_TAIL, *e = dg + 128, *lam = dg + 256;
    lds_f64* e2 = L;                                             // phase-1 work area is free
    const int tid = threadIdx.x;
    lds_f64* de = L + 256;                                        // pairs (d_i, -e_{i-1}^2); the DPP source registers below must come
                                                                  // straight from ds_read: a VALU write right before a DPP read of the
                                                                  // same register is a hazard the compiler cannot see through inline asm
    for (int t = tid; t < N; t += TTN_WG) { e2[t] = (t < N - 1) ? e[t] * e[t] : 0.0; de[2 * t] = dg[t]; de[2 * t + 1] = (t > 0) ? -(e[t - 1] * e[t - 1]) : 0.0; }
    __syncthreads();
    // Gershgorin interval and the pivot floor: wave 0, results through LDS (e2[128..130])
    if (tid < 64) {
        double glo_ = 1e300, ghi_ = -1e300, emax_ = 0.0;
        for (int t = tid; t < N; t += 64) {
            const double rad = ((t > 0) ? fabs(e[t - 1]) : 0.0) + ((t < N - 1) ? fabs(e[t]) : 0.0);
            glo_ = fmin(glo_, dg[t] - rad); ghi_ = fmax(ghi_, dg[t] + rad);
            emax_ = fmax(emax_, e2[t]);
        }
        glo_ = -wave_max(-glo_); ghi_ = wave_max(ghi_); emax_ = wave_max(emax_);
        if (tid == 0) { e2[128] = glo_; e2[129] = ghi_; e2[130] = emax_; }
    }
    __syncthreads();
    double glo = e2[128], ghi = e2[129];
    const double emax = e2[130];
    const double span = fmax(fabs(glo), fabs(ghi));
    glo -= 2.0 * DBL_EPSILON * span * N; ghi += 2.0 * DBL_EPSILON * span * N;
    const double pivmin = DBL_MIN * fmax(1.0, emax);
    double dreg[N / 16], ereg[N / 16];
#pragma unroll
    for (int g = 0; g < N / 16; ++g) { dreg[g] = de[2 * (16 * g + (tid & 15))]; ereg[g] = de[2 * (16 * g + (tid & 15)) + 1]; }
    asm volatile("s_nop 1");
    for (int g0 = 0; g0 < nev; g0 += TTN_WG / TL) {
        const int gi = g0 + tid / TL, sub = tid % TL;
        const bool act = gi < nev;
        const int jasc = N - 1 - (act ? gi : 0);            // ascending index of this group's eigenvalue
        double lo = glo, hi = ghi;
        // rounds: (TL + 1)-section; stop when the interval is at rounding level
        for (int round = 0; round < 64; ++round) {
            const double wdt = hi - lo;
            const bool done = !(wdt > 2.0 * DBL_EPSILON * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin);
            if (__syncthreads_and(done || !act)) break;
            const double x = lo + wdt * ((double)(sub + 1) / (double)(TL + 1));
            // waves whose groups are all beyond nev skip the count (wave-uniform branch): they would only compete for issue slots
            const bool wave_act = g0 + (tid & ~63) / TL < nev;
            bool zero = false;
            int cnt = wave_act ? sturm_count<N>(dreg, ereg, x, zero) : 0;
            if (zero) cnt = sturm_count_guarded(de, x, N);
            // nf = number of section points with count <= jasc (monotone in sub): the eigenvalue lies right of point nf-1
            int nf = (cnt <= jasc) ? 1 : 0;
#pragma unroll
            for (int m = 1; m < TL; m <<= 1) nf += __shfl_xor(nf, m);
            if (!done) {
                const double nlo = (nf > 0) ? lo + wdt * ((double)nf / (double)(TL + 1)) : lo;
                const double nhi = (nf < TL) ? lo + wdt * ((double)(nf + 1) / (double)(TL + 1)) : hi;
                lo = nlo; hi = nhi;
            }
        }
        if (act && sub == 0) lam[gi] = 0.5 * (lo + hi);
    }
    __syncthreads();
}

// ---- 3. eigenvectors of lam[0..r-1] (r <= 64) by twisted factorisations; z (unnormalised) stays split over the D+ / D- arrays,
//         twist index and 1/||z|| per vector in tw[], zn[].  One lane per vector; wave 0 runs the top-down recurrences (D+, then
//         z above the twist), wave 1 the bottom-up ones (D-, z below the twist) at the same time.  The recurrences are dependent
//         chains: the operands of 8 steps are fetched from LDS together. ----
// D- (and the part of z below the twist) lives in LDS behind D+ when both fit the image (2 * N * 64 doubles), otherwise in global
// memory (`Dmg`, N * 64 doubles: the dead Gram matrix) — [row][lane] there too, so every access is one coalesced 512-byte line,
// the stores are fire-and-forget and the loads of 8 steps are issued together like the LDS ones.
template <int N>
struct EigDm {
    static constexpr bool IN_LDS = 2 * N * 64 <= TTN_LDS_IMG;
    typedef __attribute__((address_space(1))) double gdouble;
    lds_f64* l;
    gdouble* g;
    __device__ __forceinline__ double ld(int idx) const { if constexpr (IN_LDS) return l[idx]; else return g[idx]; }
    __device__ __forceinline__ void st(int idx, double v) const { if constexpr (IN_LDS) l[idx] = v; else g[idx] = v; }
};
template <int N>
__device__ void wg_twisted(int r, double* lds, int* tw /*LDS 64 ints*/, lds_f64* zn /*LDS 64 + 64 (partial norms)*/, double* Dmg) {
    lds_f64* L = (lds_f64*)lds;
    lds_f64 *dg = L + EIG_TAIL, *e = dg + 128, *lam = dg + 256;
    lds_f64* Dp = L;                                           // [row][lane]
    EigDm<N> Dm;
    Dm.l = L + N * 64; Dm.g = (typename EigDm<N>::gdouble*)(unsigned long long)Dmg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (d_i, e_i) pairs in the reflector scalars' slot are not available: the bisection's pair array lives in L[256..512) = Dp rows
    // 4..7, so the chains read dg / e directly in chunks
    double emax = 0.0;
    for (int t = 0; t < N - 1; ++t) emax = fmax(emax, e[t] * e[t]);
    const double pivmin = DBL_MIN * fmax(1.0, emax);
    __syncthreads();                                             // the bisection's arrays (aliasing Dp) are no longer read
    const bool act = lane < r;
    const double lm = lam[act ? lane : 0];
    if (wave == 0) {                                             // D+_0 = d_0 - lam ; D+_i = (d_i - lam) - e_{i-1}^2 / D+_{i-1}
        double q = 1.0;
        for (int i0 = 0; i0 < N; i0 += 8) {
            double dd[8], ee[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { dd[t] = dg[i0 + t]; ee[t] = (i0 + t > 0) ? e[i0 + t - 1] : 0.0; }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                q = fma(-(ee[t] * ee[t]), fast_rcp(q), dd[t] - lm);
                if (fabs(q) < pivmin) q = -pivmin;
                Dp[(i0 + t) * 64 + lane] = q;
            }
        }
    } else if (wave == 1) {                                      // D-_{n-1} = d_{n-1} - lam ; D-_i = (d_i - lam) - e_i^2 / D-_{i+1}
        double q = 1.0;
        for (int i0 = N - 8; i0 >= 0; i0 -= 8) {
            double dd[8], ee[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { dd[t] = dg[i0 + t]; ee[t] = (i0 + t < N - 1) ? e[i0 + t] : 0.0; }
#pragma unroll
            for (int t = 7; t >= 0; --t) {
                q = fma(-(ee[t] * ee[t]), fast_rcp(q), dd[t] - lm);
                if (fabs(q) < pivmin) q = -pivmin;
                Dm.st((i0 + t) * 64 + lane, q);
            }
        }
    }
    __syncthreads();                                             // (a workgroup barrier also completes the global stores of D-)
    if (wave < 2) {
        // twist index: argmin |gamma_i|, gamma_i = D+_i + D-_i - (d_i - lam)   (both waves, redundantly: independent iterations)
        double gbest = 1e300;
        int kb = 0;
        for (int i0 = 0; i0 < N; i0 += 8) {
            double dm[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) dm[t] = Dm.ld((i0 + t) * 64 + lane);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double g = fabs(Dp[(i0 + t) * 64 + lane] + dm[t] - (dg[i0 + t] - lm));
                if (g < gbest) { gbest = g; kb = i0 + t; }
            }
        }
        // z_k = 1; wave 0: upwards with D+, wave 1: downwards with D-; the entries replace the pivots they consumed
        double z = 1.0, nrm = 0.0;
        if (wave == 0) {
            for (int i = kb - 1; i >= 0; --i) {
                z = -(e[i] * fast_rcp(Dp[i * 64 + lane])) * z;
                Dp[i * 64 + lane] = z;
                nrm = fma(z, z, nrm);
            }
            if (act) { tw[lane] = kb; zn[lane] = nrm; }
        } else {
            if constexpr (EigDm<N>::IN_LDS) {
                for (int i = kb; i < N - 1; ++i) {
                    z = -(e[i] * fast_rcp(Dm.ld((i + 1) * 64 + lane))) * z;
                    Dm.st((i + 1) * 64 + lane, z);
                    nrm = fma(z, z, nrm);
                }
            } else {
                // global D-: the twist index differs per lane, so the chain runs over ALL rows in chunks of 8 with the loads of a
                // chunk in flight together; rows at or above a lane's twist are left alone
                for (int i0 = 0; i0 < N - 1; i0 += 8) {
                    double dm[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) dm[t] = (i0 + t + 1 < N) ? Dm.ld((i0 + t + 1) * 64 + lane) : 1.0;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int i = i0 + t;
                        if (i >= kb && i < N - 1) {
                            z = -(e[i] * fast_rcp(dm[t])) * z;
                            Dm.st((i + 1) * 64 + lane, z);
                            nrm = fma(z, z, nrm);
                        }
                    }
                }
            }
            if (act) zn[64 + lane] = nrm;
        }
    }
    __syncthreads();
    if (tid < 64 && tid < r) zn[tid] = 1.0 / sqrt(1.0 + zn[tid] + zn[64 + tid]);
    __syncthreads();
}

// ---- 4. back-transformation and the driver ----
// Eigen-decomposition of the symmetric positive definite N x N matrix G (global, column-major, leading dimension ldg), N = 128 or
// 64: the `nev` largest eigenvalues (descending) -> sig[j] = sqrt(lam_j) (global), and the image X (LDS, ld 128):
// X[j*128 + row] = sqrt(lam_j) * u_j[row] for j < r (r <= 64, r <= nev).  Vst: 128 x 128 doubles of global scratch.
// Returns 0, or 1 if a wanted eigenvalue is not positive.  When the D+ / D- arrays of the twisted factorisations do not both fit
// the LDS image (N = 128 in the 512-thread build) G ITSELF IS OVERWRITTEN (its first N * 64 doubles hold D-): it is dead once
// the tridiagonalisation has loaded it.
template <int N>
__device__ __noinline__ int wg_eig_n(const double* Gg, int ldg, double* Vst, int r, int nev, double* sig, double* lds, int* iwork /*LDS 64 ints*/,
                                     double* dwork /*LDS 128*/, long long* prof) {
    Gg = unip(Gg); Vst = unip(Vst); sig = unip(sig); lds = unip(lds); iwork = unip(iwork); dwork = unip(dwork);
    r = uni32(r); nev = uni32(nev); ldg = uni32(ldg);
    constexpr int RPL = N / 16;                                   // rows per lane in the back-transformation
    lds_f64* L = (lds_f64*)lds;
    lds_f64 *lam = L + EIG_TAIL + 256, *beta = L + EIG_TAIL + 384;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define EIG_MARK(slot) if (prof && threadIdx.x == 0) prof[slot] = (long long)__builtin_amdgcn_s_memtime();
    EIG_MARK(2)
#ifdef TTN_EIG_PRIO
    __builtin_amdgcn_s_setprio(TTN_EIG_PRIO);                  // the whole solver is latency bound: ahead of a co-resident workgroup's GEMM phases
#endif
    wg_tridiag<N>(Gg, ldg, Vst, lds);
    EIG_MARK(3)
    // lanes per eigenvalue: the Sturm loop is issue bound, so the optimum is ONE busy wave per SIMD (4 waves) — 4 lanes for the
    // 33..64 eigenvalues of the headline steps (23 rounds of 5-section), 8 lanes for up to 32 (17 rounds of 9-section); more
    // eigenvalues than that (nev = N with truncerr > 0) simply occupy more waves
    if (nev <= 32) wg_bisect<N, 8>(nev, lds);
    else wg_bisect<N, 4>(nev, lds);
    // eigenvalues out; those whose vectors are wanted must be positive (the others are only reported: 0 if not positive)
    int bad = 0;
    for (int j = tid; j < nev; j += TTN_WG) { const double l = lam[j]; sig[j] = (l > 0.0) ? sqrt(l) : 0.0; bad |= (j < r) && !(l > 0.0); }
    if (__syncthreads_or(bad)) {
#ifdef TTN_EIG_PRIO
        __builtin_amdgcn_s_setprio(0); TTN_SETPRIO_BASE();
#endif
        return 1;
    }
    EIG_MARK(4)
    wg_twisted<N>(r, lds, iwork, (lds_f64*)dwork, const_cast<double*>(Gg));
    EIG_MARK(5)
    // Z into registers: waves 0..7; a row of 16 lanes owns TWO columns (8 per wave), lane rc of the row holds rows RPL*rc ..
    // RPL*rc + RPL-1 of both — the dot products v_k' z are reductions over the 16 lanes of a row (4 DPP adds each): no LDS
    // reduction and no barrier in the loop.  The reflectors are staged in LDS once (the D+ / D- arrays are dead after the load of Z).
    lds_f64* Dp = L;
    EigDm<N> Dm;
    Dm.l = L + N * 64; Dm.g = (typename EigDm<N>::gdouble*)(unsigned long long)Gg;
    // lane groups as in the Jacobi (grp_sum): the 16 lanes {B + 4g + 16k} form group g; member index rc; the 16-lane sums of the
    // two dot products per reflector then run on the matrix pipe (2 x 2 v_mfma_f64_4x4x4 instead of 2 x 12 VALU instructions)
    const int rc = (lane & 3) | ((lane >> 4) << 2), col0 = 8 * (wave & 7) + 2 * ((lane >> 2) & 3);
    double z[2][RPL];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int col = col0 + cc;
        const int kb = (col < r) ? iwork[col] : 0;
        const double zn = (col < r) ? ((lds_f64*)dwork)[col] : 0.0;
#pragma unroll
        for (int t = 0; t < RPL; ++t) {
            const int row = RPL * rc + t;
            const double v = (row < kb) ? Dp[row * 64 + col] : ((row == kb) ? 1.0 : Dm.ld(row * 64 + col));
            z[cc][t] = v * zn;
        }
    }
    // the reflectors go through the LDS image in chunks of as many rows as it holds (all N - 2 at once in the 1024-thread build)
    constexpr int VROWS = TTN_LDS_IMG / 128;
    for (int khi = N - 2; khi > 0; khi -= VROWS) {
    const int klo = (khi > VROWS) ? khi - VROWS : 0;
    __syncthreads();
    wg_batched<8>((long long)(khi - klo) * 128, [&](long long e_) { return Vst[klo * 128 + e_]; }, [&](long long e_, double v) { L[e_] = v; });
    __syncthreads();
    if (wave < 8) {
        typedef double __attribute__((ext_vector_type(2))) d2;
        typedef __attribute__((address_space(3))) d2 lds_d2;
        for (int k = khi - 1; k >= klo; --k) {
            const lds_d2* vp = (const lds_d2*)(L + (k - klo) * 128 + RPL * rc);
            double vk[RPL];
#pragma unroll
            for (int t = 0; t < RPL / 2; ++t) { const d2 w2 = vp[t]; vk[2 * t] = w2.x; vk[2 * t + 1] = w2.y; }
            const double bk = beta[k];
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int t = 0; t < RPL; ++t) { s0 = fma(vk[t], z[0][t], s0); s1 = fma(vk[t], z[1][t], s1); }
            const double t0 = grp_sum(s0) * bk, t1 = grp_sum(s1) * bk;
#pragma unroll
            for (int t = 0; t < RPL; ++t) { z[0][t] = fma(-t0, vk[t], z[0][t]); z[1][t] = fma(-t1, vk[t], z[1][t]); }
        }
    }
    }
    __syncthreads();                                              // everyone has consumed D+ / D-: the image may be written
    EIG_MARK(6)
    if (wave < 8) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int col = col0 + cc;
            if (col < r) {
                const double sg = sig[col];                       // written above by this workgroup, barriers in between
#pragma unroll
                for (int t = 0; t < RPL; ++t) L[col * 128 + RPL * rc + t] = z[cc][t] * sg;
            }
        }
    }
    __syncthreads();
#ifdef TTN_EIG_PRIO
    __builtin_amdgcn_s_setprio(0); TTN_SETPRIO_BASE();
#endif
    return 0;
#undef EIG_MARK
}
__device__ int wg_eig128(const double* Gg, double* Vst, int r, int nev, double* sig, double* lds, int* iwork, double* dwork, long long* prof) {
    return wg_eig_n<128>(Gg, 128, Vst, r, nev, sig, lds, iwork, dwork, prof);
}
__device__ int wg_eig64(const double* Gg, int ldg, double* Vst, int r, int nev, double* sig, double* lds, int* iwork, double* dwork) {
    return wg_eig_n<64>(Gg, ldg, Vst, r, nev, sig, lds, iwork, dwork, nullptr);
}

__global__ void TTN_KERNEL_BOUNDS k_selftest_eig128(const double* G, double* Vst, int n, int r, int nev, double* sig, double* Xout, long long* clk) {
    extern __shared__ double lds[];
    int* iwork = reinterpret_cast<int*>(lds + GEMM_LDS_TOTAL + 32);
    double* dwork = lds + GEMM_LDS_TOTAL + 32 + 64;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const int rc = (n == 64) ? wg_eig_n<64>(G, 64, Vst, r, nev, sig, lds, iwork, dwork, clk) : wg_eig_n<128>(G, 128, Vst, r, nev, sig, lds, iwork, dwork, clk);
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = rc; }
    if (rc == 0) for (int e = threadIdx.x; e < 128 * r; e += TTN_WG) Xout[e] = lds[e];
}
