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
// LDS map (doubles) inside the 128*128 + 512 Jacobi/GEMM image `L`:
//   [0, 16384)            phase 1: vL, wL, xcol (3 x 128), part (8 x 128)   | phase 3: D+ then D- ([row][lane], 2 x 8192)
//                         phase 4: partial dot products (2 x 16 x 64), then the output image X (ld 128)
//   [16384, 16896)        dg, e, lam, beta (4 x 128)
#define EIG_TAIL (EIG_N * EIG_N)

// ---- 1. tridiagonalisation: dg[0..127], e[0..126] (e[k] couples k, k+1), reflectors v_k (rows of Vst, zeros up to k) and beta_k ----
__device__ __noinline__ void wg_tridiag128(const double* Gg, double* Vst, double* lds) {
    Gg = unip(Gg); Vst = unip(Vst); lds = unip(lds);
    lds_f64* L = (lds_f64*)lds;
    lds_f64 *vL = L, *wL = L + 128, *xcol = L + 256, *part = L + 384;
    lds_f64 *dg = L + EIG_TAIL, *e = dg + 128, *beta = dg + 384;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = tid & 127, c = tid >> 7;
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = Gg[i + 128 * (16 * c + j)];
    for (int k = 0; k < EIG_N - 2; ++k) {
        // row k of the current matrix (= column k by symmetry) from its 8 owners
        if (i == k) {
#pragma unroll
            for (int j = 0; j < 16; ++j) xcol[16 * c + j] = a[j];
        }
        __syncthreads();
        // the reflector: wave 0 only (lane handles entries lane, lane + 64) — the other waves wait at the barrier instead of
        // spending issue slots of the shared SIMDs on redundant copies of this scalar-ish code
        if (wave == 0) {
            const double x0 = (lane > k) ? xcol[lane] : 0.0, x1 = (lane + 64 > k) ? xcol[lane + 64] : 0.0;
            const double s2 = wave_sum(fma(x0, x0, x1 * x1));
            const double xk1 = xcol[k + 1];
            const double alpha = (s2 > 0.0) ? -copysign(sqrt(s2), xk1) : 0.0;
            const double den = s2 - alpha * xk1;                   // = v'v / 2
            const double bta = (den > 0.0) ? 1.0 / den : 0.0;
            const double v0 = x0 - ((lane == k + 1) ? alpha : 0.0), v1 = x1 - ((lane + 64 == k + 1) ? alpha : 0.0);
            vL[lane] = v0; vL[lane + 64] = v1;
            Vst[k * 128 + lane] = v0; Vst[k * 128 + lane + 64] = v1;
            if (lane == 0) { dg[k] = xcol[k]; e[k] = alpha; beta[k] = bta; }
        }
        __syncthreads();
        // p = A v (partial over the thread's 16 columns)
        double pp = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) pp = fma(a[j], vL[16 * c + j], pp);
        part[c * 128 + i] = pp;
        __syncthreads();
        if (wave == 0) {
            const double bta = beta[k], v0 = vL[lane], v1 = vL[lane + 64];
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) { p0 += part[cc * 128 + lane]; p1 += part[cc * 128 + lane + 64]; }
            p0 *= bta; p1 *= bta;
            const double Kc = 0.5 * bta * wave_sum(fma(p0, v0, p1 * v1));
            wL[lane] = fma(-Kc, v0, p0); wL[lane + 64] = fma(-Kc, v1, p1);
        }
        __syncthreads();
        // A -= v w' + w v'
        const double vi = vL[i], wi = wL[i];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = fma(-vi, wL[16 * c + j], fma(-wi, vL[16 * c + j], a[j]));
    }
    // the last 2 x 2 block
    if (i == EIG_N - 2 && c == 7) { dg[EIG_N - 2] = a[14]; e[EIG_N - 2] = a[15]; }
    if (i == EIG_N - 1 && c == 7) { dg[EIG_N - 1] = a[15]; }
    __syncthreads();
}

// number of eigenvalues of the tridiagonal (dg, e2 = e^2) below x
__device__ inline int sturm_count(const lds_f64* dg, const lds_f64* e2, double x, double pivmin) {
    double q = dg[0] - x;
    int cnt = (q < 0.0) ? 1 : 0;
    for (int i = 1; i < EIG_N; ++i) {
        if (fabs(q) < pivmin) q = -pivmin;
        q = fma(-e2[i - 1], fast_rcp(q), dg[i] - x);
        cnt += (q < 0.0) ? 1 : 0;
    }
    return cnt;
}

// ---- 2. eigenvalues lam[0..nev-1] in DESCENDING order (the nev largest), multisection with TL lanes per eigenvalue ----
template <int TL>
__device__ void wg_bisect128(int nev, double* lds) {
    lds_f64* L = (lds_f64*)lds;
    lds_f64 *dg = L + EIG_TAIL, *e = dg + 128, *lam = dg + 256;
    lds_f64* e2 = L;                                             // phase-1 work area is free
    const int tid = threadIdx.x;
    for (int t = tid; t < EIG_N; t += TTN_WG) e2[t] = (t < EIG_N - 1) ? e[t] * e[t] : 0.0;
    __syncthreads();
    // Gershgorin interval and the pivot floor: wave 0, results through LDS (e2[128..130])
    if (tid < 64) {
        double glo_ = 1e300, ghi_ = -1e300, emax_ = 0.0;
        for (int t = tid; t < EIG_N; t += 64) {
            const double rad = ((t > 0) ? fabs(e[t - 1]) : 0.0) + ((t < EIG_N - 1) ? fabs(e[t]) : 0.0);
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
    glo -= 2.0 * DBL_EPSILON * span * EIG_N; ghi += 2.0 * DBL_EPSILON * span * EIG_N;
    const double pivmin = DBL_MIN * fmax(1.0, emax);
    for (int g0 = 0; g0 < nev; g0 += TTN_WG / TL) {
        const int gi = g0 + tid / TL, sub = tid % TL;
        const bool act = gi < nev;
        const int jasc = EIG_N - 1 - (act ? gi : 0);            // ascending index of this group's eigenvalue
        double lo = glo, hi = ghi;
        // rounds: (TL + 1)-section; stop when the interval is at rounding level
        for (int round = 0; round < 64; ++round) {
            const double wdt = hi - lo;
            const bool done = !(wdt > 2.0 * DBL_EPSILON * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin);
            if (__syncthreads_and(done || !act)) break;
            const double x = lo + wdt * ((double)(sub + 1) / (double)(TL + 1));
            const int cnt = sturm_count(dg, e2, x, pivmin);
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

// ---- 3. eigenvectors of lam[0..r-1] by twisted factorisations; z (unnormalised) stays split over the D+ / D- arrays,
//         twist index and 1/||z|| per vector in tw[], zn[] ----
__device__ void wg_twisted128(int r, double* lds, int* tw /*LDS r ints*/, lds_f64* zn /*LDS r*/) {
    lds_f64* L = (lds_f64*)lds;
    lds_f64 *dg = L + EIG_TAIL, *e = dg + 128, *lam = dg + 256;
    lds_f64 *Dp = L, *Dm = L + 64 * EIG_N;                       // [row][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double emax = 0.0;
    for (int t = 0; t < EIG_N - 1; ++t) emax = fmax(emax, e[t] * e[t]);
    const double pivmin = DBL_MIN * fmax(1.0, emax);
    __syncthreads();                                             // e2 (aliasing Dp) is no longer read
    // one wave, one lane per vector (r <= 64)
    {
        const int j0 = 0;
        if (wave == 0) {
            const int j = j0 + lane;
            const bool act = j < r;
            const double lm = lam[act ? j : 0];
            double q = dg[0] - lm;
            if (fabs(q) < pivmin) q = -pivmin;
            Dp[lane] = q;
            for (int i = 1; i < EIG_N; ++i) {
                q = fma(-(e[i - 1] * e[i - 1]), fast_rcp(q), dg[i] - lm);
                if (fabs(q) < pivmin) q = -pivmin;
                Dp[i * 64 + lane] = q;
            }
            q = dg[EIG_N - 1] - lm;
            if (fabs(q) < pivmin) q = -pivmin;
            Dm[(EIG_N - 1) * 64 + lane] = q;
            double gbest = fabs(Dp[(EIG_N - 1) * 64 + lane] + q - (dg[EIG_N - 1] - lm));
            int kb = EIG_N - 1;
            for (int i = EIG_N - 2; i >= 0; --i) {
                q = fma(-(e[i] * e[i]), fast_rcp(q), dg[i] - lm);
                if (fabs(q) < pivmin) q = -pivmin;
                Dm[i * 64 + lane] = q;
                const double g = fabs(Dp[i * 64 + lane] + q - (dg[i] - lm));
                if (g < gbest) { gbest = g; kb = i; }
            }
            // z_k = 1; upwards with D+, downwards with D-; the entries replace the pivots they consumed
            double z = 1.0, nrm = 1.0;
            for (int i = kb - 1; i >= 0; --i) {
                z = -(e[i] * fast_rcp(Dp[i * 64 + lane])) * z;
                Dp[i * 64 + lane] = z;
                nrm = fma(z, z, nrm);
            }
            z = 1.0;
            for (int i = kb; i < EIG_N - 1; ++i) {
                z = -(e[i] * fast_rcp(Dm[(i + 1) * 64 + lane])) * z;
                Dm[(i + 1) * 64 + lane] = z;
                nrm = fma(z, z, nrm);
            }
            if (act) { tw[j] = kb; zn[j] = 1.0 / sqrt(nrm); }
        }
        __syncthreads();
    }
}

// ---- 4. back-transformation and the driver ----
// Eigen-decomposition of the symmetric positive definite G (global, column-major, ld 128): the `nev` largest eigenvalues
// (descending) -> sig[j] = sqrt(lam_j) (global), and the image X (LDS, ld 128): X[j*128 + row] = sqrt(lam_j) * u_j[row] for j < r
// (r <= 64, r <= nev).  Vst: 128 x 128 doubles of global scratch.  Returns 0, or 1 if a wanted eigenvalue is not positive.
__device__ __noinline__ int wg_eig128(const double* Gg, double* Vst, int r, int nev, double* sig, double* lds, int* iwork /*LDS 64 ints*/,
                                      double* dwork /*LDS 64*/, long long* prof) {
    Gg = unip(Gg); Vst = unip(Vst); sig = unip(sig); lds = unip(lds); iwork = unip(iwork); dwork = unip(dwork);
    r = uni32(r); nev = uni32(nev);
    lds_f64* L = (lds_f64*)lds;
    lds_f64 *lam = L + EIG_TAIL + 256, *beta = L + EIG_TAIL + 384;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define EIG_MARK(slot) if (prof && threadIdx.x == 0) prof[slot] = (long long)__builtin_amdgcn_s_memtime();
    EIG_MARK(2)
    wg_tridiag128(Gg, Vst, lds);
    EIG_MARK(3)
    wg_bisect128<8>(nev, lds);
    // eigenvalues out; all wanted ones must be positive
    int bad = 0;
    for (int j = tid; j < nev; j += TTN_WG) { const double l = lam[j]; sig[j] = (l > 0.0) ? sqrt(l) : 0.0; bad |= !(l > 0.0); }
    if (__syncthreads_or(bad)) return 1;
    EIG_MARK(4)
    wg_twisted128(r, lds, iwork, (lds_f64*)dwork);
    EIG_MARK(5)
    // Z into registers: thread = (column lane, rows 8*wave .. 8*wave+7)
    lds_f64 *Dp = L, *Dm = L + 64 * EIG_N;
    double z[8];
    {
        const int kb = (lane < r) ? iwork[lane] : 0;
        const double zn = (lane < r) ? ((lds_f64*)dwork)[lane] : 0.0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = 8 * wave + t;
            const double v = (row < kb) ? Dp[row * 64 + lane] : ((row == kb) ? 1.0 : Dm[row * 64 + lane]);
            z[t] = v * zn;
        }
    }
    __syncthreads();
    lds_f64* P = L;                                               // [2][16][64] partial dot products
    double vk[8], vn[8];
    {
        const double* vp = Vst + (EIG_N - 3) * 128 + 8 * wave;
#pragma unroll
        for (int t = 0; t < 8; ++t) vn[t] = vp[t];
    }
    for (int k = EIG_N - 3; k >= 0; --k) {
#pragma unroll
        for (int t = 0; t < 8; ++t) vk[t] = vn[t];
        if (k > 0) {                                              // prefetch the next reflector (global memory, L2 resident)
            const double* vp = Vst + (k - 1) * 128 + 8 * wave;
#pragma unroll
            for (int t = 0; t < 8; ++t) vn[t] = vp[t];
        }
        double sdot = 0.0;
#pragma unroll
        for (int t = 0; t < 8; ++t) sdot = fma(vk[t], z[t], sdot);
        P[(k & 1) * 1024 + wave * 64 + lane] = sdot;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += P[(k & 1) * 1024 + w * 64 + lane];
        tot *= beta[k];
#pragma unroll
        for (int t = 0; t < 8; ++t) z[t] = fma(-tot, vk[t], z[t]);
    }
    __syncthreads();
    EIG_MARK(6)
    if (lane < r) {
        const double sg = sig[lane];                              // written above by this workgroup, barriers in between
#pragma unroll
        for (int t = 0; t < 8; ++t) L[lane * 128 + 8 * wave + t] = z[t] * sg;
    }
    __syncthreads();
    return 0;
}

__global__ void __launch_bounds__(TTN_WG) k_selftest_eig128(const double* G, double* Vst, int r, int nev, double* sig, double* Xout, long long* clk) {
    extern __shared__ double lds[];
    int* iwork = reinterpret_cast<int*>(lds + GEMM_LDS_TOTAL + 32);
    double* dwork = lds + GEMM_LDS_TOTAL + 32 + 64;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    const int rc = wg_eig128(G, Vst, r, nev, sig, lds, iwork, dwork, clk);
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = rc; }
    if (rc == 0) for (int e = threadIdx.x; e < 128 * r; e += TTN_WG) Xout[e] = lds[e];
}
