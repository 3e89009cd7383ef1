// ttn_densefact_kernels.h — thin QR and thin SVD of ONE small dense matrix, Float64 or ComplexF64, on device pointers.
//
// These are the dense moves of the TDVP sweeps (src/solvers/tdvp.jl:76-80, :120-126: `qr(Aqr)`, `qr(A')`; :252, :276: `_svdtrunc`) for
// local tensors that are complex in a real-time sweep — the QR / SVD machinery of k_compress / k_orthogonalize is real-valued and
// works on TT handles.  The matrices are local TDVP tensors (bond x 2 x bond): correctness first, one workgroup, global memory, no
// MFMA.  Complex numbers are interleaved (re, im) pairs, matrices column-major (Julia's layout).
//   k_dense_qr   Householder QR as LAPACK's geqr2 + org2r do it (zlarfg's beta real, H = I - tau v v^H, H^H applied from the left):
//                A (m x n, overwritten) -> Q (m x r), R (r x n), r = min(m, n).
//   k_dense_svd  one-sided Jacobi (Hestenes) on the columns of A (m x n, m >= n; the caller passes A^H otherwise): plane rotations
//                with a complex phase until every pair of columns is orthogonal to 8 eps sqrt(m) of the product of their norms, V accumulated,
//                singular values = column norms, sorted descending; U (m x n), s (n), Vh (n x n).  One wave per column pair,
//                round-robin tournament ordering: n - 1 rounds of n / 2 disjoint pairs per sweep, a barrier per round.
#pragma once
#include "ttn_common.h"

#define TTN_DF_WG 1024

template <bool CPLX> struct dfnum;
template <> struct dfnum<false> {
    typedef double T;
    static __device__ __forceinline__ T load(const double* p, long long i) { return p[i]; }
    static __device__ __forceinline__ void store(double* p, long long i, T v) { p[i] = v; }
    static __device__ __forceinline__ T zero() { return 0.0; }
    static __device__ __forceinline__ T one() { return 1.0; }
    static __device__ __forceinline__ T conj(T a) { return a; }
    static __device__ __forceinline__ T mul(T a, T b) { return a * b; }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
    static __device__ __forceinline__ T sub(T a, T b) { return a - b; }
    static __device__ __forceinline__ T scale(T a, double s_) { return a * s_; }
    static __device__ __forceinline__ double abs2(T a) { return a * a; }
    static __device__ __forceinline__ double re(T a) { return a; }
    static __device__ __forceinline__ double im(T) { return 0.0; }
    static __device__ __forceinline__ T make(double r, double) { return r; }
    static __device__ __forceinline__ T wsum(T a) { return wave_sum(a); }
};
struct dfc { double x, y; };
template <> struct dfnum<true> {
    typedef dfc T;
    static __device__ __forceinline__ T load(const double* p, long long i) { return T{p[2 * i], p[2 * i + 1]}; }
    static __device__ __forceinline__ void store(double* p, long long i, T v) { p[2 * i] = v.x; p[2 * i + 1] = v.y; }
    static __device__ __forceinline__ T zero() { return T{0.0, 0.0}; }
    static __device__ __forceinline__ T one() { return T{1.0, 0.0}; }
    static __device__ __forceinline__ T conj(T a) { return T{a.x, -a.y}; }
    static __device__ __forceinline__ T mul(T a, T b) { return T{fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x)}; }
    static __device__ __forceinline__ T add(T a, T b) { return T{a.x + b.x, a.y + b.y}; }
    static __device__ __forceinline__ T sub(T a, T b) { return T{a.x - b.x, a.y - b.y}; }
    static __device__ __forceinline__ T scale(T a, double s_) { return T{a.x * s_, a.y * s_}; }
    static __device__ __forceinline__ double abs2(T a) { return fma(a.x, a.x, a.y * a.y); }
    static __device__ __forceinline__ double re(T a) { return a.x; }
    static __device__ __forceinline__ double im(T a) { return a.y; }
    static __device__ __forceinline__ T make(double r, double i) { return T{r, i}; }
    static __device__ __forceinline__ T wsum(T a) { return T{wave_sum(a.x), wave_sum(a.y)}; }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// QR.  work: r numbers of the matrix type (the taus).  One thread per trailing column inside a step (the matrices are small).
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool CPLX>
__global__ void __launch_bounds__(TTN_DF_WG) k_dense_qr(int m, int n, double* A, double* Q, double* R, double* taus) {
    typedef dfnum<CPLX> N;
    typedef typename N::T T;
    __shared__ double red[40];
    __shared__ double sh[8];
    const int tid = threadIdx.x, r = m < n ? m : n;
    for (int k = 0; k < r; ++k) {
        // ---- the reflector of column k (zlarfg / dlarfg): beta real, v(k) = 1 implied, v(k+1:) in place ----
        double part = 0.0;
        for (int i = k + 1 + tid; i < m; i += TTN_DF_WG) part += N::abs2(N::load(A, i + (long long)m * k));
        const double xnorm2 = wg_sum(part, red);
        const T alpha = N::load(A, k + (long long)m * k);
        if (tid == 0) {
            T tau = N::zero();
            double beta = N::re(alpha);
            T scal = N::zero();
            if (xnorm2 != 0.0 || N::im(alpha) != 0.0) {
                beta = -copysign(sqrt(N::abs2(alpha) + xnorm2), N::re(alpha));
                tau = N::make((beta - N::re(alpha)) / beta, -N::im(alpha) / beta);
                const T den = N::make(N::re(alpha) - beta, N::im(alpha));                 // 1 / (alpha - beta)
                const double d2 = N::abs2(den);
                scal = N::make(N::re(den) / d2, -N::im(den) / d2);
            }
            N::store(taus, k, tau);
            sh[0] = N::re(scal); sh[1] = N::im(scal); sh[2] = beta; sh[3] = N::re(tau); sh[4] = N::im(tau);
        }
        __syncthreads();
        const T scal = N::make(sh[0], sh[1]), tau = N::make(sh[3], sh[4]);
        const double beta = sh[2];
        const bool trivial = (sh[3] == 0.0 && sh[4] == 0.0);
        if (!trivial)
            for (int i = k + 1 + tid; i < m; i += TTN_DF_WG) N::store(A, i + (long long)m * k, N::mul(N::load(A, i + (long long)m * k), scal));
        __syncthreads();
        // ---- H(k)^H = I - conj(tau) v v^H on the trailing columns ----
        if (!trivial) {
            for (int j = k + 1 + tid; j < n; j += TTN_DF_WG) {
                T w = N::load(A, k + (long long)m * j);                                       // v(k) = 1
                for (int i = k + 1; i < m; ++i) w = N::add(w, N::mul(N::conj(N::load(A, i + (long long)m * k)), N::load(A, i + (long long)m * j)));
                const T tw = N::mul(N::conj(tau), w);
                N::store(A, k + (long long)m * j, N::sub(N::load(A, k + (long long)m * j), tw));
                for (int i = k + 1; i < m; ++i)
                    N::store(A, i + (long long)m * j, N::sub(N::load(A, i + (long long)m * j), N::mul(tw, N::load(A, i + (long long)m * k))));
            }
        }
        if (tid == 0) N::store(A, k + (long long)m * k, N::make(beta, 0.0));
        __syncthreads();
    }
    // ---- R = the upper trapezoid ----
    for (long long e = tid; e < (long long)r * n; e += TTN_DF_WG) {
        const int i = (int)(e % r), j = (int)(e / r);
        N::store(R, e, (i <= j) ? N::load(A, i + (long long)m * j) : N::zero());
    }
    // ---- Q = H(0) H(1) ... H(r-1) applied to the first r columns of the identity (org2r: backwards) ----
    for (long long e = tid; e < (long long)m * r; e += TTN_DF_WG) {
        const int i = (int)(e % m), j = (int)(e / m);
        N::store(Q, e, (i == j) ? N::one() : N::zero());
    }
    __syncthreads();
    for (int k = r - 1; k >= 0; --k) {
        const T tau = N::load(taus, k);
        if (N::re(tau) != 0.0 || N::im(tau) != 0.0) {
            for (int j = k + tid; j < r; j += TTN_DF_WG) {
                T w = N::load(Q, k + (long long)m * j);
                for (int i = k + 1; i < m; ++i) w = N::add(w, N::mul(N::conj(N::load(A, i + (long long)m * k)), N::load(Q, i + (long long)m * j)));
                const T tw = N::mul(tau, w);
                N::store(Q, k + (long long)m * j, N::sub(N::load(Q, k + (long long)m * j), tw));
                for (int i = k + 1; i < m; ++i)
                    N::store(Q, i + (long long)m * j, N::sub(N::load(Q, i + (long long)m * j), N::mul(tw, N::load(A, i + (long long)m * k))));
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// SVD by one-sided Jacobi.  m >= n.  G = A in place; V (n x n) accumulated in Vw; flags / norms / permutation in iw / dw (global).
// ---------------------------------------------------------------------------------------------------------------------------------
template <bool CPLX>
__global__ void __launch_bounds__(TTN_DF_WG) k_dense_svd(int m, int n, double* A, double* U, double* sv, double* Vh, double* Vw, double* dw, int* iw,
                                                         int max_sweeps) {
    typedef dfnum<CPLX> N;
    typedef typename N::T T;
    __shared__ int rotated;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = TTN_DF_WG / 64;
    for (long long e = tid; e < (long long)n * n; e += TTN_DF_WG) N::store(Vw, e, (e % n == e / n) ? N::one() : N::zero());
    __syncthreads();
    const int np = n + (n & 1);                                         // players of the round-robin tournament (one bye when n is odd)
    const double tol_rot = 2.0 * DBL_EPSILON * sqrt((double)m), tol_conv = 8.0 * DBL_EPSILON * sqrt((double)m);
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        if (tid == 0) rotated = 0;
        __syncthreads();
        for (int round = 0; round < np - 1; ++round) {
            for (int pr = wave; pr < np / 2; pr += nwaves) {
                // the circle method: player np - 1 is fixed, the others rotate
                int a = (pr == 0) ? np - 1 : (round + pr) % (np - 1);
                int b = (round + np - 1 - pr) % (np - 1);
                int p = a < b ? a : b, q = a < b ? b : a;
                if (q >= n) continue;                                   // the bye
                double al = 0.0, be = 0.0;
                T ga = N::zero();
                for (int i = lane; i < m; i += 64) {
                    const T gp = N::load(A, i + (long long)m * p), gq = N::load(A, i + (long long)m * q);
                    al += N::abs2(gp); be += N::abs2(gq);
                    ga = N::add(ga, N::mul(N::conj(gp), gq));
                }
                al = wave_sum(al); be = wave_sum(be); ga = N::wsum(ga);
                const double ag = sqrt(N::abs2(ga));
                const double nn = sqrt(al * be);
                // rotate above 2 eps sqrt(m) of the product of the norms; only pairs above 8 eps sqrt(m) keep the sweeps going (a pair at
                // the rounding level of its own inner product would be rotated for ever: 1e-15 flat did not terminate on 16 x 16 blocks)
                if (ag > tol_rot * nn && ag > 1.0e-290) {                 // (below: 1 / ag overflows; such a pair is two null columns)
                    if (lane == 0 && ag > tol_conv * nn) rotated = 1;
                    // columns [p q] <- [p q] J,  J = [[c, s ph], [-s conj(ph), c]],  ph = ga / |ga|: zeroes the (p, q) entry of G^H G
                    const T ph = N::scale(ga, 1.0 / ag);
                    const double zeta = (be - al) / (2.0 * ag);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + hypot(1.0, zeta));       // (hypot: zeta^2 overflows for a column 1e-160 of its partner — t = 0 then, and the pair was never orthogonalised)
                    const double c = 1.0 / sqrt(1.0 + t * t), s_ = c * t;
                    const T sph = N::scale(ph, s_), sphc = N::conj(sph);
                    for (int i = lane; i < m; i += 64) {
                        const T gp = N::load(A, i + (long long)m * p), gq = N::load(A, i + (long long)m * q);
                        N::store(A, i + (long long)m * p, N::sub(N::scale(gp, c), N::mul(sphc, gq)));
                        N::store(A, i + (long long)m * q, N::add(N::mul(sph, gp), N::scale(gq, c)));
                    }
                    for (int i = lane; i < n; i += 64) {
                        const T vp = N::load(Vw, i + (long long)n * p), vq = N::load(Vw, i + (long long)n * q);
                        N::store(Vw, i + (long long)n * p, N::sub(N::scale(vp, c), N::mul(sphc, vq)));
                        N::store(Vw, i + (long long)n * q, N::add(N::mul(sph, vp), N::scale(vq, c)));
                    }
                }
            }
            __syncthreads();
        }
        if (!rotated) break;
        __syncthreads();
    }
    // ---- singular values = column norms, sorted descending by counting ----
    for (int j = wave; j < n; j += nwaves) {
        double a = 0.0;
        for (int i = lane; i < m; i += 64) a += N::abs2(N::load(A, i + (long long)m * j));
        a = wave_sum(a);
        if (lane == 0) dw[j] = sqrt(a);
    }
    __syncthreads();
    for (int j = tid; j < n; j += TTN_DF_WG) {
        int rank = 0;
        const double v = dw[j];
        for (int k = 0; k < n; ++k) { const double w = dw[k]; rank += (w > v || (w == v && k < j)) ? 1 : 0; }
        iw[rank] = j;
    }
    if (tid == 0) iw[n] = (sweep < max_sweeps) ? 0 : 1;                  // 1: not converged within max_sweeps
    __syncthreads();
    for (int k = tid; k < n; k += TTN_DF_WG) sv[k] = dw[iw[k]];
    for (long long e = tid; e < (long long)m * n; e += TTN_DF_WG) {
        const int i = (int)(e % m), k = (int)(e / m), j = iw[k];
        const double s_ = dw[j];
        N::store(U, e, s_ > 0.0 ? N::scale(N::load(A, i + (long long)m * j), 1.0 / s_) : N::zero());
    }
    for (long long e = tid; e < (long long)n * n; e += TTN_DF_WG) {
        const int k = (int)(e % n), i = (int)(e / n);                    // Vh[k][i] = conj(V[i][perm k])
        N::store(Vh, e, N::conj(N::load(Vw, i + (long long)n * iw[k])));
    }
}
