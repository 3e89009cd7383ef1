// ttn_ortho_kernels.h — orthogonalize(x; i) (src/tt_tools.jl:511-543): left QR sweep, right LQ
// sweep, centre core = FR * X_i * FL.  One workgroup per train; Householder QR with explicit thin Q.
#pragma once
#include "ttn_common.h"
#include "ttn_dense_kernels.h"

#define ORTHO_LDS_BYTES ((GEMM_LDS_TOTAL + 32 + 2 * QR_NB * QR_NB + 2 * QR_NB) * sizeof(double))

struct OrthoArgs {
    TTDev x, y;
    int center;                 // 0-based
    double* scratch;
    long long scratch_stride;
    int mmax, rmax;
};

// Thin QR  T = Q R  of the column-major mm x nn matrix Tm (ld = mm) through the blocked Householder LQ of its
// transpose (the same storage read as a row-major nn x mm matrix).  Outputs: Qb (mm x rnew column-major, i.e. the
// rnew x mm row-major Q' of the LQ), Rb (rnew x nn, ld = rnew) with zeros below the diagonal; rnew = min(mm, nn)
// (the rank the reference reads from size(Matrix(F.Q), 2), src/tt_tools.jl:522).
struct OrthoWork { double *Vb, *Wb, *Tst, *Ts, *Ss, *taus, *red; };
__device__ int wg_qr_explicit(int mm, int nn, double* Tm, double* Qb, double* Rb, const OrthoWork& W, double* lds) {
    const int tid = threadIdx.x;
    const int rnew = min(mm, nn);
    wg_lq_blocked(nn, mm, Tm, mm, W.Vb, W.Wb, W.Tst, Qb, lds, W.Ts, W.Ss, W.taus, W.red);
    for (int e = tid; e < rnew * nn; e += TTN_WG) {
        const int i = e % rnew, c = e / rnew;
        Rb[e] = (i <= c) ? Tm[(long long)c * mm + i] : 0.0;
    }
    __syncthreads();
    return rnew;
}

// device restatement of r_and_d_to_rks (src/tt_tools.jl:407-425) with Julia's wrapping Int64 products
__device__ void dev_r_and_d_to_rks(int d, const int* dims, const long long* rks, long long rmax, long long* out) {
    for (int i = 0; i <= d; ++i) out[i] = 1;
    for (int i = 0; i < d; ++i) {
        unsigned long long q = 1, p = 1;
        for (int t = i; t < d; ++t) q *= (unsigned long long)dims[t];
        for (int t = 0; t < i; ++t) p *= (unsigned long long)dims[t];
        const long long qs = (long long)q, ps = (long long)p;
        long long v = rks[i];
        if (qs > 0) {
            if (ps > 0) { v = min(v, ps); v = min(v, qs); v = min(v, rmax); }
            else { v = min(v, qs); v = min(v, rmax); }
        } else {
            if (ps > 0) { v = min(v, ps); v = min(v, rmax); }
            else v = min(v, rmax);
        }
        out[i] = v;
    }
}

__global__ void TTN_KERNEL_BOUNDS k_orthogonalize(OrthoArgs P) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const TTDev& X = P.x; const TTDev& Y = P.y;
    const int d = X.d;
    const long long* xr = X.rks + (long long)b * (d + 1);
    long long* yr = Y.rks + (long long)b * (d + 1);
    double* red = lds + GEMM_LDS_TOTAL;
    double* scr = P.scratch + (long long)b * P.scratch_stride;
    double* Tm = scr;
    double* Qb = Tm + (long long)P.mmax * P.rmax;
    double* Rb0 = Qb + (long long)P.mmax * P.rmax;
    double* Rb1 = Rb0 + (long long)P.rmax * P.rmax;
    double* Rc = Rb1 + (long long)P.rmax * P.rmax;                           // R factors of the right sweep ping-pong in Rc/Rd
    double* Rd = Rc + (long long)P.rmax * P.rmax;
    OrthoWork W;
    W.Vb = Rd + (long long)P.rmax * P.rmax;                                  // QR_NB x mmax
    W.Wb = W.Vb + (long long)QR_NB * P.mmax;                                 // mmax x QR_NB (>= max(nn, rnew) rows)
    W.Tst = W.Wb + (long long)QR_NB * P.mmax;                                // ceil(rmax/QR_NB) * QR_NB^2
    W.red = red;
    W.Ts = red + 32;
    W.Ss = W.Ts + QR_NB * QR_NB;
    W.taus = W.Ss + QR_NB * QR_NB;
    if (tid == 0) dev_r_and_d_to_rks(d, X.dims, xr, 1024, yr);
    __syncthreads();
    const int ic = P.center;

    // ---- left sweep: sites 0..ic-1 (src/tt_tools.jl:518-525) ----
    if (tid == 0) { Rb0[0] = 1.0; }
    __syncthreads();
    View FR = mkview(Rb0, plain(1), plain(1));            // (yr_j x rl)
    int which = 0;
    for (int j = 0; j < ic; ++j) {
        const int n = X.dims[j];
        const int yl = uni32((int)yr[j]), rl = uni32((int)xr[j]), rr = uni32((int)xr[j + 1]);
        double* Xj = X.data + (long long)b * X.stride + X.off[j];
        double* Yj = Y.data + (long long)b * Y.stride + Y.off[j];
        const int mm = yl * n;
        // Tm[(al + yl*s), be] = sum_ga FR[al,ga] X_j[s,ga,be]
        const View Xv = mkview(Xj, plain(n), Idx{n, 1, (long long)n * rl});            // [ga, (s + n*be)]
        const View Tv = mkview(Tm, plain(1), Idx{n, (long long)yl, (long long)mm});   // [al, (s + n*be)]
        wg_gemm(yl, n * rr, rl, FR, Xv, Tv, 1.0, 0.0, lds);
        double* Rn = which ? Rb0 : Rb1;
        const int rnew = wg_qr_explicit(mm, rr, Tm, Qb, Rn, W, lds);
        // Y_j[s, al, be] = Q[al + yl*s, be]
        for (long long e = tid; e < (long long)mm * rnew; e += TTN_WG) {
            const int row = (int)(e % mm), be = (int)(e / mm);
            const int al = row % yl, s = row / yl;
            Yj[s + (long long)n * (al + (long long)yl * be)] = Qb[e];
        }
        if (tid == 0) yr[j + 1] = rnew;
        __syncthreads();
        FR = mkview(Rn, plain(1), plain(rnew));           // (rnew x rr)
        which ^= 1;
    }
    // ---- right sweep: sites d-1..ic+1 (src/tt_tools.jl:528-536); its R factors ping-pong in Rc/Rd ----
    if (tid == 0) { Rc[0] = 1.0; }
    __syncthreads();
    View FL = mkview(Rc, plain(1), plain(1));             // (rr x yr_{j+1})
    int whichL = 0;
    for (int j = d - 1; j > ic; --j) {
        const int n = X.dims[j];
        const int ynext = uni32((int)yr[j + 1]), rl = uni32((int)xr[j]), rr = uni32((int)xr[j + 1]);
        double* Xj = X.data + (long long)b * X.stride + X.off[j];
        double* Yj = Y.data + (long long)b * Y.stride + Y.off[j];
        const int mm = ynext * n;
        // Tt[(be + ynext*s), al] = sum_ga FL[ga,be] X_j[s,al,ga]  ==  (FL^T) * X2,  X2[ga, (s + n*al)]
        const View X2 = mkview(Xj, plain((long long)n * rl), plain(1));
        const View Tv = mkview(Tm, plain(1), Idx{n, (long long)ynext, (long long)mm});
        wg_gemm(ynext, n * rl, rr, tview(FL), X2, Tv, 1.0, 0.0, lds);
        double* Rn = whichL ? Rc : Rd;
        const int rnew = wg_qr_explicit(mm, rl, Tm, Qb, Rn, W, lds);
        // Y_j[s, al, be] = Qt[(be + ynext*s), al]   (core shape (n, rnew, ynext))
        for (long long e = tid; e < (long long)mm * rnew; e += TTN_WG) {
            const int row = (int)(e % mm), al = (int)(e / mm);
            const int be = row % ynext, s = row / ynext;
            Yj[s + (long long)n * (al + (long long)rnew * be)] = Qb[e];
        }
        if (tid == 0) yr[j] = rnew;
        __syncthreads();
        FL = tview(mkview(Rn, plain(1), plain(rnew)));    // FL[ga, be] = Rt[be, ga]  (rl x rnew)
        whichL ^= 1;
    }
    // ---- centre core: Y_i[s] = FR * X_i[s] * FL  (src/tt_tools.jl:537-541) ----
    {
        const int n = X.dims[ic];
        const int yl = uni32((int)yr[ic]), yn = uni32((int)yr[ic + 1]), rl = uni32((int)xr[ic]), rr = uni32((int)xr[ic + 1]);
        double* Xi = X.data + (long long)b * X.stride + X.off[ic];
        double* Yi = Y.data + (long long)b * Y.stride + Y.off[ic];
        const int mm = yl * n;
        const View Xv = mkview(Xi, plain(n), Idx{n, 1, (long long)n * rl});
        const View Tv = mkview(Tm, plain(1), Idx{n, (long long)yl, (long long)mm});
        wg_gemm(yl, n * rr, rl, FR, Xv, Tv, 1.0, 0.0, lds);
        const View Tmv = mkview(Tm, plain(1), plain(mm));                               // (mm x rr)
        const View Yv = mkview(Yi, Idx{yl, (long long)n, 1}, plain((long long)n * yl)); // [(al + yl*s), be']
        wg_gemm(mm, yn, rr, Tmv, FL, Yv, 1.0, 0.0, lds);
    }
}
