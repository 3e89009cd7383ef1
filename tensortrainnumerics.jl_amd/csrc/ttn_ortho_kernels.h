// ttn_ortho_kernels.h — orthogonalize(x; i) (src/tt_tools.jl:511-543): left QR sweep, right LQ
// sweep, centre core = FR * X_i * FL.  One workgroup per train; Householder QR with explicit thin Q.
#pragma once
#include "ttn_common.h"
#include "ttn_dense_kernels.h"

#define ORTHO_LDS_BYTES ((GEMM_LDS_TOTAL + 64) * sizeof(double))

struct OrthoArgs {
    TTDev x, y;
    int center;                 // 0-based
    double* scratch;
    long long scratch_stride;
    int mmax, rmax;
};

// Thin Householder QR of the column-major mm x nn matrix Tm (ld = mm), in place.
// Outputs: Qb (mm x rnew, ld = mm) explicit, Rb (rnew x nn, ld = rnew) with zeros below the diagonal,
// rnew = min(mm, nn)  (the rank the reference reads from size(Matrix(F.Q), 2), src/tt_tools.jl:522).
__device__ int wg_qr_explicit(int mm, int nn, double* Tm, double* Qb, double* Rb, double* taus, double* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = TTN_WG >> 6;
    const int rnew = min(mm, nn);
    for (int j = 0; j < rnew; ++j) {
        double* col = Tm + (long long)j * mm;
        double s = 0.0;
        for (int i = j + 1 + tid; i < mm; i += TTN_WG) { const double v = col[i]; s = fma(v, v, s); }
        const double xnorm2 = wg_sum(s, red);
        const double alpha = col[j];
        double tau = 0.0, scal = 0.0, beta = alpha;
        if (xnorm2 > 0.0) {
            beta = -copysign(sqrt(fma(alpha, alpha, xnorm2)), alpha);
            tau = (beta - alpha) / beta;
            scal = 1.0 / (alpha - beta);
        }
        __syncthreads();
        if (xnorm2 > 0.0) {
            for (int i = j + 1 + tid; i < mm; i += TTN_WG) col[i] *= scal;
            if (tid == 0) col[j] = beta;
        }
        if (tid == 0) taus[j] = tau;
        __syncthreads();
        if (tau != 0.0) {
            for (int c = j + 1 + wave; c < nn; c += nwaves) {
                double* cc = Tm + (long long)c * mm;
                double w = 0.0;
                for (int i = j + 1 + lane; i < mm; i += 64) w = fma(cc[i], col[i], w);
                w = wave_sum(w) + cc[j];
                const double tw = tau * w;
                for (int i = j + 1 + lane; i < mm; i += 64) cc[i] = fma(-tw, col[i], cc[i]);
                if (lane == 0) cc[j] -= tw;
            }
        }
        __syncthreads();
    }
    // R = triu(Tm)[0:rnew, :]
    for (int e = tid; e < rnew * nn; e += TTN_WG) {
        const int i = e % rnew, c = e / rnew;
        Rb[e] = (i <= c) ? Tm[(long long)c * mm + i] : 0.0;
    }
    // Q = H_0 H_1 ... H_{rnew-1} * I[:, 0:rnew]  (backward accumulation)
    for (long long e = tid; e < (long long)mm * rnew; e += TTN_WG) {
        const int i = (int)(e % mm), c = (int)(e / mm);
        Qb[e] = (i == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int j = rnew - 1; j >= 0; --j) {
        const double tau = taus[j];
        if (tau != 0.0) {
            const double* v = Tm + (long long)j * mm;
            for (int c = j + wave; c < rnew; c += nwaves) {
                double* qc = Qb + (long long)c * mm;
                double w = 0.0;
                for (int i = j + 1 + lane; i < mm; i += 64) w = fma(qc[i], v[i], w);
                w = wave_sum(w) + qc[j];
                const double tw = tau * w;
                for (int i = j + 1 + lane; i < mm; i += 64) qc[i] = fma(-tw, v[i], qc[i]);
                if (lane == 0) qc[j] -= tw;
            }
        }
        __syncthreads();
    }
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

__global__ void __launch_bounds__(TTN_WG) k_orthogonalize(OrthoArgs P) {
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
    double* taus = Rb1 + (long long)P.rmax * P.rmax;
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
        const int yl = (int)yr[j], rl = (int)xr[j], rr = (int)xr[j + 1];
        double* Xj = X.data + (long long)b * X.stride + X.off[j];
        double* Yj = Y.data + (long long)b * Y.stride + Y.off[j];
        const int mm = yl * n;
        // Tm[(al + yl*s), be] = sum_ga FR[al,ga] X_j[s,ga,be]
        const View Xv = mkview(Xj, plain(n), Idx{n, 1, (long long)n * rl});            // [ga, (s + n*be)]
        const View Tv = mkview(Tm, plain(1), Idx{n, (long long)yl, (long long)mm});   // [al, (s + n*be)]
        wg_gemm(yl, n * rr, rl, FR, Xv, Tv, 1.0, 0.0, lds);
        double* Rn = which ? Rb0 : Rb1;
        const int rnew = wg_qr_explicit(mm, rr, Tm, Qb, Rn, taus, red);
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
    double* Rc = taus + P.rmax;                                              // third R buffer (rmax x rmax)
    double* Rd = Rc + (long long)P.rmax * P.rmax;                            // fourth
    if (tid == 0) { Rc[0] = 1.0; }
    __syncthreads();
    View FL = mkview(Rc, plain(1), plain(1));             // (rr x yr_{j+1})
    int whichL = 0;
    for (int j = d - 1; j > ic; --j) {
        const int n = X.dims[j];
        const int ynext = (int)yr[j + 1], rl = (int)xr[j], rr = (int)xr[j + 1];
        double* Xj = X.data + (long long)b * X.stride + X.off[j];
        double* Yj = Y.data + (long long)b * Y.stride + Y.off[j];
        const int mm = ynext * n;
        // Tt[(be + ynext*s), al] = sum_ga FL[ga,be] X_j[s,al,ga]  ==  (FL^T) * X2,  X2[ga, (s + n*al)]
        const View X2 = mkview(Xj, plain((long long)n * rl), plain(1));
        const View Tv = mkview(Tm, plain(1), Idx{n, (long long)ynext, (long long)mm});
        wg_gemm(ynext, n * rl, rr, tview(FL), X2, Tv, 1.0, 0.0, lds);
        double* Rn = whichL ? Rc : Rd;
        const int rnew = wg_qr_explicit(mm, rl, Tm, Qb, Rn, taus, red);
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
        const int yl = (int)yr[ic], yn = (int)yr[ic + 1], rl = (int)xr[ic], rr = (int)xr[ic + 1];
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
