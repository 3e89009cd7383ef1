// ttn_ortho_kernels.h — orthogonalize(x; i) (src/tt_tools.jl:511-543): left QR sweep, right LQ
// sweep, centre core = FR * X_i * FL.  One workgroup per train; Householder QR with explicit thin Q.
#pragma once
#include "ttn_common.h"
#include "ttn_dense_kernels.h"
#include "ttn_ortho_fused.h"

#define ORTHO_LDS_BYTES ((GEMM_LDS_TOTAL + 32 + 2 * QR_NB * QR_NB + 2 * QR_NB) * sizeof(double))

struct OrthoArgs {
    TTDev x, y;
    int center;                 // 0-based
    double* scratch;
    long long scratch_stride;
    int mmax, rmax;
    // Three-launch form for large batches of rank <= 64 QTT trains (ttn_api.hip; csrc/ttn_ortho512.h): mode 1 = the left sweep and the
    // right sweep up to the first site the 512-thread kernel takes, state saved; mode 3 = resume the right sweep from the saved state
    // with every route, then the centre core; mode 0 = everything in one launch.  state: int [batch][4] = {next right-sweep site,
    // buffer of the last right R (0: Rc, 1: Rd), buffer of the last left R (0: Rb0, 1: Rb1), train finished by k_ortho512}; behind it one
    // counter and the list of unfinished trains (int [1 + batch]).
    int mode;
    int* state;
    int ramp;                   // mode 1 only: stop the right sweep at its first site (k_ortho_ramp, csrc/ttn_ortho_ramp.h, takes it from there)
    const int* trains;          // mode 3 only: the trains k_ortho512 did not finish (workgroup w takes train trains[w]); nullptr: b = blockIdx.x
    int no_cholqr;              // bit 0: no Cholesky-QR steps on the general route, bit 1: no fused steps (TTN_ORTHO_CHOLQR = 0 sets both,
                                // 1 only bit 1: diagnostics, parity tests of every route)
    long long* prof;            // TTN_PROF=1: s_memtime stamp after every QR / LQ step of train b at prof[16 * batch + 120 * b + step] (ttn_prof_steps)
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

// The same factorisation by Cholesky-QR for tall, well-conditioned T with nn <= 64 columns (the interior sites of a rank-64 train:
// cond(T) ~ 10 measured on the benchmark's random trains — the rank-ramp sites at both ends of a chain are square and reach 1e8, they
// stay on the Householder route).  In the row-major view M = T^T (nn x mm, the storage of Tm read row by row):
//   G = M M^T (MFMA, wg_syrk) -> L1 = chol(G) in LDS -> Q1 = L1^-1 M (forward substitution, one thread per column: backward stable,
//   M = L1 Q1 + E with |E| <= eps |L1| |Q1|) -> C = Q1 Q1^T (MFMA): dev = max |C - I| ~ eps cond(T)^2 MEASURES the orthogonality.
//   dev <= ORTHO_CHOLQR_ACCEPT: Q1 is the orthonormal factor (1e-12 is the parity bar of orthogonalize; 2.7e-15 at cond 10);
//   dev <= ORTHO_CHOLQR_SECOND: second pass L2 = chol(C), Q = L2^-1 Q1, L = L1 L2 (CholeskyQR2, as the rank-ramp steps of k_compress);
//   otherwise, or when a pivot fails: return 0 with Tm untouched — the caller runs the Householder route.
// Against the in-LDS Householder LQ + explicit Q of round 2 (64 reflectors, each applied twice, one barrier per reflector and
// pass: 446 k clk per 64 x 128 site) this is two Gram products, one 64 x 64 factorisation and one triangular solve.
// Outputs as wg_qr_explicit: Qb (mm x nn column-major), Rb (nn x nn, ld = nn, zeros below the diagonal) — R has a positive diagonal.
#define ORTHO_CHOLQR_ACCEPT 2.0e-13
#define ORTHO_CHOLQR_SECOND 1.0e-6
#define ORTHO_CHOLQR_PIVOT_MAX 1.0e8         // pivot ratio (a lower bound of cond^2) above which the attempt stops after the factorisation
struct CholQrWork { double *Ga, *Cc, *L1g; double* scal; int* iflag; double* red; long long* stamps; };
#define OQ_STAMP(i) if (Cw.stamps && threadIdx.x == 0) Cw.stamps[i] += (long long)__builtin_amdgcn_s_memtime() - t_prev_; if (Cw.stamps) t_prev_ = (long long)__builtin_amdgcn_s_memtime();
__device__ __noinline__ int wg_qr_cholqr(int mm, int nn, double* Tm, double* Qb, double* Rb, CholQrWork Cw, double* lds) {
    mm = uni32(mm); nn = uni32(nn); Tm = unip(Tm); Qb = unip(Qb); Rb = unip(Rb); lds = unip(lds);
    Cw.Ga = unip(Cw.Ga); Cw.Cc = unip(Cw.Cc); Cw.L1g = unip(Cw.L1g); Cw.scal = unip(Cw.scal); Cw.iflag = unip(Cw.iflag); Cw.red = unip(Cw.red);
    const int tid = threadIdx.x;
    const int p = nn, q = mm;
    long long t_prev_ = Cw.stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const View Mv = mkview(Tm, plain(q), plain(1));                          // M[i, c] = Tm[i * mm + c] = T[c, i]
    wg_syrk(p, q, Mv, mkview(Cw.Ga, plain(1), plain(128)), 1.0, lds);
    OQ_STAMP(0)
    wg_img_load(lds, 0, Cw.Ga, p, Cw.red);
    if (wg_chol_lds128(p, lds, Cw.red, Cw.iflag, Cw.scal + 1) != 0) return 0;
    if (!(unif64(Cw.scal[1]) <= ORTHO_CHOLQR_PIVOT_MAX)) return 0;
    for (int e = tid; e < p * p; e += TTN_WG) { const int i = e % p, j = e / p; Cw.L1g[i + 128 * j] = lds[i + 128 * j]; }      // L1 (the Gram product below takes the image)
    OQ_STAMP(1)
    wg_trsm_lower_cols(p, q, lds, Tm, q, 1.0, Qb, q);                        // Q1 = L1^-1 M, row-major p x q = Qb column-major mm x nn
    OQ_STAMP(2)
    wg_syrk(p, q, mkview(Qb, plain(q), plain(1)), mkview(Cw.Cc, plain(1), plain(128)), 1.0, lds);
    const double dev = wg_img_load(lds, 0, Cw.Cc, p, Cw.red);
    OQ_STAMP(3)
    const double* Lsrc = Cw.L1g;                                             // L[i, k] at Lsrc[i + 128 k]
    if (!(dev <= ORTHO_CHOLQR_ACCEPT)) {
        if (!(dev <= ORTHO_CHOLQR_SECOND)) return 0;
        if (wg_chol_lds128(p, lds, Cw.red, Cw.iflag, Cw.scal + 1) != 0) return 0;
        wg_trsm_lower_cols(p, q, lds, Qb, q, 1.0, Qb, q);                    // Q = L2^-1 Q1 (in place: a thread owns its column)
        wg_img_load(lds, 64, Cw.L1g, p, Cw.red);
        wg_tril_mul_lds(p, lds);                                             // L = L1 L2 in the image
        for (int e = tid; e < p * p; e += TTN_WG) { const int i = e % p, j = e / p; Cw.L1g[i + 128 * j] = lds[i + 128 * j]; }
        __syncthreads();
    }
    for (int e = tid; e < p * p; e += TTN_WG) {                              // R = L^T: Rb[i + nn c] = L[c, i] for i <= c
        const int i = e % p, c = e / p;
        Rb[e] = (i <= c) ? Lsrc[c + 128 * i] : 0.0;
    }
    __syncthreads();
    OQ_STAMP(4)
    return p;
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

// the sites csrc/ttn_ortho512.h takes: tall QTT cores of rank <= 64 in the right-to-left sweep (the same test in all three launches)
__device__ __forceinline__ bool ortho512_eligible(int n, int rl, int rr, int ynext) {
    return n == 2 && rl <= 64 && rr <= 64 && ynext <= 64 && 2 * ynext > rl;
}

__global__ void TTN_KERNEL_BOUNDS k_orthogonalize(OrthoArgs P) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int b = P.trains ? P.trains[blockIdx.x] : blockIdx.x;
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
    CholQrWork Cw;                                                           // Cholesky-QR of the well-conditioned tall steps (rank <= 64)
    Cw.Ga = W.Tst + (long long)((P.rmax + QR_NB - 1) / QR_NB) * QR_NB * QR_NB;
    Cw.Cc = Cw.Ga + 128 * 128;
    Cw.L1g = Cw.Cc + 128 * 128;
    Cw.scal = W.Ss;                                                          // LDS words of the Householder panel matrices, dead outside wg_lq_blocked
    Cw.iflag = (int*)(W.Ss + 8);
    Cw.red = red;
    Cw.stamps = P.prof ? P.prof + 136LL * gridDim.x + 64LL * b : nullptr;
    if (P.prof && tid == 0) P.prof[16LL * gridDim.x + 120LL * b + 100] = (long long)__builtin_amdgcn_s_memtime();
    if (P.mode != 3 && tid == 0) dev_r_and_d_to_rks(d, X.dims, xr, 1024, yr);
    __syncthreads();
    const int ic = P.center;
    int* st = P.state ? P.state + 4 * b : nullptr;
    if (P.mode == 3 && uni32(st[3]) == 1) return;        // k_ortho512 finished this train, centre core included

    // ---- left sweep: sites 0..ic-1 (src/tt_tools.jl:518-525) ----
    if (P.mode != 3 && tid == 0) { Rb0[0] = 1.0; }
    __syncthreads();
    View FR = mkview(Rb0, plain(1), plain(1));            // (yr_j x rl)
    int which = 0;
    if (P.mode == 3) {                                    // the left sweep ran in the mode-1 launch: its last R is in buffer st[2]
        which = uni32(st[2]);
        if (ic > 0) FR = mkview(which ? Rb1 : Rb0, plain(1), plain(uni32((int)yr[ic])));      // (which was toggled after the last step)
    }
    for (int j = 0; j < ic && P.mode != 3; ++j) {
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
        int rnew = (rr <= 64 && mm > rr && !(P.no_cholqr & 1)) ? wg_qr_cholqr(mm, rr, Tm, Qb, Rn, Cw, lds) : 0;
        if (!rnew) rnew = wg_qr_explicit(mm, rr, Tm, Qb, Rn, W, lds);
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
    if (P.mode != 3 && tid == 0) { Rc[0] = 1.0; }
    __syncthreads();
    View FL = mkview(Rc, plain(1), plain(1));             // (rr x yr_{j+1})
    int whichL = 0;
    int jstart = d - 1;
    if (P.mode == 3) {                                    // resume: sites above st[0] are done, their last R is in buffer st[1]
        jstart = uni32(st[0]);
        whichL = uni32(st[1]);
        if (jstart < d - 1) FL = tview(mkview(whichL ? Rd : Rc, plain(1), plain(uni32((int)yr[jstart + 1]))));
    }
    int jstop = ic;                                       // (mode 1: the site the 512-thread kernel starts with)
    for (int j = jstart; j > ic; --j) {
        const int n = X.dims[j];
        const int ynext = uni32((int)yr[j + 1]), rl = uni32((int)xr[j]), rr = uni32((int)xr[j + 1]);
        if (P.mode == 1 && (P.ramp || ortho512_eligible(n, rl, rr, ynext))) { jstop = j; break; }
        double* Xj = X.data + (long long)b * X.stride + X.off[j];
        double* Yj = Y.data + (long long)b * Y.stride + Y.off[j];
        const int mm = ynext * n;
        double* Rn = whichL ? Rc : Rd;
        // the whole step in registers and LDS when the site is a tall QTT core of rank <= 64 (ttn_ortho_fused.h); 0 = not taken / refused
        int rnew = 0;
        if (n == 2 && rl <= 64 && rr <= 64 && ynext <= 64 && mm > rl && !(P.no_cholqr & 2))
            rnew = ortho_step_fused(Xj, Yj, whichL ? Rd : Rc, Rn, rl, rr, ynext, lds, Cw.stamps ? Cw.stamps + 8 : nullptr);
        if (!rnew) {
        // Tt[(be + ynext*s), al] = sum_ga FL[ga,be] X_j[s,al,ga]  ==  (FL^T) * X2,  X2[ga, (s + n*al)]
        const View X2 = mkview(Xj, plain((long long)n * rl), plain(1));
        const View Tv = mkview(Tm, plain(1), Idx{n, (long long)ynext, (long long)mm});
        wg_gemm(ynext, n * rl, rr, tview(FL), X2, Tv, 1.0, 0.0, lds);
        rnew = (rl <= 64 && mm > rl && !(P.no_cholqr & 1)) ? wg_qr_cholqr(mm, rl, Tm, Qb, Rn, Cw, lds) : 0;
        if (!rnew) rnew = wg_qr_explicit(mm, rl, Tm, Qb, Rn, W, lds);
        // Y_j[s, al, be] = Qt[(be + ynext*s), al]   (core shape (n, rnew, ynext))
        for (int e = tid; e < mm * rnew; e += TTN_WG) {
            const int row = e % mm, al = e / mm;
            const int be = row % ynext, s = row / ynext;
            Yj[s + (long long)n * (al + (long long)rnew * be)] = Qb[e];
        }
        }
        if (tid == 0) yr[j] = rnew;
        __syncthreads();
        FL = tview(mkview(Rn, plain(1), plain(rnew)));    // FL[ga, be] = Rt[be, ga]  (rl x rnew)
        whichL ^= 1;
        if (P.prof && tid == 0 && d - 1 - j < 120) P.prof[16LL * gridDim.x + 120LL * b + (d - 1 - j)] = (long long)__builtin_amdgcn_s_memtime();
    }
    if (P.mode == 1) {                                    // hand over to k_ortho512 (and then to the mode-3 launch)
        if (tid == 0) { st[0] = jstop; st[1] = whichL; st[2] = which; st[3] = 0; }
        return;
    }
#define OSTAMP(i) if (P.prof && tid == 0) P.prof[16LL * gridDim.x + 120LL * b + 100 + (i)] = (long long)__builtin_amdgcn_s_memtime();
    OSTAMP(1)
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
        OSTAMP(2)
        const View Tmv = mkview(Tm, plain(1), plain(mm));                               // (mm x rr)
        const View Yv = mkview(Yi, Idx{yl, (long long)n, 1}, plain((long long)n * yl)); // [(al + yl*s), be']
        wg_gemm(mm, yn, rr, Tmv, FL, Yv, 1.0, 0.0, lds);
        OSTAMP(3)
    }
}
