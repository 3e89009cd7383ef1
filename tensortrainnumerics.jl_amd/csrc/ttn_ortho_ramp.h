// ttn_ortho_ramp.h — the ramp sites of the right-to-left orthogonalize sweep, ONE WAVE per train, everything in registers.
// (test infrastructure: none — product code; reference: orthogonalize, src/tt_tools.jl:528-536.)
//
// The sweep starts at the right end, where the ranks double from site to site (1, 2, 4 ... 64): the LQ step there factors a
// WIDE or SQUARE matrix (rows = 2 y_{j+1} <= columns = r_j <= 64).  These matrices are small, and ill conditioned (the R factors of
// the sites before multiply into them: cond 1e2 ... 1e8 measured on the benchmark trains), so they need Householder reflections —
// 64 dependent column steps, each a norm, a rank-one update and several barriers when a 1024-thread workgroup runs it (650 k clk for
// the six ramp sites of a rank-64 train, one train per CU: 1.4 ms of the 5.9 ms a batch of 1024 takes).  A wave needs no barriers:
//   * lane c holds COLUMN c of the site's matrix W (rows i = 2 be + s in registers a[0 .. ROWS-1]);
//   * the reflector of step k lives in lane k: x_i = readlane(a[i], k) is a scalar, the rank-one update is two FMAs per row;
//   * the R factor stays in registers for the next site's carry product  W'[2 be + s][al] = sum_ga R[be][ga] X'[s, al, ga]
//     (again readlane scalars times per-lane 16-byte loads of X');
//   * Q^T is accumulated on the fly (the same reflector, the same readlane scalars, applied to a second register array that starts as
//     the identity) and written as the core Y_j.
// Four waves per workgroup, one per SIMD: 1024 trains run in one round over the chip.
#pragma once

#define ORAMP_WG 256

__device__ __forceinline__ double oramp_readlane(double v, int lane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

// a site the ramp kernel takes (the complement of ortho512_eligible among QTT sites of rank <= 64)
__device__ __forceinline__ bool ortho_ramp_eligible(int n, int rl, int rr, int ynext) {
    return n == 2 && rl <= 64 && rr <= 64 && 2 * ynext <= rl;
}

// Householder steps k in [kbeg, kend) of the phase whose rows start at LO (k lies in rows LO .. LO + CH - 1: rows below LO are finished and
// are not touched, rows from LO + CH on are below every k of the phase and need no masks).  The reflector is kept UNSCALED:
// H = I - t u u^T, u = x - beta e_k, t = -1 / (beta u_k): the dot products with x need neither beta nor a stored copy of x — every
// x_i is a readlane scalar used on the spot (two passes over the rows, no uniform array, no SGPR spills).
// The same reflector is applied, in the same two passes, to B (b[], starts as the identity): after the last step B = H_{r-2} ... H_0 =
// Q^T, lane c holding column c of Q^T = row c of Q.  The first version formed Q afterwards from the stored reflectors (the dorg2r
// recurrence, a second sweep over k with its own readlanes, and a special case for lane k that cost a multiply per row to do
// accurately); accumulating Q^T on the fly shares the readlanes with the factorisation and has no special lane.
template <int ROWS, int LO, int CH>
__device__ __forceinline__ void oramp_phase(double (&a)[ROWS], double (&b)[ROWS], int kbeg, int kend) {
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int kk = kbeg; kk < kend; ++kk) {
        const int k = __builtin_amdgcn_readfirstlane(kk);
        double xk = 0.0, ak = 0.0, bk = 0.0;
        double n0 = 0.0, n1 = 0.0, d0 = 0.0, d1 = 0.0, e0 = 0.0, e1 = 0.0;
#pragma unroll
        for (int i = LO; i < ROWS; ++i) {
            double xi = oramp_readlane(a[i], k);
            if (i < LO + CH) {
                xk = (i == k) ? xi : xk;
                ak = (i == k) ? a[i] : ak;
                bk = (i == k) ? b[i] : bk;
                xi = (i > k) ? xi : 0.0;
            }
            if (i & 1) { n1 = fma(xi, xi, n1); d1 = fma(xi, a[i], d1); e1 = fma(xi, b[i], e1); }
            else       { n0 = fma(xi, xi, n0); d0 = fma(xi, a[i], d0); e0 = fma(xi, b[i], e0); }
        }
        const double nrm2 = n0 + n1;
        if (nrm2 == 0.0) continue;                                        // H = I: the column is already in its final form
        const double beta = -copysign(sqrt(fma(xk, xk, nrm2)), xk);
        const double uk = xk - beta;
        const double t = -1.0 / (beta * uk);
        const bool mine = lane == k;
        const double tsa = (lane > k) ? t * fma(uk, ak, d0 + d1) : 0.0;   // (the lanes left of k hold finished columns and their reflectors)
        const double tsb = t * fma(uk, bk, e0 + e1);
#pragma unroll
        for (int i = LO; i < ROWS; ++i) {
            const double xi = oramp_readlane(a[i], k);
            if (i < LO + CH) {
                const double ui = (i > k) ? xi : ((i == k) ? uk : 0.0);
                const double u = fma(-tsa, ui, a[i]);
                a[i] = (mine && i == k) ? beta : u;
                b[i] = fma(-tsb, ui, b[i]);
            } else {
                a[i] = fma(-tsa, xi, a[i]);                               // (lane k: tsa = 0, its rows below the diagonal keep u_i = x_i)
                b[i] = fma(-tsb, xi, b[i]);
            }
        }
    }
}

// One site.  In: Rp[be] (lane ga) = FL[ga][be] for be < ROWS / 2 (zero beyond ynext and beyond rr).  Out: Y_j, R to Rn (ld = rows),
// Rp = this site's R for the next one (when ROWS <= 32).
template <int ROWS>
__device__ __forceinline__ void oramp_site(double (&Rp)[32], const double* __restrict__ Xj, double* __restrict__ Yj, double* __restrict__ Rn,
                                           int rl, int rr, int ynext) {
    const int lane = threadIdx.x & 63;
    const int rows = 2 * ynext;
    constexpr int HB = ROWS / 2;
    constexpr int CH = ROWS < 16 ? ROWS : 16;
    double a[ROWS], b[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) { a[i] = 0.0; b[i] = (i == lane) ? 1.0 : 0.0; }
    // ---- carry: W[2 be + s][al = lane] = sum_ga FL[ga][be] X_j[s, al, ga] ----
    {
        typedef double __attribute__((ext_vector_type(2))) d2;
        const d2* xg = reinterpret_cast<const d2*>(Xj);
        const bool in = lane < rl;
        d2 xn = in ? xg[lane] : (d2){0.0, 0.0};
        for (int ga = 0; ga < rr; ++ga) {
            const d2 x = xn;
            if (ga + 1 < rr) xn = in ? xg[lane + rl * (ga + 1)] : (d2){0.0, 0.0};
            const int gs = __builtin_amdgcn_readfirstlane(ga);
#pragma unroll
            for (int be = 0; be < HB; ++be) {
                const double f = oramp_readlane(Rp[be], gs);
                a[2 * be] = fma(f, x.x, a[2 * be]);
                a[2 * be + 1] = fma(f, x.y, a[2 * be + 1]);
            }
        }
    }
    // ---- Householder steps k = 0 .. rows - 2 (the last row needs none) on W, accumulated into B = Q^T ----
    const int klast = rows - 1;
    oramp_phase<ROWS, 0, CH>(a, b, 0, klast < CH ? klast : CH);
    if constexpr (ROWS > 16) oramp_phase<ROWS, 16, CH>(a, b, 16, klast < 32 ? klast : 32);
    if constexpr (ROWS > 32) {
        oramp_phase<ROWS, 32, CH>(a, b, 32, klast < 48 ? klast : 48);
        oramp_phase<ROWS, 48, CH>(a, b, 48, klast);
    }
    // ---- R (rows x rl, upper trapezoid) to global memory and to the registers the next site's carry reads ----
    {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const double r = (i <= lane && i < rows) ? a[i] : 0.0;
            if (lane < rl && i < rows) Rn[i + rows * lane] = r;
            if (i < 32) Rp[i] = (lane < rl) ? r : 0.0;
        }
#pragma unroll
        for (int i = (ROWS < 32 ? ROWS : 32); i < 32; ++i) Rp[i] = 0.0;
    }
    // ---- Y_j[s, al', be] = Q[2 be + s][al'] = B[al'][2 be + s]: lane i = 2 be + s holds column i of B, register al' its row al' ----
    if (lane < rows) {
        double* yl = Yj + (lane & 1) + 2LL * rows * (lane >> 1);
#pragma unroll
        for (int al = 0; al < ROWS; ++al)
            if (al < rows) yl[2 * al] = b[al];
    }
}

// mode 5: no launch ran before (the centre is site 0: there is no left sweep) — this kernel also initialises the ranks of y and the
// state; mode 1 ran before otherwise (left sweep; it stopped the right sweep at its first site because P.ramp is set).
__global__ void __launch_bounds__(ORAMP_WG) k_ortho_ramp(OrthoArgs P, int batch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * (ORAMP_WG / 64) + wave;
    if (b >= batch) return;
    const TTDev& X = P.x; const TTDev& Y = P.y;
    const int d = X.d, ic = P.center;
    const long long* xr = X.rks + (long long)b * (d + 1);
    long long* yr = Y.rks + (long long)b * (d + 1);
    double* scr = P.scratch + (long long)b * P.scratch_stride;
    double* Rb0 = scr + 2LL * P.mmax * P.rmax;                                                // (the layout of k_orthogonalize)
    double* Rc = Rb0 + 2LL * P.rmax * P.rmax;
    double* Rd = Rc + (long long)P.rmax * P.rmax;
    int* st = P.state + 4 * b;
    int j, whichL, which;
    if (P.mode == 5) {
        if (lane == 0) { dev_r_and_d_to_rks(d, X.dims, xr, 1024, yr); Rb0[0] = 1.0; Rc[0] = 1.0; }
        j = d - 1; whichL = 0; which = 0;
    } else {
        j = __builtin_amdgcn_readfirstlane(st[0]); whichL = __builtin_amdgcn_readfirstlane(st[1]); which = __builtin_amdgcn_readfirstlane(st[2]);
    }
    if (j == d - 1) {                                   // (a sweep some other launch began is left to the kernels after this one)
        double Rp[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) Rp[i] = 0.0;
        Rp[0] = (lane == 0) ? 1.0 : 0.0;                // FL of the last site: the 1 x 1 identity
        int ynext = 1;
        const double* Xbase = X.data + (long long)b * X.stride;
        double* Ybase = Y.data + (long long)b * Y.stride;
        while (j > ic) {
            const int rl = __builtin_amdgcn_readfirstlane((int)xr[j]), rr = __builtin_amdgcn_readfirstlane((int)xr[j + 1]);
            const int n = X.dims[j];
            if (!ortho_ramp_eligible(n, rl, rr, ynext)) break;
            const double* Xj = Xbase + X.off[j];
            double* Yj = Ybase + Y.off[j];
            double* Rn = whichL ? Rc : Rd;
            const int rows = 2 * ynext;
            if (rows <= 2) oramp_site<2>(Rp, Xj, Yj, Rn, rl, rr, ynext);
            else if (rows <= 4) oramp_site<4>(Rp, Xj, Yj, Rn, rl, rr, ynext);
            else if (rows <= 8) oramp_site<8>(Rp, Xj, Yj, Rn, rl, rr, ynext);
            else if (rows <= 16) oramp_site<16>(Rp, Xj, Yj, Rn, rl, rr, ynext);
            else if (rows <= 32) oramp_site<32>(Rp, Xj, Yj, Rn, rl, rr, ynext);
            else oramp_site<64>(Rp, Xj, Yj, Rn, rl, rr, ynext);
            if (lane == 0) yr[j] = rows;
            ynext = rows;
            whichL ^= 1;
            --j;
        }
    }
    if (lane == 0) { st[0] = j; st[1] = whichL; st[2] = which; st[3] = 0; }
}
