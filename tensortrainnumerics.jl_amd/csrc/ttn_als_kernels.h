// ttn_als_kernels.h — als_linsolve (src/solvers/als.jl:161-222) for a batch of independent right-hand sides / start trains and
// one operator: one workgroup owns one train for the whole solve (the half sweeps are sequential along the chain, like
// tt_compress!).  Ranks are FIXED by the start train (als.jl:177), so every shape below is known on the host.
//   environments   G_i (n, r_{i-1}, n, r_{i-1}, R_i), Gb_i (n, r_{i-1}, rb_i), H_i (R_i, r_i, r_i), Hb_i (r_i, rb_i)   (als.jl:9-55)
//   local system   K[(a,b,c),(d,e,f)] = sum_z G_i[a,b,d,e,z] H_i[z,c,f] ;  Pb = Gb_i Hb_i' ;  V = K \ Pb               (als.jl:58-70)
//   core moves     thin QR of V, R pushed into the neighbour                                                              (als.jl:102-135)
// The reference assembles K densely and calls LAPACK's LU (`K \ b`; its it_solver keyword is ignored, :161,203): the device
// does the same — dense K in global memory, blocked right-looking LU with partial pivoting by the whole workgroup (trailing updates on the MFMA) — so results agree
// to rounding * cond(K).  The contractions are staged exactly as the reference's @tensoropt orders suggest (three small stages
// each); at ALS sizes they are latency-, not flop-bound, and run on the VALU.  Local sizes n*r*r up to 2048 are supported.
#pragma once
#include "ttn_ortho_kernels.h"

struct AlsArgs {
    TTODev A;
    TTDev b, x;
    int sweep_count;
    double* scratch;
    long long scratch_stride;
    const long long* off;        // device [4][d]: offsets of G_i, Gb_i, H_i, Hb_i in the per-train scratch
    long long offK, offPb, offPiv, offT1, offT2, offTm, offQb, offRb, offVb, offWb, offTst;
    int mmax, rmax;
    int* status;                 // [batch]: 0 ok, 3 singular local system, 4 ranks differ from the handle's bound
    const long long* rfix;       // device [d+1]: the fixed ranks of x
    // The GRID form for local systems too large for one workgroup (ttn_api.hip: als_grid_path): the host walks the half sweeps and runs
    // the same kernel phase by phase for ONE train (train0; its scratch at `scratch`), the assembly of K and the LU on the whole chip
    // in between.  phase 0: the whole solve in one launch (the one-workgroup form); 1: rank check + initial environments; 2 / 3: the
    // core move and environment update after the local solve of `site` in a forward / backward half sweep (V in the Pb slot).
    int phase, site, train0;
};

// Dense solve K v = rhs (in place, v overwrites rhs); K is N x N column-major (destroyed).  Blocked right-looking LU with
// partial pivoting (first maximal |entry| of the column, like LAPACK's idamax): panels of LU_NB columns are factored column
// by column inside the panel, the row interchanges are then applied to the rest of the matrix and to the right-hand side,
// U12 = L11^-1 A12 by forward substitution (one column per thread), and the trailing block gets A22 -= L21 U12 as ONE MFMA
// GEMM (wg_gemm, alpha = -1, beta = 1) — that GEMM carries the 2/3 N^3 flops.  The right-hand side is eliminated on the fly;
// back substitution at the end.  Returns 0, or 1 if a pivot is exactly zero (LAPACK: SingularException).
#define LU_NB 32
__device__ __noinline__ int wg_lu_solve(int N, double* K_, double* rhs_, int* piv_, double* red, int* iflag_, double* lds) {
    N = uni32(N); K_ = unip(K_); rhs_ = unip(rhs_); piv_ = unip(piv_); red = unip(red); iflag_ = unip(iflag_); lds = unip(lds);
    // typed address spaces: through the generic pointers every access below was a FLAT instruction
    typedef __attribute__((address_space(1))) int gmem_i32;
    gmem_wf64* K = (gmem_wf64*)K_;
    gmem_wf64* rhs = (gmem_wf64*)rhs_;
    gmem_i32* piv = (gmem_i32*)piv_;
    lds_i32* iflag = (lds_i32*)iflag_;
    const int tid = threadIdx.x;
    for (int k0 = 0; k0 < N; k0 += LU_NB) {
        const int w = min(LU_NB, N - k0);
        // ---- (a) the panel, column by column.  A panel of at most 16384 doubles is factored in LDS (every access of the
        //      column loop is then an LDS access instead of a global-memory round trip: the loop is latency bound) ----
        const int mrows = N - k0;
        const bool pan_lds = (long long)mrows * LU_NB <= 128 * 128;
        if (pan_lds) {
            lds_f64* Pn = (lds_f64*)lds;
            const int ldp = mrows | 1;                            // odd leading dimension
            for (int e = tid; e < mrows * w; e += TTN_WG) { const int c = e / mrows, i = e - c * mrows; Pn[c * ldp + i] = K[(long long)(k0 + c) * N + k0 + i]; }
            __syncthreads();
            for (int jl = 0; jl < w; ++jl) {
                lds_f64* colj = Pn + jl * ldp;
                double vm = 0.0;
                for (int i = jl + tid; i < mrows; i += TTN_WG) vm = fmax(vm, fabs(colj[i]));
                if (tid == 0) iflag[0] = N;
                vm = unif64(wg_max(vm, red));
                if (!(vm > 0.0)) return 1;
                for (int i = jl + tid; i < mrows; i += TTN_WG) if (fabs(colj[i]) == vm) atomicMin((int*)iflag_, i);
                __syncthreads();
                const int pvl = uni32(iflag[0]);
                if (tid == 0) piv[k0 + jl] = k0 + pvl;
                if (pvl != jl && tid < w) { lds_f64* c = Pn + tid * ldp; const double t = c[jl]; c[jl] = c[pvl]; c[pvl] = t; }
                __syncthreads();
                const double pivot = colj[jl];
                __syncthreads();                                  // everybody has the pivot before column jl is scaled
                for (int i = jl + 1 + tid; i < mrows; i += TTN_WG) colj[i] = colj[i] / pivot;
                __syncthreads();
                const int m = mrows - jl - 1, nc = w - jl - 1;
                for (int e = tid; e < m * nc; e += TTN_WG) {
                    const int i = jl + 1 + e % m, c = jl + 1 + e / m;
                    Pn[c * ldp + i] = fma(-colj[i], Pn[c * ldp + jl], Pn[c * ldp + i]);
                }
                __syncthreads();
            }
            for (int e = tid; e < mrows * w; e += TTN_WG) { const int c = e / mrows, i = e - c * mrows; K[(long long)(k0 + c) * N + k0 + i] = Pn[c * ldp + i]; }
            __syncthreads();
        } else
        for (int j = k0; j < k0 + w; ++j) {
            gmem_wf64* colj = K + (long long)j * N;
            double vm = 0.0;
            for (int i = j + tid; i < N; i += TTN_WG) vm = fmax(vm, fabs(colj[i]));
            if (tid == 0) iflag[0] = N;
            vm = unif64(wg_max(vm, red));                   // barriers inside: iflag[0] is visible after it
            if (!(vm > 0.0)) return 1;
            for (int i = j + tid; i < N; i += TTN_WG) if (fabs(colj[i]) == vm) atomicMin((int*)iflag_, i);
            __syncthreads();
            const int pv = uni32(iflag[0]);
            if (tid == 0) piv[j] = pv;
            if (pv != j && tid < w) {                        // interchange inside the panel now, outside it in (b)
                gmem_wf64* c = K + (long long)(k0 + tid) * N;
                const double t = c[j]; c[j] = c[pv]; c[pv] = t;
            }
            __syncthreads();
            const double pivot = colj[j];
            for (int i = j + 1 + tid; i < N; i += TTN_WG) colj[i] = colj[i] / pivot;
            __syncthreads();
            const int m = N - j - 1, nc = k0 + w - j - 1;    // rank-1 update of the rest of the panel
            for (long long e = tid; e < (long long)m * nc; e += TTN_WG) {
                const int i = j + 1 + (int)(e % m), c = j + 1 + (int)(e / m);
                K[(long long)c * N + i] = fma(-colj[i], K[(long long)c * N + j], K[(long long)c * N + i]);
            }
            __syncthreads();
        }
        // ---- (b) the panel's interchanges on the other columns (a thread owns a column: order kept) and on the rhs ----
        for (int c = tid; c < N + 1; c += TTN_WG) {
            if (c >= k0 && c < k0 + w) continue;
            gmem_wf64* col = (c == N) ? rhs : K + (long long)c * N;
            for (int j = k0; j < k0 + w; ++j) {
                const int pv = piv[j];
                if (pv != j) { const double t = col[j]; col[j] = col[pv]; col[pv] = t; }
            }
        }
        __syncthreads();
        // ---- (c) U12 = L11^-1 A12 and the same for the rhs: forward substitution with the unit lower triangle of the panel,
        //      staged in LDS (zero padded to LU_NB) so that a thread keeps its column segment in registers ----
        lds_f64* L11 = (lds_f64*)lds;
        for (int e = tid; e < LU_NB * LU_NB; e += TTN_WG) {
            const int ii = e % LU_NB, jj = e / LU_NB;
            L11[e] = (ii < w && jj < w && ii > jj) ? K[(long long)(k0 + jj) * N + k0 + ii] : 0.0;
        }
        __syncthreads();
        for (int c = k0 + w + tid; c < N + 1; c += TTN_WG) {
            gmem_wf64* col = (c == N) ? rhs : K + (long long)c * N;
            double u[LU_NB];
#pragma unroll
            for (int ii = 0; ii < LU_NB; ++ii) u[ii] = (ii < w) ? col[k0 + ii] : 0.0;
#pragma unroll
            for (int jj = 0; jj < LU_NB - 1; ++jj) {
#pragma unroll
                for (int ii = jj + 1; ii < LU_NB; ++ii) u[ii] = fma(-L11[jj * LU_NB + ii], u[jj], u[ii]);
            }
#pragma unroll
            for (int ii = 0; ii < LU_NB; ++ii) if (ii < w) col[k0 + ii] = u[ii];
        }
        __syncthreads();
        const int m = N - k0 - w;
        if (m > 0) {
            // rhs[i] -= sum_jj L21[i, jj] rhs[k0 + jj]
            for (int i = k0 + w + tid; i < N; i += TTN_WG) {
                double a = rhs[i];
                for (int jj = 0; jj < w; ++jj) a = fma(-K[(long long)(k0 + jj) * N + i], rhs[k0 + jj], a);
                rhs[i] = a;
            }
            // ---- (d) A22 -= L21 U12 ----
            const View L21 = mkview(K_ + (long long)k0 * N + (k0 + w), plain(1), plain(N));                 // m x w
            const View U12 = mkview(K_ + (long long)(k0 + w) * N + k0, plain(1), plain(N));                 // w x m
            const View A22 = mkview(K_ + (long long)(k0 + w) * N + (k0 + w), plain(1), plain(N));           // m x m
            wg_gemm(m, m, w, L21, U12, A22, -1.0, 1.0, lds);
        }
        __syncthreads();
    }
    // back substitution with U, in blocks of 32 unknowns: the triangular block is solved in LDS by wave 0, the rows above it take
    // the block's contribution in parallel (same operations in the same order as the unblocked loop, one global round trip per
    // block instead of one per unknown)
    {
        lds_f64* Ub = (lds_f64*)lds;                              // [c * 33 + r], r <= c
        lds_f64* yb = Ub + 33 * 32;
        const int lane = tid & 63, wave = tid >> 6;
        for (int kb = ((N - 1) / 32) * 32; kb >= 0; kb -= 32) {
            const int wb = min(32, N - kb);
            for (int e = tid; e < wb * wb; e += TTN_WG) { const int c = e / wb, r_ = e - c * wb; if (r_ <= c) Ub[c * 33 + r_] = K[(long long)(kb + c) * N + kb + r_]; }
            if (tid < wb) yb[tid] = rhs[kb + tid];
            __syncthreads();
            if (wave == 0) {
                for (int c = wb - 1; c >= 0; --c) {
                    const double xc = yb[c] / Ub[c * 33 + c];
                    if (lane < c) yb[lane] = fma(-Ub[c * 33 + lane], xc, yb[lane]);
                    if (lane == c) yb[c] = xc;
                }
            }
            __syncthreads();
            for (int i = tid; i < kb; i += TTN_WG) {
                double a = rhs[i];
                for (int c = wb - 1; c >= 0; --c) a = fma(-K[(long long)(kb + c) * N + i], yb[c], a);
                rhs[i] = a;
            }
            if (tid < wb) rhs[kb + tid] = yb[tid];
            __syncthreads();
        }
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Matrix-free local solve of the two-site solvers (src/solvers/dmrg.jl:92-171, the branch `it_solver || N > itslv_thresh`):
// conjugate gradients on the SYMMETRISED local operator
//     K_s v = 1/2 (K + K^T) v,   K[(ab,cd),(ef,gh)] = sum_z G_z[ab,ef] H_z[cd,gh]
// (the reference symmetrises the same way, dmrg.jl:135-166; its operator is the three-tensor sandwich G (x) Amid (x) H, which is
// this Kronecker sum with the window's first / second operator core folded into G / H).  With the unknown viewed as the na x nb
// matrix V, K v = sum_z G_z V H_z^T and K^T v = sum_z G_z^T V H_z: per application 2 R_z products na x nb x na into the slab
// W = [W_1 ... W_R] and two products na x nb x (R_z nb) against the stacked H — all on the fp64 MFMA GEMM (wg_gemm), no K anywhere:
// the 65 536-unknown systems of rank-128 trains (a 34 GB matrix) need 2 MB of G, 2 MB of H and a 2 MB slab.
// The iteration is KrylovKit's CG (the reference's `linsolve(...; issymmetric, isposdef)`): start from x0, stop when ||r||_2 < tol
// (ABSOLUTE, as `tol` is passed there) or after maxiter iterations, in which case the current iterate is returned like there.
// G: (na, na, Rz) column-major; H: (Rz, nb, nb) with z fastest; x (in: x0, out: solution), rhs, and the work vectors r, p, q: N = na*nb
// doubles each; W: na*nb*Rz.  Returns the number of iterations.
// -------------------------------------------------------------------------------------------------
__device__ __noinline__ int wg_cg_two_site(int na, int nb, int Rz, double* G, double* H, double* x, const double* rhs, double* r, double* p, double* q,
                                           double* W, double tol, int maxiter, double* red, double* lds) {
    na = uni32(na); nb = uni32(nb); Rz = uni32(Rz); maxiter = uni32(maxiter);
    G = unip(G); H = unip(H); x = unip(x); rhs = unip(rhs); r = unip(r); p = unip(p); q = unip(q); W = unip(W); red = unip(red); lds = unip(lds);
    const int N = na * nb;
    const int tid = threadIdx.x;
    // out = K_s v
    auto apply = [&](double* v, double* out) {
        const View Vv = mkview(v, plain(1), plain(na));
        const View Ov = mkview(out, plain(1), plain(na));
        const View Wcat = mkview(W, plain(1), plain(na));                                        // na x (Rz nb): column gh + nb z
        for (int z = 0; z < Rz; ++z)                                                              // W_z = G_z V
            wg_gemm(na, nb, na, mkview(G + (long long)na * na * z, plain(1), plain(na)), Vv, mkview(W + (long long)N * z, plain(1), plain(na)), 1.0, 0.0, lds);
        // out = 1/2 sum_z W_z H_z^T:  B[(gh, z), cd] = H[z, cd, gh]
        wg_gemm(na, nb, Rz * nb, Wcat, mkview(H, Idx{nb, (long long)Rz * nb, 1}, plain(Rz)), Ov, 0.5, 0.0, lds);
        for (int z = 0; z < Rz; ++z)                                                              // W_z = G_z^T V
            wg_gemm(na, nb, na, tview(mkview(G + (long long)na * na * z, plain(1), plain(na))), Vv, mkview(W + (long long)N * z, plain(1), plain(na)), 1.0, 0.0, lds);
        // out += 1/2 sum_z W_z H_z:    B[(gh, z), cd] = H[z, gh, cd]
        wg_gemm(na, nb, Rz * nb, Wcat, mkview(H, Idx{nb, (long long)Rz, 1}, plain((long long)Rz * nb)), Ov, 0.5, 1.0, lds);
    };
    auto dot = [&](const double* u, const double* v) {
        double a = 0.0;
        for (int e = tid; e < N; e += TTN_WG) a = fma(u[e], v[e], a);
        return unif64(wg_sum(a, red));
    };
    apply(x, q);
    for (int e = tid; e < N; e += TTN_WG) { const double re = rhs[e] - q[e]; r[e] = re; p[e] = re; }
    __syncthreads();
    double rho = dot(r, r);
    int it = 0;
    while (!(sqrt(rho) < tol) && it < maxiter) {
        apply(p, q);
        const double pq = dot(p, q);
        const double alpha = rho / pq;
        for (int e = tid; e < N; e += TTN_WG) { x[e] = fma(alpha, p[e], x[e]); r[e] = fma(-alpha, q[e], r[e]); }
        __syncthreads();
        const double rho_new = dot(r, r);
        const double beta = rho_new / rho;
        rho = rho_new;
        ++it;
        if (sqrt(rho) < tol) break;
        for (int e = tid; e < N; e += TTN_WG) p[e] = fma(beta, p[e], r[e]);
        __syncthreads();
    }
    __syncthreads();
    return it;
}

// Shared by als_linsolve and mals_linsolve: G_{i+1}, Gb_{i+1} from site i (als.jl:47-55).  `xr` = the ranks of x to use.
struct AlsEnv {
    TTODev A; TTDev b, x;
    int tb;                      // train
    double* scr; const long long* off; const long long *xr, *br;
    double *T1, *T2;
    int d;
};
#define XC(i) (E.x.data + (long long)E.tb * E.x.stride + E.x.off[i])
#define BC(i) (E.b.data + (long long)E.tb * E.b.stride + E.b.off[i])
#define AC(i) (E.A.data + E.A.off[i])
#define GP(i) (E.scr + E.off[i])
#define GBP(i) (E.scr + E.off[E.d + (i)])
#define HP(i) (E.scr + E.off[2 * E.d + (i)])
#define HBP(i) (E.scr + E.off[3 * E.d + (i)])
#define SITE(i) AlsSite{uni32(E.x.dims[i]), uni32((int)E.xr[i]), uni32((int)E.xr[(i) + 1]), uni32((int)E.A.rks[i]), uni32((int)E.A.rks[(i) + 1]), \
                        uni32((int)E.br[i]), uni32((int)E.br[(i) + 1])}
#define WG_FOR(total) for (long long e_ = threadIdx.x; e_ < (long long)(total); e_ += TTN_WG)
struct AlsSite { int n, rl, rr, Rl, Rr, bl, br; };
__device__ __noinline__ void als_update_G(const AlsEnv& E, int i) {
    double* T1 = E.T1; double* T2 = E.T2;
    const AlsSite s = SITE(i);
    const AlsSite s2 = SITE(i + 1);
    const double *x = XC(i), *A2 = AC(i + 1), *Gi = GP(i), *Gbi = GBP(i), *b2 = BC(i + 1);
    double *Go = GP(i + 1), *Gbo = GBP(i + 1);
    // T1[l, ph, be, L] = sum_{m, ch} Gi[l, ph, m, ch, L] x[m, ch, be]
    WG_FOR((long long)s.n * s.rl * s.rr * s.Rr) {
        long long t = e_; const int l = t % s.n; t /= s.n; const int ph = t % s.rl; t /= s.rl; const int be = t % s.rr; const int L = (int)(t / s.rr);
        double a = 0.0;
        for (int ch = 0; ch < s.rl; ++ch)
            for (int m = 0; m < s.n; ++m)
                a = fma(Gi[l + s.n * (ph + (long long)s.rl * (m + (long long)s.n * (ch + (long long)s.rl * L)))], x[m + s.n * (ch + (long long)s.rl * be)], a);
        T1[e_] = a;
    }
    __syncthreads();
    // T2[al, be, L] = sum_{l, ph} x[l, ph, al] T1[l, ph, be, L]
    WG_FOR((long long)s.rr * s.rr * s.Rr) {
        long long t = e_; const int al = t % s.rr; t /= s.rr; const int be = t % s.rr; const int L = (int)(t / s.rr);
        double a = 0.0;
        for (int ph = 0; ph < s.rl; ++ph)
            for (int l = 0; l < s.n; ++l)
                a = fma(x[l + s.n * (ph + (long long)s.rl * al)], T1[l + s.n * (ph + (long long)s.rl * (be + (long long)s.rr * L))], a);
        T2[e_] = a;
    }
    __syncthreads();
    // G_{i+1}[j, al, k, be, J] = sum_L T2[al, be, L] A2[j, k, L, J]
    WG_FOR((long long)s2.n * s.rr * s2.n * s.rr * s2.Rr) {
        long long t = e_; const int j = t % s2.n; t /= s2.n; const int al = t % s.rr; t /= s.rr; const int k = t % s2.n; t /= s2.n; const int be = t % s.rr; const int J = (int)(t / s.rr);
        double a = 0.0;
        for (int L = 0; L < s.Rr; ++L)
            a = fma(T2[al + s.rr * (be + (long long)s.rr * L)], A2[j + s2.n * (k + s2.n * (L + (long long)s2.Rl * J))], a);
        Go[e_] = a;
    }
    __syncthreads();
    // Gb: T1[al, ph] = sum_{j, ch} x[j, ch, al] Gb_i[j, ch, ph] ; Gb_{i+1}[i', al, be] = sum_ph b2[i', ph, be] T1[al, ph]
    WG_FOR((long long)s.rr * s.br) {
        const int al = (int)(e_ % s.rr), ph = (int)(e_ / s.rr);
        double a = 0.0;
        for (int ch = 0; ch < s.rl; ++ch)
            for (int j = 0; j < s.n; ++j)
                a = fma(x[j + s.n * (ch + (long long)s.rl * al)], Gbi[j + s.n * (ch + (long long)s.rl * ph)], a);
        T1[e_] = a;
    }
    __syncthreads();
    WG_FOR((long long)s2.n * s.rr * s2.br) {
        long long t = e_; const int ii = t % s2.n; t /= s2.n; const int al = t % s.rr; const int be = (int)(t / s.rr);
        double a = 0.0;
        for (int ph = 0; ph < s.br; ++ph) a = fma(b2[ii + s2.n * (ph + (long long)s2.bl * be)], T1[al + (long long)s.rr * ph], a);
        Gbo[e_] = a;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(TTN_WG) k_als_linsolve(AlsArgs P) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, b = blockIdx.x + P.train0;
    const int d = P.x.d;
    double* scr = P.scratch + (long long)blockIdx.x * P.scratch_stride;
    double* red = lds + GEMM_LDS_TOTAL;
    int* iflag = reinterpret_cast<int*>(red + 32 + 2 * QR_NB * QR_NB + QR_NB + 8);
    OrthoWork W;
    W.Vb = scr + P.offVb; W.Wb = scr + P.offWb; W.Tst = scr + P.offTst;
    W.red = red; W.Ts = red + 32; W.Ss = W.Ts + QR_NB * QR_NB; W.taus = W.Ss + QR_NB * QR_NB;
    double* K = scr + P.offK;
    double* Pb = scr + P.offPb;
    double* T1 = scr + P.offT1;
    double* T2 = scr + P.offT2;
    double* Tm = scr + P.offTm;
    double* Qb = scr + P.offQb;
    double* Rb = scr + P.offRb;
    // the ranks are fixed (als.jl:177): every train must carry exactly the handle's ranks
    if (P.phase <= 1) {
        const long long* xr = P.x.rks + (long long)b * (d + 1);
        bool bad = false;
        for (int k = 0; k <= d; ++k) bad |= (xr[k] != P.rfix[k]);
        if (bad) { if (tid == 0) ttn_set_status(&P.status[b], 4); return; }
    }
    const long long* br_ = P.b.rks + (long long)b * (d + 1);
    AlsEnv E;
    E.A = P.A; E.b = P.b; E.x = P.x; E.tb = b; E.scr = scr; E.off = P.off; E.xr = P.rfix; E.br = br_; E.T1 = T1; E.T2 = T2; E.d = d;
    // H_{i-1} from site i (als.jl:23-26) and Hb_{i-1} (als.jl:42-45)
    auto update_H = [&](int i) {
        const AlsSite s = SITE(i);
        const double *x = XC(i), *A = AC(i), *Hi = HP(i), *Hbi = HBP(i), *bb = BC(i);
        double *Ho = HP(i - 1), *Hbo = HBP(i - 1);
        // T1[z, ph, k, be] = sum_ch H[z, ph, ch] x[k, be, ch]
        WG_FOR((long long)s.Rr * s.rr * s.n * s.rl) {
            long long t = e_; const int z = t % s.Rr; t /= s.Rr; const int ph = t % s.rr; t /= s.rr; const int k = t % s.n; const int be = (int)(t / s.n);
            double a = 0.0;
            for (int ch = 0; ch < s.rr; ++ch) a = fma(Hi[z + s.Rr * (ph + (long long)s.rr * ch)], x[k + s.n * (be + (long long)s.rl * ch)], a);
            T1[e_] = a;
        }
        __syncthreads();
        // T2[j, ph, a, be] = sum_{z, k} A[j, k, a, z] T1[z, ph, k, be]
        WG_FOR((long long)s.n * s.rr * s.Rl * s.rl) {
            long long t = e_; const int j = t % s.n; t /= s.n; const int ph = t % s.rr; t /= s.rr; const int a_ = t % s.Rl; const int be = (int)(t / s.Rl);
            double a = 0.0;
            for (int z = 0; z < s.Rr; ++z)
                for (int k = 0; k < s.n; ++k)
                    a = fma(A[j + s.n * (k + s.n * (a_ + (long long)s.Rl * z))], T1[z + s.Rr * (ph + (long long)s.rr * (k + (long long)s.n * be))], a);
            T2[e_] = a;
        }
        __syncthreads();
        // H_{i-1}[a, al, be] = sum_{j, ph} x[j, al, ph] T2[j, ph, a, be]
        WG_FOR((long long)s.Rl * s.rl * s.rl) {
            long long t = e_; const int a_ = t % s.Rl; t /= s.Rl; const int al = t % s.rl; const int be = (int)(t / s.rl);
            double a = 0.0;
            for (int ph = 0; ph < s.rr; ++ph)
                for (int j = 0; j < s.n; ++j)
                    a = fma(x[j + s.n * (al + (long long)s.rl * ph)], T2[j + s.n * (ph + (long long)s.rr * (a_ + (long long)s.Rl * be))], a);
            Ho[e_] = a;
        }
        __syncthreads();
        // Hb: T1[ph, i, be] = sum_ch Hb[ph, ch] b[i, be, ch] ; Hb_{i-1}[al, be] = sum_{i, ph} x[i, al, ph] T1[ph, i, be]
        WG_FOR((long long)s.rr * s.n * s.bl) {
            long long t = e_; const int ph = t % s.rr; t /= s.rr; const int ii = t % s.n; const int be = (int)(t / s.n);
            double a = 0.0;
            for (int ch = 0; ch < s.br; ++ch) a = fma(Hbi[ph + (long long)s.rr * ch], bb[ii + s.n * (be + (long long)s.bl * ch)], a);
            T1[e_] = a;
        }
        __syncthreads();
        WG_FOR((long long)s.rl * s.bl) {
            const int al = (int)(e_ % s.rl), be = (int)(e_ / s.rl);
            double a = 0.0;
            for (int ph = 0; ph < s.rr; ++ph)
                for (int ii = 0; ii < s.n; ++ii)
                    a = fma(x[ii + s.n * (al + (long long)s.rl * ph)], T1[ph + s.rr * (ii + (long long)s.n * be)], a);
            Hbo[e_] = a;
        }
        __syncthreads();
    };
    // V = K \ Pb for site i (als.jl:58-70); V lands in Pb as (n, rl, rr) column-major.  Returns false on a singular system.
    auto ksolve = [&](int i) -> bool {
        const AlsSite s = SITE(i);
        const double *Gi = GP(i), *Gbi = GBP(i), *Hi = HP(i), *Hbi = HBP(i);
        const int nr = s.n * s.rl, N = nr * s.rr;
        WG_FOR((long long)N * N) {
            const int row = (int)(e_ % N), col = (int)(e_ / N);
            const int ab = row % nr, c = row / nr, de = col % nr, f = col / nr;
            double a = 0.0;
            for (int z = 0; z < s.Rr; ++z) a = fma(Gi[ab + (long long)nr * (de + (long long)nr * z)], Hi[z + s.Rr * (c + (long long)s.rr * f)], a);
            K[e_] = a;
        }
        WG_FOR(N) {
            const int ia = (int)(e_ % nr), a2 = (int)(e_ / nr);
            double a = 0.0;
            for (int be = 0; be < s.br; ++be) a = fma(Gbi[ia + (long long)nr * be], Hbi[a2 + (long long)s.rr * be], a);
            Pb[e_] = a;
        }
        __syncthreads();
        return wg_lu_solve(N, K, Pb, reinterpret_cast<int*>(scr + P.offPiv), red, iflag, lds) == 0;
    };

    // the core move and environment update that follow the local solve of site i (V in Pb): forward half sweep (als.jl:122-135) ...
    auto fwd_move = [&](int i) {
        const AlsSite s = SITE(i);
        const int mm = s.n * s.rl;
        WG_FOR((long long)mm * s.rr) Tm[e_] = Pb[e_];
        __syncthreads();
        wg_qr_explicit(mm, s.rr, Tm, Qb, Rb, W, lds);                           // right_core_move (als.jl:122-135)
        double* xi = XC(i);
        WG_FOR((long long)mm * s.rr) xi[e_] = Qb[e_];
        // x_{i+1}[a, b, c] = sum_z R[b, z] x_{i+1}[a, z, c]
        const AlsSite s2 = SITE(i + 1);
        double* xn = XC(i + 1);
        WG_FOR((long long)s2.n * s2.rl * s2.rr) {
            long long t = e_; const int a_ = t % s2.n; t /= s2.n; const int bq = t % s2.rl; const int c = (int)(t / s2.rl);
            double a = 0.0;
            for (int z = 0; z < s2.rl; ++z) a = fma(Rb[bq + (long long)s.rr * z], xn[a_ + s2.n * (z + (long long)s2.rl * c)], a);
            T1[e_] = a;
        }
        __syncthreads();
        WG_FOR((long long)s2.n * s2.rl * s2.rr) xn[e_] = T1[e_];
        __syncthreads();
        als_update_G(E, i);
    };
    // ... and backward half sweep (als.jl:102-120)
    auto bwd_move = [&](int i) {
        const AlsSite s = SITE(i);
        const int mm = s.n * s.rr;
        // M[(x + n*a2), a1] = V[x, a1, a2]                                      left_core_move (als.jl:102-120)
        WG_FOR((long long)mm * s.rl) {
            const int row = (int)(e_ % mm), a1 = (int)(e_ / mm);
            const int xx = row % s.n, a2 = row / s.n;
            Tm[e_] = Pb[xx + s.n * (a1 + (long long)s.rl * a2)];
        }
        __syncthreads();
        wg_qr_explicit(mm, s.rl, Tm, Qb, Rb, W, lds);
        double* xi = XC(i);
        WG_FOR((long long)mm * s.rl) {
            const int row = (int)(e_ % mm), a1 = (int)(e_ / mm);
            const int xx = row % s.n, a2 = row / s.n;
            xi[xx + s.n * (a1 + (long long)s.rl * a2)] = Qb[e_];
        }
        // x_{i-1}[a, b, c] = sum_z x_{i-1}[a, b, z] R[c, z]
        const AlsSite s0 = SITE(i - 1);
        double* xp = XC(i - 1);
        WG_FOR((long long)s0.n * s0.rl * s0.rr) {
            const long long ab = e_ % ((long long)s0.n * s0.rl);
            const int c = (int)(e_ / ((long long)s0.n * s0.rl));
            double a = 0.0;
            for (int z = 0; z < s0.rr; ++z) a = fma(xp[ab + (long long)s0.n * s0.rl * z], Rb[c + (long long)s.rl * z], a);
            T1[e_] = a;
        }
        __syncthreads();
        WG_FOR((long long)s0.n * s0.rl * s0.rr) xp[e_] = T1[e_];
        __syncthreads();
        update_H(i);
    };

    if (P.phase == 2) { fwd_move(P.site); return; }
    if (P.phase == 3) { bwd_move(P.site); return; }
    // ---- initial environments (als.jl:183-193): G_1, Gb_1 from the first cores, H / Hb from the right ----
    {
        const AlsSite s = SITE(0);
        WG_FOR((long long)s.n * s.n * s.Rr) GP(0)[e_] = AC(0)[e_];                 // A_1[:, :, 1, :] as (n, 1, n, 1, R_1)
        WG_FOR((long long)s.n * s.br) GBP(0)[e_] = BC(0)[e_];
        if (tid == 0) { HP(d - 1)[0] = 1.0; HBP(d - 1)[0] = 1.0; }
        __syncthreads();
    }
    for (int i = d - 1; i >= 1; --i) update_H(i);
    if (P.phase == 1) return;
    bool ok = true;
    int nsweeps = 0;
    while (nsweeps < P.sweep_count && ok) {
        ++nsweeps;
        for (int i = 0; i < d - 1 && ok; ++i) {                                     // first half sweep (als.jl:199-207)
            ok = ksolve(i);
            if (!ok) break;
            fwd_move(i);
        }
        if (nsweeps == P.sweep_count || !ok) break;
        ++nsweeps;
        for (int i = d - 1; i >= 1 && ok; --i) {                                    // second half sweep (als.jl:213-219)
            ok = ksolve(i);
            if (!ok) break;
            bwd_move(i);
        }
    }
    if (!ok && tid == 0) ttn_set_status(&P.status[b], 3);
}

// -------------------------------------------------------------------------------------------------
// mals_linsolve (src/solvers/mals.jl:240-312): one forward and one backward half sweep of TWO-site solves; the ranks adapt
// through the truncated SVD of every local solution (sv_trunc, :42-56, clamped to rmax).  Same machinery as als_linsolve:
// dense K[(a,b,c,d),(e,f,g,h)] = sum_z G_i[a,b,e,f,z] H_i[z,c,d,g,h] (:148-157) solved by the blocked LU (the reference's
// Hermitian(K) \ b reads one triangle; K is symmetric to rounding), the split by the Householder-LQ + Jacobi SVD of the bond
// step (wg_hsvd_step, layouts 2 / 1).  Ranks change per train, so the environments are stored compactly with the CURRENT
// ranks inside slots sized by the handle's capacity.
//   H_i  (R_{i+1}, n_{i+1}, r_{i+2}, n_{i+1}, r_{i+2})   couples sites i, i+1          (:10-40)
//   Hb_i (rb_{i+1}, n_{i+1}, r_{i+2})                                                  (:60-92)
// -------------------------------------------------------------------------------------------------
#define TTN_DMRG_MAX_SWEEPS 32
struct MalsArgs {
    AlsArgs L;                   // operator, handles, scratch offsets (off[2d..], off[3d..] = the MALS H / Hb slots)
    CompressArgs C;              // Jacobi knobs, status, sweep statistics for wg_hsvd_step (C.scratch unused)
    double tol;
    int rmax;
    long long offM2, offXg, offUs, offSig;     // SVD scratch
    int pmax, qmax;
    // mode 0: mals_linsolve (one forward and one backward half sweep over all d-1 windows, rank rule sv_trunc).
    // mode 1: dmrg_linsolve with N = 2 (src/solvers/dmrg.jl:421-472): `nsweeps` sweeps (windows 0..d-3 forward, d-2..1 backward,
    //         sweep s capped at rmax_sweep[s]) and the closing solve at window 0 with a left move capped at rmax_final;
    //         rank rule cut_off_index (dmrg.jl:179-185).
    int mode, nsweeps, rmax_final;
    int rmax_sweep[TTN_DMRG_MAX_SWEEPS];
    // local solver (dmrg.jl:92-97): conjugate gradients (wg_cg_two_site) if cg_all or the system has more than cg_above unknowns,
    // dense LU otherwise.  offCg: per-train work area (4 vectors of Nmax doubles, then the slab Rzmax * Nmax); cg_iters: [batch]
    // total CG iterations (diagnostics) or null.
    int cg_all, cg_above, cg_maxiter;
    double cg_tol;
    long long offCg, cg_nmax;
    int* cg_iters;
};

__global__ void __launch_bounds__(TTN_WG) k_mals_linsolve(MalsArgs Q) {
    extern __shared__ double lds[];
    const AlsArgs& P = Q.L;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int d = P.x.d;
    double* scr = P.scratch + (long long)b * P.scratch_stride;
    double* red = lds + GEMM_LDS_TOTAL;
    BondCtx S;
    S.ldsX = lds;
    S.red = red;
    S.Ts = S.red + 32;
    S.Ss = S.Ts + QR_NB * QR_NB;
    S.taus = S.Ss + QR_NB * QR_NB;
    S.scal = S.taus + QR_NB;
    S.iflag = reinterpret_cast<int*>(S.scal + 8);
    S.nrm2 = S.scal + 16;
    S.M = nullptr; S.M2 = nullptr;
    S.Vb = scr + P.offVb; S.Wb = scr + P.offWb;
    S.Us = scr + Q.offUs; S.Xg = scr + Q.offXg;
    S.sig = scr + Q.offSig; S.sigs = S.sig + Q.pmax; S.perm = reinterpret_cast<int*>(S.sigs + Q.pmax);
    S.Ga = S.Gb = S.Cc = S.T1 = S.T2 = S.T3 = nullptr;
    double* K = scr + P.offK;
    double* Pb = scr + P.offPb;
    double* T1 = scr + P.offT1;
    double* T2 = scr + P.offT2;
    double* M2 = scr + Q.offM2;
    int* piv = reinterpret_cast<int*>(scr + P.offPiv);
    if (tid == 0) Q.C.sweep_stats[b] = 0;
    long long* xr = P.x.rks + (long long)b * (d + 1);
    const long long* br_ = P.b.rks + (long long)b * (d + 1);
    AlsEnv E;
    E.A = P.A; E.b = P.b; E.x = P.x; E.tb = b; E.scr = scr; E.off = P.off; E.xr = xr; E.br = br_; E.T1 = T1; E.T2 = T2; E.d = d;
    __syncthreads();

    // H_{i-1}, Hb_{i-1} from core i+1 of x, A_i, b_i and H_i, Hb_i   (mals.jl:10-13, :60-66)
    auto update_H = [&](int i) {
        const int n1 = uni32(P.x.dims[i]), n2 = uni32(P.x.dims[i + 1]);
        const int r1 = uni32((int)xr[i + 1]), r2 = uni32((int)xr[i + 2]);
        const int Ra = uni32((int)P.A.rks[i]), Rz = uni32((int)P.A.rks[i + 1]);
        const int ba = uni32((int)br_[i]), bz = uni32((int)br_[i + 1]);
        const double *x = XC(i + 1), *A = AC(i), *Hi = HP(i), *Hbi = HBP(i), *bb = BC(i);
        double *Ho = HP(i - 1), *Hbo = HBP(i - 1);
        // T1[z, j, xx, be] = sum_{k, y} Hi[z, j, xx, k, y] x[k, be, y]
        WG_FOR((long long)Rz * n2 * r2 * r1) {
            long long t = e_; const int z = t % Rz; t /= Rz; const int j = t % n2; t /= n2; const int xx = t % r2; const int be = (int)(t / r2);
            double a = 0.0;
            for (int y = 0; y < r2; ++y)
                for (int k = 0; k < n2; ++k)
                    a = fma(Hi[z + Rz * (j + n2 * (xx + (long long)r2 * (k + (long long)n2 * y)))], x[k + n2 * (be + (long long)r1 * y)], a);
            T1[e_] = a;
        }
        __syncthreads();
        // T2[z, al, be] = sum_{j, xx} x[j, al, xx] T1[z, j, xx, be]
        WG_FOR((long long)Rz * r1 * r1) {
            long long t = e_; const int z = t % Rz; t /= Rz; const int al = t % r1; const int be = (int)(t / r1);
            double a = 0.0;
            for (int xx = 0; xx < r2; ++xx)
                for (int j = 0; j < n2; ++j)
                    a = fma(x[j + n2 * (al + (long long)r1 * xx)], T1[z + Rz * (j + n2 * (xx + (long long)r2 * be))], a);
            T2[e_] = a;
        }
        __syncthreads();
        // H_{i-1}[a, ii, al, l, be] = sum_z T2[z, al, be] A[ii, l, a, z]
        WG_FOR((long long)Ra * n1 * r1 * n1 * r1) {
            long long t = e_; const int a_ = t % Ra; t /= Ra; const int ii = t % n1; t /= n1; const int al = t % r1; t /= r1; const int l = t % n1; const int be = (int)(t / n1);
            double a = 0.0;
            for (int z = 0; z < Rz; ++z) a = fma(T2[z + Rz * (al + (long long)r1 * be)], A[ii + n1 * (l + n1 * (a_ + (long long)Ra * z))], a);
            Ho[e_] = a;
        }
        __syncthreads();
        // Hb: T1[ga, ch] = sum_{j, a} x[j, ch, a] Hbi[ga, j, a] ; Hb_{i-1}[be, ii, ch] = sum_ga b[ii, be, ga] T1[ga, ch]
        WG_FOR((long long)bz * r1) {
            const int ga = (int)(e_ % bz), ch = (int)(e_ / bz);
            double a = 0.0;
            for (int aa = 0; aa < r2; ++aa)
                for (int j = 0; j < n2; ++j)
                    a = fma(x[j + n2 * (ch + (long long)r1 * aa)], Hbi[ga + bz * (j + (long long)n2 * aa)], a);
            T1[e_] = a;
        }
        __syncthreads();
        WG_FOR((long long)ba * n1 * r1) {
            long long t = e_; const int be = t % ba; t /= ba; const int ii = t % n1; const int ch = (int)(t / n1);
            double a = 0.0;
            for (int ga = 0; ga < bz; ++ga) a = fma(bb[ii + n1 * (be + (long long)ba * ga)], T1[ga + (long long)bz * ch], a);
            Hbo[e_] = a;
        }
        __syncthreads();
    };
    // two-site solve at sites i, i+1: V (n1, r_i, n2, r_{i+2}) column-major in Pb
    auto ksolve = [&](int i, int& a_out, int& b_out, bool v0_swapped) -> bool {
        const int n1 = uni32(P.x.dims[i]), n2 = uni32(P.x.dims[i + 1]);
        const int rl = uni32((int)xr[i]), rr = uni32((int)xr[i + 2]);
        const int Rz = uni32((int)P.A.rks[i + 1]), bz = uni32((int)br_[i + 1]);
        const double *Gi = GP(i), *Gbi = GBP(i), *Hi = HP(i), *Hbi = HBP(i);
        const int na = n1 * rl, nb = n2 * rr, N = na * nb;
        a_out = na; b_out = nb;
        if (Q.cg_all || N > Q.cg_above) {
            // matrix-free: right-hand side, the start vector V0 = the current two-site block x_i x_{i+1} (what update_right /
            // update_left hand to the next solve, dmrg.jl:311-316), CG, solution into Pb
            WG_FOR(N) {
                const int ab = (int)(e_ % na), cd = (int)(e_ / na);
                double a = 0.0;
                for (int z = 0; z < bz; ++z) a = fma(Gbi[ab + (long long)na * z], Hbi[z + (long long)bz * cd], a);
                Pb[e_] = a;
            }
            double* cg = scr + Q.offCg;
            double *xv = cg, *rv = cg + Q.cg_nmax, *pv = cg + 2 * Q.cg_nmax, *qv = cg + 3 * Q.cg_nmax, *Wv = cg + 4 * Q.cg_nmax;
            const int rm = uni32((int)xr[i + 1]);
            if (v0_swapped && n1 == n2) {
                // after a LEFT move the reference builds the start vector with the two physical indices exchanged (update_left reshapes
                // [alpha, J, i_k, gamma] with the index of site i+1 running faster than that of site i, dmrg.jl:331-334): restated as
                // it is — V0[(j, al), (k, be)] = sum_ga x_i[k, al, ga] x_{i+1}[j, ga, be], one rl x rr x rm product per (j, k)
                for (int j = 0; j < n1; ++j)
                    for (int k = 0; k < n2; ++k)
                        wg_gemm(rl, rr, rm, mkview(XC(i) + k, plain(n1), plain((long long)n1 * rl)), mkview(XC(i + 1) + j, plain(n2), plain((long long)n2 * rm)),
                                mkview(xv + j + (long long)na * k, plain(n1), plain((long long)na * n2)), 1.0, 0.0, lds);
            } else {
                // V0[(j, al), (k, be)] = sum_ga x_i[j, al, ga] x_{i+1}[k, ga, be]: the current two-site block (b_mid / update_right)
                wg_gemm(na, nb, rm, mkview(XC(i), plain(1), plain(na)), mkview(XC(i + 1), plain(n2), Idx{n2, 1, (long long)n2 * rm}),
                        mkview(xv, plain(1), plain(na)), 1.0, 0.0, lds);
            }
            const int iters = wg_cg_two_site(na, nb, Rz, const_cast<double*>(Gi), const_cast<double*>(Hi), xv, Pb, rv, pv, qv, Wv, Q.cg_tol, Q.cg_maxiter, red, lds);
            WG_FOR(N) Pb[e_] = xv[e_];
            if (Q.cg_iters && tid == 0) Q.cg_iters[b] += iters;
            __syncthreads();
            return true;
        }
        WG_FOR((long long)N * N) {
            const int row = (int)(e_ % N), col = (int)(e_ / N);
            const int ab = row % na, cd = row / na, ef = col % na, gh = col / na;
            double a = 0.0;
            for (int z = 0; z < Rz; ++z) a = fma(Gi[ab + (long long)na * (ef + (long long)na * z)], Hi[z + Rz * (cd + (long long)nb * gh)], a);
            K[e_] = a;
        }
        WG_FOR(N) {
            const int ab = (int)(e_ % na), cd = (int)(e_ / na);
            double a = 0.0;
            for (int z = 0; z < bz; ++z) a = fma(Gbi[ab + (long long)na * z], Hbi[z + (long long)bz * cd], a);
            Pb[e_] = a;
        }
        __syncthreads();
        return wg_lu_solve(N, K, Pb, piv, red, S.iflag, lds) == 0;
    };

    // ---- initial environments (mals.jl:255-265) ----
    {
        const AlsSite s = SITE(0);
        WG_FOR((long long)s.n * s.n * s.Rr) GP(0)[e_] = AC(0)[e_];
        WG_FOR((long long)s.n * s.br) GBP(0)[e_] = BC(0)[e_];
        // H_{d-2}[z, j, 1, k, 1] = A_d[j, k, z, 1] ; Hb_{d-2}[ga, j, 1] = b_d[j, ga, 1]
        const int n = uni32(P.x.dims[d - 1]), Rz = uni32((int)P.A.rks[d - 1]), bz = uni32((int)br_[d - 1]);
        WG_FOR((long long)Rz * n * n) {
            long long t = e_; const int z = t % Rz; t /= Rz; const int j = t % n; const int k = (int)(t / n);
            HP(d - 2)[e_] = AC(d - 1)[j + n * (k + (long long)n * z)];
        }
        WG_FOR((long long)bz * n) { const int ga = (int)(e_ % bz), j = (int)(e_ / bz); HBP(d - 2)[e_] = BC(d - 1)[j + (long long)n * ga]; }
        __syncthreads();
    }
    for (int i = d - 2; i >= 1; --i) update_H(i);
    int status = 0;
    const int mode = uni32(Q.mode);
    const int per = mode == 0 ? 2 * (d - 1) : 2 * (d - 2);                 // windows visited by one sweep
    const int total = mode == 0 ? per : uni32(Q.nsweeps) * per + 1;
    const int rule = mode == 0 ? 1 : 2;
    int prev_dir = 0;                                                      // direction of the move before this solve (start: none)
    for (int t = 0; t < total && !status; ++t) {
        int i, dir, rmax;
        if (mode == 0) { dir = t >= d - 1; i = dir ? 2 * (d - 1) - 1 - t : t; rmax = Q.rmax; }
        else if (t == total - 1) { i = 0; dir = 1; rmax = Q.rmax_final; }
        else { const int u = t % per; dir = u >= d - 2; i = dir ? 2 * (d - 2) - u : u; rmax = Q.rmax_sweep[t / per]; }
        i = uni32(i); dir = uni32(dir); rmax = uni32(rmax);
        int na, nb;
        if (!ksolve(i, na, nb, prev_dir == 1)) { status = 3; break; }
        prev_dir = dir;
        const int n2 = uni32(P.x.dims[i + 1]);
        double* xi = XC(i);
        double* xn = XC(i + 1);
        int r;
        if (dir == 0)        // right_core_move_mals / right_core_move!: x_i <- U, x_{i+1} <- S V'
            r = wg_hsvd_step(Q.C, b, S, mkview(Pb, plain(1), plain(na)), na, nb, M2, 2, n2, 0, xi, xn, Q.tol, (int)P.x.cap[i + 1], lds, rule, rmax);
        else                 // left_core_move_mals / left_core_move!: x_{i+1} <- V', x_i <- U S  (the step on the transposed view)
            r = wg_hsvd_step(Q.C, b, S, mkview(Pb, plain(na), plain(1)), nb, na, M2, 1, n2, 0, xn, xi, Q.tol, (int)P.x.cap[i + 1], lds, rule, rmax);
        if (r < 0) { status = 2; break; }
        if (tid == 0) xr[i + 1] = r;
        __syncthreads();
        if (dir == 0) als_update_G(E, i);
        else if (i > 0) update_H(i);
    }
    if (status && tid == 0) ttn_set_status(&P.status[b], status);
}
#undef XC
#undef BC
#undef AC
#undef GP
#undef GBP
#undef HP
#undef HBP
#undef SITE
#undef WG_FOR
