// ttn_hsvd_kernels.h — ttv_decomp (src/tt_tools.jl:186-252): TT decomposition of a dense tensor by successive SVDs
// (hierarchical SVD), one workgroup per tensor of a batch.  Every step is the route-H SVD of the bond step
// (Householder LQ of the short side, one-sided Jacobi on L) applied to an unfolding of what is left of the tensor:
//   A (a x b) = W S Z^T ;  core <- W[:, 1:r] ;  remainder <- S Z^T (r x b) ;  r = count(s >= tol)   (absolute threshold, :203,224)
// The reference's reshapes are index arithmetic: the unfoldings, the cores and the remainder are all addressed through Views, and
// the right-to-left half of the algorithm (sites d .. index+1) is the same step on the transposed View.
#pragma once
#include "ttn_dense_kernels.h"

struct HsvdArgs {
    CompressArgs C;              // scratch / status / Jacobi knobs (C.tt = the output handle)
    const double* tensors;       // device [batch][total], column-major like the reference's Array
    long long total;
    int index;                   // 0-based root site
    double tol;
    double* work;                // per train: cur0 | cur1 | M2, `total` doubles each, then the BondCtx-style scratch (C.scratch)
    long long work_stride;
};

// A = W S Z^T for the a x b matrix behind `Av`; outputs as described above.  `left` selects the output layouts:
//   left  (sites i < index): A[(al + rl*x), col]; core[x, al, j] = W[al + rl*x, j]; remainder (r x b) column-major, ld = r
//   right (sites i > index): A[(x + n*be), row] (the transposed unfolding); core[x, j, be] = W[x + n*be, j];
//                            remainder[j, row] stored as the column-major (rows x r) matrix U S of the reference
// `layout` 2 / 3 are the two-site core moves of mals_linsolve (src/solvers/mals.jl:94-146) on V = (n1 r_l) x (n2 r_r):
//   2 (right move): core = x_i (n1, r_l, r) <- U, contiguous; remainder S V' -> x_{i+1}[x, j, c] = (S V')[j, x + n c]   (n = n2)
//   3 (left move, called on the TRANSPOSED view): core = x_{i+1}[x, j, c] <- V'[j, x + n c]; remainder (U S) -> x_i, contiguous
// `rule` 0: r = count(s >= tol) (absolute); 1: sv_trunc (mals.jl:42-56: drop the tail while its weight stays below
// tol * ||s||^2, keep the value that crossed the line) clamped to `rclamp`; 2: cut_off_index (dmrg.jl:179-185: count(s > ||s|| tol),
// extended over values within 1e-10 (relative and absolute) of the last kept one) clamped to `rclamp`.
// Returns r (>= 1), or -1 if r exceeds `cap` (nothing written).
__device__ __noinline__ int wg_hsvd_step(const CompressArgs& P, int b, const BondCtx& S, View Av, int a, int bcols, double* M2,
                                         int layout, int n, int rfix, double* core, double* rem, double tol, int cap, double* lds,
                                         int rule = 0, int rclamp = 1 << 30) {
    const bool left = layout == 0;
    a = uni32(a); bcols = uni32(bcols); n = uni32(n); rfix = uni32(rfix); cap = uni32(cap);
    Av = uniView(Av); M2 = unip(M2); core = unip(core); rem = unip(rem); lds = unip(lds);
    const int tid = threadIdx.x;
    const bool tall = a > bcols;                           // then the LQ works on A^T
    const int p = tall ? bcols : a, q = tall ? a : bcols;
    const View Mv = tall ? tview(Av) : Av;                 // p x q, p <= q
    // scale
    double mx = 0.0;
    for (long long e = tid; e < (long long)p * q; e += TTN_WG) {
        const int i = (int)(e / q), j = (int)(e % q);
        mx = fmax(mx, fabs(Mv.p[ix(Mv.r, i) + ix(Mv.c, j)]));
    }
    mx = unif64(wg_max(mx, S.red));
    const double s0 = (mx > 0.0) ? mx : 1.0, inv_s0 = 1.0 / s0;
    for (long long e = tid; e < (long long)p * q; e += TTN_WG) {
        const int i = (int)(e / q), j = (int)(e % q);
        M2[e] = Mv.p[ix(Mv.r, i) + ix(Mv.c, j)] * inv_s0;
    }
    __syncthreads();
    const bool need_lq = q > p;
    if (need_lq) wg_lq_blocked(p, q, M2, q, S.Vb, S.Wb, nullptr, nullptr, lds, S.Ts, S.Ss, S.taus, S.red);
    const bool x_in_lds = p <= 128;
    double* X = x_in_lds ? S.ldsX : S.Xg;
    const int ldx = x_in_lds ? 128 : p;
    for (long long e = tid; e < (long long)p * ldx; e += TTN_WG) {
        const int r_ = (int)(e % ldx), c = (int)(e / ldx);
        const double v = (r_ < p) ? M2[(long long)r_ * q + c] : 0.0;
        X[(long long)c * ldx + r_] = (need_lq && c > r_) ? 0.0 : v;
    }
    __syncthreads();
    const int nsw = uni32(wg_svd_cols(P, S, p, X, ldx, x_in_lds));
    if (tid == 0) { P.sweep_stats[b] += (nsw < 0 ? -nsw : nsw); if (nsw < 0) ttn_set_status(&P.status[b], 1); }
    // rank: count(s >= tol), absolute (src/tt_tools.jl:203, :224); an all-zero remainder keeps one (zero) direction
    if (tid == 0) {
        int r = 0;
        if (rule == 0) { for (int i = 0; i < p; ++i) r += (S.sigs[i] * s0 >= tol) ? 1 : 0; }
        else if (rule == 2) {
            double norm2 = 0.0;
            for (int i = 0; i < p; ++i) { const double sv = S.sigs[i] * s0; norm2 = fma(sv, sv, norm2); }
            const double thr = sqrt(norm2) * tol;
            for (int i = 0; i < p; ++i) r += (S.sigs[i] * s0 > thr) ? 1 : 0;
            while (r > 0 && r < p) {                    // isapprox(s[k], s[k+1]; rtol = atol = 1e-10)
                const double u = S.sigs[r - 1] * s0, v = S.sigs[r] * s0;
                if (fabs(u - v) <= fmax(1.0e-10, 1.0e-10 * fmax(fabs(u), fabs(v)))) ++r; else break;
            }
            if (r > rclamp) r = rclamp;
        } else {
            r = p;
            if (tol != 0.0) {
                double norm2 = 0.0, weight = 0.0;
                for (int i = 0; i < p; ++i) { const double sv = S.sigs[i] * s0; norm2 = fma(sv, sv, norm2); }
                int i = 0;
                while (i < p && weight < tol * norm2) { const double sv = S.sigs[p - i - 1] * s0; weight = fma(sv, sv, weight); ++i; }
                r = p - i + 1;
                if (r > p) r = p;
            }
            if (r > rclamp) r = rclamp;
        }
        S.iflag[1] = r < 1 ? 1 : r;
    }
    __syncthreads();
    const int r = uni32(S.iflag[1]);
    __syncthreads();
    if (r > cap) return -1;
    const double aneg = S.scal[0];
    // short-side vectors (x_j = sigma_j * vector, scaled units); Us[j*p + row] holds the multiplier form for the GEMM
    for (int e = tid; e < p * r; e += TTN_WG) {
        const int row = e % p, j = e / p;
        const double sj = S.sigs[j], xv = X[(long long)S.perm[j] * ldx + row];
        const bool keep = (sj > 0.0) && (sj * sj > aneg);
        S.Us[(long long)j * p + row] = keep ? (tall ? xv / (sj * sj) : xv / sj) : 0.0;
    }
    __syncthreads();
    // output views
    const View coreV = left ? mkview(core, Idx{rfix, (long long)n, 1}, plain((long long)n * rfix))          // (a x r): row al + rl*x
                     : (layout == 2) ? mkview(core, plain(1), plain(a))                                     // (a x r) contiguous
                                     : mkview(core, Idx{n, 1, (long long)n * r}, plain(n));                 // (a x r): row x + n*be
    const View remV = left ? mkview(rem, plain(1), plain(r))                                                // (r x b) column-major
                    : (layout == 2) ? mkview(rem, plain(n), Idx{n, 1, (long long)n * r})                    // x_{i+1}[x, j, c]
                                    : mkview(rem, plain(bcols), plain(1));                                  // (r x rows): U S column-major
    if (!tall) {
        // W = short-side vectors; remainder = W^T A
        for (int e = tid; e < p * r; e += TTN_WG) {
            const int row = e % p, j = e / p;
            coreV.p[ix(coreV.r, row) + ix(coreV.c, j)] = S.Us[(long long)j * p + row];
        }
        wg_gemm(r, bcols, a, mkview(S.Us, plain(p), plain(1)), Av, remV, 1.0, 0.0, lds);
    } else {
        // the Jacobi vectors are the Z side: remainder[j, :] = s0 * x_j^T ; W = (A / s0) x_j / sigma_j^2
        for (int e = tid; e < p * r; e += TTN_WG) {
            const int col = e % p, j = e / p;
            const double sj = S.sigs[j], xv = X[(long long)S.perm[j] * ldx + col];
            const bool keep = (sj > 0.0) && (sj * sj > aneg);
            remV.p[ix(remV.r, j) + ix(remV.c, col)] = keep ? xv * s0 : 0.0;
        }
        wg_gemm(a, r, bcols, Av, mkview(S.Us, plain(1), plain(p)), coreV, inv_s0, 0.0, lds);
    }
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(TTN_WG) k_ttv_decomp(HsvdArgs H) {
    extern __shared__ double lds[];
    const CompressArgs& P = H.C;
    const TTDev& T = P.tt;
    const int b = blockIdx.x, tid = threadIdx.x, d = T.d;
    if (tid == 0) P.sweep_stats[b] = 0;
    BondCtx S;
    S.ldsX = lds;
    S.red = lds + GEMM_LDS_TOTAL;
    S.Ts = S.red + 32;
    S.Ss = S.Ts + QR_NB * QR_NB;
    S.taus = S.Ss + QR_NB * QR_NB;
    S.scal = S.taus + QR_NB;
    S.iflag = reinterpret_cast<int*>(S.scal + 8);
    S.nrm2 = S.scal + 16;
    double* scr = P.scratch + (long long)b * P.scratch_stride;
    S.M = nullptr; S.M2 = nullptr;
    S.Vb = scr;                                           // QR_NB x qmax
    S.Wb = S.Vb + (long long)QR_NB * P.qmax;              // pmax x QR_NB
    S.Us = S.Wb + (long long)P.pmax * QR_NB;              // pmax x pmax
    S.Xg = S.Us + (long long)P.pmax * P.pmax;             // pmax x pmax
    S.sig = S.Xg + (long long)P.pmax * P.pmax;
    S.sigs = S.sig + P.pmax;
    S.perm = reinterpret_cast<int*>(S.sigs + P.pmax);
    S.Ga = S.Gb = S.Cc = S.T1 = S.T2 = S.T3 = nullptr;
    double* cur = H.work + (long long)b * H.work_stride;
    double* nxt = cur + H.total;
    double* M2 = nxt + H.total;
    const double* src = H.tensors + (long long)b * H.total;
    for (long long e = tid; e < H.total; e += TTN_WG) cur[e] = src[e];
    long long* rks = T.rks + (long long)b * (d + 1);
    if (tid == 0) { rks[0] = 1; rks[d] = 1; }
    __syncthreads();
    long long len = H.total;                               // doubles in `cur`
    bool alive = true;
    int rleft = 1;
    // ---- sites left of the root: unfold (r_i n_i) x rest, U -> core i, S V' -> remainder   (src/tt_tools.jl:197-211) ----
    for (int i = 0; i < H.index && alive; ++i) {
        const int n = T.dims[i];
        const int a = rleft * n;
        const int bc = (int)(len / a);
        double* core = T.data + (long long)b * T.stride + T.off[i];
        const int r = wg_hsvd_step(P, b, S, mkview(cur, plain(1), plain(a)), a, bc, M2, 0, n, rleft, core, nxt, H.tol, (int)T.cap[i + 1], lds);
        if (r < 0) { alive = false; break; }
        if (tid == 0) rks[i + 1] = r;
        rleft = r;
        len = (long long)r * bc;
        double* t = cur; cur = nxt; nxt = t;
        __syncthreads();
    }
    // ---- sites right of the root, from the last one: unfold rest x (n_i r_{i+1}), V' -> core i, U S -> remainder (:214-233) ----
    int rright = 1;
    for (int i = d - 1; i > H.index && alive; --i) {
        const int n = T.dims[i];
        const int a = n * rright;                          // short-ish side: (x + n*be)
        const int rows = (int)(len / a);
        double* core = T.data + (long long)b * T.stride + T.off[i];
        // the reference's unfolding is (rows x a) column-major = this a x rows matrix row-major
        const int r = wg_hsvd_step(P, b, S, mkview(cur, plain(rows), plain(1)), a, rows, M2, 1, n, rright, core, nxt, H.tol, (int)T.cap[i], lds);
        if (r < 0) { alive = false; break; }
        if (tid == 0) rks[i] = r;
        rright = r;
        len = (long long)rows * r;
        double* t = cur; cur = nxt; nxt = t;
        __syncthreads();
    }
    // ---- root: what is left IS the core, (rleft * n) x rright column-major -> core[x, al, be]   (:236-245) ----
    if (alive) {
        const int n = T.dims[H.index];
        double* core = T.data + (long long)b * T.stride + T.off[H.index];
        for (long long e = tid; e < (long long)n * rleft * rright; e += TTN_WG) {
            const int x = (int)(e % n);
            const long long t2 = e / n;
            const int al = (int)(t2 % rleft), be = (int)(t2 / rleft);
            core[e] = cur[(al + (long long)rleft * x) + (long long)rleft * n * be];
        }
    } else if (tid == 0) ttn_set_status(&P.status[b], 2);
}
