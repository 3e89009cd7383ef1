// ttn_als_grid.h — als_linsolve's local systems beyond one workgroup (src/solvers/als.jl:58-70: dense K_full, then `K \ Pb`).
//
// BASELINE config C5 names ranks up to 128: the one-site system then has n r^2 = 32 768 unknowns and K is an 8.6 GB matrix; the
// one-workgroup LU of ttn_als_kernels.h stops at 2048 unknowns (rank 32).  Here the same algorithm — dense assembly, right-looking
// blocked LU with partial pivoting (first maximal |entry| of the column, like LAPACK's idamax), panels of 32 columns — runs on the
// whole chip, one kernel per stage, driven panel by panel from the host (ttn_api.hip: als_grid_path):
//   k_als_assemble   K[(ab, c), (de, f)] = sum_z G[ab, de, z] H[z, c, f] and Pb = Gb Hb^T            grid over tiles of K
//   k_lu_panel       the panel, column by column (one workgroup; in LDS while mrows * 32 <= 16384)     1 workgroup
//   k_lu_rows        the panel's interchanges on every other column and the right-hand side, then
//                    U12 = L11^-1 A12 (one column per thread, the column segment in registers)        grid over columns
//   k_lu_trail       A22 -= L21 U12 (fp64 MFMA, one 128 x 128 tile per workgroup) and the rhs           grid over tiles
//   k_lu_back_tri / k_lu_back_rows   back substitution in blocks of 32 unknowns                        1 workgroup / grid over rows
// The stages are separate launches on one stream: the launch boundary is the grid-wide barrier.  Same operations in the same order
// as wg_lu_solve, so the two forms agree to rounding (tested on systems both can take).
#pragma once
#include "ttn_als_kernels.h"

struct AlsAssembleArgs {
    const double *G, *H, *Gb, *Hb;      // G (nr, nr, Rr), H (Rr, rr, rr), Gb (nr, br), Hb (rr, br)
    double *K, *Pb;
    int nr, rr, Rr, br;
};

// one thread per ROW of a 256-row strip, walking a strip of columns: ab / c of the row once, de / f of a column advance without divisions
#define ALS_ASM_ROWS 256
#define ALS_ASM_COLS 64
__global__ void __launch_bounds__(ALS_ASM_ROWS) k_als_assemble(AlsAssembleArgs Q) {
    const int N = Q.nr * Q.rr;
    const int row = blockIdx.x * ALS_ASM_ROWS + threadIdx.x;
    const int col0 = blockIdx.y * ALS_ASM_COLS;
    if (row < N) {
        const int ab = row % Q.nr, c = row / Q.nr;
        int de = col0 % Q.nr, f = col0 / Q.nr;
        const int cend = min(col0 + ALS_ASM_COLS, N);
        for (int col = col0; col < cend; ++col) {
            double a = 0.0;
            for (int z = 0; z < Q.Rr; ++z)
                a = fma(Q.G[ab + (long long)Q.nr * (de + (long long)Q.nr * z)], Q.H[z + Q.Rr * (c + (long long)Q.rr * f)], a);
            Q.K[(long long)col * N + row] = a;
            if (++de == Q.nr) { de = 0; ++f; }
        }
        if (blockIdx.y == 0) {                                            // Pb[(ia, a2)] = sum_be Gb[ia, be] Hb[a2, be]
            double a = 0.0;
            for (int be = 0; be < Q.br; ++be) a = fma(Q.Gb[ab + (long long)Q.nr * be], Q.Hb[c + (long long)Q.rr * be], a);
            Q.Pb[row] = a;
        }
    }
}

// ---- the panel k0 .. k0 + w - 1: the column loop of wg_lu_solve, one workgroup.  flag[0] = 1 on an exactly zero pivot column. ----
__global__ void __launch_bounds__(TTN_WG) k_lu_panel(double* K_, int N, int k0, int w, int* piv_, int* flag) {
    extern __shared__ double lds[];
    typedef __attribute__((address_space(1))) int gmem_i32;
    gmem_wf64* K = (gmem_wf64*)K_;
    gmem_i32* piv = (gmem_i32*)piv_;
    double* red = lds + 128 * 128 + 64;
    int* iflag_ = reinterpret_cast<int*>(red + 40);
    lds_i32* iflag = (lds_i32*)iflag_;
    const int tid = threadIdx.x;
    if (*flag) return;                                                    // an earlier panel found the matrix singular
    const int mrows = N - k0;
    if ((long long)mrows * LU_NB <= 128 * 128) {
        lds_f64* Pn = (lds_f64*)lds;
        const int ldp = mrows | 1;
        for (int e = tid; e < mrows * w; e += TTN_WG) { const int c = e / mrows, i = e - c * mrows; Pn[c * ldp + i] = K[(long long)(k0 + c) * N + k0 + i]; }
        __syncthreads();
        for (int jl = 0; jl < w; ++jl) {
            lds_f64* colj = Pn + jl * ldp;
            double vm = 0.0;
            for (int i = jl + tid; i < mrows; i += TTN_WG) vm = fmax(vm, fabs(colj[i]));
            if (tid == 0) iflag[0] = N;
            vm = unif64(wg_max(vm, red));
            if (!(vm > 0.0)) { if (tid == 0) *flag = 1; return; }
            for (int i = jl + tid; i < mrows; i += TTN_WG) if (fabs(colj[i]) == vm) atomicMin(iflag_, i);
            __syncthreads();
            const int pvl = uni32(iflag[0]);
            if (tid == 0) piv[k0 + jl] = k0 + pvl;
            if (pvl != jl && tid < w) { lds_f64* c = Pn + tid * ldp; const double t = c[jl]; c[jl] = c[pvl]; c[pvl] = t; }
            __syncthreads();
            const double pivot = colj[jl];
            __syncthreads();
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
        return;
    }
    for (int j = k0; j < k0 + w; ++j) {
        gmem_wf64* colj = K + (long long)j * N;
        double vm = 0.0;
        for (int i = j + tid; i < N; i += TTN_WG) vm = fmax(vm, fabs(colj[i]));
        if (tid == 0) iflag[0] = N;
        vm = unif64(wg_max(vm, red));
        if (!(vm > 0.0)) { if (tid == 0) *flag = 1; return; }
        for (int i = j + tid; i < N; i += TTN_WG) if (fabs(colj[i]) == vm) atomicMin(iflag_, i);
        __syncthreads();
        const int pv = uni32(iflag[0]);
        if (tid == 0) piv[j] = pv;
        if (pv != j && tid < w) { gmem_wf64* c = K + (long long)(k0 + tid) * N; const double t = c[j]; c[j] = c[pv]; c[pv] = t; }
        __syncthreads();
        const double pivot = colj[j];
        __syncthreads();
        for (int i = j + 1 + tid; i < N; i += TTN_WG) colj[i] = colj[i] / pivot;
        __syncthreads();
        const int m = N - j - 1, nc = k0 + w - j - 1;
        for (long long e = tid; e < (long long)m * nc; e += TTN_WG) {
            const int i = j + 1 + (int)(e % m), c = j + 1 + (int)(e / m);
            K[(long long)c * N + i] = fma(-colj[i], K[(long long)c * N + j], K[(long long)c * N + i]);
        }
        __syncthreads();
    }
}

// ---- interchanges of panel k0 on every column outside it (and on the rhs = column N), then U12 = L11^-1 A12 for the columns right
//      of the panel (and the rhs): one thread per column ----
__global__ void __launch_bounds__(256) k_lu_rows(double* K_, double* rhs_, int N, int k0, int w, const int* piv, const int* flag) {
    __shared__ double L11[LU_NB * LU_NB];
    __shared__ int pv[LU_NB];
    gmem_wf64* K = (gmem_wf64*)K_;
    gmem_wf64* rhs = (gmem_wf64*)rhs_;
    if (*flag) return;
    for (int e = threadIdx.x; e < LU_NB * LU_NB; e += 256) {
        const int ii = e % LU_NB, jj = e / LU_NB;
        L11[e] = (ii < w && jj < w && ii > jj) ? K[(long long)(k0 + jj) * N + k0 + ii] : 0.0;
    }
    if (threadIdx.x < LU_NB) pv[threadIdx.x] = threadIdx.x < w ? piv[k0 + threadIdx.x] : 0;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;                         // 0 .. N (N = the rhs)
    if (c > N || (c >= k0 && c < k0 + w)) return;
    gmem_wf64* col = (c == N) ? rhs : K + (long long)c * N;
    for (int j = 0; j < w; ++j) {
        const int p = pv[j];
        if (p != k0 + j) { const double t = col[k0 + j]; col[k0 + j] = col[p]; col[p] = t; }
    }
    if (c < k0 + w) return;                                               // left of the panel: interchanges only
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

// ---- A22 -= L21 U12, one 128 x 128 tile per workgroup (fp64 MFMA: the 2/3 N^3 flops); tile column 0 also eliminates the rhs ----
#define LU_TILE 128
__global__ void __launch_bounds__(TTN_WG) k_lu_trail(double* K_, double* rhs_, int N, int k0, int w, const int* flag) {
    extern __shared__ double lds[];
    if (*flag) return;
    const int s0 = k0 + w, m = N - s0;
    const int r0 = blockIdx.x * LU_TILE, c0 = blockIdx.y * LU_TILE;
    if (r0 >= m || c0 >= m) return;
    const int tm = min(LU_TILE, m - r0), tn = min(LU_TILE, m - c0);
    if (blockIdx.y == 0) {                                                // rhs[i] -= sum_jj L21[i, jj] rhs[k0 + jj]
        gmem_wf64* K = (gmem_wf64*)K_;
        gmem_wf64* rhs = (gmem_wf64*)rhs_;
        for (int i = threadIdx.x; i < tm; i += TTN_WG) {
            double a = rhs[s0 + r0 + i];
            for (int jj = 0; jj < w; ++jj) a = fma(-K[(long long)(k0 + jj) * N + s0 + r0 + i], rhs[k0 + jj], a);
            rhs[s0 + r0 + i] = a;
        }
    }
    const View L21 = mkview(K_ + (long long)k0 * N + (s0 + r0), plain(1), plain(N));                  // tm x w
    const View U12 = mkview(K_ + (long long)(s0 + c0) * N + k0, plain(1), plain(N));                  // w x tn
    const View A22 = mkview(K_ + (long long)(s0 + c0) * N + (s0 + r0), plain(1), plain(N));           // tm x tn
    wg_gemm(tm, tn, w, L21, U12, A22, -1.0, 1.0, lds);
}

// ---- back substitution: the triangular 32-block kb (one workgroup, wave 0), then its contribution to the rows above (grid) ----
__global__ void __launch_bounds__(64) k_lu_back_tri(const double* K_, double* rhs_, int N, int kb, int wb, const int* flag) {
    __shared__ double Ub[33 * 32];
    __shared__ double yb[32];
    if (*flag) return;
    gmem_f64* K = (gmem_f64*)K_;
    gmem_wf64* rhs = (gmem_wf64*)rhs_;
    const int lane = threadIdx.x;
    for (int e = lane; e < wb * wb; e += 64) { const int c = e / wb, r_ = e - c * wb; if (r_ <= c) Ub[c * 33 + r_] = K[(long long)(kb + c) * N + kb + r_]; }
    if (lane < wb) yb[lane] = rhs[kb + lane];
    __syncthreads();
    for (int c = wb - 1; c >= 0; --c) {
        const double xc = yb[c] / Ub[c * 33 + c];
        __syncthreads();
        if (lane < c) yb[lane] = fma(-Ub[c * 33 + lane], xc, yb[lane]);
        if (lane == c) yb[c] = xc;
        __syncthreads();
    }
    if (lane < wb) rhs[kb + lane] = yb[lane];
}
__global__ void __launch_bounds__(256) k_lu_back_rows(const double* K_, double* rhs_, int N, int kb, int wb, const int* flag) {
    __shared__ double yb[32];
    if (*flag) return;
    gmem_f64* K = (gmem_f64*)K_;
    gmem_wf64* rhs = (gmem_wf64*)rhs_;
    if (threadIdx.x < wb) yb[threadIdx.x] = rhs[kb + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kb) return;
    double a = rhs[i];
    for (int c = wb - 1; c >= 0; --c) a = fma(-K[(long long)(kb + c) * N + i], yb[c], a);
    rhs[i] = a;
}
