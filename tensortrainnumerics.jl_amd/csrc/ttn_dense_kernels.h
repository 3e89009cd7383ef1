// ttn_dense_kernels.h — one-workgroup-per-train dense linear algebra for the chain recurrences:
// tt_compress! bond steps (merge GEMM -> LQ -> one-sided Jacobi SVD -> truncate -> split),
// dot (transfer-matrix GEMMs) and orthogonalize (Householder QR/LQ sweeps).
//
// Design: a sweep is strictly sequential along the chain, so ONE 1024-thread workgroup owns one
// train for the whole sweep (no inter-workgroup synchronisation anywhere); a batch of trains fills
// the 256 CUs.  All matrices are addressed through `View`s (2-level strides) so the reference's
// permutedims/reshape copies (src/tt_tools.jl:746-767) become index arithmetic.
//
// Map of this file (top to bottom):
//   wg_sum / wg_max                 workgroup reductions
//   wg_gemm                         fp64 MFMA GEMM on Views: tiled (2 LDS stages, offset tables) and one-shot small form
//   lq_lds_whole(_q), wg_lq_blocked Householder LQ: whole matrix in LDS / TSQR column chunks / blocked compact WY
//   wg_jacobi_cols                  one-sided Jacobi in global memory (fallback for short sides > 256)
//   jacobi_lds128_body, wg_jacobi_lds128, wg_jacobi_blocked256
//                                   the LDS Jacobi (register-resident stationary columns, DPP hand-over of the moving
//                                   block, MFMA all-reduce) and its blocked driver for short sides 129..256
//   chol_lds128_teams               blocked LDS Cholesky with lookahead, one or two matrices at once
//   wg_svd_cols / wg_rank_rule / wg_check_diag, wg_materialize_core / wg_fused_merge, wg_bond_step, k_compress
//                                   the bond step (routes F / G / H, fused apply) and the persistent sweep kernel
//   k_dot, k_selftest_gemm, k_bench_gemm, k_bench_lds
// Sibling headers (included after this one by ttn_api.hip): ttn_eig_kernels.h (symmetric eigensolver of the Gram routes, called
// from the bond step through the forward declarations below), ttn_ortho_kernels.h, ttn_hsvd_kernels.h (ttv_decomp, SVD moves),
// ttn_als_kernels.h (als_linsolve / mals_linsolve).
// Rule learnt the hard way: arguments of out-of-line device functions arrive in VGPRs, and anything
// loaded from memory the kernel also writes is a per-lane value to the compiler — pin workgroup-uniform values with
// uni32/uni64/unip, or every derived size, view and pointer occupies VGPRs and ends up in the stack frame.
#pragma once
#include "ttn_common.h"
#include <float.h>

#define GEMM_BK 16
// Wave priorities (s_setprio, a per-wave hint to the SIMD's issue arbiter): with two workgroups per CU one of them is usually in a
// latency-bound phase (an eigensolver's serial chain, a Cholesky pivot, a Jacobi sweep) while the other streams MFMAs; the
// latency-bound waves issue first.  Levels: TTN_PRIO_BASE for everything that is not a matrix product, 0 inside the matrix-product
// routines, higher inside the eigensolver (ttn_eig_kernels.h).
#ifdef TTN_PRIO_BASE
#define TTN_SETPRIO_GEMM() __builtin_amdgcn_s_setprio(0)
#define TTN_SETPRIO_BASE() __builtin_amdgcn_s_setprio(TTN_PRIO_BASE)
#else
#define TTN_SETPRIO_GEMM()
#define TTN_SETPRIO_BASE()
#endif
// LDS leading dimension (doubles) of the staged chunks, 145 = 17 mod 32.  Two access patterns meet here:
//  * a k-fast operand is staged by 16 lanes that walk k (coalesced global reads) and store to As[kk*LD + r]: an even LD
//    puts all 16 stores on ONE bank pair (16-way conflict: measured 49 % of the MFMA peak with LD = 144, stores alone
//    cost 13 points); with LD odd, 2*kk*LD mod 32 = 2*kk: conflict free;
//  * the MFMA fragment reads take two k-rows per 32-lane half (ds_read_b64, 64 banks): LD*8 mod 256 = 136 puts the second
//    row 2 banks short of the other half: one 2-way conflict per read instead of none.
#define GEMM_LD 145
// The LDS IMAGE region (Jacobi / Cholesky / eigensolver images with leading dimension 128, LQ images, staged GEMM operands):
//   1024-thread build: 128 x 128 doubles, one workgroup per CU (its 16 waves at 128 VGPRs fill the register file anyway);
//   512-thread build:   64 x 128 doubles — the whole kernel then needs < 80 KB of LDS and TWO workgroups share a CU (8 waves at 128
//   VGPRs each), so the serial phases of one train (reflectors, pivots, bisection rounds, barriers) overlap the other's work.
//   Images of more than TTN_LDS_COLS columns take the blocked / global-memory forms there.
#if TTN_WG == 512
#define TTN_LDS_IMG (64 * 128)
#else
#define TTN_LDS_IMG (128 * 128)
#endif
#define TTN_LDS_COLS (TTN_LDS_IMG / 128)           // columns of a leading-dimension-128 image
#define GEMM_LDS_DOUBLES (TTN_LDS_IMG + 512)       // LDS region every GEMM may use (the one-shot small GEMM uses all of it): the
                                                   // image + room for the odd leading dimensions of a 64x64x128 one-shot product
// Call-boundary policy of the big building blocks (experiments: -DTTN_NI_JACOBI=inline etc.)
#ifndef TTN_NI_JACOBI
#define TTN_NI_JACOBI __noinline__
#endif
#ifndef TTN_NI_CHOL
#define TTN_NI_CHOL __noinline__
#endif
#ifndef TTN_NI_GEMM
#define TTN_NI_GEMM __noinline__
#endif
#ifndef TTN_NI_GEMMS
#define TTN_NI_GEMMS __noinline__
#endif
#define QR_NB 16                         // Householder panel width
#define JACOBI_MAX_SWEEPS 40

// -------------------------------------------------------------------------------------------------
// workgroup reductions (1024 threads = 16 waves of 64)
// -------------------------------------------------------------------------------------------------
// Wave reductions on DPP row operations + lane reads: __shfl_xor compiles to ds_bpermute (an LDS round trip per step: 1250 clk
// per 64-lane sum measured with the workgroup busy, 440 for this form).  All 64 lanes must be active.
template <int CTRL>
__device__ inline double dpp_mov_f64(double v) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xF, 0xF, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xF, 0xF, true);
    return r.d;
}
__device__ inline double row16_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);    // row_half_mirror
    v += dpp_mov_f64<0x140>(v);    // row_mirror
    return v;
}
__device__ inline double row16_max(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v));
    v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v));
    v = fmax(v, dpp_mov_f64<0x140>(v));
    return v;
}
__device__ inline double wave64_sum_fast(double v) {
    v = row16_sum(v);                                   // every lane of a 16-lane row holds the row sum
    union { double d; int i[2]; } u;
    u.d = v;
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        union { double d; int i[2]; } w;
        w.i[0] = __builtin_amdgcn_readlane(u.i[0], 16 * r);
        w.i[1] = __builtin_amdgcn_readlane(u.i[1], 16 * r);
        t += w.d;
    }
    return t;
}
__device__ inline double wave64_max_fast(double v) {
    v = row16_max(v);
    union { double d; int i[2]; } u;
    u.d = v;
    double t = -1.7976931348623157e308;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        union { double d; int i[2]; } w;
        w.i[0] = __builtin_amdgcn_readlane(u.i[0], 16 * r);
        w.i[1] = __builtin_amdgcn_readlane(u.i[1], 16 * r);
        t = fmax(t, w.d);
    }
    return t;
}
__device__ inline double wave_sum(double v) { return wave64_sum_fast(v); }
__device__ inline double wave_max(double v) { return wave64_max_fast(v); }
// red: LDS scratch of >= 32 doubles.  All threads get the result.
__device__ inline double wg_sum(double v, double* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    double t = (l < nw) ? red[l] : 0.0;
    __syncthreads();                                   // red is free again: the caller's next routine may write it at once
    t = wave_sum(t);
    return t;
}
__device__ inline double wg_max(double v, double* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    double t = (l < nw) ? red[l] : 0.0;
    __syncthreads();                                   // (as above)
    t = wave_max(t);
    return t;
}

// -------------------------------------------------------------------------------------------------
// wg_gemm: C[m x n] = alpha * A[m x k] * B[k x n] + beta * C, all operands through Views.
// fp64 MFMA (v_mfma_f64_16x16x4_f64): the 16 waves form a 4x4 grid, each wave owns a 32x32 block of a
// 128x128 output tile (2x2 MFMA tiles, 16 accumulator doubles per lane).  K is consumed in chunks of
// 16 staged through LDS (k-major, leading dimension 144 doubles so the two 16-lane k-rows a half-wave
// reads land on disjoint bank halves); the next chunk's global loads are issued before the MFMAs of
// the current one.  Fragment maps (one f64 per lane): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D[row = (lane>>4) + 4*reg][col = lane&15].
// Every thread of the workgroup must call it (it contains barriers).
// -------------------------------------------------------------------------------------------------
typedef double mfma_acc_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) double lds_f64;

struct GemmDesc {
    int m, n, k, pad;
    View A, B, C;
    double alpha, beta;
    unsigned long long* amax;   // null, or an LDS word that receives max |C_ij| over the stored values (bits of a non-negative double)
};
// Behind the LDS tiles: the descriptor (sizeof(GemmDesc) = 208 bytes, 32 doubles reserved) and the OFFSET TABLES.
// Operands are 2-level strided Views, so an element address costs two integer divisions; a GEMM evaluates ix() once per
// row / column / k index into these tables and the staging loops only add table entries.
// (32-bit entries, two per double of the region: element offsets fit 32 bits, operands are at most 4096 x 16384 doubles)
#if TTN_WG == 512
#define GEMM_TAB_ENTRIES 704                     // doubles: tiled 2 x GEMM_KSLAB + BM + BN ints; one-shot 4 x 128 + 2 x 256 ints
#define GEMM_KSLAB 512
#define GEMM_SMALL_KMAX 256
#else
#define GEMM_TAB_ENTRIES 1536
#define GEMM_KSLAB 768                           // k indices tabulated at a time by the tiled GEMM (A and B tables)
#define GEMM_SMALL_KMAX 512                      // one-shot GEMM: k <= GEMM_LDS_DOUBLES / 34
#endif
#define GEMM_DESC_DOUBLES (32 + GEMM_TAB_ENTRIES)
#define GEMM_LDS_TOTAL (GEMM_LDS_DOUBLES + GEMM_DESC_DOUBLES)
typedef __attribute__((address_space(3))) long long lds_i64;
typedef __attribute__((address_space(3))) int lds_i32;
// GEMM A/B operands always live in global memory.  Loading them through the generic `double*` of a View makes the
// compiler emit flat_load, which counts on lgkmcnt as well as vmcnt — every `s_waitcnt lgkmcnt(0)` in front of an MFMA
// group (meant for the LDS fragment reads) then also waits for the global loads of the NEXT chunk: measured 49 % -> 7x %
// of the per-CU MFMA peak for the tiled GEMM once the loads are global_load.
typedef const __attribute__((address_space(1))) double gmem_f64;
typedef __attribute__((address_space(1))) double gmem_wf64;

// The body is deliberately NOT inlined (17 call sites) and takes its operands through a descriptor in LDS: only two
// pointers cross the call boundary.
// workgroup-uniform values read from LDS land in VGPRs; readfirstlane moves them to SGPRs (the GEMM is VGPR-starved)
__device__ inline int uni32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline long long uni64(long long v) {
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
}
template <typename PT> __device__ inline PT* unip(PT* q_) { return (PT*)uni64((long long)q_); }
__device__ inline double unif64(double v) { union { double d; long long i; } u; u.d = v; u.i = uni64(u.i); return u.d; }
__device__ inline Idx uniIdx(const Idx& d) { return Idx{uni32(d.q), uni64(d.lo), uni64(d.hi)}; }
__device__ inline View uniView(const View& v) { return View{(double*)uni64((long long)v.p), uniIdx(v.r), uniIdx(v.c)}; }

// The descriptor lives in LDS: read it through an LDS-address-space pointer (ds_read, ~100 clk and batched) — through the generic
// pointer the ~25 fields were FLAT loads, 3.5 k clk per GEMM call before the first useful instruction.
typedef __attribute__((address_space(3))) GemmDesc lds_gdesc;
__device__ inline Idx ldsIdx(const __attribute__((address_space(3))) Idx* d) { return Idx{uni32(d->q), uni64(d->lo), uni64(d->hi)}; }
__device__ inline View ldsView(const __attribute__((address_space(3))) View* v) { return View{(double*)uni64((long long)v->p), ldsIdx(&v->r), ldsIdx(&v->c)}; }

// -------------------------------------------------------------------------------------------------
// Element loops over global memory.  `for (e = tid; e < n; e += TTN_WG) dst[e] = f(src[e])` compiles to load, s_waitcnt vmcnt(0),
// store per trip (no unrolling, possible aliasing): a full memory round trip per ELEMENT of a thread.  wg_batched runs such a loop
// with U loads of a thread in flight before the first is used: load(e) returns what the element needs (index clamped: it must be
// harmless to call it twice for an e < n), use(e, value) consumes it.  It is used inside OUT-OF-LINE leaf routines only: inlined
// into the bond step the same loops cost 300 more spilled VGPRs (the step runs at the 128-register limit) and 19 % of its speed.
// -------------------------------------------------------------------------------------------------
template <int U, typename LoadF, typename UseF>
__device__ __forceinline__ void wg_batched(long long n, LoadF load, UseF use) {
    for (long long e0 = threadIdx.x; e0 < n; e0 += (long long)TTN_WG * U) {
        decltype(load(0LL)) v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long e = e0 + (long long)u * TTN_WG; v[u] = load(e < n ? e : e0); }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long e = e0 + (long long)u * TTN_WG; if (e < n) use(e, v[u]); }
    }
}
// dst[e] = s * src[e], e < n
__device__ __noinline__ void wg_copy_scale(double* dst, const double* src, long long n, double s_) {
    dst = unip(dst); src = unip(src); n = uni64(n); s_ = unif64(s_);
    wg_batched<8>(n, [&](long long e) { return src[e]; }, [&](long long e, double v) { dst[e] = v * s_; });
}
// max |src[e]|, e < n (all threads get it)
__device__ __noinline__ double wg_absmax(const double* src, long long n, double* red) {
    src = unip(src); n = uni64(n); red = unip(red);
    double m_ = 0.0;
    wg_batched<8>(n, [&](long long e) { return src[e]; }, [&](long long, double v) { m_ = fmax(m_, fabs(v)); });
    return unif64(wg_max(m_, red));
}
// dst(i, j) = (lim_first ? i : j) < lim ? src[i + ld_first... : the commit copy of the factored route: element (row, col) of the rows x cols
// matrix `src` (element stride rs between rows, cs between columns) goes to the View `dst`; entries whose column (col_lim) or row
// (!col_lim) index is >= lim are written as zeros
__device__ __noinline__ void wg_copy_to_view(View dst, const double* src, int rows, int cols, long long rs, long long cs, int lim, int col_lim) {
    dst = uniView(dst); src = unip(src); rows = uni32(rows); cols = uni32(cols); rs = uni64(rs); cs = uni64(cs); lim = uni32(lim); col_lim = uni32(col_lim);
    const bool row_fast = rs <= cs;
    const int fast = row_fast ? rows : cols;                       // 32-bit index arithmetic (rows * cols < 2^31: at most 4096 x 16384)
    wg_batched<8>((long long)rows * cols,
                  [&](long long e) {
                      const int ei = (int)e, lo = ei % fast, hi = ei / fast;
                      const int i = row_fast ? lo : hi, j = row_fast ? hi : lo;
                      return ((col_lim ? j : i) < lim) ? src[i * rs + j * cs] : 0.0;
                  },
                  [&](long long e, double v) {
                      const int ei = (int)e, lo = ei % fast, hi = ei / fast;
                      const int i = row_fast ? lo : hi, j = row_fast ? hi : lo;
                      dst.p[ix(dst.r, i) + ix(dst.c, j)] = v;
                  });
}


// max |C_ij| of the values a GEMM stored, for callers that rescale by it: saves a pass over C (wave max, then one LDS
// atomic per wave; non-negative doubles order like their bit patterns)
__device__ inline void gemm_publish_amax(const lds_gdesc* dsc, double cmax) {
    unsigned long long* am = (unsigned long long*)uni64((long long)dsc->amax);
    if (am) {
        cmax = wave_max(cmax);
        if ((threadIdx.x & 63) == 0) atomicMax(am, (unsigned long long)__double_as_longlong(cmax));
    }
}

// WR = wave rows of the 16-wave grid (WR x 16/WR waves, 32x32 outputs per wave): 4 -> 128x128 output tiles, 2 -> 64x256
// (for m <= 64 and wide n, where half the waves of the square grid would idle).
template <int WR>
__device__ TTN_NI_GEMM void wg_gemm_impl(const GemmDesc* dsc_, double* lds) {
    const lds_gdesc* dsc = (const lds_gdesc*)dsc_;
    constexpr int WCN = TTN_NWAVES / WR;                 // wave columns
    constexpr int BM = 32 * WR, BN = 32 * WCN;
    constexpr int LDA = (BM <= 64) ? 81 : GEMM_LD, LDB = (BN <= 64) ? 81 : (BN <= 128) ? GEMM_LD : 273;     // all == 17 mod 32 (see GEMM_LD)
    constexpr int STAGE = GEMM_BK * (LDA + LDB);
    static_assert(2 * STAGE <= GEMM_LDS_DOUBLES, "two LDS stages must fit");
    const int m = uni32(dsc->m), n = uni32(dsc->n), k = uni32(dsc->k);
    const View A = ldsView(&dsc->A), B = ldsView(&dsc->B), C = ldsView(&dsc->C);
    const double alpha = unif64(dsc->alpha), beta = unif64(dsc->beta);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WCN, wc = wave % WCN;
    const int li = lane & 15, lk = lane >> 4;
    gmem_f64* Ag = (gmem_f64*)A.p;
    gmem_f64* Bg = (gmem_f64*)B.p;
    // element offsets fit 32 bits (operands are at most 4096 x 16384 doubles): half the registers of 64-bit ones
    lds_i32* tabA = (lds_i32*)(lds + GEMM_LDS_DOUBLES + 32);      // ix(A.c, slab0 + j), j < GEMM_KSLAB
    lds_i32* tabB = tabA + GEMM_KSLAB;                            // ix(B.r, slab0 + j)
    const bool a_kfast = minstride(A.c) < minstride(A.r);
    const bool b_kfast = minstride(B.r) < minstride(B.c);
    // staging assignment: element e = tid + TTN_WG*u of the BM x BK (A) and BK x BN (B) chunk
    constexpr int NUA = BM * GEMM_BK / TTN_WG, NUB = BN * GEMM_BK / TTN_WG;
    int ar[NUA], akk[NUA], bc[NUB], bkk[NUB];
#pragma unroll
    for (int u = 0; u < NUA; ++u) {
        const int e = tid + TTN_WG * u;
        if (a_kfast) { akk[u] = e & (GEMM_BK - 1); ar[u] = e / GEMM_BK; } else { ar[u] = e & (BM - 1); akk[u] = e / BM; }
    }
#pragma unroll
    for (int u = 0; u < NUB; ++u) {
        const int e = tid + TTN_WG * u;
        if (b_kfast) { bkk[u] = e & (GEMM_BK - 1); bc[u] = e / GEMM_BK; } else { bc[u] = e & (BN - 1); bkk[u] = e / BN; }
    }
    double cmax = 0.0;
    const int nch = (k + GEMM_BK - 1) / GEMM_BK;
    int slab0 = -1;                                               // first k index the tables currently hold
    for (int m0 = 0; m0 < m; m0 += BM) {
        for (int n0 = 0; n0 < n; n0 += BN) {
            int aoff[NUA], boff[NUB];
            bool aok[NUA], bok[NUB];
#pragma unroll
            for (int u = 0; u < NUA; ++u) { aok[u] = (m0 + ar[u]) < m; aoff[u] = aok[u] ? (int)ix(A.r, m0 + ar[u]) : 0; }
#pragma unroll
            for (int u = 0; u < NUB; ++u) { bok[u] = (n0 + bc[u]) < n; boff[u] = bok[u] ? (int)ix(B.c, n0 + bc[u]) : 0; }
            const bool live = (m0 + wr * 32 < m) && (n0 + wc * 32 < n);      // wave-uniform
            mfma_acc_t acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
            double av[NUA], bv[NUB];
            // Software pipeline over the K chunks, two LDS stages, ONE barrier per chunk:
            //   iteration c:  MFMAs of chunk c (stage c&1) | registers (chunk c+1) -> stage (c+1)&1 | global loads of chunk c+2
#define GEMM_FILL_TAB(K0)                                                                                       \
            {                                                                                                   \
                __syncthreads();                                                                                \
                slab0 = (K0);                                                                                   \
                for (int j = tid; j < GEMM_KSLAB; j += TTN_WG) {                                                \
                    const int kk = slab0 + j;                                                                   \
                    tabA[j] = (kk < k) ? (int)ix(A.c, kk) : 0;                                                     \
                    tabB[j] = (kk < k) ? (int)ix(B.r, kk) : 0;                                                     \
                }                                                                                               \
                __syncthreads();                                                                                \
            }
#define GEMM_LOAD(K0)  /* branch-free: out-of-range elements read element 0 of the operand and are then zeroed */        \
            {                                                                                                   \
                int oa_[NUA], ob_[NUB];                                                                         \
                bool va_[NUA], vb_[NUB];                                                                        \
                _Pragma("unroll") for (int u = 0; u < NUA; ++u) {                                               \
                    const int ga = (K0) + akk[u];                                                               \
                    va_[u] = aok[u] && ga < k;                                                                  \
                    oa_[u] = va_[u] ? aoff[u] + tabA[ga - slab0] : 0;                                           \
                }                                                                                               \
                _Pragma("unroll") for (int u = 0; u < NUB; ++u) {                                               \
                    const int gb = (K0) + bkk[u];                                                               \
                    vb_[u] = bok[u] && gb < k;                                                                  \
                    ob_[u] = vb_[u] ? boff[u] + tabB[gb - slab0] : 0;                                           \
                }                                                                                               \
                _Pragma("unroll") for (int u = 0; u < NUA; ++u) av[u] = Ag[oa_[u]];                             \
                _Pragma("unroll") for (int u = 0; u < NUB; ++u) bv[u] = Bg[ob_[u]];                             \
                _Pragma("unroll") for (int u = 0; u < NUA; ++u) av[u] = va_[u] ? av[u] : 0.0;                   \
                _Pragma("unroll") for (int u = 0; u < NUB; ++u) bv[u] = vb_[u] ? bv[u] : 0.0;                   \
            }
#define GEMM_STORE(STG)                                                                                         \
            {                                                                                                   \
                lds_f64* As_ = (lds_f64*)lds + (STG) * STAGE;                                                    \
                lds_f64* Bs_ = As_ + GEMM_BK * LDA;                                                             \
                _Pragma("unroll") for (int u = 0; u < NUA; ++u) As_[akk[u] * LDA + ar[u]] = av[u];              \
                _Pragma("unroll") for (int u = 0; u < NUB; ++u) Bs_[bkk[u] * LDB + bc[u]] = bv[u];              \
            }
            if (slab0 != 0) GEMM_FILL_TAB(0)
            else __syncthreads();                      // the previous tile's MFMAs have consumed both stages
            {   // prologue: the loads of chunk 0 AND chunk 1 are in flight together (one exposed global latency, not two)
                GEMM_LOAD(0)
                double av0[NUA], bv0[NUB];
                _Pragma("unroll") for (int u = 0; u < NUA; ++u) av0[u] = av[u];
                _Pragma("unroll") for (int u = 0; u < NUB; ++u) bv0[u] = bv[u];
                if (nch > 1) GEMM_LOAD(GEMM_BK)
                lds_f64* As0 = (lds_f64*)lds;
                lds_f64* Bs0 = As0 + GEMM_BK * LDA;
                _Pragma("unroll") for (int u = 0; u < NUA; ++u) As0[akk[u] * LDA + ar[u]] = av0[u];
                _Pragma("unroll") for (int u = 0; u < NUB; ++u) Bs0[bkk[u] * LDB + bc[u]] = bv0[u];
            }
            __syncthreads();
#ifdef TTN_GEMM_PRIO
            __builtin_amdgcn_s_setprio(TTN_GEMM_PRIO);       // experiment: the matrix-pipe phase ahead of a co-resident workgroup's latency-bound phase
#endif
            for (int c = 0; c < nch; ++c) {
                // The staging work of the other chunks is placed BETWEEN this wave's MFMA groups: an MFMA occupies the
                // matrix pipe for 64 clk, so the VALU/LDS/global instructions issued behind it run under the MFMAs of
                // the SIMD's other waves instead of after them (the waves run in lockstep from barrier to barrier).
                lds_f64* As = (lds_f64*)lds + (c & 1) * STAGE;
                lds_f64* Bs = As + GEMM_BK * LDA;
#define GEMM_MFMA_STEP(T)                                                                                       \
                if (live) {                                                                                     \
                    const int kra = (4 * (T) + lk) * LDA, krb = (4 * (T) + lk) * LDB;                                                 \
                    const double a0 = As[kra + wr * 32 + li], a1 = As[kra + wr * 32 + 16 + li];                 \
                    const double b0 = Bs[krb + wc * 32 + li], b1 = Bs[krb + wc * 32 + 16 + li];                 \
                    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);               \
                    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);               \
                    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);               \
                    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);               \
                }
                GEMM_MFMA_STEP(0)
                if (c + 1 < nch) GEMM_STORE((c + 1) & 1)
                GEMM_MFMA_STEP(1)
                if (c + 2 < nch) {
                    const int k2 = (c + 2) * GEMM_BK;
                    if (k2 >= slab0 + GEMM_KSLAB) GEMM_FILL_TAB(k2)       // workgroup-uniform; rare (k > GEMM_KSLAB)
                    GEMM_LOAD(k2)
                }
                GEMM_MFMA_STEP(2)
                GEMM_MFMA_STEP(3)
#undef GEMM_MFMA_STEP
                __syncthreads();
            }
#ifdef TTN_GEMM_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#undef GEMM_FILL_TAB
#undef GEMM_LOAD
#undef GEMM_STORE
            // C offsets of this tile through tables too (ten ix() per thread otherwise: two integer divisions each)
            lds_i32* rowC = tabB + GEMM_KSLAB;                            // BM entries
            lds_i32* colC = rowC + BM;                                    // BN entries
            for (int i = tid; i < BM + BN; i += TTN_WG) {
                if (i < BM) rowC[i] = (m0 + i < m) ? (int)ix(C.r, m0 + i) : 0;
                else colC[i - BM] = (n0 + i - BM < n) ? (int)ix(C.c, n0 + i - BM) : 0;
            }
            __syncthreads();
            if (live) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int li_ = wr * 32 + ti * 16 + lk + 4 * reg;
                        if (m0 + li_ >= m) continue;
                        const int ro = rowC[li_];
#pragma unroll
                        for (int tj = 0; tj < 2; ++tj) {
                            const int lj_ = wc * 32 + tj * 16 + li;
                            if (n0 + lj_ >= n) continue;
                            gmem_wf64* cp = (gmem_wf64*)C.p + ((long long)ro + colC[lj_]);      // C is always global memory: global_store, not flat
                            double v = alpha * acc[ti][tj][reg];
                            if (beta != 0.0) v += beta * (*cp);
                            *cp = v;
                            cmax = fmax(cmax, fabs(v));
                        }
                    }
            }
        }
    }
    gemm_publish_amax(dsc, cmax);
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------
// One-shot small GEMM: m, n <= 128, m*n <= 8192 (at most 2 MFMA tiles per wave) and the WHOLE K extent of both
// operands fits the LDS region: k*(lda+ldb) <= GEMM_LDS_DOUBLES.  All global loads are issued at once (these GEMMs
// are latency bound: a 64x64x128 product is 1 Mflop), one barrier, k/4 MFMAs per tile, store.  Small register
// footprint on purpose (8 accumulator doubles per tile): the factored route issues ten of these per bond step.
// -------------------------------------------------------------------------------------------------
__device__ inline int small_ld(int x) { const int t = (x + 15) & ~15; return (((t & 31) == 16) ? t : t + 16) + 1; }   // == 17 mod 32 (see GEMM_LD)
__device__ inline int tight_ld(int x) { return ((x + 15) & ~15) + 1; }                                             // odd: conflict-free k-fast stores

__device__ TTN_NI_GEMMS void wg_gemm_small_impl(const GemmDesc* dsc_, double* lds) {
    const lds_gdesc* dsc = (const lds_gdesc*)dsc_;
    const int m = uni32(dsc->m), n = uni32(dsc->n), k = uni32(dsc->k);
    const View A = ldsView(&dsc->A), B = ldsView(&dsc->B), C = ldsView(&dsc->C);
    const double alpha = unif64(dsc->alpha), beta = unif64(dsc->beta);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    // dsc->pad = 1: leading dimensions == 17 mod 32 (fragment reads nearly conflict free); 0: tight (2-way read conflicts, fits more)
    const int lda = uni32(dsc->pad) ? small_ld(m) : tight_ld(m), ldb = uni32(dsc->pad) ? small_ld(n) : tight_ld(n);
    lds_f64* As = (lds_f64*)lds;                       // As[kk*lda + row]
    lds_f64* Bs = As + k * lda;                        // Bs[kk*ldb + col]
    const bool a_kfast = minstride(A.c) < minstride(A.r);
    const bool b_kfast = minstride(B.r) < minstride(B.c);
    const int mp = (m + 15) & ~15, np = (n + 15) & ~15;
    // ---- offset tables (m, n <= 128; k <= 512 because k*(lda+ldb) <= GEMM_LDS_DOUBLES and lda, ldb >= 16) ----
    gmem_f64* Ag = (gmem_f64*)A.p;
    gmem_f64* Bg = (gmem_f64*)B.p;
    lds_i32* rowA = (lds_i32*)(lds + GEMM_LDS_DOUBLES + 32);
    lds_i32* colB = rowA + 128;
    lds_i32* rowC = colB + 128;
    lds_i32* colC = rowC + 128;
    lds_i32* kA = colC + 128;
    lds_i32* kB = kA + GEMM_SMALL_KMAX;
    for (int i = tid; i < 128; i += TTN_WG) {
        rowA[i] = (i < m) ? (int)ix(A.r, i) : 0;
        rowC[i] = (i < m) ? (int)ix(C.r, i) : 0;
        colB[i] = (i < n) ? (int)ix(B.c, i) : 0;
        colC[i] = (i < n) ? (int)ix(C.c, i) : 0;
    }
    for (int i = tid; i < k; i += TTN_WG) { kA[i] = (int)ix(A.c, i); kB[i] = (int)ix(B.r, i); }
    __syncthreads();
    // ---- stage all of A (m x k) and B (k x n), zero padding rows/cols up to the tile edge; 16 lanes walk the
    //      operand's fast (small-stride) index, the 64 lane groups its slow index: no divisions in the loops ----
    const int fx = tid & 15, sy = tid >> 4;
    // The loads of a thread are issued in batches of GS_U before the first of them is awaited (written naively — load, wait, store
    // per element — every element of a thread paid its own global-memory round trip: 16 of them in a row for a 64 x 64 x 128
    // product, most of the time of these latency-bound GEMMs).  Out-of-range elements read element 0 and are zeroed.
#define GS_U 8
#define GEMM_SMALL_STAGE(DST, LDD, SRC, SLOWTAB, FASTTAB, NSLOW, NSLOW_VALID, NFAST, NFAST_VALID, SLOW_IS_K)                      \
    for (int s_ = sy; s_ < (NSLOW); s_ += TTN_WG / 16) {                                                                        \
        const int so_ = (s_ < (NSLOW_VALID)) ? SLOWTAB[s_] : 0;                                                                 \
        for (int f0_ = fx; f0_ < (NFAST); f0_ += 16 * GS_U) {                                                                   \
            double v_[GS_U];                                                                                                    \
            _Pragma("unroll") for (int u_ = 0; u_ < GS_U; ++u_) {                                                               \
                const int f_ = f0_ + 16 * u_;                                                                                   \
                const bool ok_ = (s_ < (NSLOW_VALID)) && (f_ < (NFAST_VALID));                                                  \
                v_[u_] = SRC[ok_ ? so_ + FASTTAB[f_] : 0];                                                                      \
            }                                                                                                                   \
            _Pragma("unroll") for (int u_ = 0; u_ < GS_U; ++u_) {                                                               \
                const int f_ = f0_ + 16 * u_;                                                                                   \
                const bool ok_ = (s_ < (NSLOW_VALID)) && (f_ < (NFAST_VALID));                                                  \
                if (f_ < (NFAST)) DST[(SLOW_IS_K) ? s_ * (LDD) + f_ : f_ * (LDD) + s_] = ok_ ? v_[u_] : 0.0;                    \
            }                                                                                                                   \
        }                                                                                                                       \
    }
    if (a_kfast) { GEMM_SMALL_STAGE(As, lda, Ag, rowA, kA, mp, m, k, k, 0) }          // slow = row, fast = k
    else { GEMM_SMALL_STAGE(As, lda, Ag, kA, rowA, k, k, mp, m, 1) }                  // slow = k, fast = row
    if (b_kfast) { GEMM_SMALL_STAGE(Bs, ldb, Bg, colB, kB, np, n, k, k, 0) }          // slow = column, fast = k
    else { GEMM_SMALL_STAGE(Bs, ldb, Bg, kB, colB, k, k, np, n, 1) }                  // slow = k, fast = column
#undef GEMM_SMALL_STAGE
#undef GS_U
    __syncthreads();
    const int tm = mp >> 4, tn = np >> 4, ntile = tm * tn;
    const int k4 = k >> 2, krem = k & 3;
    double cmax = 0.0;
    for (int tile = wave; tile < ntile; tile += (TTN_WG >> 6)) {
        const int r0 = (tile % tm) << 4, c0 = (tile / tm) << 4;
        mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
        const lds_f64* ap = As + lk * lda + r0 + li;
        const lds_f64* bp = Bs + lk * ldb + c0 + li;
        int t = 0;
        for (; t + 4 <= k4; t += 4) {                   // four k-steps per trip: the eight LDS reads go out before the MFMAs
            double a_[4], b_[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a_[q] = ap[(4 * (t + q)) * lda]; b_[q] = bp[(4 * (t + q)) * ldb]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_[q], b_[q], acc, 0, 0, 0);
        }
        for (; t < k4; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[(4 * t) * lda], bp[(4 * t) * ldb], acc, 0, 0, 0);
        if (krem) {                                     // k not a multiple of 4: pad the last step with zeros
            const double a = (lk < krem) ? ap[(4 * k4) * lda] : 0.0;
            const double b = (lk < krem) ? bp[(4 * k4) * ldb] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int gi = r0 + lk + 4 * reg, gj = c0 + li;
            if (gi < m && gj < n) {
                gmem_wf64* cp = (gmem_wf64*)C.p + ((long long)rowC[gi] + colC[gj]);
                double v = alpha * acc[reg];
                if (beta != 0.0) v += beta * (*cp);
                *cp = v;
                cmax = fmax(cmax, fabs(v));
            }
        }
    }
    gemm_publish_amax(dsc, cmax);
    __syncthreads();
}

__device__ inline void wg_gemm(int m, int n, int k, View A, View B, View C, double alpha, double beta, double* lds,
                               double* amax_lds = nullptr /* LDS word: receives max |C_ij| */) {
    GemmDesc* dsc = reinterpret_cast<GemmDesc*>(lds + GEMM_LDS_DOUBLES);
    __syncthreads();                         // nobody still reads what the tiles / descriptor alias
    if (threadIdx.x == 0) {
        dsc->m = m; dsc->n = n; dsc->k = k; dsc->pad = 0;
        dsc->A = A; dsc->B = B; dsc->C = C;
        dsc->alpha = alpha; dsc->beta = beta;
        dsc->amax = (unsigned long long*)amax_lds;
        if (amax_lds) *amax_lds = 0.0;
    }
    __syncthreads();
    TTN_SETPRIO_GEMM();
    const bool shape_ok = (m <= 128) && (n <= 128) && (((m + 15) >> 4) * ((n + 15) >> 4) <= 32);
    const int wpad = small_ld(m) + small_ld(n), wtight = tight_ld(m) + tight_ld(n);
    if (shape_ok && (long long)k * wpad <= GEMM_LDS_DOUBLES) {
        if (threadIdx.x == 0) dsc->pad = 1;
        __syncthreads();
        wg_gemm_small_impl(dsc, lds);
    } else if (shape_ok && (long long)k * wtight <= GEMM_LDS_DOUBLES) {
        wg_gemm_small_impl(dsc, lds);                    // pad = 0: tight leading dimensions
    } else if (shape_ok && A.c.q == 0 && B.r.q == 0 && k <= 8 * (GEMM_LDS_DOUBLES / wtight)) {
        // few output tiles but a long K: the tiled GEMM would keep most waves idle; run the one-shot kernel over
        // K chunks, accumulating into C.  Plain-stride k indices only: a chunk is addressed by shifting the operands' base pointers,
        // which a two-level k index (the stacked H of the matrix-free two-site operator) does not allow — that takes the tiled form
        const int kcmax = (GEMM_LDS_DOUBLES / wtight) & ~3;
        const int nck = (k + kcmax - 1) / kcmax;
        const int kc = (((k + nck - 1) / nck) + 3) & ~3;                  // balanced chunks, multiples of 4
        for (int k0 = 0; k0 < k; k0 += kc) {
            if (k0 > 0) __syncthreads();
            if (threadIdx.x == 0) {
                dsc->k = (k - k0 < kc) ? k - k0 : kc;
                dsc->A.p = A.p + ix(A.c, k0);
                dsc->B.p = B.p + ix(B.r, k0);
                dsc->beta = (k0 == 0) ? beta : 1.0;
                dsc->amax = (k0 + kc >= k) ? (unsigned long long*)amax_lds : nullptr;     // the last chunk stores the final values
            }
            __syncthreads();
            wg_gemm_small_impl(dsc, lds);
        }
    } else if (m <= 64 && n > 128) {
        wg_gemm_impl<2>(dsc, lds);                       // 64 x 256 output tiles: all 16 waves busy on a short, wide product
    } else {
        wg_gemm_impl<4>(dsc, lds);
    }
    TTN_SETPRIO_BASE();
}

// -------------------------------------------------------------------------------------------------
// wg_syrk: G = alpha * A A^T, A (p x q, p <= 128) and G (p x p, both triangles written) through Views in global memory.
// The Gram products of a bond step (M M^T, the a-posteriori checks Rf Rf^T / Lf^T Lf, route F's A'^T A' and B' B'^T) are
// 64..128 rows x a long K: as general GEMMs they ran the tiled form at 25 % of the matrix pipe (128 x 128 x 384: both operands staged
// although they are the same matrix, a barrier per 16 k) or the chunked one-shot form (64 x 64 x 384: seven K chunks, each read-modify-
// writing C in memory).  Here A is staged ONCE per K chunk (k-major, leading dimension == 17 mod 32 like the GEMM's) and serves both
// fragment operands — frag(tb) = A[16 tb + lane&15][k + lane>>4] is the A fragment of tile row tb and the B fragment of tile column
// tb alike —, only the tiles on and below the diagonal are computed (36 of 64 for p = 128) with the accumulators in registers over all
// of K, and the global loads of chunk c + 1 are in flight (registers) while chunk c is multiplied.
//   KC16 x 16 = k per chunk, JMAX = rows per thread of a chunk, MAXT = tiles per wave (all sized per build by the wrapper below).
// -------------------------------------------------------------------------------------------------
#define SYRK_QMAX (2 * GEMM_TAB_ENTRIES - 128)           // k offsets tabulated once (32-bit entries behind the descriptor)
template <int KC16, int JMAX, int MAXT>
__device__ TTN_NI_GEMM void wg_syrk_impl(const GemmDesc* dsc_, double* lds) {
    const lds_gdesc* dsc = (const lds_gdesc*)dsc_;
    constexpr int KC = 16 * KC16, RG = TTN_WG / 16;                 // k per chunk; row groups (16 lanes walk k)
    const int p = uni32(dsc->m), q = uni32(dsc->k);
    const View A = ldsView(&dsc->A), C = ldsView(&dsc->C);
    const double alpha = unif64(dsc->alpha);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int pp = (p + 15) & ~15, LD = small_ld(p);
    lds_f64* Cs = (lds_f64*)lds;                                    // Cs[kk * LD + row]
    lds_i32* rowT = (lds_i32*)(lds + GEMM_LDS_DOUBLES + 32);       // 128 entries
    lds_i32* colT = rowT + 128;                                     // q entries
    gmem_f64* Ag = (gmem_f64*)A.p;
    for (int i = tid; i < 128; i += TTN_WG) rowT[i] = (i < p) ? (int)ix(A.r, i) : 0;
    for (int i = tid; i < q; i += TTN_WG) colT[i] = (int)ix(A.c, i);
    // tiles of this wave: t = wave, wave + NWAVES, ... over the lower triangle, t = tr (tr + 1) / 2 + tc
    const int nt = pp >> 4, ntile = nt * (nt + 1) / 2;
    int t_r[MAXT], t_c[MAXT];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        const int t = wave + u * TTN_NWAVES;
        int tr = 0, base = 0;
        while (base + tr + 1 <= t) { base += tr + 1; ++tr; }
        t_r[u] = uni32((t < ntile) ? tr : -1); t_c[u] = uni32(t - base);
    }
    mfma_acc_t acc[MAXT];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) acc[u] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
    const int fx = tid & 15, sy = tid >> 4;
    double v[JMAX][KC16];
    __syncthreads();
#define SYRK_LOAD(K0)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < JMAX; ++j) {                                                                   \
        const int row = sy + RG * j;                                                                                    \
        const int ro = (row < p) ? rowT[row] : 0;                                                                       \
        _Pragma("unroll") for (int u = 0; u < KC16; ++u) {                                                               \
            const int kk = (K0) + fx + 16 * u;                                                                          \
            v[j][u] = Ag[(row < p && kk < q) ? ro + colT[kk] : 0];                                                      \
        }                                                                                                               \
    }
    SYRK_LOAD(0)
    for (int k0 = 0; k0 < q; k0 += KC) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int row = sy + RG * j;
#pragma unroll
            for (int u = 0; u < KC16; ++u) {
                const int kl = fx + 16 * u;
                if (row < pp) Cs[kl * LD + row] = (row < p && k0 + kl < q) ? v[j][u] : 0.0;
            }
        }
        __syncthreads();
        if (k0 + KC < q) { SYRK_LOAD(k0 + KC) }
        const int ksteps = (q - k0 < KC) ? (q - k0 + 3) >> 2 : KC / 4;
        for (int ks = 0; ks < ksteps; ++ks) {
            const lds_f64* rowp = Cs + (4 * ks + lk) * LD + li;
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {
                if (t_r[u] >= 0) {                                   // wave-uniform
                    const double a = rowp[16 * t_r[u]], b = rowp[16 * t_c[u]];
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#undef SYRK_LOAD
    gmem_wf64* Cg = (gmem_wf64*)C.p;
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        if (t_r[u] < 0) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int gi = 16 * t_r[u] + lk + 4 * reg, gj = 16 * t_c[u] + li;
            if (gi < p && gj < p) {
                const double val = alpha * acc[u][reg];
                Cg[ix(C.r, gi) + ix(C.c, gj)] = val;
                if (t_r[u] != t_c[u]) Cg[ix(C.r, gj) + ix(C.c, gi)] = val;
            }
        }
    }
    __syncthreads();
}

__device__ inline void wg_syrk(int p, int q, View A, View G, double alpha, double* lds) {
    if (p > 128 || q > SYRK_QMAX) { wg_gemm(p, p, q, A, tview(A), G, alpha, 0.0, lds); return; }
    GemmDesc* dsc = reinterpret_cast<GemmDesc*>(lds + GEMM_LDS_DOUBLES);
    __syncthreads();                         // nobody still reads what the tiles / descriptor alias
    if (threadIdx.x == 0) { dsc->m = p; dsc->n = p; dsc->k = q; dsc->pad = 0; dsc->A = A; dsc->C = G; dsc->alpha = alpha; dsc->beta = 0.0; dsc->amax = nullptr; }
    __syncthreads();
    TTN_SETPRIO_GEMM();
#if TTN_WG == 512
    if (p > 64) wg_syrk_impl<3, 4, 5>(dsc, lds);         // 128 rows: chunks of 48 k (145 x 48 doubles), 36 tiles on 8 waves
    else wg_syrk_impl<6, 2, 2>(dsc, lds);                // <= 64 rows: chunks of 96 k (81 x 96), 10 tiles
#else
    if (p > 64) wg_syrk_impl<6, 2, 3>(dsc, lds);         // 145 x 96 doubles, 36 tiles on 16 waves
    else wg_syrk_impl<6, 1, 1>(dsc, lds);
#endif
    TTN_SETPRIO_BASE();
}

// -------------------------------------------------------------------------------------------------
// wg_gemm_ra: C = alpha * A B for a SHORT, SHALLOW A (m <= 64 rows, k <= 128) and any n — the right factor of a bond step
// (sqrt(S) V^T = X^T M: 64 x 384 x 128), route F's two output products (64 x 128 x 64; the left one is passed transposed).
// Every wave keeps the A fragments of its 16-row tile for ALL of k in registers (k/4 doubles per lane, read once), B streams through
// LDS in chunks of 16 x (NWAVES / 4) columns — all of k at once, so one barrier pair per chunk of k/4 MFMAs per wave instead of
// one per 16 k, no A staging at all, and the loads of the next chunk are in flight (registers) during the MFMAs.  The general
// tiled GEMM ran these shapes at 21 % of the matrix pipe with two workgroups per CU.
// -------------------------------------------------------------------------------------------------
#define GEMM_RA_NMAX (2 * GEMM_TAB_ENTRIES - 320)           // column offsets of B tabulated once
__device__ TTN_NI_GEMM void wg_gemm_ra_impl(const GemmDesc* dsc_, double* lds) {
    const lds_gdesc* dsc = (const lds_gdesc*)dsc_;
    constexpr int CG = TTN_NWAVES / 4, NC = 16 * CG;                // column groups of waves; columns per chunk (32 / 64)
    constexpr int LD = (NC == 32) ? 49 : 81;                       // == 17 mod 32
    constexpr int KR = 128 / (TTN_WG / NC);                         // k rows per thread of a chunk (8)
    static_assert(128 * LD <= GEMM_LDS_DOUBLES, "a chunk holds all of k");
    const int m = uni32(dsc->m), n = uni32(dsc->n), k = uni32(dsc->k);
    const View A = ldsView(&dsc->A), B = ldsView(&dsc->B), C = ldsView(&dsc->C);
    const double alpha = unif64(dsc->alpha);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int rt = uni32(wave & 3), cg = uni32(wave >> 2);
    lds_f64* Bs = (lds_f64*)lds;                                    // Bs[kk * LD + j]
    lds_i32* rowA = (lds_i32*)(lds + GEMM_LDS_DOUBLES + 32);       // 64
    lds_i32* kA = rowA + 64;                                        // 128
    lds_i32* kB = kA + 128;                                         // 128
    lds_i32* colB = kB + 128;                                       // n
    for (int i = tid; i < 64; i += TTN_WG) rowA[i] = (i < m) ? (int)ix(A.r, i) : 0;
    for (int i = tid; i < 128; i += TTN_WG) { kA[i] = (i < k) ? (int)ix(A.c, i) : 0; kB[i] = (i < k) ? (int)ix(B.r, i) : 0; }
    for (int i = tid; i < n; i += TTN_WG) colB[i] = (int)ix(B.c, i);
    __syncthreads();
    gmem_f64* Ag = (gmem_f64*)A.p;
    gmem_f64* Bg = (gmem_f64*)B.p;
    gmem_wf64* Cg = (gmem_wf64*)C.p;
    const int ksteps = (k + 3) >> 2;
    double areg[32];
    {
        const int row = rt * 16 + li, ro = rowA[row < 64 ? row : 0];
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int kk = 4 * ks + lk;
            areg[ks] = Ag[(row < m && kk < k) ? ro + kA[kk] : 0];
        }
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) { const int kk = 4 * ks + lk; if (!(row < m && kk < k)) areg[ks] = 0.0; }
    }
    const int fx = tid % NC, sy = tid / NC;                         // column of the chunk; first k row (rows sy + (TTN_WG / NC) u)
    double v[KR];
#define RA_LOAD(C0)                                                                                                     \
    {                                                                                                                   \
        const int col = (C0) + fx;                                                                                      \
        const int co = (col < n) ? colB[col] : 0;                                                                       \
        _Pragma("unroll") for (int u = 0; u < KR; ++u) {                                                                 \
            const int kk = sy + (TTN_WG / NC) * u;                                                                      \
            v[u] = Bg[(col < n && kk < k) ? co + kB[kk] : 0];                                                           \
        }                                                                                                               \
    }
    RA_LOAD(0)
    const bool live = rt * 16 < m;                                  // wave-uniform
    for (int c0 = 0; c0 < n; c0 += NC) {
#pragma unroll
        for (int u = 0; u < KR; ++u) {
            const int kk = sy + (TTN_WG / NC) * u;
            Bs[kk * LD + fx] = (c0 + fx < n && kk < k) ? v[u] : 0.0;
        }
        __syncthreads();
        if (c0 + NC < n) RA_LOAD(c0 + NC)
        if (live && c0 + cg * 16 < n) {
            mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
            const lds_f64* bp = Bs + lk * LD + cg * 16 + li;
#pragma unroll
            for (int ks = 0; ks < 32; ++ks)
                if (ks < ksteps) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(areg[ks], bp[4 * ks * LD], acc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = rt * 16 + lk + 4 * reg, gj = c0 + cg * 16 + li;
                if (gi < m && gj < n) Cg[ix(C.r, gi) + ix(C.c, gj)] = alpha * acc[reg];
            }
        }
        __syncthreads();
    }
#undef RA_LOAD
}

// C = alpha A B; takes the register-A form when the shape allows (m <= 64, k <= 128), else the general GEMM
__device__ inline void wg_gemm_ra(int m, int n, int k, View A, View B, View C, double alpha, double* lds) {
#if TTN_WG != 512
    // (measured: inside the 1024-thread kernel — one workgroup per CU, 16 waves — the general forms are as fast or faster)
    wg_gemm(m, n, k, A, B, C, alpha, 0.0, lds);
#else
    if (m > 64 || k > 128 || n > GEMM_RA_NMAX || n < 64) { wg_gemm(m, n, k, A, B, C, alpha, 0.0, lds); return; }
    GemmDesc* dsc = reinterpret_cast<GemmDesc*>(lds + GEMM_LDS_DOUBLES);
    __syncthreads();
    if (threadIdx.x == 0) { dsc->m = m; dsc->n = n; dsc->k = k; dsc->pad = 0; dsc->A = A; dsc->B = B; dsc->C = C; dsc->alpha = alpha; dsc->beta = 0.0; dsc->amax = nullptr; }
    __syncthreads();
    TTN_SETPRIO_GEMM();
    wg_gemm_ra_impl(dsc, lds);
    TTN_SETPRIO_BASE();
#endif
}

// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ inline double fast_rcp(double x) {       // ~2^-50 relative after one Newton step
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}
__device__ inline double fast_rsqrt2(double w) {    // two Newton steps on v_rsq_f64: full fp64 accuracy
    double y = __builtin_amdgcn_rsq(w);
    double h = 0.5 * w;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}


// -------------------------------------------------------------------------------------------------
// Blocked Householder LQ (compact WY, panel QR_NB) of a row-major p x q matrix M2 (any p, q), in place.
// rr = min(p, q) row reflectors H_j = I - tau_j v_j v_j^T act from the right:  M2 * H_0 ... H_{rr-1} = [L 0].
// On exit M2[i][c], c <= i, c < rr holds L; M2[j][j+1:] holds the scaled v_j (v_j[j] = 1 implicit).
// If Qout != null it receives the rr x q matrix Q' = first rr rows of H_{rr-1} ... H_0 (orthonormal rows,
// M = L Q'), row-major with leading dimension q.  A column-major mm x nn matrix is the row-major nn x mm
// matrix of its transpose, so the same routine is the thin QR  T = Q R  with Q = Q'^T, R = L^T.
//
// Panel: the jb x len panel is factored in LDS (global memory if it does not fit) with ONE barrier per reflector:
// every wave recomputes the reflector of row r (no broadcast), wave w applies it to panel row r+1+w, and row r is
// left unscaled until the panel is done.  Trailing matrix: C <- C - ((C V^T) T) V by three GEMM calls.
// Scratch: Vb (QR_NB x q), Wb (max(p, rr) x QR_NB), Tst (ceil(rr/QR_NB) * QR_NB^2, only with Qout) in global memory;
// Ts, Ss (QR_NB^2 each), taus (QR_NB), red in LDS behind the GEMM region.
// -------------------------------------------------------------------------------------------------

// Householder LQ of a p x qc matrix (row-major, leading dimension lds_) entirely in LDS: p reflectors, one barrier each,
// wave w applies H_r to rows r+1+w, r+1+w+16, ...  Writes L (lower triangle, p x min(p,qc)) to dst (row-major, leading
// dimension ldd); with `full` also zeros above the diagonal up to column p-1 (the L blocks of a TSQR level are read whole).
// Ss: >= 128 doubles of LDS for the diagonal.  A: the LDS image (>= p*qc doubles).
__device__ inline void lq_lds_whole(int p, int qc, const double* src, int lds_, double* dst, int ldd, double* A_, double* Ss_, bool full) {
    // LDS address space pointers: through generic pointers every access would be a FLAT instruction
    lds_f64* A = (lds_f64*)A_;
    lds_f64* Ss = (lds_f64*)Ss_;
    const int tid = threadIdx.x;
    const int rr = min(p, qc);
    // One row per group of 16 lanes (64 rows per pass): the dot product <row_i, v> is a reduction over the 16 lanes of a DPP row
    // (row16_sum) instead of a whole-wave reduction per row, and the group that updates row r+1 also produces the NEXT reflector's
    // scalars (beta, tau, scale) as a by-product — nothing in a step is computed redundantly by all waves.
    const int grp = tid >> 4, l16 = tid & 15;
    lds_f64* par = Ss + 128;                            // [2][4]: beta, tau, scale of the current / next reflector
    __syncthreads();
    gmem_f64* srcg = (gmem_f64*)src;                              // M2 is global memory: global_load / global_store, not FLAT
    gmem_wf64* dstg = (gmem_wf64*)dst;
    wg_batched<8>((long long)p * qc, [&](long long e) { return srcg[(e / qc) * lds_ + (e % qc)]; }, [&](long long e, double v) { A[e] = v; });
    __syncthreads();
    if (grp == 0) {                                               // reflector 0
        double s = 0.0;
        for (int c = 1 + l16; c < qc; c += 16) { const double v = A[c]; s = fma(v, v, s); }
        const double xnorm2 = row16_sum(s), alpha = A[0];
        double tau = 0.0, scal = 0.0, beta = alpha;
        if (xnorm2 > 0.0) { beta = -copysign(sqrt(fma(alpha, alpha, xnorm2)), alpha); tau = (beta - alpha) / beta; scal = 1.0 / (alpha - beta); }
        if (l16 == 0) { par[0] = beta; par[1] = tau; par[2] = scal; }
    }
    __syncthreads();
    for (int r = 0; r < rr; ++r) {
        const lds_f64* pr = par + 4 * (r & 1);
        lds_f64* pn = par + 4 * ((r + 1) & 1);
        const double beta = pr[0], tau = pr[1], scal = pr[2];
        const lds_f64* row = A + r * qc;
        if (tid == 0) Ss[r] = beta;
        for (int i = r + 1 + grp; i < p; i += TTN_WG / 16) {
            lds_f64* ri = A + i * qc;
            // (LDS latency: four independent column groups per trip, so the loads of a trip are in flight together)
            if (tau != 0.0) {
                double w = 0.0;
                for (int c = r + 1 + l16; c < qc; c += 64) {
                    const bool k1 = c + 16 < qc, k2 = c + 32 < qc, k3 = c + 48 < qc;
                    const double a0 = ri[c], b0 = row[c];
                    const double a1 = k1 ? ri[c + 16] : 0.0, b1 = k1 ? row[c + 16] : 0.0;
                    const double a2 = k2 ? ri[c + 32] : 0.0, b2 = k2 ? row[c + 32] : 0.0;
                    const double a3 = k3 ? ri[c + 48] : 0.0, b3 = k3 ? row[c + 48] : 0.0;
                    w = fma(a0, b0, fma(a1, b1, fma(a2, b2, fma(a3, b3, w))));
                }
                w = fma(scal, row16_sum(w), ri[r]);
                const double tws = tau * w * scal;
                for (int c = r + 1 + l16; c < qc; c += 64) {
                    const bool k1 = c + 16 < qc, k2 = c + 32 < qc, k3 = c + 48 < qc;
                    const double a0 = ri[c], b0 = row[c];
                    const double a1 = k1 ? ri[c + 16] : 0.0, b1 = k1 ? row[c + 16] : 0.0;
                    const double a2 = k2 ? ri[c + 32] : 0.0, b2 = k2 ? row[c + 32] : 0.0;
                    const double a3 = k3 ? ri[c + 48] : 0.0, b3 = k3 ? row[c + 48] : 0.0;
                    ri[c] = fma(-tws, b0, a0);
                    if (k1) ri[c + 16] = fma(-tws, b1, a1);
                    if (k2) ri[c + 32] = fma(-tws, b2, a2);
                    if (k3) ri[c + 48] = fma(-tws, b3, a3);
                }
                if (l16 == 0) ri[r] -= tau * w;
            }
            if (i == r + 1 && r + 1 < rr) {                        // the next reflector from the freshly updated row r+1
                double s = 0.0;
                for (int c = r + 2 + l16; c < qc; c += 64) {
                    const double v0 = ri[c], v1 = (c + 16 < qc) ? ri[c + 16] : 0.0, v2 = (c + 32 < qc) ? ri[c + 32] : 0.0, v3 = (c + 48 < qc) ? ri[c + 48] : 0.0;
                    s = fma(v0, v0, fma(v1, v1, fma(v2, v2, fma(v3, v3, s))));
                }
                const double xnorm2 = row16_sum(s), alpha = ri[r + 1];
                double tau2 = 0.0, scal2 = 0.0, beta2 = alpha;
                if (xnorm2 > 0.0) { beta2 = -copysign(sqrt(fma(alpha, alpha, xnorm2)), alpha); tau2 = (beta2 - alpha) / beta2; scal2 = 1.0 / (alpha - beta2); }
                if (l16 == 0) { pn[0] = beta2; pn[1] = tau2; pn[2] = scal2; }
            }
        }
        __syncthreads();
    }
    const int ncol = full ? p : rr;
    for (int e = tid; e < p * ncol; e += TTN_WG) {
        const int r = e / ncol, c = e % ncol;
        if (c <= r && c < rr) dstg[(long long)r * ldd + c] = (c == r) ? Ss[r] : A[r * qc + c];
        else if (full) dstg[(long long)r * ldd + c] = 0.0;
    }
    __syncthreads();
}

// The same with the explicit thin Q' (p x q, orthonormal rows, M = L Q'): factor in LDS, then E = [I 0] times
// H_{p-1} ... H_0 in a second LDS image (2*p*q doubles in all: 64x128 cores of rank-64 trains fit).  In place: M2 receives L.
// Ts (>= 128 doubles): tau_r; Ss (>= 256 doubles): beta_r and the scale 1/(alpha - beta) of the stored reflectors.
__device__ inline void lq_lds_whole_q(int p, int q, double* M2, int ld, double* Qout, double* A_, double* Ss_, double* Ts_) {
    // same organisation as lq_lds_whole: LDS address-space pointers (no FLAT instructions), one row per group of 16 lanes with DPP
    // row sums, the next reflector's scalars as a by-product of the update of row r+1
    const int tid = threadIdx.x;
    const int grp = tid >> 4, l16 = tid & 15;
    lds_f64* A = (lds_f64*)A_;
    lds_f64* Ss = (lds_f64*)Ss_;
    lds_f64* Ts = (lds_f64*)Ts_;
    lds_f64* E = A + p * q;
    lds_f64* scl = Ss + 128;
    lds_f64* par = Ts + 128;                                      // [2][4]: beta, tau, scale of the current / next reflector
    gmem_wf64* M2g = (gmem_wf64*)M2;
    gmem_wf64* Qg = (gmem_wf64*)Qout;
    __syncthreads();
    for (int e = tid; e < p * q; e += TTN_WG) {
        const int r = e / q, c = e % q;
        A[e] = M2g[(long long)r * ld + c];
        E[e] = (r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (grp == 0) {
        double s = 0.0;
        for (int c = 1 + l16; c < q; c += 16) { const double v = A[c]; s = fma(v, v, s); }
        const double xnorm2 = row16_sum(s), alpha = A[0];
        double tau = 0.0, scal = 0.0, beta = alpha;
        if (xnorm2 > 0.0) { beta = -copysign(sqrt(fma(alpha, alpha, xnorm2)), alpha); tau = (beta - alpha) / beta; scal = 1.0 / (alpha - beta); }
        if (l16 == 0) { par[0] = beta; par[1] = tau; par[2] = scal; }
    }
    __syncthreads();
    for (int r = 0; r < p; ++r) {
        const lds_f64* pr = par + 4 * (r & 1);
        lds_f64* pn = par + 4 * ((r + 1) & 1);
        const double beta = pr[0], tau = pr[1], scal = pr[2];
        const lds_f64* row = A + r * q;
        if (tid == 0) { Ss[r] = beta; scl[r] = scal; Ts[r] = tau; }
        for (int i = r + 1 + grp; i < p; i += TTN_WG / 16) {
            lds_f64* ri = A + i * q;
            if (tau != 0.0) {
                double w = 0.0;
                for (int c = r + 1 + l16; c < q; c += 64) {
                    const bool k1 = c + 16 < q, k2 = c + 32 < q, k3 = c + 48 < q;
                    const double a0 = ri[c], b0 = row[c];
                    const double a1 = k1 ? ri[c + 16] : 0.0, b1 = k1 ? row[c + 16] : 0.0;
                    const double a2 = k2 ? ri[c + 32] : 0.0, b2 = k2 ? row[c + 32] : 0.0;
                    const double a3 = k3 ? ri[c + 48] : 0.0, b3 = k3 ? row[c + 48] : 0.0;
                    w = fma(a0, b0, fma(a1, b1, fma(a2, b2, fma(a3, b3, w))));
                }
                w = fma(scal, row16_sum(w), ri[r]);
                const double tws = tau * w * scal;
                for (int c = r + 1 + l16; c < q; c += 16) ri[c] = fma(-tws, row[c], ri[c]);
                if (l16 == 0) ri[r] -= tau * w;
            }
            if (i == r + 1 && r + 1 < p) {
                double s = 0.0;
                for (int c = r + 2 + l16; c < q; c += 16) { const double v = ri[c]; s = fma(v, v, s); }
                const double xnorm2 = row16_sum(s), alpha = ri[r + 1];
                double tau2 = 0.0, scal2 = 0.0, beta2 = alpha;
                if (xnorm2 > 0.0) { beta2 = -copysign(sqrt(fma(alpha, alpha, xnorm2)), alpha); tau2 = (beta2 - alpha) / beta2; scal2 = 1.0 / (alpha - beta2); }
                if (l16 == 0) { pn[0] = beta2; pn[1] = tau2; pn[2] = scal2; }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < p * p; e += TTN_WG) {                 // L into M2 (the caller reads c <= r only)
        const int r = e / p, c = e % p;
        if (c <= r) M2g[(long long)r * ld + c] = (c == r) ? Ss[r] : A[r * q + c];
    }
    // E <- E H_r for r = p-1 .. 0: rows i < r are still unit vectors orthogonal to v_r, so only rows i >= r change
    for (int r = p - 1; r >= 0; --r) {
        const lds_f64* row = A + r * q;                          // v_r = [0.., 1 at r, row[c] * scl[r] for c > r]
        const double tau = Ts[r], sc = scl[r];
        for (int i = r + grp; tau != 0.0 && i < p; i += TTN_WG / 16) {
            lds_f64* ei = E + i * q;
            double w = 0.0;
            for (int c = r + 1 + l16; c < q; c += 64) {
                const bool k1 = c + 16 < q, k2 = c + 32 < q, k3 = c + 48 < q;
                const double a0 = ei[c], b0 = row[c];
                const double a1 = k1 ? ei[c + 16] : 0.0, b1 = k1 ? row[c + 16] : 0.0;
                const double a2 = k2 ? ei[c + 32] : 0.0, b2 = k2 ? row[c + 32] : 0.0;
                const double a3 = k3 ? ei[c + 48] : 0.0, b3 = k3 ? row[c + 48] : 0.0;
                w = fma(a0, b0, fma(a1, b1, fma(a2, b2, fma(a3, b3, w))));
            }
            w = fma(sc, row16_sum(w), ei[r]);
            const double tws = tau * w * sc;
            for (int c = r + 1 + l16; c < q; c += 16) ei[c] = fma(-tws, row[c], ei[c]);
            if (l16 == 0) ei[r] -= tau * w;
        }
        __syncthreads();
    }
    for (int e = tid; e < p * q; e += TTN_WG) Qg[e] = E[e];
    __syncthreads();
}

// Block-reflector update with everything but C in LDS:  C (rows x len, row-major in GLOBAL memory, leading dimension ldc)
//     C <- C - ((C V^T) Top) V,   V (16 x len, LDS, leading dimension ldv), Top = T or T^T (16 x 16, LDS, row-major).
// One wave per tile of 16 rows of C, three fp64-MFMA passes (W = C V^T: len/4 MFMAs; W' = W Top: 4; C -= W' V: len/4), no
// workgroup barrier inside (a wave only needs its own W).  Replaces three generic GEMM calls per panel whose inner /
// outer dimension is 16 (prologue-dominated: ~150 k clk per panel against ~25 k here).  Rows of V beyond the panel height
// must be zero.  Wl: 256 doubles of LDS per wave.
__device__ inline int blkref_ldv(int len) { return ((len + 27) / 32) * 32 + 4; }     // >= len, == 4 mod 32: the 16 V rows a pass-1 read touches spread over the banks
__device__ inline void wg_block_reflector_apply(int rows, int len, double* Cg, int ldc, const double* Vl_, int ldv, const double* Tl_,
                                                bool transT, double* Wl_) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = TTN_WG >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const lds_f64* V = (const lds_f64*)Vl_;
    const lds_f64* T = (const lds_f64*)Tl_;
    lds_f64* W = (lds_f64*)Wl_ + wave * 256;
    const int ntile = (rows + 15) >> 4;
    for (int tile = wave; tile < ntile; tile += nwaves) {
        const int r0 = tile << 4;
        const bool rok = r0 + li < rows;
        const double* crow = Cg + (long long)(r0 + (rok ? li : 0)) * ldc;
        mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < len; k0 += 4) {                                   // W = C_tile V^T
            const int kk = k0 + lk;
            const double a = (rok && kk < len) ? crow[kk] : 0.0;
            const double b = (kk < len) ? V[li * ldv + kk] : 0.0;               // B[k][j] = V[j][k]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) W[(lk + 4 * reg) * 16 + li] = acc[reg];
        __builtin_amdgcn_wave_barrier();
        mfma_acc_t acc2 = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                           // W' = W Top
            const int kk = 4 * t + lk;
            const double a = W[li * 16 + kk];
            const double b = transT ? T[li * 16 + kk] : T[kk * 16 + li];        // B[k][j] = Top[k][j]
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) W[(lk + 4 * reg) * 16 + li] = acc2[reg];
        __builtin_amdgcn_wave_barrier();
        double at[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) at[t] = W[li * 16 + 4 * t + lk];
        for (int c0 = 0; c0 < len; c0 += 16) {                                  // C_tile -= W' V
            const bool cok = c0 + li < len;
            mfma_acc_t acc3 = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double b = cok ? V[(4 * t + lk) * ldv + c0 + li] : 0.0;
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(at[t], b, acc3, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = r0 + lk + 4 * reg;
                if (row < rows && cok) Cg[(long long)row * ldc + c0 + li] -= acc3[reg];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __noinline__ void wg_lq_blocked(int p, int q, double* M2, int ld, double* Vb, double* Wb, double* Tst, double* Qout,
                                           double* lds_gemm, double* Ts, double* Ss, double* taus, double* red) {
    // arguments of an out-of-line function arrive in VGPRs: pin the workgroup-uniform ones to SGPRs
    p = uni32(p); q = uni32(q); ld = uni32(ld);
    M2 = unip(M2); Vb = unip(Vb); Wb = unip(Wb); Tst = unip(Tst); Qout = unip(Qout);
    lds_gemm = unip(lds_gemm); Ts = unip(Ts); Ss = unip(Ss); taus = unip(taus); red = unip(red);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = TTN_WG >> 6;
    const int rr = min(p, q);
#ifndef TTN_NO_LDS_LQ
    if (Qout && rr == p && p <= 128 && 2LL * p * q <= GEMM_LDS_DOUBLES) {       // small cores: factor AND explicit Q' in LDS
        lq_lds_whole_q(p, q, M2, ld, Qout, lds_gemm, Ss, Ts);
        return;
    }
    // ---- matrices that fit the LDS whole (the small H-route steps of a sweep: 96x128, 64x128, ...): ONE "panel", no
    //      trailing GEMMs at all.  Wider ones with p <= 88 rows go through the LDS in column chunks of width w >= 2p
    //      (TSQR): L_i of every chunk is written back side by side, [L_1 L_2 ...] is factored again until one chunk is left
    //      — M M^T = sum L_i L_i^T, so the final L is an L factor of M (the callers use L only, never Q) ----
    if (!Qout && rr == p && p <= 128) {
        const int w = GEMM_LDS_DOUBLES / p;                // columns of a chunk
        if (q <= w || w >= 2 * p) {
            int qc = q;                                     // current width of the matrix held in M2[:, 0:qc]
            for (;;) {
                const int nchunk = (qc + w - 1) / w;
                for (int ci = 0; ci < nchunk; ++ci) {
                    const int c0 = ci * w, cw = min(w, qc - c0);
                    lq_lds_whole(p, cw, M2 + c0, ld, M2 + (long long)ci * p, ld, lds_gemm, Ss, nchunk > 1);
                }
                if (nchunk == 1) break;
                qc = nchunk * p;
            }
            return;
        }
    }
#endif
    double* betas = Ss;                                  // QR_NB (Ss is free while a panel is being factored)
    double* scl = Ss + QR_NB;                            // QR_NB
    for (int j0 = 0; j0 < rr; j0 += QR_NB) {
        const int jb = min(QR_NB, rr - j0);
        const int len = q - j0;
        const bool in_lds = (long long)jb * len <= GEMM_LDS_DOUBLES;
        // fast trailing update (wg_block_reflector_apply): the panel is factored in LDS with the padded leading dimension
        // ldv and then turned into the explicit V in place; 16 rows of V + 256 doubles of W per wave must fit the region
        const int ldv = blkref_ldv(len);
        const bool fastp = (long long)QR_NB * ldv + 256 * (TTN_WG / 64) <= GEMM_LDS_DOUBLES;
        double* Pn = in_lds ? lds_gemm : (M2 + (long long)j0 * ld + j0);
        const int ldp = fastp ? ldv : (in_lds ? len : ld);
        __syncthreads();
        if (in_lds) {
            for (int e = tid; e < jb * len; e += TTN_WG) { const int r = e / len, c = e - r * len; Pn[(long long)r * ldp + c] = M2[(long long)(j0 + r) * ld + j0 + c]; }
            __syncthreads();
        }
        // ---- panel factorisation ----
        for (int r = 0; r < jb; ++r) {
            const double* row = Pn + (long long)r * ldp;
            double s = 0.0;
            for (int c = r + 1 + lane; c < len; c += 64) { const double v = row[c]; s = fma(v, v, s); }
            const double xnorm2 = wave64_sum_fast(s);
            const double alpha = row[r];
            double tau = 0.0, scal = 0.0, beta = alpha;
            if (xnorm2 > 0.0) {
                beta = -copysign(sqrt(fma(alpha, alpha, xnorm2)), alpha);
                tau = (beta - alpha) / beta;
                scal = 1.0 / (alpha - beta);
            }
            if (tid == 0) { taus[r] = tau; scl[r] = scal; betas[r] = beta; }
            for (int i = r + 1 + wave; tau != 0.0 && i < jb; i += nwaves) {   // wave w applies H_r to panel rows r+1+w, ...
                double* ri = Pn + (long long)i * ldp;
                double w = 0.0;
                for (int c = r + 1 + lane; c < len; c += 64) w = fma(ri[c], row[c], w);
                w = fma(scal, wave64_sum_fast(w), ri[r]);
                const double tws = tau * w * scal;
                for (int c = r + 1 + lane; c < len; c += 64) ri[c] = fma(-tws, row[c], ri[c]);
                if (lane == 0) ri[r] -= tau * w;
            }
            __syncthreads();
        }
        // ---- write the panel back (L entries, beta on the diagonal, scaled v to the right) and the explicit V ----
        for (int e = tid; e < QR_NB * len; e += TTN_WG) {
            const int r = e / len, c = e - r * len;
            if (r < jb) {
                const double x = Pn[(long long)r * ldp + c];
                const double out = (c < r) ? x : ((c == r) ? betas[r] : x * scl[r]);
                const double vv = (c < r) ? 0.0 : ((c == r) ? 1.0 : out);
                if (fastp) Pn[(long long)r * ldp + c] = vv;               // the LDS panel becomes V (same element, same thread)
                else Vb[(long long)r * len + c] = vv;
                M2[(long long)(j0 + r) * ld + j0 + c] = out;
            } else if (fastp) Pn[(long long)r * ldp + c] = 0.0;            // rows of V beyond a short last panel
        }
        __syncthreads();
        if (fastp) {                                                      // betas / scl (which live in Ss) have been consumed
            for (int e = tid; e < QR_NB * QR_NB; e += TTN_WG) Ss[e] = 0.0;
            __syncthreads();
        }
        const int rows_t = p - (j0 + jb);
        if (rows_t <= 0 && !Qout) break;
        View Vv = mkview(Vb, plain(len), plain(1));                       // jb x len
        // S = V V^T, then T (upper triangular): T[j][j] = tau_j ; T[i][j] = -tau_j * sum_{l=i}^{j-1} T[i][l] S[l][j]
        if (fastp) {
            // split-K MFMA: wave w sums its slice of k, the 16 partial tiles meet in Ss by LDS atomic adds
            const int li = lane & 15, lk = lane >> 4;
            const int kc = (((len + nwaves - 1) / nwaves) + 3) & ~3;
            const lds_f64* V = (const lds_f64*)Pn;
            mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
            for (int k0 = wave * kc; k0 < min(len, (wave + 1) * kc); k0 += 4) {
                const int kk = k0 + lk;
                const double a = (kk < len) ? V[li * ldp + kk] : 0.0;      // A[i][k] = V[i][k], B[k][j] = V[j][k]: the same register
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) atomicAdd(&Ss[(lk + 4 * reg) * QR_NB + li], acc[reg]);
            __syncthreads();
        } else {
            // panel too long for the LDS form (V is in global memory): 256 dot products, one wave each.  NOT a wg_gemm call: Ss is
            // LDS and the GEMM stores through global-address-space pointers (that call faulted on the first matrix long enough to
            // get here: q > 412 in the 512-thread build, q > 796 in the 1024-thread one)
            for (int e = wave; e < jb * jb; e += nwaves) {
                const int i = e / jb, j = e - i * jb;
                double a = 0.0;
                for (int c = lane; c < len; c += 64) a = fma(Vb[(long long)i * len + c], Vb[(long long)j * len + c], a);
                a = wave_sum(a);
                if (lane == 0) Ss[i * QR_NB + j] = a;
            }
            __syncthreads();
        }
        if (tid < QR_NB) {                                // lane i builds row i of T: it only needs its own row
            const int i = tid;
            for (int j = 0; j < QR_NB; ++j) Ts[i * QR_NB + j] = 0.0;
            if (i < jb) {
                Ts[i * QR_NB + i] = taus[i];
                for (int j = i + 1; j < jb; ++j) {
                    double acc = 0.0;
                    for (int l = i; l < j; ++l) acc = fma(Ts[i * QR_NB + l], Ss[l * QR_NB + j], acc);
                    Ts[i * QR_NB + j] = -taus[j] * acc;
                }
            }
        }
        __syncthreads();
        if (Qout) for (int e = tid; e < QR_NB * QR_NB; e += TTN_WG) Tst[(long long)(j0 / QR_NB) * QR_NB * QR_NB + e] = Ts[e];
        if (rows_t > 0 && fastp) {
            wg_block_reflector_apply(rows_t, len, M2 + (long long)(j0 + jb) * ld + j0, ld, Pn, ldp, Ts, false, Pn + (long long)QR_NB * ldp);
        } else if (rows_t > 0) {
            // W = C V^T (rows_t x jb), C = M2[j0+jb:, j0:] ; W <- W T ; C <- C - W V
            View Cv = mkview(M2 + (long long)(j0 + jb) * ld + j0, plain(ld), plain(1));
            View Wv = mkview(Wb, plain(QR_NB), plain(1));
            wg_gemm(rows_t, jb, len, Cv, tview(Vv), Wv, 1.0, 0.0, lds_gemm);
            for (int i = tid; i < rows_t; i += TTN_WG) {
                double w[QR_NB], o[QR_NB];
                for (int l = 0; l < jb; ++l) w[l] = Wb[(long long)i * QR_NB + l];
                for (int c = 0; c < jb; ++c) {
                    double acc = 0.0;
                    for (int l = 0; l <= c; ++l) acc = fma(w[l], Ts[l * QR_NB + c], acc);
                    o[c] = acc;
                }
                for (int c = 0; c < jb; ++c) Wb[(long long)i * QR_NB + c] = o[c];
            }
            __syncthreads();
            wg_gemm(rows_t, len, jb, Wv, Vv, Cv, -1.0, 1.0, lds_gemm);
        }
    }
    __syncthreads();
    if (!Qout) return;
    // ---- Q' = [I 0] H_{rr-1} ... H_0, panels in reverse: Qs <- Qs - ((Qs V^T) T^T) V on Qs = Q'[j0:, j0:] ----
    for (long long e = tid; e < (long long)rr * q; e += TTN_WG) { const int i = (int)(e / q), c = (int)(e - (long long)i * q); Qout[e] = (i == c) ? 1.0 : 0.0; }
    __syncthreads();
    for (int j0 = ((rr - 1) / QR_NB) * QR_NB; j0 >= 0; j0 -= QR_NB) {
        const int jb = min(QR_NB, rr - j0);
        const int len = q - j0;
        const int ldv = blkref_ldv(len);
        const bool fastp = (long long)QR_NB * ldv + 256 * (TTN_WG / 64) <= GEMM_LDS_DOUBLES;
        __syncthreads();
        if (fastp) {
            for (int e = tid; e < QR_NB * len; e += TTN_WG) {
                const int r = e / len, c = e - r * len;
                lds_gemm[(long long)r * ldv + c] = (r >= jb || c < r) ? 0.0 : ((c == r) ? 1.0 : M2[(long long)(j0 + r) * ld + j0 + c]);
            }
        } else
        for (int e = tid; e < jb * len; e += TTN_WG) {
            const int r = e / len, c = e - r * len;
            Vb[e] = (c < r) ? 0.0 : ((c == r) ? 1.0 : M2[(long long)(j0 + r) * ld + j0 + c]);
        }
        for (int e = tid; e < QR_NB * QR_NB; e += TTN_WG) Ts[e] = Tst[(long long)(j0 / QR_NB) * QR_NB * QR_NB + e];
        __syncthreads();
        const int rows_q = rr - j0;
        if (fastp) {
            wg_block_reflector_apply(rows_q, len, Qout + (long long)j0 * q + j0, q, lds_gemm, ldv, Ts, true, lds_gemm + (long long)QR_NB * ldv);
            continue;
        }
        View Vv = mkview(Vb, plain(len), plain(1));
        View Qs = mkview(Qout + (long long)j0 * q + j0, plain(q), plain(1));
        View Wv = mkview(Wb, plain(QR_NB), plain(1));
        wg_gemm(rows_q, jb, len, Qs, tview(Vv), Wv, 1.0, 0.0, lds_gemm);
        for (int i = tid; i < rows_q; i += TTN_WG) {       // W <- W T^T : o[c] = sum_{l >= c} w[l] T[c][l]
            double w[QR_NB], o[QR_NB];
            for (int l = 0; l < jb; ++l) w[l] = Wb[(long long)i * QR_NB + l];
            for (int c = 0; c < jb; ++c) {
                double acc = 0.0;
                for (int l = c; l < jb; ++l) acc = fma(w[l], Ts[c * QR_NB + l], acc);
                o[c] = acc;
            }
            for (int c = 0; c < jb; ++c) Wb[(long long)i * QR_NB + c] = o[c];
        }
        __syncthreads();
        wg_gemm(rows_q, len, jb, Wv, Vv, Qs, -1.0, 1.0, lds_gemm);
    }
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------
// One-sided (Hestenes) Jacobi on the columns of X (m x p, column-major, leading dimension ldx):
// X <- X*W with orthogonal columns; on exit column i = sigma_i * u_i.  Round-robin parallel
// ordering, one wave per column pair, lanes over rows.  Returns the number of sweeps used
// (negative if the sweep limit was hit).  X may live in LDS or in global memory.
// -------------------------------------------------------------------------------------------------
__device__ __noinline__ int wg_jacobi_cols(int m, int p, double* X, int ldx, int* flag /*LDS*/, double* red /*LDS*/, double tol_mult, double neg_mult, double* aneg_out /*LDS*/) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = TTN_WG >> 6;
    if (p < 2) { if (tid == 0) *aneg_out = 0.0; __syncthreads(); return 0; }
    const int pe = p + (p & 1);               // even number of slots; slot p (if odd) is a bye
    const int half = pe >> 1;
    const double tol = tol_mult * sqrt((double)m) * DBL_EPSILON;
    // columns whose norm is below eps*sqrt(m)*(largest column norm) carry singular values below what fp64
    // can resolve against sigma_max; they are left alone (rotating pure rounding noise never converges)
    double amax = 0.0;
    for (int c = wave; c < p; c += nwaves) {
        double a = 0.0;
        for (int r = lane; r < m; r += 64) { const double v = X[(long long)c * ldx + r]; a = fma(v, v, a); }
        amax = fmax(amax, wave_sum(a));
    }
    amax = wg_max(amax, red);
    const double aneg = neg_mult * neg_mult * (double)m * DBL_EPSILON * DBL_EPSILON * amax;
    if (tid == 0) *aneg_out = aneg;
    int sweep = 0;
    for (; sweep < JACOBI_MAX_SWEEPS; ++sweep) {
        if (tid == 0) *flag = 0;
        __syncthreads();
        int rotated = 0;
        for (int round = 0; round < pe - 1; ++round) {
            for (int kk = wave; kk < half; kk += nwaves) {
                int i, j;
                if (kk == 0) { i = round; j = pe - 1; }
                else { i = (round + kk) % (pe - 1); j = (round + pe - 1 - kk) % (pe - 1); }
                if (i > j) { const int t = i; i = j; j = t; }
                if (j >= p) continue;          // bye
                double* xi = X + (long long)i * ldx;
                double* xj = X + (long long)j * ldx;
                double a = 0.0, b = 0.0, g = 0.0;
                for (int r = lane; r < m; r += 64) {
                    const double u = xi[r], v = xj[r];
                    a = fma(u, u, a); b = fma(v, v, b); g = fma(u, v, g);
                }
                a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
                if (a <= aneg || b <= aneg) continue;
                if (fabs(g) <= tol * sqrt(a) * sqrt(b)) continue;
                // rotation annihilating g (Rutishauser formulas)
                const double zeta = (b - a) / (2.0 * g);
                const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
                const double cs = 1.0 / sqrt(fma(t, t, 1.0));
                const double sn = cs * t;
                for (int r = lane; r < m; r += 64) {
                    const double u = xi[r], v = xj[r];
                    xi[r] = cs * u - sn * v;
                    xj[r] = sn * u + cs * v;
                }
                rotated = 1;
            }
            __syncthreads();
        }
        if (rotated && lane == 0) atomicOr(flag, 1);
        __syncthreads();
        const int any = *flag;
        __syncthreads();
        if (!any) return sweep + 1;
    }
    return -sweep;
}


// -------------------------------------------------------------------------------------------------
// Fast path of the one-sided Jacobi for p <= 128 columns of length m <= 128, X resident in LDS with a
// FIXED leading dimension of 128 doubles (column c at X + 128*c).
//   * a wave works on 4 column pairs at once: 16 lanes per pair, lane `sub` owns rows sub + 16*t;
//     groups 1 and 3 walk t rotated by one so the two pairs of a 32-lane half hit disjoint LDS banks;
//   * the 64 pairs of a round-robin round are spread over the 16 waves (pair = 4*wave + group);
//   * squared column norms are cached in LDS (recomputed at the start of every sweep, updated by
//     a' = a - t*g, b' = b + t*g), so a rotation costs one dot product;
//   * reductions over the 16 lanes of a pair are 4 DPP steps (quad_perm, row_half_mirror, row_mirror);
//   * the rotation (t, c, s) uses v_rcp_f64 / v_rsq_f64 seeds: t only needs ~2^-40 relative accuracy
//     (it sets the speed of convergence), c = rsqrt(1+t^2) gets two Newton steps so that c^2+s^2 = 1
//     to rounding (that is what makes every applied rotation orthogonal, i.e. backward stable).
// -------------------------------------------------------------------------------------------------
typedef double lds_f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) lds_f64x2 lds_v2;

// All-reduce over the 16 lanes of a Jacobi lane group, result in every lane of the group.  MUST be called in
// wave-uniform control flow (the MFMA involves all 64 lanes).  The group of lane l is (l >> 2) & 3, i.e. the quad at
// position g of each of the four 16-lane rows.  Two v_mfma_f64_4x4x4 against a matrix of ones do the reduction in the
// matrix pipe: the instruction computes D_B[i][j] = sum_k A_B[i][k] with A_B[i][k] in lane B + 4i + 16k and D_B[i][j]
// in lane 16B + 4i + j (layout measured on gfx950, scratch/mfma444_test.hip), so the first sums over k, the second
// over B — together the 16 lanes with the same i.  Replaces 8 v_mov_dpp + 4 v_add_f64 + wait states on the VALU.
__device__ inline double grp_sum(double v) {
    const double s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    return __builtin_amdgcn_mfma_f64_4x4x4f64(s1, 1.0, 0.0, 0, 0, 0);
}
// Hand a value to the previous lane group: lane l receives from lane l + 4 of its 16-lane row (row_ror:12), i.e. member
// s of group g receives from member s of group (g + 1) mod 4.  Wave-uniform control flow only.
// (the same hand-over through the LDS crossbar, ds_bpermute_b32, measured 25 % slower per sweep: 279.5 vs 224.1 k clk)
__device__ inline double grp_from_next(double v) { return dpp_mov_f64<0x12C>(v); }

// One sweep-loop instance.  A column is owned by a group of 16 lanes; a lane holds NT2 16-byte pieces of it, so the
// instance reads 32*NT2 rows of every column (rows [m, 32*NT2) must be zero on entry).  A wave has 4 lane groups and
// works on blocks of 4 columns.  Group g = (lane >> 2) & 3, member s = (lane & 3) + 4*(lane >> 4) (see grp_sum); the
// member owns the row pairs {2s, 2s+1} + 32*t.
//
// What bounds the sweep (measured, profiles/README.md "Jacobi"): every rotation used to store its moving column to
// LDS (1 KiB) and ds_write_b128 moves 79 B/clk per CU — 2.5 M stores x 13 clk = 2/3 of the Jacobi time, which neither
// fewer VALU instructions (-20 %: no change), nor 8 lanes per column, nor 8 waves instead of 16 changed.  So:
//   * the stationary column of each lane group stays in REGISTERS for a whole level;
//   * the moving block is loaded from LDS once per block round and then handed from lane group to lane group in
//     registers (16 v_mov_dpp per hand-over, VALU) — one LDS store per column and block round instead of four;
//   * the 16-lane reductions run on the matrix pipe (grp_sum).
// Control flow: which waves work is wave-uniform (branches); which lane groups of a working wave hold real columns
// (only the last block of a p not divisible by 4 has phantom columns) is a MASK — phantom groups load whatever the
// LDS holds, their dot product is discarded and they never write — so that grp_sum / grp_from_next run converged.
// LD = leading dimension of the LDS image (128: up to 128 columns of <= 128 rows; 256: up to 64 columns of <= 256 rows, the
// blocked driver below).  max_sweeps / cross_only / aneg_fixed serve that driver: one sweep per visit, optionally only the
// TOP tournament level (= exactly the cross pairs between the first and the second half of the columns), and the
// negligible-column threshold of the WHOLE matrix instead of the one of the columns at hand (aneg_fixed < 0: compute it).
template <int NT2, int LD>
__device__ int jacobi_lds128_body(int m, int p, lds_f64* X, lds_f64* nrm2, int* flag, double* red,
                                  double tol_mult, double neg_mult, double* aneg_out, int max_sweeps = JACOBI_MAX_SWEEPS,
                                  bool cross_only = false, double aneg_fixed = -1.0) {
    constexpr int CMASK = TTN_LDS_IMG / LD - 1;         // columns the LDS image holds, minus one
    constexpr int G = 4;                                // lane groups per wave = columns per block
    constexpr int CH = 32;                              // doubles per 16-byte-per-lane piece of a column
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = TTN_WG >> 6;
    lds_i32* flagL = (lds_i32*)flag;                    // the convergence flag lives in LDS: ds ops, not FLAT ones through the generic pointer
    const int grp = (lane >> 2) & 3;
    const int sub = (lane & 3) | ((lane >> 4) << 2);
    const int roff = 2 * sub;                           // row offset inside a piece
#define JOFF(t) (roff + CH * (t))
    const double tol = tol_mult * sqrt((double)m) * DBL_EPSILON;
    const double tol2 = tol * tol;
    double aneg = 0.0;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        // ---- refresh the cached squared norms ----
        double amax = 0.0;
        for (int cb = wave * G; cb < p; cb += nwaves * G) {
            const int c = (cb + grp) & CMASK;            // inside the LDS image even when >= p
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const lds_f64x2 v = *(lds_v2*)(X + c * LD + JOFF(t));
                a0 = fma(v.x, v.x, a0);
                a1 = fma(v.y, v.y, a1);
            }
            const double a = grp_sum(a0 + a1);
            if (c < p) {
                if (sub == 0) nrm2[c] = a;
                amax = fmax(amax, a);
            }
        }
        if (sweep == 0) {
            if (aneg_fixed >= 0.0) aneg = aneg_fixed;
            else {
                amax = wg_max(amax, red);
                aneg = neg_mult * neg_mult * (double)m * DBL_EPSILON * DBL_EPSILON * amax;
                if (tid == 0) *aneg_out = aneg;
            }
        }
        if (tid == 0) *flagL = 0;
        __syncthreads();
        int rotated = 0;
        // Block ordering.  The columns form nb blocks of 4 (nbp = nb rounded up to a power of two, phantom blocks idle).
        //  phase 0: the 6 pairs inside every block (round-robin, both columns through LDS);
        //  then log2(nbp) LEVELS of a recursive tournament: at a level the blocks are split into groups of gs, each
        //  group into a stationary half and a moving half; in round r wave (group, a) rotates its stationary block
        //  against moving block (a + r) mod h — 16 cross pairs in 4 inner rounds, the 4 lane groups of the wave taking
        //  the 4 disjoint pairs of an inner round.  nbp-1 block rounds per sweep, every pair exactly once.
        // Workgroup barriers: one per block round (the moving blocks change hands).
        const int nb = (p + G - 1) / G;
        int nbp = 1;
        while (nbp < nb) nbp <<= 1;
// t = tan(theta) of the Jacobi rotation from d = b - a, h = 2g:  t = sign(d) h / (|d| + sqrt(d^2 + h^2)); raw
// v_sqrt/v_rcp seeds are enough for t (it only sets the speed of convergence), c = rsqrt(1 + t^2) gets two Newton steps
// so that c^2 + s^2 = 1 to rounding (that is what makes every applied rotation orthogonal, i.e. backward stable).
#define JROT_MATH                                                                                             \
                const double dd_ = b - a, hh_ = g + g;                                                        \
                const double hyp_ = __builtin_amdgcn_sqrt(fma(dd_, dd_, hh_ * hh_));                          \
                const double t_ = copysign(hh_ * __builtin_amdgcn_rcp(fabs(dd_) + hyp_), hh_ * dd_ );         \
                const double w_ = fma(t_, t_, 1.0);                                                           \
                const double cs = fast_rsqrt2(w_);                                                            \
                const double sn = cs * t_;
        // ---- phase 0: pairs inside the blocks 2*slot and 2*slot+1 (circle method: local column 3 stays put) ----
        if (!cross_only)
        for (int slot0 = wave; 2 * slot0 < nb; slot0 += nwaves) {
            const int blk = 2 * slot0 + (grp >> 1), hq = grp & 1;
            const int c0 = G * blk;
#pragma unroll
            for (int rr = 0; rr < G - 1; ++rr) {
                const int li = (hq == 0) ? G - 1 : (rr + hq) % (G - 1);
                const int lj = (hq == 0) ? rr : (rr + G - 1 - hq) % (G - 1);
                const int i = (c0 + (li < lj ? li : lj)) & CMASK, j = (c0 + (li < lj ? lj : li)) & CMASK;
                const bool act = j < p && blk < nb;
                lds_f64* xi = X + i * LD;
                lds_f64* xj = X + j * LD;
                const double a = nrm2[i], b = nrm2[j];
                lds_f64x2 u[NT2], v[NT2];
                double g0 = 0.0, g1 = 0.0;
#pragma unroll
                for (int t = 0; t < NT2; ++t) {
                    u[t] = *(lds_v2*)(xi + JOFF(t));
                    v[t] = *(lds_v2*)(xj + JOFF(t));
                    g0 = fma(u[t].x, v[t].x, g0);
                    g1 = fma(u[t].y, v[t].y, g1);
                }
                const double g = grp_sum(g0 + g1);
                if (act && (a > aneg) && (b > aneg) && (g * g > tol2 * a * b)) {
                    JROT_MATH
#pragma unroll
                    for (int t = 0; t < NT2; ++t) {
                        lds_f64x2 nu, nv;
                        nu.x = fma(cs, u[t].x, -sn * v[t].x);
                        nu.y = fma(cs, u[t].y, -sn * v[t].y);
                        nv.x = fma(sn, u[t].x, cs * v[t].x);
                        nv.y = fma(sn, u[t].y, cs * v[t].y);
                        *(lds_v2*)(xi + JOFF(t)) = nu;
                        *(lds_v2*)(xj + JOFF(t)) = nv;
                    }
                    if (sub == 0) { nrm2[i] = fmax(a - t_ * g, 0.0); nrm2[j] = b + t_ * g; }
                    rotated = 1;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        // ---- levels ----
        constexpr int PPW = ((CMASK + 1) / 8 + TTN_NWAVES - 1) / TTN_NWAVES;   // block pairs a wave owns per block round (image full)
        for (int gs = nbp; gs >= (cross_only ? nbp : 2); gs >>= 1) {
            const int h = gs >> 1;
            int gam[PPW], aa[PPW], ci[PPW];
            bool wact[PPW], iact[PPW];
            lds_f64x2 u[PPW][NT2];
            double an[PPW];
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                const int slot = wave + q * nwaves;
                wact[q] = slot < (nbp >> 1);                                // wave-uniform
                gam[q] = wact[q] ? slot / h : 0;
                aa[q] = wact[q] ? slot % h : 0;
                ci[q] = (G * (gam[q] * gs + aa[q]) + grp) & CMASK;            // stationary column of this lane group
                iact[q] = wact[q] && ci[q] < p;
                an[q] = nrm2[ci[q]];
#pragma unroll
                for (int t = 0; t < NT2; ++t) u[q][t] = *(lds_v2*)(X + ci[q] * LD + JOFF(t));
            }
            for (int r = 0; r < h; ++r) {
#pragma unroll
                for (int q = 0; q < PPW; ++q) {
                    if (wact[q]) {
                        int mrot = aa[q] + r; if (mrot >= h) mrot -= h;
                        const int cjb = G * (gam[q] * gs + h + mrot);
                        // the moving column this lane group starts with; it then takes the next group's column each inner round
                        const int cj0 = (cjb + grp) & CMASK;
                        lds_f64x2 v[NT2];
#pragma unroll
                        for (int t = 0; t < NT2; ++t) v[t] = *(lds_v2*)(X + cj0 * LD + JOFF(t));
                        double b = nrm2[cj0];
                        int dirty = 0;
#pragma unroll
                        for (int sft = 0; sft < G; ++sft) {
                            const int cj = (cjb + ((grp + sft) & (G - 1))) & CMASK;
                            const bool act = iact[q] && cj < p;
                            double g0 = 0.0, g1 = 0.0;
#pragma unroll
                            for (int t = 0; t < NT2; ++t) {
                                g0 = fma(u[q][t].x, v[t].x, g0);
                                g1 = fma(u[q][t].y, v[t].y, g1);
                            }
                            const double g = grp_sum(g0 + g1);
                            const double a = an[q];
                            if (act && (a > aneg) && (b > aneg) && (g * g > tol2 * a * b)) {
                                JROT_MATH
                                const double ic = w_ * cs;                 // 1/c
                                // u' = c (u - t v);  v' = t u' + v / c  — both in place (v_fmac / v_mul on their own registers)
#pragma unroll
                                for (int t = 0; t < NT2; ++t) {
                                    u[q][t].x = cs * fma(-t_, v[t].x, u[q][t].x);
                                    u[q][t].y = cs * fma(-t_, v[t].y, u[q][t].y);
                                    v[t].x = fma(t_, u[q][t].x, ic * v[t].x);
                                    v[t].y = fma(t_, u[q][t].y, ic * v[t].y);
                                }
                                b = b + t_ * g;
                                an[q] = fmax(a - t_ * g, 0.0);
                                dirty = 1;
                            }
                            if (sft < G - 1) {
#pragma unroll
                                for (int t = 0; t < NT2; ++t) { v[t].x = grp_from_next(v[t].x); v[t].y = grp_from_next(v[t].y); }
                                b = grp_from_next(b);
                                dirty = __builtin_amdgcn_update_dpp(0, dirty, 0x12C, 0xF, 0xF, true);
                            }
                        }
                        const int cjl = (cjb + ((grp + G - 1) & (G - 1))) & CMASK;     // the column this group ends up with
                        if (dirty && cjl < p) {
#pragma unroll
                            for (int t = 0; t < NT2; ++t) *(lds_v2*)(X + cjl * LD + JOFF(t)) = v[t];
                            if (sub == 0) nrm2[cjl] = b;
                            rotated = 1;
                        }
                    }
                }
                __syncthreads();                                           // the moving blocks change hands
            }
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                if (iact[q]) {
#pragma unroll
                    for (int t = 0; t < NT2; ++t) *(lds_v2*)(X + ci[q] * LD + JOFF(t)) = u[q][t];
                    if (sub == 0) nrm2[ci[q]] = an[q];
                }
            }
            __syncthreads();
        }
#undef JROT_MATH
#undef JOFF
        if (rotated) *flagL = 1;
        __syncthreads();
        const int any = *flagL;
        __syncthreads();
        if (!any) return sweep + 1;
    }
    return -sweep;
}

// Fast path of the one-sided Jacobi: p <= 128 columns of length m <= 128 in LDS, leading dimension 128; rows
// [m, 128) of every column must be ZERO (the caller pads).  See jacobi_lds128_body.
__device__ TTN_NI_JACOBI int wg_jacobi_lds128(int m, int p, double* Xg, double* nrm2g, int* flag, double* red,
                                             double tol_mult, double neg_mult, double* aneg_out /*LDS*/) {
    if (p < 2) { if (threadIdx.x == 0) *aneg_out = 0.0; __syncthreads(); return 0; }
    lds_f64* X = (lds_f64*)Xg;
    lds_f64* nrm2 = (lds_f64*)nrm2g;
    if (m <= 32) return jacobi_lds128_body<1, 128>(m, p, X, nrm2, flag, red, tol_mult, neg_mult, aneg_out);
    if (m <= 64) return jacobi_lds128_body<2, 128>(m, p, X, nrm2, flag, red, tol_mult, neg_mult, aneg_out);
    if (m <= 96) return jacobi_lds128_body<3, 128>(m, p, X, nrm2, flag, red, tol_mult, neg_mult, aneg_out);     // the 96-column ramp step
    return jacobi_lds128_body<4, 128>(m, p, X, nrm2, flag, red, tol_mult, neg_mult, aneg_out);
}

// -------------------------------------------------------------------------------------------------
// Blocked one-sided Jacobi for 128 < p <= 256 columns of length m <= 256 (ranks 65..128): the matrix (p x p doubles, up
// to 512 KB) lives in global memory (column-major, leading dimension ldx) and is processed in column blocks of 32 through
// the LDS image (leading dimension 256, 64 columns) by the same lane-group kernel as the p <= 128 fast path:
//   per sweep   (a) every block alone: one full sweep over its 496 pairs;
//               (b) every pair of blocks (I < J): I in columns 0..31, J in 32..63, only the top tournament level =
//                   exactly the 32 x 32 cross pairs —
//   i.e. every pair of columns once per sweep, like the cyclic orderings.  A block visit reads and writes its columns once
//   (coalesced); a visit that rotated nothing is not written back.  Converged when a whole sweep rotates nothing.
// Replaces the global-memory Jacobi (wg_jacobi_cols, 3-4x slower at these sizes), which remains the fallback for p > 256.
// -------------------------------------------------------------------------------------------------
// LD = leading dimension of the LDS image (>= m), JBW = columns per block: 2 * JBW * LD doubles of image.
template <int LD, int JBW>
__device__ inline void jb_load(lds_f64* X, int col0_lds, const double* Xg, int ldx, int c0, int ncols, int m) {
    // columns c0 .. c0+ncols-1 of Xg -> LDS columns col0_lds .., rows >= m and missing columns zero-filled (JBW columns always)
    for (int e = threadIdx.x; e < JBW * LD; e += TTN_WG) {
        const int r = e % LD, c = e / LD;
        X[(col0_lds + c) * LD + r] = (c < ncols && r < m) ? Xg[(long long)(c0 + c) * ldx + r] : 0.0;
    }
}
template <int LD, int JBW>
__device__ inline void jb_store(const lds_f64* X, int col0_lds, double* Xg, int ldx, int c0, int ncols, int m) {
    for (int e = threadIdx.x; e < JBW * LD; e += TTN_WG) {
        const int r = e % LD, c = e / LD;
        if (c < ncols && r < m) Xg[(long long)(c0 + c) * ldx + r] = X[(col0_lds + c) * LD + r];
    }
}

template <int LD, int JBW>
__device__ int jacobi_blocked_body(int m, int p, double* Xg, int ldx, double* Xlds, double* nrm2g, int* flag, double* red,
                                   double tol_mult, double neg_mult, double* aneg_out /*LDS*/) {
    static_assert(2 * JBW * LD <= TTN_LDS_IMG, "two column blocks must fit the LDS image");
    m = uni32(m); p = uni32(p); ldx = uni32(ldx);
    Xg = unip(Xg); Xlds = unip(Xlds); nrm2g = unip(nrm2g); flag = unip(flag); red = unip(red); aneg_out = unip(aneg_out);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = TTN_WG >> 6;
    lds_f64* X = (lds_f64*)Xlds;
    lds_f64* nrm2 = (lds_f64*)nrm2g;
    // negligible-column threshold of the whole matrix
    double amax = 0.0;
    for (int c = wave; c < p; c += nwaves) {
        double a = 0.0;
        for (int r = lane; r < m; r += 64) { const double v = Xg[(long long)c * ldx + r]; a = fma(v, v, a); }
        amax = fmax(amax, wave_sum(a));
    }
    amax = wg_max(amax, red);
    const double aneg = neg_mult * neg_mult * (double)m * DBL_EPSILON * DBL_EPSILON * amax;
    if (tid == 0) *aneg_out = aneg;
    __syncthreads();
    const int nbk = (p + JBW - 1) / JBW;
    for (int sweep = 0; sweep < JACOBI_MAX_SWEEPS; ++sweep) {
        int any = 0;
        for (int I = 0; I < nbk; ++I) {                                           // (a) inside every block
            const int nI = min(JBW, p - I * JBW);
            __syncthreads();
            jb_load<LD, JBW>(X, 0, Xg, ldx, I * JBW, nI, m);
            __syncthreads();
            const int r = jacobi_lds128_body<LD / 32, LD>(m, nI, X, nrm2, flag, red, tol_mult, neg_mult, aneg_out, 1, false, aneg);
            if (r < 0) { any = 1; jb_store<LD, JBW>(X, 0, Xg, ldx, I * JBW, nI, m); }
        }
        for (int I = 0; I + 1 < nbk; ++I)                                          // (b) between every two blocks
            for (int J = I + 1; J < nbk; ++J) {
                const int nJ = min(JBW, p - J * JBW);
                __syncthreads();
                jb_load<LD, JBW>(X, 0, Xg, ldx, I * JBW, JBW, m);
                jb_load<LD, JBW>(X, JBW, Xg, ldx, J * JBW, nJ, m);
                __syncthreads();
                const int r = jacobi_lds128_body<LD / 32, LD>(m, JBW + nJ, X, nrm2, flag, red, tol_mult, neg_mult, aneg_out, 1, true, aneg);
                if (r < 0) {
                    any = 1;
                    jb_store<LD, JBW>(X, 0, Xg, ldx, I * JBW, JBW, m);
                    jb_store<LD, JBW>(X, JBW, Xg, ldx, J * JBW, nJ, m);
                }
            }
        __syncthreads();
        if (!any) return sweep + 1;
    }
    return -JACOBI_MAX_SWEEPS;
}
// 128 < p <= 256 (ranks 65..128): image leading dimension 256, as many columns per block as two blocks fit the image
__device__ __noinline__ int wg_jacobi_blocked256(int m, int p, double* Xg, int ldx, double* Xlds, double* nrm2g, int* flag, double* red,
                                                 double tol_mult, double neg_mult, double* aneg_out /*LDS*/) {
    return jacobi_blocked_body<256, TTN_LDS_IMG / 512>(m, p, Xg, ldx, Xlds, nrm2g, flag, red, tol_mult, neg_mult, aneg_out);
}
#if TTN_LDS_COLS < 128
// TTN_LDS_COLS < p <= 128 in the small-image build: the same blocked scheme with leading dimension 128
__device__ __noinline__ int wg_jacobi_blocked128(int m, int p, double* Xg, int ldx, double* Xlds, double* nrm2g, int* flag, double* red,
                                                 double tol_mult, double neg_mult, double* aneg_out /*LDS*/) {
    return jacobi_blocked_body<128, TTN_LDS_IMG / 256>(m, p, Xg, ldx, Xlds, nrm2g, flag, red, tol_mult, neg_mult, aneg_out);
}
#endif

// -------------------------------------------------------------------------------------------------
// Cholesky G = L L^T in LDS (column-major, leading dimension 128, n <= 128), in place: on exit the lower
// triangle holds L and the strict upper triangle is zeroed.  Returns 0, or 1 if a pivot is not safely
// positive (d_j <= n*eps*max_diag): the caller then falls back to the Householder path.
// Blocked right-looking, panel width 16, three workgroup barriers per PANEL (the unblocked form it replaces took one
// per column and ~2.7 k clk per column of LDS read-modify-write traffic: 355 k clk for n = 128):
//   (a) wave 0 factors the 16x16 diagonal block in LDS (wave-level ordering only) and leaves 1/l_kk in red[0..15];
//   (b) one thread per row below the block solves its row of L21 = A21 L11^-T in registers;
//   (c) the trailing matrix gets A22 -= L21 L21^T by fp64 MFMA, one 16x16 tile of the lower triangle per wave and trip,
//       operands read straight from the LDS image.
// -------------------------------------------------------------------------------------------------
// NTEAM = 2 (n <= 64): the workgroup splits into two teams of 8 waves that factor TWO matrices at once (team t: the matrix
// at Gg + t*64*128, scratch red + 32*t, result slot pivmin_out[t]) — the route-F step needs chol(A'^T A') and chol(B' B'^T),
// and a factorisation is dominated by the dependent pivot chain of one wave, so two in parallel cost about what one does.
// The barriers are workgroup-wide: both teams run the same control flow (same n); a bad pivot in either stops both.
template <int NTEAM>
__device__ int chol_lds128_teams(int n, double* Gg, double* red, int* flag, double* pivmin_out /*LDS*/) {
    n = uni32(n); Gg = unip(Gg); red = unip(red); flag = unip(flag); pivmin_out = unip(pivmin_out);     // VGPR arguments -> SGPRs
    constexpr int WPT = TTN_NWAVES / NTEAM;              // waves per team
    const int lane = threadIdx.x & 63;
    const int team = (threadIdx.x >> 6) / WPT, wave = (threadIdx.x >> 6) % WPT, nwaves = WPT;
    const int tid = threadIdx.x - team * WPT * 64;       // thread index inside the team
    const int li = lane & 15, lk = lane >> 4;
    lds_f64* G = (lds_f64*)Gg + team * (TTN_LDS_IMG / 2);
    lds_f64* invd = (lds_f64*)red + 32 * team;           // 1 / l_kk of the current panel ([0..15]) and the column buffer ([16..31])
    double dmax = 0.0;
    if (wave == 0) {
        for (int j = lane; j < n; j += 64) dmax = fmax(dmax, G[j * 128 + j]);
        dmax = wave_max(dmax);
        if (lane == 0) invd[0] = dmax;
    }
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    dmax = invd[0];
    __syncthreads();
    const double dmin = (double)n * DBL_EPSILON * dmax;
    double pmin = dmax;                                  // tracked by wave 0 (every lane sees every pivot)
    int bad = 0;
    // (a) diagonal block j0, wave 0, in REGISTERS: lane (li, lk) holds D[li][lk + 4q], q = 0..3.  Per column one LDS round
    //     trip: the owners publish the unscaled column, every lane reads the pivot, its row's and its columns' entries and
    //     applies the rank-1 update to its four elements.  A lone wave issues one fp64 instruction per ~9 clk, so the step
    //     is written with as few as possible: no per-element predicates.  Entries above the diagonal are PUBLISHED as zero,
    //     which makes the update of an already final column (c < jj) vanish by itself; the unpublished upper entries a lane
    //     holds just carry bounded garbage that is never read or written back.
#define CHOL_DIAG_BLOCK(J0)                                                                                         \
    {                                                                                                               \
        const int jb_ = (n - (J0) < 16) ? n - (J0) : 16;                                                            \
        lds_f64* D = G + (J0) * 128 + (J0);              /* D[c*128 + i] */                                          \
        lds_f64* colbuf = invd + 16;                     /* red[16..31] */                                           \
        double e[4];                                                                                                \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) { const int c = lk + 4 * q; e[q] = (li < jb_ && c <= li) ? D[c * 128 + li] : 0.0; } \
        _Pragma("unroll") for (int jj = 0; jj < 16; ++jj) {                                                         \
            if (jj >= jb_) break;                                                                                   \
            if (lk == (jj & 3)) colbuf[li] = (li >= jj) ? e[jj >> 2] : 0.0;                                         \
            __builtin_amdgcn_wave_barrier();                                                                        \
            const double d = colbuf[jj], ali = colbuf[li];                                                          \
            double ac[4];                                                                                           \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) ac[q] = colbuf[lk + 4 * q];                               \
            if (!(d > dmin)) { if (lane == 0) *flag = 1; break; }                                                   \
            pmin = fmin(pmin, d);                                                                                   \
            const double rs = fast_rsqrt2(d);                                                                       \
            const double lli = ali * rs, nl = -lli * rs;                                                            \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) e[q] = fma(nl, ac[q], e[q]);                              \
            /* column jj is final: l_ij = a_ij / sqrt(d) below the diagonal, sqrt(d) = d * rs on it */              \
            if (lk == (jj & 3) && li >= jj) e[jj >> 2] = lli;                                                       \
            if (lane == 0) invd[jj] = rs;                                                                           \
            __builtin_amdgcn_wave_barrier();                                                                        \
        }                                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) { const int c = lk + 4 * q; if (li < jb_ && c <= li) D[c * 128 + li] = e[q]; } \
    }
    // (c) one 16x16 tile (tr >= tc) of the trailing update A22 -= L21 L21^T of panel j0, by one wave, straight from LDS
#define CHOL_TILE(TILE)                                                                                             \
    {                                                                                                               \
        int tr = 0, base = 0;                                                                                       \
        while (base + tr + 1 <= (TILE)) { base += tr + 1; ++tr; }      /* tile = tr(tr+1)/2 + tc */                  \
        const int tc = (TILE) - base;                                                                               \
        const int r0 = s0 + 16 * tr, c0 = s0 + 16 * tc;                                                             \
        mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};                                                          \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                             \
            const lds_f64* col = G + (j0 + 4 * t + lk) * 128;                                                       \
            const double av = (r0 + li < n) ? col[r0 + li] : 0.0;                                                   \
            const double bv = (c0 + li < n) ? col[c0 + li] : 0.0;                                                   \
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                       \
            const int row = r0 + lk + 4 * reg, colj = c0 + li;                                                      \
            if (row < n && colj < n && row >= colj) G[colj * 128 + row] -= acc[reg];                                \
        }                                                                                                           \
    }
    if (wave == 0) CHOL_DIAG_BLOCK(0)
    for (int j0 = 0; j0 < n; j0 += 16) {
        const int jb = (n - j0 < 16) ? n - j0 : 16;
        __syncthreads();                                 // diagonal block j0 factored (by wave 0, during the previous trailing update)
        if (*flag) { bad = 1; break; }
        const int s0 = j0 + jb, sr = n - s0;             // first row / number of rows below the block
        if (sr > 0) {                                    // then jb == 16
            // ---- (b) rows below the block: x L11^T = a, right-looking in registers ----
            if (tid < sr) {
                const int r = s0 + tid;
                double a[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = G[(j0 + c) * 128 + r];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    a[k] *= invd[k];
#pragma unroll
                    for (int c = k + 1; c < 16; ++c) a[c] = fma(-a[k], G[(j0 + k) * 128 + j0 + c], a[c]);
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) G[(j0 + c) * 128 + r] = a[c];
            }
            __syncthreads();
            // ---- (c) trailing update with LOOKAHEAD: wave 0 takes tile 0 (= the next diagonal block) and factors it right
            //      away — the long dependent chain of (a) — while the other waves do the remaining tiles ----
            const int nt = (sr + 15) >> 4, ntiles = nt * (nt + 1) / 2;
            if (wave == 0) {
                CHOL_TILE(0)
                CHOL_DIAG_BLOCK(s0)
            } else {
                for (int tile = wave; tile < ntiles; tile += nwaves - 1) CHOL_TILE(tile)
            }
        }
    }
    __syncthreads();
#undef CHOL_DIAG_BLOCK
#undef CHOL_TILE
    // pivots lie between the extreme eigenvalues of G, so dmax/pmin is a LOWER bound of cond(G) = cond(M)^2
    if (tid == 0) pivmin_out[team] = (!bad && pmin > 0.0) ? dmax / pmin : 1.0e300;
    for (int e = tid; e < n * 128; e += WPT * 64) { const int c = e >> 7, i = e & 127; if (i < c) G[c * 128 + i] = 0.0; }
    __syncthreads();
    return bad;
}

__device__ TTN_NI_CHOL int wg_chol_lds128(int n, double* Gg, double* red, int* flag, double* pivmin_out /*LDS*/) {
    return chol_lds128_teams<1>(n, Gg, red, flag, pivmin_out);
}
// two n x n matrices (n <= TTN_LDS_COLS / 2) at Gg and Gg + TTN_LDS_IMG / 2; red: 64 doubles; pivmin_out: 2 doubles
__device__ TTN_NI_CHOL int wg_chol2_lds128(int n, double* Gg, double* red64, int* flag, double* pivmin_out2 /*LDS*/) {
    return chol_lds128_teams<2>(n, Gg, red64, flag, pivmin_out2);
}

// -------------------------------------------------------------------------------------------------
// Parameters of the compress / bond-truncate kernel
// -------------------------------------------------------------------------------------------------
struct CompressArgs {
    TTDev tt;
    long long max_bond;
    double truncerr;
    int sweeps;
    int k_single;          // 0: full tt_compress! sweeps; > 0: 1-based bond for _tt_bond_truncate!; < 0: the bond range below
    int k_first, k_last;   // k_single < 0: 0-based bonds k_first .. k_last in that order (descending if k_first > k_last)
    double* scratch;       // per-train global scratch
    long long scratch_stride;
    int pmax, qmax;        // bounds on the short / long side of any merged matrix
    double* sv_out;        // [batch][sv_steps][pmax] or null
    int sv_steps;
    int* status;           // [batch] device: 0 ok, 1 = Jacobi did not converge
    int* sweep_stats;      // [batch] device: total Jacobi sweeps (diagnostics)
    double jtol_mult;      // Jacobi convergence threshold = jtol_mult * sqrt(m) * eps
    long long* prof;       // null, or cycle counters per phase (TTN_PROF=1 diagnostic launches only)
    double jneg_mult;      // columns below jneg_mult * sqrt(m) * eps * max column norm are treated as zero
    int fast;              // 0: Householder route only; odd: try the Gram / factored fast paths (verified a posteriori) first.
                           // Diagnostic bits (TTN_FAST): 2 no eigensolver in route G (Cholesky + Jacobi), 4 none in route F,
                           // 8 no diagonal-left shortcut in route F, 16 no Jacobi polish after a failed conditioning test, 32 no CholeskyQR2,
                           // 64 Gram / reflector / check matrices in their own scratch instead of the dead T buffer, 128 two-pass fused merge,
                           // 512 no barrier-free (direct) form of the one-pass merge, 1024 no LDS-only path for the tiny steps
    // fused apply (ttn_apply_compress): psi = A * x is never materialised.  During the FIRST L->R sweep core k+1 of psi
    // is still virtual (= A_{k+1} applied to x_{k+1}); psi's ranks already hold A.rks .* x.rks.
    int fused;
    TTODev op;
    TTDev x;
    int rank_rule;         // 0: relative tail norm (_svdtrunc, tt_cross_interpolation.jl:149-166); 1: count(s > truncerr * s[1]), at least 1
                           //    (_swap_adjacent_sites, src/qtt_tools.jl:680-685)
    int fused_first_real;  // fused bond range (k_single < 0, ascending): the left core of the first bond is already real (a boundary core
                           // imported from the left neighbour of a core-wise sharded chain); otherwise it is written out first
    int prof_step;         // TTN_PROF_STEP: the phase counters of P.prof collect this step only (-1: every step)
    int* next_train;       // null: one workgroup per train (grid = batch).  Else a device counter (zeroed before the launch): the grid is
                           // PERSISTENT — workgroup w starts with train w and then pulls train gridDim.x + atomicAdd(next_train, 1)
                           // until the batch is exhausted (dynamic balancing of the data-dependent sweep counts, scratch per slot)
};

#define COMPRESS_LDS_X_DOUBLES (128 * 128)
#define COMPRESS_LDS_BYTES ((GEMM_LDS_TOTAL + 32 + 2 * QR_NB * QR_NB + QR_NB + 8 + 8 + 128) * sizeof(double))
#define FAST_KAPPA_MAX 128.0          // fast paths are used only when sigma_max/sigma_min <= this (error ~ eps*kappa^2)
#define FAST_KAPPA_POLISH 32768.0     // Gram route with a Jacobi polish of U^T M up to this conditioning of the kept block
#define FAST_POLISH_DROP 1.0e-12      // ... and only if the singular values it drops carry at most this share of ||M||_F^2 (i.e. are noise)
#define FAST_DIAG_TOL 2.0e-12         // route F: A'^T A' counts as diagonal below this (relative to sqrt(G_ii G_jj))
#define FAST_CHECK_TOL 2.0e-11        // a-posteriori bound on |Rf Rf^T - Sigma| (and Lf^T Lf - Sigma), relative

struct BondCtx {
    // LDS
    double *ldsX, *red, *Ts, *Ss, *taus, *scal, *nrm2;
    int* iflag;
    // global scratch
    double *M, *M2, *Vb, *Wb, *Us, *Xg, *sig, *sigs, *Ga, *Gb, *Cc, *T1, *T2, *T3;
    int* perm;
};

// -------------------------------------------------------------------------------------------------
// CholeskyQR2 for the rank-ramp bond steps (short side 16..64, moderately ill-conditioned: kappa 1e2..1e5 on the benchmark's
// L->R step 64 x 384).  The Householder LQ of such a matrix is 64 dependent reflectors over 384 columns (1.75 M clk with two
// workgroups per CU — more than a whole 128-row Gram step); the Gram route alone squares the condition number.  CholeskyQR2:
//   L1 = chol(M M^T);  Q1 = L1^-1 M (forward substitution, backward stable column by column);  L2 = chol(Q1 Q1^T);  L = L1 L2.
// Q1 is orthonormal up to eps kappa^2 << 1, so ITS Gram matrix loses nothing, and M = L1 Q1 + E with |E| <= eps |L1| |Q1|: L is
// the triangular factor of M + dM, ||dM|| ~ eps ||M|| — the backward error of the Householder factorisation — as long as
// eps kappa^2 stays well below 1 (measured: |Q1 Q1^T - I| <= CHOLQR_ORTH_MAX, i.e. kappa up to ~3e6); two MFMA Gram products, two
// 64 x 64 factorisations in LDS and one triangular solve instead of the reflector chain.
// -------------------------------------------------------------------------------------------------
#define CHOLQR_PIVOT_MAX 1.0e11       // first-pass pivot ratio (a lower bound of kappa^2) up to which CholeskyQR2 is taken
#define CHOLQR_ORTH_MAX 1.0e-3        // ... and the second pass must see a nearly orthonormal Q1: max |Q1 Q1^T - I| (~ eps kappa^2)
#define CHOLQR_CHECK_TOL 1.0e-9       // a-posteriori |Rf Rf^T - Sigma| bound of the route (the Householder route it replaces has none)

// p x p block of a leading-dimension-128 matrix in global memory -> rows [row_off, row_off + p) of the first p columns of the LDS image,
// eight loads of a thread in flight; returns max |src - I| (every thread; contains barriers)
__device__ __noinline__ double wg_img_load(double* img, int row_off, const double* src, int p, double* red) {
    img = unip(img); src = unip(src); p = uni32(p); row_off = uni32(row_off); red = unip(red);
    lds_f64* X = (lds_f64*)img;
    double dev = 0.0;
    wg_batched<8>((long long)p * p, [&](long long e) { const int ei = (int)e; return src[ei % p + 128 * (ei / p)]; },
                  [&](long long e, double v) {
                      const int ei = (int)e, i = ei % p, j = ei / p;
                      X[row_off + i + 128 * j] = v;
                      dev = fmax(dev, fabs(v - ((i == j) ? 1.0 : 0.0)));
                  });
    return unif64(wg_max(dev, red));
}

// Q[i][c] = (L^-1 (s M))[i][c], i < p, c < q: one thread per column, 16 rows at a time in registers; L (lower triangular, p <= 64) in
// the LDS image (L[i][k] at img[i + 128 k]), M and Q row-major in global memory.  Ends with a barrier.
__device__ __noinline__ void wg_trsm_lower_cols(int p, int q, const double* img, const double* M, long long ldm, double s_, double* Q, long long ldq) {
    p = uni32(p); q = uni32(q); img = unip(img); M = unip(M); Q = unip(Q); ldm = uni64(ldm); ldq = uni64(ldq); s_ = unif64(s_);
    const lds_f64* L = (const lds_f64*)img;
    gmem_f64* Mg = (gmem_f64*)M;
    gmem_wf64* Qg = (gmem_wf64*)Q;
    for (int c = threadIdx.x; c < q; c += TTN_WG) {
        for (int i0 = 0; i0 < p; i0 += 16) {
            const int nb = (p - i0 < 16) ? p - i0 : 16;
            double v[16];
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) v[ii] = Mg[(long long)(i0 + (ii < nb ? ii : 0)) * ldm + c];
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) v[ii] *= s_;
            for (int j0 = 0; j0 < i0; j0 += 16) {            // the finished blocks of this column (the thread's own stores)
                double qv[16];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) qv[jj] = Qg[(long long)(j0 + jj) * ldq + c];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj)
#pragma unroll
                    for (int ii = 0; ii < 16; ++ii) v[ii] = fma(-L[(j0 + jj) * 128 + i0 + ii], qv[jj], v[ii]);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < nb) {
                    v[k] = v[k] / L[(i0 + k) * 128 + i0 + k];
#pragma unroll
                    for (int ii = k + 1; ii < 16; ++ii) v[ii] = fma(-L[(i0 + k) * 128 + i0 + ii], v[k], v[ii]);
                }
            }
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) if (ii < nb) Qg[(long long)(i0 + ii) * ldq + c] = v[ii];
        }
    }
    __syncthreads();
}

// L = L1 L2 in the LDS image (p <= 64): L2 at img[i + 128 k], L1 at img[64 + i + 128 k] (the rows the p x p Jacobi image pads with
// zeros); on exit the first p columns hold L (lower triangle) and zeros everywhere else.  Ends with a barrier.
__device__ __noinline__ void wg_tril_mul_lds(int p, double* img) {
    p = uni32(p); img = unip(img);
    lds_f64* X = (lds_f64*)img;
    constexpr int NE = (64 * 64 + TTN_WG - 1) / TTN_WG;
    double acc[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = threadIdx.x + u * TTN_WG, i = e % p, j = e / p;
        double a = 0.0;
        if (e < p * p && i >= j)
            for (int k = j; k <= i; ++k) a = fma(X[64 + i + 128 * k], X[k + 128 * j], a);
        acc[u] = a;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < p * 128; e += TTN_WG) X[e] = 0.0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = threadIdx.x + u * TTN_WG, i = e % p, j = e / p;
        if (e < p * p && i >= j) X[i + 128 * j] = acc[u];
    }
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------
// Element loops of a bond step that scale the image columns by functions of sigma_j.  Written inline as
// `for (e = tid; e < p * r; e += WG) { row = e % p; j = e / p; ... sqrt(sigs[j]) ... ix(view, row) ... }` they spent ~3 k clk per
// element of a thread on two runtime integer divisions, the two-level index of the output view, an fp64 square root and two fp64
// divisions (57 k clk per 128-row Gram step for 8192 elements).  Out of line: everything that depends on j alone (factors, the
// permutation, column offsets) or on the row alone (row offsets) goes to small LDS tables first, the element loop walks (row = lane,
// j = wave) without divisions.  `tab`: 512 doubles of LDS (the Householder panel matrices S.Ts / S.Ss, dead outside the LQ).
// -------------------------------------------------------------------------------------------------
// Lo[row, j] = keep_j ? X[row, perm_j] f1_j : 0,  Us[j p + row] = keep_j ? X[row, perm_j] f2_j : 0   (row < p <= 128, j < r <= 128)
//   mode 0 (tt_compress!):  f1 = sq0 / sqrt(s_j), f2 = sq0 / (s_j sqrt(s_j));  mode 1 (swap, wide): 1 / s_j, s0 / s_j;  mode 2 (swap, tall): s0, 1 / s_j^2
__device__ __noinline__ void wg_scale_image(const double* X, int ldx, int x_in_lds, const double* sigs, const int* perm, View Lo, double* Us, int p, int r,
                                            double s0, double aneg, int mode, double* tab) {
    X = unip(X); ldx = uni32(ldx); x_in_lds = uni32(x_in_lds); sigs = unip(sigs); perm = unip(perm); Lo = uniView(Lo); Us = unip(Us);
    p = uni32(p); r = uni32(r); s0 = unif64(s0); aneg = unif64(aneg); mode = uni32(mode); tab = unip(tab);
    lds_f64* f1 = (lds_f64*)tab; lds_f64* f2 = f1 + 128;
    lds_i32* pj = (lds_i32*)(f2 + 128); lds_i32* coff = pj + 128; lds_i32* roff = coff + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double sq0 = sqrt(s0);
    for (int j = tid; j < r; j += TTN_WG) {
        const double sj = sigs[j];
        const bool keep = (sj > 0.0) && (sj * sj > aneg);
        double a, b_;
        if (mode == 0) { const double rs = sqrt(sj); a = sq0 / rs; b_ = sq0 / (sj * rs); }
        else if (mode == 1) { a = 1.0 / sj; b_ = s0 / sj; }
        else { a = s0; b_ = 1.0 / (sj * sj); }
        f1[j] = keep ? a : 0.0; f2[j] = keep ? b_ : 0.0;
        pj[j] = perm[j]; coff[j] = (int)ix(Lo.c, j);
    }
    for (int i = tid; i < p; i += TTN_WG) roff[i] = (int)ix(Lo.r, i);
    __syncthreads();
    gmem_wf64* Lg = (gmem_wf64*)Lo.p;
    gmem_wf64* Ug = (gmem_wf64*)Us;
    for (int j = wave; j < r; j += TTN_NWAVES) {
        const double a = f1[j], b_ = f2[j];
        const int c = pj[j], co = coff[j];
        for (int row = lane; row < p; row += 64) {
            const double xv = x_in_lds ? ((const lds_f64*)X)[c * ldx + row] : X[(long long)c * ldx + row];
            Lg[roff[row] + co] = xv * a;
            Ug[j * p + row] = xv * b_;
        }
    }
    __syncthreads();
}

// max over i, j < r of |D[i + ldd j] - delta_ij sigs[i] s0| / (s0 sqrt(sigs[i] sigs[j]))   (r <= 128; every thread gets it)
__device__ __noinline__ double wg_check_diag_tab(const double* sigs, const double* D, int ldd, int r, double s0, double* tab, double* red) {
    sigs = unip(sigs); D = unip(D); ldd = uni32(ldd); r = uni32(r); s0 = unif64(s0); tab = unip(tab); red = unip(red);
    lds_f64* t = (lds_f64*)tab; lds_f64* ref = t + 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < r; i += TTN_WG) { const double si = sigs[i]; t[i] = 1.0 / sqrt(s0 * si); ref[i] = si * s0; }
    __syncthreads();
    gmem_f64* Dg = (gmem_f64*)D;
    double worst = 0.0;
    for (int j = wave; j < r; j += TTN_NWAVES) {
        const double tj = t[j];
        for (int i = lane; i < r; i += 64) {
            const double dv = Dg[i + ldd * j];
            const double dev = fabs(dv - ((i == j) ? ref[i] : 0.0)) * (t[i] * tj);
            worst = fmax(worst, dev);
        }
    }
    return unif64(wg_max(worst, red));
}

// Route F with a 64 x 64 left Gram matrix Ga (ld 128): w = max_{i != j} |Ga_ij| / sqrt(Ga_ii Ga_jj) (1 if a diagonal entry is not
// positive) and T3 = D^(1/2) Gb D^(1/2), D = diag(Ga) — one pass over both matrices.  Every thread gets w.
__device__ __noinline__ double wg_diag_test_t3(const double* Ga, const double* Gb, double* T3, double* tab, double* red) {
    Ga = unip(Ga); Gb = unip(Gb); T3 = unip(T3); tab = unip(tab); red = unip(red);
    lds_f64* sd = (lds_f64*)tab; lds_f64* isd = sd + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    gmem_f64* Gag = (gmem_f64*)Ga; gmem_f64* Gbg = (gmem_f64*)Gb; gmem_wf64* Tg = (gmem_wf64*)T3;
    for (int i = tid; i < 64; i += TTN_WG) { const double d = Gag[i * 129]; const double sq = (d > 0.0) ? sqrt(d) : 0.0; sd[i] = sq; isd[i] = (d > 0.0) ? 1.0 / sq : -1.0; }
    __syncthreads();
    double w = 0.0;
    const double si = sd[lane], ii = isd[lane];
    constexpr int NJ = 64 / TTN_NWAVES;                       // columns per wave (8 / 4)
    double ga[NJ], gb[NJ];
#pragma unroll
    for (int u = 0; u < NJ; ++u) { const int j = wave + TTN_NWAVES * u; ga[u] = Gag[lane + 128 * j]; gb[u] = Gbg[lane + 128 * j]; }
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
        const int j = wave + TTN_NWAVES * u;
        const double ij = isd[j];
        if (lane != j) w = fmax(w, (ii > 0.0 && ij > 0.0) ? fabs(ga[u]) * (ii * ij) : 1.0);
        Tg[lane + 128 * j] = si * gb[u] * sd[j];
    }
    w = unif64(wg_max(w, red));
    return w;
}

// Route F, diagonal-left form: T1[j 128 + row] = X[row, perm_j] fa / (sd_row sqrt(s_j)),  T2[j 128 + row] = X[row, perm_j] sd_row fb / (s_j sqrt(s_j)),
// sd_row = sqrt(Ga[row, row]); row < 64, j < rk <= 64; X = the LDS image (ld 128)
__device__ __noinline__ void wg_scale_t12_diag(const double* img, const double* sigs, const int* perm, const double* Ga, double* T1, double* T2, int rk,
                                               double fa, double fb, double* tab) {
    img = unip(img); sigs = unip(sigs); perm = unip(perm); Ga = unip(Ga); T1 = unip(T1); T2 = unip(T2); rk = uni32(rk); fa = unif64(fa); fb = unif64(fb); tab = unip(tab);
    lds_f64* c1 = (lds_f64*)tab; lds_f64* c2 = c1 + 64; lds_f64* sd = c2 + 64; lds_f64* isd = sd + 64;
    lds_i32* pj = (lds_i32*)(isd + 64);
    const lds_f64* X = (const lds_f64*)img;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = tid; j < 64; j += TTN_WG) {
        const double d = sqrt(Ga[j * 129]); sd[j] = d; isd[j] = 1.0 / d;
        if (j < rk) { const double sj = sigs[j], rs = sqrt(sj); c1[j] = fa / rs; c2[j] = fb / (sj * rs); pj[j] = perm[j]; }
    }
    __syncthreads();
    gmem_wf64* T1g = (gmem_wf64*)T1; gmem_wf64* T2g = (gmem_wf64*)T2;
    const double sdr = sd[lane], isdr = isd[lane];
    for (int j = wave; j < rk; j += TTN_NWAVES) {
        const double xv = X[pj[j] * 128 + lane];
        T1g[j * 128 + lane] = xv * (c1[j] * isdr);
        T2g[j * 128 + lane] = xv * (sdr * c2[j]);
    }
    __syncthreads();
}

// Jacobi on the pj columns (length pj) of X, then singular values sigma_c = ||x_c|| sorted descending with a
// stable order: perm[pos] = column, sigs[pos] = sigma (scaled units).  Returns the sweep count (<0: limit hit).
__device__ int wg_svd_cols(const CompressArgs& P, const BondCtx& S, int pj, double* X, int ldx, bool in_lds, int mlen = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = TTN_WG >> 6;
    // mlen: column length if it is not pj (LDS image only: the polish step of the Gram route runs r columns of length q)
    const int nsw = (in_lds && mlen) ? wg_jacobi_lds128(mlen, pj, X, S.nrm2, S.iflag, S.red, P.jtol_mult, P.jneg_mult, S.scal)
                  : in_lds ? wg_jacobi_lds128(pj, pj, X, S.nrm2, S.iflag, S.red, P.jtol_mult, P.jneg_mult, S.scal)
#if TTN_LDS_COLS < 128
                  : (pj <= 128) ? wg_jacobi_blocked128(pj, pj, X, ldx, S.ldsX, S.nrm2, S.iflag, S.red, P.jtol_mult, P.jneg_mult, S.scal)
#endif
                  : (pj <= 256) ? wg_jacobi_blocked256(pj, pj, X, ldx, S.ldsX, S.nrm2, S.iflag, S.red, P.jtol_mult, P.jneg_mult, S.scal)
                                : wg_jacobi_cols(pj, pj, X, ldx, S.iflag, S.red, P.jtol_mult, P.jneg_mult, S.scal);
    for (int c = wave; c < pj; c += nwaves) {
        double a = 0.0;
        for (int r = lane; r < (mlen ? mlen : pj); r += 64) { const double v = X[(long long)c * ldx + r]; a = fma(v, v, a); }
        a = wave_sum(a);
        if (lane == 0) S.sig[c] = sqrt(a);
    }
    __syncthreads();
    for (int c = tid; c < pj; c += TTN_WG) {
        const double sc = S.sig[c];
        int pos = 0;
        for (int j = 0; j < pj; ++j) { const double sj = S.sig[j]; pos += (sj > sc) || (sj == sc && j < c); }
        S.perm[pos] = c;
        S.sigs[pos] = sc;
    }
    __syncthreads();
    return nsw;
}

// The effective _svdtrunc rank rule (src/tt_cross_interpolation.jl:149-166) on `ns` computed singular values
// (scaled by s0) padded with zeros to the reference's length `plen` = min(size(M)).  All threads get r.
__device__ __forceinline__ int wg_rank_rule_core(int rank_rule, double truncerr, long long max_bond, const double* sigs, int* iflag, int ns, int plen, double s0) {
    if (threadIdx.x == 0) {
        int r = plen;
        if (rank_rule == 1) {
            if (truncerr > 0.0) {
                r = 0;
                const double thr = truncerr * (sigs[0] * s0);
                for (int i = 0; i < ns; ++i) r += (sigs[i] * s0 > thr) ? 1 : 0;
                if (r < 1) r = 1;
            }
        } else if (truncerr > 0.0) {
            double n2 = 0.0;
            for (int i = 0; i < ns; ++i) { const double s = sigs[i] * s0; n2 = fma(s, s, n2); }
            const double nrm = sqrt(n2);
            double cum = 0.0;
            for (int i = plen; i >= 1; --i) {
                const double s = (i <= ns) ? sigs[i - 1] * s0 : 0.0;
                cum = fma(s, s, cum);
                if (sqrt(cum) > truncerr * nrm) { r = i; break; }
            }
        }
        if ((long long)r > max_bond) r = (int)max_bond;
        iflag[1] = r;
    }
    __syncthreads();
    const int r = uni32(iflag[1]);
    __syncthreads();
    return r;
}
__device__ int wg_rank_rule(const CompressArgs& P, const BondCtx& S, int ns, int plen, double s0) {
    return wg_rank_rule_core(P.rank_rule, P.truncerr, P.max_bond, S.sigs, S.iflag, ns, plen, s0);
}

// max over i,j < r of |D[i + ldd*j] - delta_ij * sigs[i]*s0| / (s0*sqrt(sigs[i]*sigs[j]))
__device__ double wg_check_diag(const BondCtx& S, const double* D, int ldd, int r, double s0) {
    double worst = 0.0;
    for (int e = threadIdx.x; e < r * r; e += TTN_WG) {
        const int i = e % r, j = e / r;
        const double si = S.sigs[i], sj = S.sigs[j];
        const double ref = (i == j) ? si * s0 : 0.0;
        worst = fmax(worst, fabs(D[i + (long long)ldd * j] - ref) / (s0 * sqrt(si * sj)));
    }
    return wg_max(worst, S.red);
}

// Core k of psi = A * x written out (src/tt_operations.jl:101-111): psi_k[s, a' + Rl*nu', a + Rr*nu] = sum_j A_k[s,j,a',a] x_k[j,nu',nu].
// Used by the fused apply+compress for the cores the fused merge does not cover (first core, tall / tiny steps).
__device__ void wg_materialize_core(const CompressArgs& P, int b, int k) {
    const TTDev& T = P.tt;
    const int n = T.dims[k];
    const long long* xr = P.x.rks + (long long)b * (P.x.d + 1);
    const int rl = uni32((int)xr[k]), rr = uni32((int)xr[k + 1]);
    const int Rl = uni32((int)P.op.rks[k]), Rr = uni32((int)P.op.rks[k + 1]);
    const double* xc = P.x.data + (long long)b * P.x.stride + P.x.off[k];
    const double* ac = P.op.data + P.op.off[k];
    double* yc = T.data + (long long)b * T.stride + T.off[k];
    const long long Dl = (long long)Rl * rl, tot = (long long)n * Dl * Rr * rr;
    for (long long e = threadIdx.x; e < tot; e += TTN_WG) {
        const int s = (int)(e % n);
        const long long t1 = e / n;
        const int ga = (int)(t1 % Dl), be = (int)(t1 / Dl);
        const int a1 = ga % Rl, nu1 = ga / Rl, a2 = be % Rr, nu2 = be / Rr;
        double v = 0.0;
        for (int j = 0; j < n; ++j)
            v = fma(ac[s + n * (j + n * (a1 + (long long)Rl * a2))], xc[j + n * (nu1 + (long long)rl * nu2)], v);
        yc[e] = v;
    }
    __syncthreads();
}

// Fused merge of an L->R step whose right core is still virtual (wide case, p = n1*Dl rows):
//   M[row, s2 + n2*(a + Rr*nu)] = sum_{a', nu'} C_k[row, a' + Rl*nu'] * sum_j A[s2, j, a', a] x[j, nu', nu]
// = (i)  T_a'[row, (j, nu)] = sum_nu' C_k[row, a' + Rl*nu'] x[j, nu', nu]      Rl GEMMs  p x (n2*rho_r) x rho_l  (MFMA)
//   (ii) M[row, (s2, a, nu)] = sum_{a', j} T_a'[row, (j, nu)] A[s2, j, a', a]     Rl*n2 terms per output     (VALU)
// 2*p*Rl*rho_l*n2*rho_r flops instead of 2*p*(Rl*rho_l)*(n2*Rr*rho_r) — Rr times fewer — and the core never exists in HBM.
// T lives in `Tbuf` (>= Rl*p*n2*rho_r doubles).  max|M| goes to *amax_lds.  Returns false (nothing done) if the shapes
// do not qualify; the caller then materialises the core.
// ---- the same merge in ONE pass, with the operator contraction in the GEMM epilogue (n2 = 2, small operator ranks) -----------------
// The first version (wg_fused_merge below, still the general path) runs Rl separate GEMMs into a buffer T (Rl p n2 rho_r doubles,
// 393 KB for the benchmark's steps), and a second pass reads T back, contracts with the operator core and writes M: measured inside
// the kernel (TTN_PROF, 512-thread build, two workgroups per CU) 349 k clk for the three GEMMs — 14 % of the matrix pipe each: K = 64
// is all prologue — plus 191 k clk for the contraction pass, per bond step.  Here a wave keeps the Rl accumulators T_a'[16 rows,
// 16 (j, nu) columns] of its tiles in registers: they share every B fragment (x is read once, not Rl times), and when the K loop is
// done the lane that holds column (j, nu) forms its share sum_a' T_a' A[s, j, a', a] of the n2 Rr outputs of that (row, nu), adds its
// neighbour's (the other j: one DPP quad permutation) and writes M.  T never exists.
// Layout: chunks of FM_BK = 8 values of nu' through two LDS stages (As[(a' FM_BK + kk) LDA + row], Bs[kk LDB + col]); a wave owns
// one 16-row tile and two 16-column tiles ((j, nu) columns: 8 nu each) of a pass over CP = 32 * (waves / row tiles) columns.
#define FM_BK 8
__device__ __noinline__ bool wg_fused_merge_mfma(double* ck, const double* xc, const double* ac, double* M, int p, int q, int n1, int Dl,
                                                 int rhl, int rhr, int Rl, int Rr, double* lds, double* amax_lds, double* red) {
    ck = unip(ck); xc = unip(xc); ac = unip(ac); M = unip(M); lds = unip(lds); amax_lds = unip(amax_lds); red = unip(red);
    p = uni32(p); q = uni32(q); n1 = uni32(n1); Dl = uni32(Dl); rhl = uni32(rhl); rhr = uni32(rhr); Rl = uni32(Rl); Rr = uni32(Rr);
    constexpr int n2 = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int RT = p >> 4;                                           // row tiles
    const int WR = RT < 8 ? RT : 8;                                  // waves along the rows (the A stage holds 128 rows: LDA = 145)
    const int WC = TTN_NWAVES / WR;                                  // wave columns (>= 1)
    const int CP = 32 * WC;                                          // (j, nu) columns per pass
    const int ncol = n2 * rhr, nin = Rl * n2, nout = n2 * Rr;
    constexpr int LDA = GEMM_LD;
    const int LDB = (CP <= 32) ? 49 : (CP <= 64) ? 81 : GEMM_LD;     // all == 17 mod 32
    const int stage = Rl * FM_BK * LDA + FM_BK * LDB;
    if (Rl < 1 || Rl > 3 || Rr > 4 || (p & 15) || (rhl % FM_BK) || (ncol % CP) || (RT % WR) || WR * WC != TTN_NWAVES || CP > 128 ||
        Rl * FM_BK * 16 * WR > 6 * TTN_WG || FM_BK * CP > 2 * TTN_WG || 2 * stage + 64 > GEMM_LDS_DOUBLES)
        return false;
    lds_f64* opl = (lds_f64*)lds + 2 * stage;                        // operator core: opl[o * nin + i], o = s + n2 a, i = j + n2 a'
    __syncthreads();
    for (int e = tid; e < nin * nout; e += TTN_WG) {
        const int i = e % nin, o = e / nin;
        opl[e] = ac[(o % n2) + n2 * ((i % n2) + n2 * ((i / n2) + (long long)Rl * (o / n2)))];
    }
    const long long ldk = (long long)n1 * Dl;                        // column stride of C_k as a (n1 Dl) x (Rl rho_l) matrix
    const int wr = wave % WR, wc = wave / WR;
    const int nchunk = rhl / FM_BK;
    gmem_f64* Cg = (gmem_f64*)ck;
    gmem_f64* Xg = (gmem_f64*)xc;
    gmem_wf64* Mg = (gmem_wf64*)M;
    double cmax = 0.0;
    // staging assignment: A chunk = Rl * FM_BK * (16 WR) elements, row fastest; B chunk = FM_BK * CP elements, column fastest
    const int arows = 16 * WR;
    const int na_el = Rl * FM_BK * arows, nb_el = FM_BK * CP;
    constexpr int NUA = 6, NUB = 2;                                  // elements per thread and chunk (enough for 3 * 8 * 128 / 512 and 8 * 128 / 512)
    for (int rb = 0; rb < RT; rb += WR) {                            // row blocks of 16 WR rows
        for (int c0 = 0; c0 < ncol; c0 += CP) {                      // column passes
            mfma_acc_t acc[3][2];
#pragma unroll
            for (int a1 = 0; a1 < 3; ++a1) { acc[a1][0] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0}; acc[a1][1] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0}; }
            int aoff[NUA], alds[NUA], boff[NUB], blds[NUB];
            bool aok[NUA], bok[NUB];
#pragma unroll
            for (int u = 0; u < NUA; ++u) {
                const int e = tid + TTN_WG * u;
                aok[u] = e < na_el;
                const int r = e % arows, rest = e / arows, kk = rest % FM_BK, a1 = rest / FM_BK;
                const int row = 16 * rb + r;
                // C_k[(al, s1), ga] at s1 + n1 (al + Dl ga), row = al + Dl s1, ga = a1 + Rl nu'
                aoff[u] = aok[u] ? (row % Dl) * n1 + row / Dl + (int)(ldk * (a1 + (long long)Rl * kk)) : 0;
                alds[u] = (a1 * FM_BK + kk) * LDA + r;
            }
#pragma unroll
            for (int u = 0; u < NUB; ++u) {
                const int e = tid + TTN_WG * u;
                bok[u] = e < nb_el;
                const int c = e % CP, kk = e / CP, col = c0 + c;
                // x[j, nu', nu] at j + n2 (nu' + rho_l nu), col = j + n2 nu
                boff[u] = bok[u] ? (col & 1) + n2 * (kk + rhl * (col >> 1)) : 0;
                blds[u] = kk * LDB + c;
            }
            const int astep = (int)(ldk * Rl * FM_BK), bstep = n2 * FM_BK;       // address advance per chunk
            double av[NUA], bv[NUB];
#define FM_LOAD(CH)                                                                                                     \
            { _Pragma("unroll") for (int u = 0; u < NUA; ++u) av[u] = Cg[aoff[u] + (CH) * astep];                       \
              _Pragma("unroll") for (int u = 0; u < NUB; ++u) bv[u] = Xg[boff[u] + (CH) * bstep]; }
#define FM_STORE(STG)                                                                                                   \
            { lds_f64* As_ = (lds_f64*)lds + (STG) * stage; lds_f64* Bs_ = As_ + Rl * FM_BK * LDA;                      \
              _Pragma("unroll") for (int u = 0; u < NUA; ++u) if (aok[u]) As_[alds[u]] = av[u];                         \
              _Pragma("unroll") for (int u = 0; u < NUB; ++u) if (bok[u]) Bs_[blds[u]] = bv[u]; }
            __syncthreads();                                         // the previous pass has consumed both stages (and opl is written)
            FM_LOAD(0)
            FM_STORE(0)
            if (nchunk > 1) FM_LOAD(1)
            __syncthreads();
            for (int ch = 0; ch < nchunk; ++ch) {
                const lds_f64* As = (lds_f64*)lds + (ch & 1) * stage;
                const lds_f64* Bs = As + Rl * FM_BK * LDA;
#pragma unroll
                for (int t = 0; t < FM_BK / 4; ++t) {
                    const int kr = 4 * t + lk;
                    const double b0 = Bs[kr * LDB + wc * 32 + li], b1 = Bs[kr * LDB + wc * 32 + 16 + li];
#pragma unroll
                    for (int a1 = 0; a1 < 3; ++a1) {
                        if (a1 < Rl) {
                            const double a = As[(a1 * FM_BK + kr) * LDA + wr * 16 + li];
                            acc[a1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[a1][0], 0, 0, 0);
                            acc[a1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[a1][1], 0, 0, 0);
                        }
                    }
                    if (t == 0 && ch + 1 < nchunk) FM_STORE((ch + 1) & 1)
                }
                if (ch + 2 < nchunk) FM_LOAD(ch + 2)
                __syncthreads();
            }
#undef FM_LOAD
#undef FM_STORE
            // ---- epilogue: lane (li, lk) holds T_a'[row = lk + 4 reg][column li] of its two column tiles; column = j + n2 nu ----
            const int jl = li & 1;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int nu = (c0 + wc * 32 + ct * 16 + li) >> 1;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * (rb + wr) + lk + 4 * reg;
                    gmem_wf64* mrow = Mg + ((long long)row * q + (long long)nout * nu);
                    for (int o = 0; o < nout; ++o) {
                        double v = 0.0;
#pragma unroll
                        for (int a1 = 0; a1 < 3; ++a1)
                            if (a1 < Rl) v = fma(acc[a1][ct][reg], opl[o * nin + jl + n2 * a1], v);
                        v += dpp_mov_f64<0xB1>(v);                   // + the other j (quad_perm [1,0,3,2]): both lanes of the pair hold the sum
                        if ((o & 1) == jl) mrow[o] = v;              // the pair shares the stores
                        cmax = fmax(cmax, fabs(v));
                    }
                }
            }
        }
    }
    cmax = wg_max(cmax, red);
    if (tid == 0) *amax_lds = cmax;
    __syncthreads();
    return true;
}

// ---- the one-pass merge without barriers in its main loop ("direct" form) ------------------------------------------------------------
// wg_fused_merge_mfma stages chunks of FM_BK = 8 values of nu' of BOTH operands through LDS: 8 chunks x 4 column passes = 32 barrier
// rounds of 12 MFMAs per wave each for a 128-row step — the barriers and the staging, not the matrix pipe, set its time (29 % of the
// pipe alone, 18 % with two workgroups per CU).  Here the x core (rho_l x n2 rho_r, 64 KB for rank 64) is staged ONCE, whole, and the
// A fragments — lane (li, lk) needs C_k[row 16 rt + li, a' + Rl (4 ks + lk)] — are loaded from global memory (L2) straight into the
// MFMA operand registers, four k-steps ahead of their use: no LDS staging of C_k, no barrier between the staging of x and the epilogue.
// Same tiling (a wave = one 16-row tile x two 16-column tiles x Rl accumulators), same epilogue.
__device__ __noinline__ bool wg_fused_merge_direct(double* ck, const double* xc, const double* ac, double* M, int p, int q, int n1, int Dl,
                                                   int rhl, int rhr, int Rl, int Rr, double* lds, double* amax_lds, double* red, long long* stamps) {
    stamps = unip(stamps);
#define FMD_STAMP(i) if (stamps && threadIdx.x == 0) stamps[i] = (long long)__builtin_amdgcn_s_memtime();
    FMD_STAMP(0)
    ck = unip(ck); xc = unip(xc); ac = unip(ac); M = unip(M); lds = unip(lds); amax_lds = unip(amax_lds); red = unip(red);
    p = uni32(p); q = uni32(q); n1 = uni32(n1); Dl = uni32(Dl); rhl = uni32(rhl); rhr = uni32(rhr); Rl = uni32(Rl); Rr = uni32(Rr);
    constexpr int n2 = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int RT = p >> 4;
    const int WR = RT < 8 ? RT : 8;
    const int WC = TTN_NWAVES / WR;
    const int CP = 32 * WC;
    const int ncol = n2 * rhr, nin = Rl * n2, nout = n2 * Rr;
    const int LDB = ncol + 1;                                        // odd: the four k rows of a fragment read land two banks apart
    if (Rl < 1 || Rl > 3 || Rr > 4 || (p & 15) || (rhl & 15) || (ncol % CP) || (RT % WR) || WR * WC != TTN_NWAVES ||
        (long long)rhl * LDB + 64 > GEMM_LDS_DOUBLES)
        return false;
    lds_f64* Bs = (lds_f64*)lds;                                     // Bs[kk * LDB + col], kk = nu' < rho_l, col = j + n2 nu
    lds_f64* opl = Bs + rhl * LDB;                                   // operator core: opl[o * nin + i], o = s + n2 a, i = j + n2 a'
    gmem_f64* Cg = (gmem_f64*)ck;
    gmem_f64* Xg = (gmem_f64*)xc;
    gmem_wf64* Mg = (gmem_wf64*)M;
    __syncthreads();
    for (int e = tid; e < nin * nout; e += TTN_WG) {
        const int i = e % nin, o = e / nin;
        opl[e] = ac[(o % n2) + n2 * ((i % n2) + n2 * ((i / n2) + (long long)Rl * (o / n2)))];
    }
    // x[j, nu', nu] at j + n2 (nu' + rho_l nu): consecutive threads walk (j, nu') of one nu — contiguous in memory
    {
        const int nx = rhl * ncol, kspan = n2 * rhl;                 // elements of one nu
        for (int e0 = tid; e0 < nx; e0 += 8 * TTN_WG) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + u * TTN_WG; v[u] = Xg[e < nx ? e : e0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * TTN_WG;
                if (e < nx) { const int nu = e / kspan, w_ = e - nu * kspan, j = w_ & 1, kk = w_ >> 1; Bs[kk * LDB + j + n2 * nu] = v[u]; }
            }
        }
    }
    __syncthreads();
    FMD_STAMP(1)
    const long long ldk = (long long)n1 * Dl;                        // column stride of C_k as a (n1 Dl) x (Rl rho_l) matrix
    const int wr = wave % WR, wc = wave / WR;
    const int ksteps = rhl >> 2, ngrp = ksteps >> 2;                 // groups of four k-steps
    const int kstride = (int)(ldk * Rl);                             // address advance per nu'
    double cmax = 0.0;
    const int jl = li & 1;
    for (int rb = 0; rb < RT; rb += WR) {
        const int arow = 16 * (rb + wr) + li;
        // C_k[(al, s1), ga] at s1 + n1 (al + Dl ga), row = al + Dl s1, ga = a' + Rl nu'
        const int abase = (arow % Dl) * n1 + arow / Dl + lk * kstride;
        for (int c0 = 0; c0 < ncol; c0 += CP) {
            mfma_acc_t acc[3][2];
#pragma unroll
            for (int a1 = 0; a1 < 3; ++a1) { acc[a1][0] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0}; acc[a1][1] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0}; }
            double cur[4][3], nxt[4][3];
#define FMD_LOAD(DST, G)                                                                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                \
                _Pragma("unroll") for (int a1 = 0; a1 < 3; ++a1)                                                         \
                    DST[t][a1] = (a1 < Rl) ? Cg[abase + (4 * (4 * (G) + t)) * kstride + a1 * (int)ldk] : 0.0;
            FMD_LOAD(cur, 0)
            const lds_f64* bcol = Bs + lk * LDB + c0 + wc * 32 + li;
            for (int g = 0; g < ngrp; ++g) {
                if (g + 1 < ngrp) { FMD_LOAD(nxt, g + 1) }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kr = 4 * (4 * g + t);
                    const double b0 = bcol[kr * LDB], b1 = bcol[kr * LDB + 16];
#pragma unroll
                    for (int a1 = 0; a1 < 3; ++a1) {
                        if (a1 < Rl) {
                            acc[a1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[t][a1], b0, acc[a1][0], 0, 0, 0);
                            acc[a1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[t][a1], b1, acc[a1][1], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int a1 = 0; a1 < 3; ++a1) cur[t][a1] = nxt[t][a1];
            }
#undef FMD_LOAD
            if (rb == 0 && c0 == 0) { FMD_STAMP(2) }
            // ---- epilogue: lane (li, lk) holds T_a'[row = lk + 4 reg][column li] of its two column tiles; column = j + n2 nu.
            // Both lanes of a (j = 0, 1) pair end up with all nout sums of their (row, nu); the pair then writes the nout consecutive
            // doubles M[row, nout nu ...] as two contiguous halves (lane j = 0 the first nout / 2, lane j = 1 the rest), 16 bytes at a
            // time where the address allows: the 16 lanes of a row of lanes cover 8 nu = 8 nout contiguous doubles of one row of M —
            // whole cache lines per store instruction.  (Stored one output at a time, lanes 6 doubles apart and half of them idle,
            // these stores, not the MFMAs, set the time of the merge.) ----
            const int half = nout >> 1;
            const bool vec_ok = (half == 3) && ((((unsigned long long)Mg) & 15ull) == 0) && ((q & 1) == 0);
            if (vec_ok && Rl == 3) {
                // Operator ranks 3 x 3 (the Laplacian's interior cores): the pair first swaps its raw T values (one DPP move per a'), then
                // each lane forms ITS three outputs from all six inputs — weights w[oo][a'][own / other j] read from LDS once per pass —
                // and stores them as 16 + 8 bytes.
                double w[3][3][2];
#pragma unroll
                for (int oo = 0; oo < 3; ++oo)
#pragma unroll
                    for (int a1 = 0; a1 < 3; ++a1) {
                        const int o = 3 * jl + oo;
                        w[oo][a1][0] = opl[o * nin + jl + n2 * a1];          // own j
                        w[oo][a1][1] = opl[o * nin + (1 - jl) + n2 * a1];    // the neighbour's j
                    }
                typedef double dbl2 __attribute__((ext_vector_type(2)));
                typedef __attribute__((address_space(1))) dbl2 gmem_wd2;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int nu = (c0 + wc * 32 + ct * 16 + li) >> 1;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = 16 * (rb + wr) + lk + 4 * reg;
                        gmem_wf64* mrow = Mg + ((long long)row * q + (long long)nout * nu);
                        const double t0 = acc[0][ct][reg], t1 = acc[1][ct][reg], t2 = acc[2][ct][reg];
                        const double u0 = dpp_mov_f64<0xB1>(t0), u1 = dpp_mov_f64<0xB1>(t1), u2 = dpp_mov_f64<0xB1>(t2);      // quad_perm [1,0,3,2]
                        double v[3];
#pragma unroll
                        for (int oo = 0; oo < 3; ++oo) {
                            double x_ = t0 * w[oo][0][0];
                            x_ = fma(u0, w[oo][0][1], x_);
                            x_ = fma(t1, w[oo][1][0], x_); x_ = fma(u1, w[oo][1][1], x_);
                            x_ = fma(t2, w[oo][2][0], x_); x_ = fma(u2, w[oo][2][1], x_);
                            v[oo] = x_;
                            cmax = fmax(cmax, fabs(x_));
                        }
                        // j = 0: (o0, o1) as 16 bytes at +0, o2 at +2;  j = 1: o3 at +3, (o4, o5) as 16 bytes at +4
                        *(gmem_wd2*)(mrow + (jl ? 4 : 0)) = jl ? (dbl2){v[1], v[2]} : (dbl2){v[0], v[1]};
                        mrow[jl ? 3 : 2] = jl ? v[0] : v[2];
                    }
                }
            } else {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int nu = (c0 + wc * 32 + ct * 16 + li) >> 1;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = 16 * (rb + wr) + lk + 4 * reg;
                    gmem_wf64* mrow = Mg + ((long long)row * q + (long long)nout * nu);
                    for (int o = 0; o < nout; ++o) {
                        double v = 0.0;
#pragma unroll
                        for (int a1 = 0; a1 < 3; ++a1)
                            if (a1 < Rl) v = fma(acc[a1][ct][reg], opl[o * nin + jl + n2 * a1], v);
                        v += dpp_mov_f64<0xB1>(v);                   // + the other j (quad_perm [1,0,3,2]): both lanes of the pair hold the sum
                        if ((o >= half) == (jl == 1)) mrow[o] = v;   // the pair shares the stores: each lane a contiguous half
                        cmax = fmax(cmax, fabs(v));
                    }
                }
            }
            }
            if (rb == 0 && c0 == 0) { FMD_STAMP(3) }
        }
    }
    FMD_STAMP(4)
    cmax = wg_max(cmax, red);
    if (tid == 0) *amax_lds = cmax;
    __syncthreads();
    FMD_STAMP(5)
#undef FMD_STAMP
    return true;
}


#define FUSE_MAX_TERMS 16
__device__ bool wg_fused_merge(const CompressArgs& P, int b, int k, int p, int q, const View& Am, double* M, double* Tbuf,
                               long long tbuf_doubles, double* lds, double* amax_lds, double* red, long long* stamps = nullptr) {
    const TTDev& T = P.tt;
    const int n1 = T.dims[k], n2 = T.dims[k + 1];
    const long long* rks = T.rks + (long long)b * (T.d + 1);
    const int Dl = uni32((int)rks[k]);
    const long long* xr = P.x.rks + (long long)b * (P.x.d + 1);
    const int rhl = uni32((int)xr[k + 1]), rhr = uni32((int)xr[k + 2]);
    const int Rl = uni32((int)P.op.rks[k + 1]), Rr = uni32((int)P.op.rks[k + 2]);
    const int ncol = n2 * rhr;
    if (Rl * n2 > FUSE_MAX_TERMS || n2 * Rr > FUSE_MAX_TERMS || (long long)Rl * p * ncol > tbuf_doubles) return false;
    if ((long long)Rl * rhl != rks[k + 1] || (long long)n2 * Rr * rhr != q || p != n1 * Dl) return false;
    double* ck = T.data + (long long)b * T.stride + T.off[k];
    double* xc = P.x.data + (long long)b * P.x.stride + P.x.off[k + 1];
    const double* ac = P.op.data + P.op.off[k + 1];
    TTN_SETPRIO_GEMM();
    struct PrioRestore { __device__ ~PrioRestore() { TTN_SETPRIO_BASE(); } } prio_restore_;
    if (n2 == 2 && !(P.fast & 128) && !(P.fast & 512) && wg_fused_merge_direct(ck, xc, ac, M, p, q, n1, Dl, rhl, rhr, Rl, Rr, lds, amax_lds, red, stamps)) return true;
    if (n2 == 2 && !(P.fast & 128) && wg_fused_merge_mfma(ck, xc, ac, M, p, q, n1, Dl, rhl, rhr, Rl, Rr, lds, amax_lds, red)) return true;
    const long long ldk = (long long)n1 * Dl;                           // column stride of C_k viewed as (n1*Dl) x r_mid
    for (int a1 = 0; a1 < Rl; ++a1) {
        const View Av = mkview(ck + a1 * ldk, Am.r, plain((long long)Rl * ldk));
        const View Bv = mkview(xc, plain(n2), Idx{n2, 1, (long long)n2 * rhl});
        const View Cv = mkview(Tbuf + (long long)a1 * p * ncol, plain(ncol), plain(1));
        wg_gemm(p, ncol, rhl, Av, Bv, Cv, 1.0, 0.0, lds);
    }
    // (ii): one thread per (row, nu): Rl*n2 inputs, n2*Rr outputs.  The operator core is staged in LDS as opl[o*nin + i]
    // (o = s2 + n2*a, i = j + n2*a'; the GEMM region is free between GEMM calls): the inner product then costs one broadcast
    // LDS read and one FMA per term instead of an indexed global load.
    double mx = 0.0;
    const int nin = Rl * n2, nout = n2 * Rr;
    lds_f64* opl = (lds_f64*)lds;
    for (int e = threadIdx.x; e < nin * nout; e += TTN_WG) {
        const int i = e % nin, o = e / nin;
        opl[e] = ac[(o % n2) + n2 * ((i % n2) + n2 * ((i / n2) + (long long)Rl * (o / n2)))];
    }
    __syncthreads();
    const long long tstride = (long long)p * ncol;                          // doubles between T_a' and T_a'+1
    for (int e = threadIdx.x; e < p * rhr; e += TTN_WG) {
        const int nu = e % rhr, row = e / rhr;
        const double* tp = Tbuf + (long long)row * ncol + n2 * nu;
        double t[FUSE_MAX_TERMS];
#pragma unroll
        for (int i = 0; i < FUSE_MAX_TERMS; ++i) t[i] = (i < nin) ? tp[(i / n2) * tstride + (i % n2)] : 0.0;
        double* mrow = M + (long long)row * q + (long long)nout * nu;
        for (int o = 0; o < nout; ++o) {
            const lds_f64* w = opl + o * nin;
            double v = 0.0;
#pragma unroll
            for (int i = 0; i < FUSE_MAX_TERMS; ++i)
                if (i < nin) v = fma(t[i], w[i], v);
            mrow[o] = v;
            mx = fmax(mx, fabs(v));
        }
    }
    mx = wg_max(mx, red);
    if (threadIdx.x == 0) *amax_lds = mx;
    __syncthreads();
    return true;
}

// -------------------------------------------------------------------------------------------------
// Bond steps with a SHORT side of at most 8 rows (the first and last three steps of each half sweep: p = 2, 4, 8, q <= 256; with
// 12..16 rows the row-wise Jacobi below loses to LQ + Jacobi on the triangle: measured).
// Through the general machinery — GEMM calls, Householder LQ, Jacobi image, output GEMM, each with its descriptor, barriers and
// global round trips — such a step costs 80..300 k clk, almost all of it fixed overhead (a 2 x 24 matrix: 154 k clk; twenty of them
// are 5 % of a train alone on a CU).  Here the merged matrix lives in LDS from the product to the outputs:
//   M = A' B' by one thread per entry;  one-sided Jacobi on the ROWS of M (one wave per pair, round-robin rounds; no LQ first:
//   the rows are at most 256 long) with the same rotations applied to a p x p identity — rows of the result are sigma_i v_i^T, the
//   rotated identity is U^T;  sort, rank rule, and U sqrt(S) / sqrt(S) V^T are written straight from LDS.
// Same conventions as the Householder route: units of s0 = max |M|, negligible rows (norm^2 <= aneg) are left alone and come out as
// exact zeros, convergence when a whole sweep rotates nothing (|g| <= tol sqrt(a b), tol = jtol_mult sqrt(q) eps).
// -------------------------------------------------------------------------------------------------
#ifndef SMALL_STEP_PMAX
#define SMALL_STEP_PMAX 8
#endif
#define SMALL_STEP_QMAX 256
// (Everything it needs of the kernel's argument block and of the step's context comes BY VALUE in SmallStepArgs: a reference to either
// would force the caller — the force-inlined bond step, at its register limit — to keep them in memory: +280 spilled VGPRs, -6 % measured.)
struct SmallStepArgs {
    double jneg_mult, jtol_mult, truncerr;
    long long max_bond;
    int rank_rule, pmax;
    double* sv_row;              // null, or the pmax singular values of this (train, step)
    int* status_b;               // the train's status word
    double *red, *scal, *sigs;   // LDS reduction scratch, LDS scalars, singular values out
    int *perm, *iflag;
};
__device__ __noinline__ int wg_bond_small(SmallStepArgs Q, double* ck, double* ck1, int n1, int n2, int Dl, int rm, int Dr, long long* rank_out, double* lds) {
    rm = uni32(rm); n1 = uni32(n1); n2 = uni32(n2); Dl = uni32(Dl); Dr = uni32(Dr);
    ck = unip(ck); ck1 = unip(ck1); rank_out = unip(rank_out); lds = unip(lds);
    // A_mat[(al + Dl*s1), ga] = core_k[s1, al, ga] ; B_mat[ga, (s2 + n2*be)] = core_{k+1}[s2, ga, be]   (as in wg_bond_step_io)
    const int mr = n1 * Dl, mc = n2 * Dr;
    const View Am = mkview(ck, Idx{Dl, (long long)n1, 1}, plain((long long)n1 * Dl));
    const View Bm = mkview(ck1, plain(n2), Idx{n2, 1, (long long)n2 * rm});
    const int wide = (mr <= mc) ? 1 : 0;
    const int p = wide ? mr : mc, q = wide ? mc : mr;
    const View Ap = wide ? Am : tview(Bm);       // p x rm
    const View Bp = wide ? Bm : tview(Am);       // rm x q
    Q.jneg_mult = unif64(Q.jneg_mult); Q.jtol_mult = unif64(Q.jtol_mult); Q.truncerr = unif64(Q.truncerr); Q.max_bond = uni64(Q.max_bond);
    Q.rank_rule = uni32(Q.rank_rule); Q.pmax = uni32(Q.pmax); Q.sv_row = unip(Q.sv_row); Q.status_b = unip(Q.status_b);
    Q.red = unip(Q.red); Q.scal = unip(Q.scal); Q.sigs = unip(Q.sigs); Q.perm = unip(Q.perm); Q.iflag = unip(Q.iflag);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LDQ = q + 1;
    lds_f64* Ms = (lds_f64*)lds;                                   // Ms[i * LDQ + j]
    lds_f64* Es = Ms + SMALL_STEP_PMAX * (SMALL_STEP_QMAX + 1);    // Es[i * 17 + j]: the rotated identity (row i = U[:, i]^T)
    lds_f64* nr = Es + SMALL_STEP_PMAX * 17;                       // squared row norms (16)
    lds_i32* flg = (lds_i32*)(nr + 16);
    gmem_f64* Ag = (gmem_f64*)Ap.p;
    gmem_f64* Bg = (gmem_f64*)Bp.p;
    __syncthreads();
    // ---- M = A' B' (p x q, inner rm), max |M| ----
    double mx = 0.0;
    for (int e = tid; e < p * q; e += TTN_WG) {
        const int i = e / q, j = e - i * q;
        const long long ao = ix(Ap.r, i), bo = ix(Bp.c, j);
        double acc = 0.0;
        for (int kk = 0; kk < rm; ++kk) acc = fma(Ag[ao + ix(Ap.c, kk)], Bg[bo + ix(Bp.r, kk)], acc);
        Ms[i * LDQ + j] = acc;
        mx = fmax(mx, fabs(acc));
    }
    for (int e = tid; e < p * 17; e += TTN_WG) Es[e] = ((e / 17) == (e % 17)) ? 1.0 : 0.0;
    mx = unif64(wg_max(mx, Q.red));
    const double s0 = (mx > 0.0) ? mx : 1.0, inv_s0 = 1.0 / s0;
    for (int e = tid; e < p * q; e += TTN_WG) { const int i = e / q, j = e - i * q; Ms[i * LDQ + j] *= inv_s0; }
    __syncthreads();
    // ---- squared row norms, negligible threshold ----
    double amax = 0.0;
    for (int i = wave; i < p; i += TTN_NWAVES) {
        double a = 0.0;
        for (int j = lane; j < q; j += 64) { const double v = Ms[i * LDQ + j]; a = fma(v, v, a); }
        a = wave_sum(a);
        if (lane == 0) nr[i] = a;
        amax = fmax(amax, a);
    }
    amax = unif64(wg_max(amax, Q.red));
    const double aneg = Q.jneg_mult * Q.jneg_mult * (double)q * DBL_EPSILON * DBL_EPSILON * amax;
    const double tol = Q.jtol_mult * sqrt((double)q) * DBL_EPSILON, tol2 = tol * tol;
    if (tid == 0) Q.scal[0] = aneg;
    // ---- one-sided Jacobi on the rows ----
    const int pe = p + (p & 1), half = pe >> 1;
    int sweeps = 0, conv = (p < 2);
    for (; !conv && sweeps < JACOBI_MAX_SWEEPS; ++sweeps) {
        if (tid == 0) *flg = 0;
        __syncthreads();
        int rotated = 0;
        for (int round = 0; round < pe - 1; ++round) {
            for (int kk = wave; kk < half; kk += TTN_NWAVES) {
                int i, j;
                if (kk == 0) { i = round; j = pe - 1; }
                else { i = (round + kk) % (pe - 1); j = (round + pe - 1 - kk) % (pe - 1); }
                if (i > j) { const int t = i; i = j; j = t; }
                if (j >= p) continue;                                  // bye
                lds_f64* xi = Ms + i * LDQ;
                lds_f64* xj = Ms + j * LDQ;
                // (rows of up to 256 entries: four per lane in registers between the dot products and the rotation)
                double u[4], v[4];
                double a = 0.0, b_ = 0.0, g = 0.0;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int c = lane + 64 * t4;
                    u[t4] = (c < q) ? xi[c] : 0.0; v[t4] = (c < q) ? xj[c] : 0.0;
                    a = fma(u[t4], u[t4], a); b_ = fma(v[t4], v[t4], b_); g = fma(u[t4], v[t4], g);
                }
                a = wave_sum(a); b_ = wave_sum(b_); g = wave_sum(g);
                if (a <= aneg || b_ <= aneg) continue;
                if (g * g <= tol2 * a * b_) continue;
                // t = tan(theta) = sign(d) h / (|d| + sqrt(d^2 + h^2)), d = b - a, h = 2 g: hardware seeds are enough for t (it only sets the
                // speed of convergence); c = rsqrt(1 + t^2) with Newton steps so that c^2 + s^2 = 1 to rounding (as in the image Jacobi)
                const double dd = b_ - a, hh = g + g;
                const double hyp = __builtin_amdgcn_sqrt(fma(dd, dd, hh * hh));
                const double t = copysign(hh * __builtin_amdgcn_rcp(fabs(dd) + hyp), hh * dd);
                const double cs = fast_rsqrt2(fma(t, t, 1.0)), sn = cs * t;
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4) {
                    const int c = lane + 64 * t4;
                    if (c < q) { xi[c] = cs * u[t4] - sn * v[t4]; xj[c] = sn * u[t4] + cs * v[t4]; }
                }
                if (lane < p) { const double u = Es[i * 17 + lane], v = Es[j * 17 + lane]; Es[i * 17 + lane] = cs * u - sn * v; Es[j * 17 + lane] = sn * u + cs * v; }
                rotated = 1;
            }
            __syncthreads();
        }
        if (rotated && lane == 0) atomicOr((int*)flg, 1);
        __syncthreads();
        conv = !(*flg);
        __syncthreads();
    }
    // ---- sigma_i = ||row i||, sorted descending (stable) ----
    for (int i = wave; i < p; i += TTN_NWAVES) {
        double a = 0.0;
        for (int j = lane; j < q; j += 64) { const double v = Ms[i * LDQ + j]; a = fma(v, v, a); }
        a = wave_sum(a);
        if (lane == 0) nr[i] = sqrt(a);
    }
    __syncthreads();
    for (int c = tid; c < p; c += TTN_WG) {
        const double sc = nr[c];
        int pos = 0;
        for (int j = 0; j < p; ++j) { const double sj = nr[j]; pos += (sj > sc) || (sj == sc && j < c); }
        Q.perm[pos] = c;
        Q.sigs[pos] = sc;
    }
    __syncthreads();
    const int r = wg_rank_rule_core(Q.rank_rule, Q.truncerr, Q.max_bond, Q.sigs, Q.iflag, p, p, s0);
    if (Q.sv_row)
        for (int i = tid; i < Q.pmax; i += TTN_WG) Q.sv_row[i] = (i < p) ? Q.sigs[i] * s0 : -1.0;
    // ---- outputs: left factor (p x r) = U sqrt(s0 Sigma), right factor (r x q) = sqrt(s0 / Sigma) (sigma v^T) ----
    const View Lfv = mkview(ck, Idx{Dl, (long long)n1, 1}, plain((long long)n1 * Dl));
    const View Rfv = mkview(ck1, plain(n2), Idx{n2, 1, (long long)n2 * r});
    const View Lo = wide ? Lfv : tview(Rfv);
    const View Ro = wide ? Rfv : tview(Lfv);
    const double sq0 = sqrt(s0);
    for (int e = tid; e < p * r; e += TTN_WG) {
        const int row = e % p, j = e / p;
        const double sj = Q.sigs[j];
        const int pj = Q.perm[j];
        const bool keep = (sj > 0.0) && (sj * sj > aneg);
        Lo.p[ix(Lo.r, row) + ix(Lo.c, j)] = keep ? Es[pj * 17 + row] * (sq0 * sqrt(sj)) : 0.0;
    }
    for (int e = tid; e < r * q; e += TTN_WG) {
        const int col = e % q, j = e / q;
        const double sj = Q.sigs[j];
        const int pj = Q.perm[j];
        const bool keep = (sj > 0.0) && (sj * sj > aneg);
        Ro.p[ix(Ro.r, j) + ix(Ro.c, col)] = keep ? Ms[pj * LDQ + col] * (sq0 / sqrt(sj)) : 0.0;
    }
    if (tid == 0) { *rank_out = r; if (!conv) *Q.status_b = 1; }
    __syncthreads();
    return sweeps;
}

// One bond step on (core_k, core_{k+1}), 0-based k.  src/tt_tools.jl:743-768 with the effective
// _svdtrunc of src/tt_cross_interpolation.jl:149-166.
//
// With A' (p x rm), B' (rm x q) the two cores viewed as matrices (transposed roles when the merged matrix
// is tall) and M = A'B' (p <= q), three numerically distinct routes produce U sqrt(S) and sqrt(S) V^T:
//   F  (rm < p)  factored: Cholesky of A'^T A' and B' B'^T, Jacobi SVD of the rm x rm core L_A^T L_B,
//                outputs by small GEMMs; never forms M.                     [R->L steps of a sweep]
//   G  (q > p)   Gram: L = chol(M M^T), Jacobi on the columns of L.         [L->R steps]
//   H            robust: blocked Householder LQ of M (or M itself if square), Jacobi on its columns.
// F and G square the condition number, so they are taken only if sigma_max/sigma_min <= FAST_KAPPA_MAX and are
// verified a posteriori (Rf Rf^T = Sigma, Lf^T Lf = Sigma to FAST_CHECK_TOL); otherwise the step is redone by H.
// SWAP != 0 is the site-swap variant of the same step (_ttm_swap!, src/tt_operations.jl:365-382, and _swap_adjacent_sites,
// src/qtt_tools.jl:660-695; both cores have the same physical dimension here): the merged matrix takes its row index from
// (left rank, physical index of core k+1) and its column index from (physical index of core k, right rank), and the factors
// are U and S*Vt instead of U sqrt(S), sqrt(S) Vt.  Only route H is used (the swapped matrices are rank deficient by design).
// symmetric eigensolver of the Gram route (ttn_eig_kernels.h, included after this header by the translation unit)
__device__ int wg_eig128(const double* Gg, double* Vst, int r, int nev, double* sig, double* lds, int* iwork, double* dwork, long long* prof);
__device__ int wg_eig64(const double* Gg, int ldg, double* Vst, int r, int nev, double* sig, double* lds, int* iwork, double* dwork);

struct BondIO {
    double *ck, *ck1;            // the two cores (slots)
    int n1, n2, Dl, rm, Dr;      // physical dimensions and the three ranks
    long long* rank_out;         // receives the new middle rank (SWAP: -1 if it exceeds `cap`; nothing is written then)
    int cap;                     // SWAP only: largest middle rank the two slots can hold
};
template <int SWAP>
__device__ __forceinline__ void wg_bond_step_io(const CompressArgs& P, int b, const BondIO& io, int k, int step, double* lds, bool virt) {
    const int tid = threadIdx.x;
    const int n1 = io.n1, n2 = io.n2;
    const int Dl = io.Dl, rm = io.rm, Dr = io.Dr;
    const int mr = n1 * Dl, mc = n2 * Dr;
    double* ck = io.ck;
    double* ck1 = io.ck1;
    // A_mat[(al + Dl*s1), ga] = core_k[s1, al, ga] ; B_mat[ga, (s2 + n2*be)] = core_{k+1}[s2, ga, be]
    const View Am = mkview(ck, Idx{Dl, (long long)n1, 1}, plain((long long)n1 * Dl));
    const View Bm = mkview(ck1, plain(n2), Idx{n2, 1, (long long)n2 * rm});
    const bool wide = mr <= mc;
    const int p = wide ? mr : mc, q = wide ? mc : mr;
    const View Ap = wide ? Am : tview(Bm);       // p x rm
    const View Bp = wide ? Bm : tview(Am);       // rm x q

    BondCtx S;
    S.ldsX = lds;                                         // COMPRESS_LDS_X_DOUBLES (aliases the GEMM tiles)
    S.red = lds + GEMM_LDS_TOTAL;                         // 32 (the GEMM descriptor sits right behind the X region)
    S.Ts = S.red + 32;                                    // QR_NB*QR_NB
    S.Ss = S.Ts + QR_NB * QR_NB;                          // QR_NB*QR_NB
    S.taus = S.Ss + QR_NB * QR_NB;                        // QR_NB
    S.scal = S.taus + QR_NB;                              // 8 misc doubles (scal[0] = Jacobi's negligible threshold)
    S.iflag = reinterpret_cast<int*>(S.scal + 8);         // 8 ints
    S.nrm2 = S.scal + 16;                                 // 128 cached squared column norms
    // scratch belongs to the WORKGROUP SLOT, not to the train: a persistent grid (k_compress with a train counter) reuses the slot for
    // every train the workgroup pulls, so the hot footprint is #slots x per_slot whatever the batch size (every other caller
    // launches one workgroup per train: blockIdx.x == b there)
    double* scr = P.scratch + (long long)blockIdx.x * P.scratch_stride;
    const long long pq = (long long)P.pmax * P.qmax;
    S.M = scr;                                            // p x q row-major
    S.M2 = S.M + pq;                                      // copy for LQ (F path: Lf/Rf staging uses M..M2)
    S.Vb = S.M2 + pq;                                     // QR_NB x qmax
    S.Wb = S.Vb + (long long)QR_NB * P.qmax;              // pmax x QR_NB
    S.Us = S.Wb + (long long)P.pmax * QR_NB;              // pmax x pmax
    S.Xg = S.Us + (long long)P.pmax * P.pmax;             // pmax x pmax Jacobi fallback when LDS is too small
    S.sig = S.Xg + (long long)P.pmax * P.pmax;            // pmax
    S.sigs = S.sig + P.pmax;                              // pmax
    S.perm = reinterpret_cast<int*>(S.sigs + P.pmax);     // pmax ints
    S.Ga = S.sigs + 2 * P.pmax;                           // 6 x (128 x 128) fast-path matrices
    S.Gb = S.Ga + 128 * 128;
    S.Cc = S.Gb + 128 * 128;
    S.T1 = S.Cc + 128 * 128;
    S.T2 = S.T1 + 128 * 128;
    S.T3 = S.T2 + 128 * 128;
    if (pq >= 3 * 128 * 128 && !(P.fast & 64)) {
        // the Gram matrix, the reflector store and the check matrix take the place of the fused merge's T buffer / the LQ copy
        // (dead by the time they are written): 288 KB less footprint per train, and the lines are warm
        S.Ga = S.M2; S.Gb = S.M2 + 128 * 128; S.T2 = S.M2 + 2 * 128 * 128;
    }

    // fused apply: the right core is still A_{k+1} x_{k+1}.  Wide steps with r_mid >= p get the fused merge below; the
    // others (first steps of the ramp that are tall, tiny cores) write the core out first and proceed as usual.
    bool virt_live = false;
    if (virt) {
        if (wide && rm >= p) virt_live = true;
        else wg_materialize_core(P, b, k + 1);
    }

    long long t_prev = P.prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    // fine-grained marks of ONE step (TTN_PROF_STEP): 64 more counters per train behind the phase and step tables
#define FINE_MARK(id) if (P.prof && P.prof_step == step) { __syncthreads(); if (tid == 0) { long long t_now = (long long)__builtin_amdgcn_s_memtime(); P.prof[(long long)P.tt.batch * 136 + (long long)b * 64 + (id)] += t_now - t_fine; t_fine = t_now; } }
    long long t_fine = t_prev;
#define PROF_MARK(slot) if (P.prof) { __syncthreads(); if (tid == 0) { long long t_now = (long long)__builtin_amdgcn_s_memtime(); if (P.prof_step < 0 || P.prof_step == step) P.prof[(long long)b * 16 + (slot)] += t_now - t_prev; t_prev = t_now; } }

    const View Lfv = mkview(ck, Idx{Dl, (long long)n1, 1}, plain((long long)n1 * Dl));     // (mr x r)
    int route = 2;                                        // 0 = F, 1 = G, 2 = H (for the diagnostics)
    int nsw_total = 0;
    bool done = false;

    // =============================== route F: factored ===============================
    if (SWAP == 0 && P.fast && rm < p && rm <= TTN_LDS_COLS && rm >= 2) {
        // scales
        double sa = 0.0, sb = 0.0;
        sa = wg_absmax(ck, (long long)n1 * Dl * rm, S.red);
        sb = wg_absmax(ck1, (long long)n2 * rm * Dr, S.red);
        FINE_MARK(0)
        bool ok = (sa > 0.0) && (sb > 0.0);
        const double sA = wide ? sa : sb, sB = wide ? sb : sa;       // scale of A', B'
        const double s0 = sA * sB;
        const View Gav = mkview(S.Ga, plain(1), plain(128));
        const View Gbv = mkview(S.Gb, plain(1), plain(128));
        const View Ccv = mkview(S.Cc, plain(1), plain(128));
        bool diagA = false;
        if (ok) {
            wg_syrk(rm, p, tview(Ap), Gav, 1.0 / (sA * sA), lds);          // A'^T A'
            FINE_MARK(1)
            wg_syrk(rm, q, Bp, Gbv, 1.0 / (sB * sB), lds);          // B' B'^T
            FINE_MARK(2)
            PROF_MARK(8)
            // A' = U D^(1/2) with orthonormal U (the left core of a bond step is left as U sqrt(S) by the step before it, so every
            // R->L step of a sweep that follows an L->R sweep sees this): then M = U (D^(1/2) B') and the SVD of M is U times the SVD
            // of the rm x q matrix N = D^(1/2) B' — no Cholesky factors, no core matrix, two output GEMMs instead of five.
            // Detected, not assumed: |G_ij| <= FAST_DIAG_TOL sqrt(G_ii G_jj) for every off-diagonal entry of A'^T A'.
            if (rm == 64 && !(P.fast & 4) && !(P.fast & 8)) {
                const double wmax = wg_diag_test_t3(S.Ga, S.Gb, S.T3, S.Ss, S.red);      // (T3 = D^(1/2) (B' B'^T) D^(1/2), used if the test passes)
                diagA = wmax <= FAST_DIAG_TOL;
                if (P.prof && tid == 0) {            // diagnostics: largest off-diagonal level seen (1e-18 units), steps tested / taken
                    long long* pf = P.prof + (long long)b * 16;
                    const long long wl = (long long)fmin(wmax * 1e18, 9e18);
                    if (wl > pf[12]) pf[12] = wl;
                    pf[13] += 1; pf[14] += diagA ? 1 : 0;
                }
            }
            FINE_MARK(3)
            if (diagA) {
                // (the Gram matrix of N = D^(1/2) B', D^(1/2) (B' B'^T) D^(1/2), is in T3 already)
                FINE_MARK(4)
            } else if (rm <= TTN_LDS_COLS / 2) {
                // both Cholesky factorisations at once, one per half of the workgroup (wg_chol2_lds128)
                for (int e = tid; e < rm * 128; e += TTN_WG) if ((e & 127) < rm) { S.ldsX[e] = S.Ga[e]; S.ldsX[TTN_LDS_IMG / 2 + e] = S.Gb[e]; }
                __syncthreads();
                ok = wg_chol2_lds128(rm, S.ldsX, S.Ts, S.iflag, S.scal + 1) == 0;     // Ts: 256 doubles of LDS, free here
                for (int e = tid; e < rm * 128; e += TTN_WG) if ((e & 127) < rm) { S.Ga[e] = S.ldsX[e]; S.Gb[e] = S.ldsX[TTN_LDS_IMG / 2 + e]; }
                __syncthreads();
            } else {
                // L_A
                for (int e = tid; e < rm * 128; e += TTN_WG) if ((e & 127) < rm) S.ldsX[e] = S.Ga[e];
                __syncthreads();
                ok = wg_chol_lds128(rm, S.ldsX, S.red, S.iflag, S.scal + 1) == 0;
                for (int e = tid; e < rm * 128; e += TTN_WG) if ((e & 127) < rm) S.Ga[e] = S.ldsX[e];
                __syncthreads();
                if (ok) {
                    for (int e = tid; e < rm * 128; e += TTN_WG) if ((e & 127) < rm) S.ldsX[e] = S.Gb[e];
                    __syncthreads();
                    ok = wg_chol_lds128(rm, S.ldsX, S.red, S.iflag, S.scal + 1) == 0;
                    for (int e = tid; e < rm * 128; e += TTN_WG) if ((e & 127) < rm) S.Gb[e] = S.ldsX[e];
                    __syncthreads();
                }
            }
        }
        int r = 0, rk = 0;
        if (ok) {
            PROF_MARK(9)
            int nsw = 1;
            if (diagA) {
                ok = wg_eig64(S.T3, 128, S.T2, 64, 64, S.sigs, lds, reinterpret_cast<int*>(S.Ts), S.Ts + 64) == 0;
                FINE_MARK(5)
                for (int j = tid; j < 64; j += TTN_WG) S.perm[j] = j;
                if (tid == 0) S.scal[0] = P.jneg_mult * P.jneg_mult * 64.0 * DBL_EPSILON * DBL_EPSILON * S.sigs[0] * S.sigs[0];
                __syncthreads();
            } else {
            // core C = L_A^T L_B  (rm x rm)
            wg_gemm(rm, rm, rm, tview(Gav), Gbv, Ccv, 1.0, 0.0, lds);
            if (rm == 64 && !(P.fast & 4)) {
                // the 64 x 64 core through the symmetric eigensolver: C C' = U S^2 U' gives the same image x_j = sigma_j u_j the
                // Jacobi on the columns of C leaves (ttn_eig_kernels.h; the a-posteriori check below covers the squared condition)
                wg_gemm(64, 64, 64, Ccv, tview(Ccv), mkview(S.T3, plain(1), plain(128)), 1.0, 0.0, lds);
                ok = wg_eig64(S.T3, 128, S.T2, 64, 64, S.sigs, lds, reinterpret_cast<int*>(S.Ts), S.Ts + 64) == 0;
                for (int j = tid; j < 64; j += TTN_WG) S.perm[j] = j;
                if (tid == 0) S.scal[0] = P.jneg_mult * P.jneg_mult * 64.0 * DBL_EPSILON * DBL_EPSILON * S.sigs[0] * S.sigs[0];
                __syncthreads();
            } else {
                for (int e = tid; e < rm * 128; e += TTN_WG) { const int c = e >> 7, r_ = e & 127; S.ldsX[e] = (r_ < rm) ? S.Cc[c * 128 + r_] : 0.0; }
                __syncthreads();
                nsw = uni32(wg_svd_cols(P, S, rm, S.ldsX, 128, true));
                nsw_total += (nsw < 0 ? -nsw : nsw);
            }
            }
            ok = ok && (nsw > 0) && (S.sigs[rm - 1] * FAST_KAPPA_MAX >= S.sigs[0]) && (S.sigs[rm - 1] * S.sigs[rm - 1] > S.scal[0]);
            PROF_MARK(10)
        }
        if (ok) {
            r = wg_rank_rule(P, S, rm, p, s0);
            rk = r < rm ? r : rm;                                           // columns that carry data
            FINE_MARK(6)
            // X_s[:, j] = x_j * (sqrt(s0)/sA) / sigma_j^2.5 -> T1 ;  X_t[:, j] = x_j * (sqrt(s0)/sB) / sigma_j^1.5 -> T2
            const double fa = sqrt(s0) / sA, fb = sqrt(s0) / sB;
            if (diagA) {
                // x_j = sigma_j w_j (w_j: left singular vector of N).  Lf = A' (D^(-1/2) w_j sqrt(s0 sigma_j) / sA),
                // Rf = (D^(1/2) w_j sqrt(s0) / (sB sqrt(sigma_j)))^T B'
                wg_scale_t12_diag(S.ldsX, S.sigs, S.perm, S.Ga, S.T1, S.T2, rk, fa, fb, S.Ts);
            } else {
                for (int e = tid; e < rm * rk; e += TTN_WG) {
                    const int row = e % rm, j = e / rm;
                    const double sj = S.sigs[j], xv = S.ldsX[S.perm[j] * 128 + row];
                    const double rs = sqrt(sj);
                    S.T1[j * 128 + row] = xv * (fa / (sj * sj * rs));
                    S.T2[j * 128 + row] = xv * (fb / (sj * rs));
                }
                __syncthreads();
            }
            const View T1v = mkview(S.T1, plain(1), plain(128));
            const View T2v = mkview(S.T2, plain(1), plain(128));
            double* LfT = S.M;                                              // p x rk, column-major (ld = p)
            double* RfT = S.M + (long long)p * rk;                          // rk x q, row-major (ld = q)
            const View Lft = mkview(LfT, plain(1), plain(p));
            const View Rft = mkview(RfT, plain(q), plain(1));
            FINE_MARK(7)
            if (diagA) {
                wg_gemm_ra(rk, p, rm, tview(T1v), tview(Ap), tview(Lft), 1.0, lds);          // Lf^T = T1^T A'^T: the short operand in registers
                FINE_MARK(8)
                wg_gemm_ra(rk, q, rm, tview(T2v), Bp, Rft, 1.0, lds);
                FINE_MARK(9)
            } else {
            // Lf = A' * (L_B * (C^T * X_s))     (ldsX is free again: use it as the second temporary)
            wg_gemm(rm, rk, rm, tview(Ccv), T1v, mkview(S.T3, plain(1), plain(128)), 1.0, 0.0, lds);
            wg_gemm(rm, rk, rm, Gbv, mkview(S.T3, plain(1), plain(128)), T1v, 1.0, 0.0, lds);
            wg_gemm(p, rk, rm, Ap, T1v, Lft, 1.0, 0.0, lds);
            // Rf = (L_A * X_t)^T * B'
            wg_gemm(rm, rk, rm, Gav, T2v, mkview(S.T3, plain(1), plain(128)), 1.0, 0.0, lds);
            wg_gemm(rk, q, rm, tview(mkview(S.T3, plain(1), plain(128))), Bp, Rft, 1.0, 0.0, lds);
            }
            // a-posteriori check: Lf^T Lf = Sigma, Rf Rf^T = Sigma
            wg_syrk(rk, p, tview(Lft), mkview(S.T1, plain(1), plain(128)), 1.0, lds);
            FINE_MARK(10)
            wg_syrk(rk, q, Rft, mkview(S.T2, plain(1), plain(128)), 1.0, lds);
            FINE_MARK(11)
            const double e1 = wg_check_diag_tab(S.sigs, S.T1, 128, rk, s0, S.Ts, S.red);
            const double e2 = wg_check_diag_tab(S.sigs, S.T2, 128, rk, s0, S.Ts, S.red);
            ok = (e1 <= FAST_CHECK_TOL) && (e2 <= FAST_CHECK_TOL);
            FINE_MARK(12)
            if (ok) {
                if (P.sv_out && step < P.sv_steps) {
                    double* so = P.sv_out + ((long long)b * P.sv_steps + step) * P.pmax;
                    for (int i = tid; i < P.pmax; i += TTN_WG) so[i] = (i < rm) ? S.sigs[i] * s0 : (i < p ? 0.0 : -1.0);
                }
                // commit: cores are overwritten only now
                const View Rfv = mkview(ck1, plain(n2), Idx{n2, 1, (long long)n2 * r});
                const View Lo = wide ? Lfv : tview(Rfv);       // p x r
                const View Ro = wide ? Rfv : tview(Lfv);       // r x q
                __syncthreads();
                wg_copy_to_view(Lo, LfT, p, r, 1, p, rk, 1);               // Lo[row, j] = LfT[row + p j], columns j >= rk zero
                FINE_MARK(13)
                wg_copy_to_view(Ro, RfT, r, q, q, 1, rk, 0);               // Ro[j, col] = RfT[j q + col], rows j >= rk zero
                FINE_MARK(14)
                if (tid == 0) *io.rank_out = r;
                __syncthreads();
                done = true;
                route = diagA ? 3 : 0;
            }
        }
        PROF_MARK(6)
    }

    if (!done) {
        // ---- merge: M = A' * B'  (p x q), scaled to max|M| = 1 ----
        const View Mv = mkview(S.M, plain(q), plain(1));
        // M stays UNSCALED in memory: max|M| comes out of the GEMM epilogue and the consumers (Gram, LQ copy, output GEMM)
        // apply 1/s0 as their alpha — no extra read-modify-write pass over the p x q matrix.
        bool merged = false;
        if (virt_live) {
            FINE_MARK(26)
            merged = wg_fused_merge(P, b, k, p, q, Am, S.M, S.M2, pq, lds, S.scal + 6, S.red,
                                    (P.prof && P.prof_step == step) ? P.prof + (long long)P.tt.batch * 136 + (long long)b * 64 + 48 : nullptr);
            FINE_MARK(27)
            if (!merged) wg_materialize_core(P, b, k + 1);
        }
        double mx_swap = 0.0;
        if (SWAP != 0) {
            // M[(m + Dl*sB), (sA + n1*nn)] = sum_a core_k[sA, m, a] * core_{k+1}[sB, a, nn]: one Dl x Dr x rm product per (sA, sB),
            // written into its strided block of M (transposed when the merged matrix is tall)
            for (int sA = 0; sA < n1; ++sA)
                for (int sB = 0; sB < n2; ++sB) {
                    const View Av = mkview(ck + sA, plain(n1), plain((long long)n1 * Dl));          // Dl x rm
                    const View Bv = mkview(ck1 + sB, plain(n2), plain((long long)n2 * rm));         // rm x Dr
                    if (wide) wg_gemm(Dl, Dr, rm, Av, Bv, mkview(S.M + (long long)Dl * sB * q + sA, plain(q), plain(n1)), 1.0, 0.0, lds, S.scal + 6);
                    else wg_gemm(Dr, Dl, rm, tview(Bv), tview(Av), mkview(S.M + (long long)sA * q + (long long)Dl * sB, plain((long long)n1 * q), plain(1)), 1.0, 0.0, lds, S.scal + 6);
                    mx_swap = fmax(mx_swap, unif64(S.scal[6]));
                    __syncthreads();
                }
        } else if (!merged) wg_gemm(p, q, rm, Ap, Bp, Mv, 1.0, 0.0, lds, S.scal + 6);
        PROF_MARK(0)
        const double mx = (SWAP != 0) ? mx_swap : unif64(S.scal[6]);
        const double s0 = (mx > 0.0) ? mx : 1.0;
        const double inv_s0 = 1.0 / s0;
        PROF_MARK(1)
        const bool need_lq = q > p;
        const bool x_in_lds = p <= TTN_LDS_COLS;              // fast Jacobi: X in LDS with leading dimension 128
        double* X = x_in_lds ? S.ldsX : S.Xg;
        int ldx = x_in_lds ? 128 : p;

        // small merged matrices (the rank-ramp steps) fit the LDS whole: their Householder LQ needs no GEMM calls and costs
        // about what the Gram + Cholesky do, without the conditioning gamble — route H directly
        const bool lq_in_lds = (long long)p * q <= GEMM_LDS_DOUBLES || GEMM_LDS_DOUBLES / p >= 2 * p;     // whole, or TSQR chunks (wg_lq_blocked)
        const bool eig_ok = SWAP == 0 && P.fast && !(P.fast & 2) && need_lq && ((p > 64 && p <= 128 && P.max_bond <= 64) || (p == 64 && P.max_bond < 64));
        // (the eigensolver leaves its image — at most 64 columns — in LDS whatever p is; the Cholesky + Jacobi form needs the whole L there)
        // CholeskyQR2 (above) instead of the Householder LQ when the plain Gram route finds the matrix too ill-conditioned for itself
        // (only where the Householder LQ cannot keep the whole matrix in LDS — 64 x 384, 32 x 384: there it wins 3x; against the in-LDS
        // LQ of a 64 x 128 step it is a draw, and the R->L ramp matrices are often too ill-conditioned for it anyway)
        const bool cholqr_ok = SWAP == 0 && P.fast && !(P.fast & 32) && need_lq && x_in_lds && p >= 16 && p <= 64 && (long long)p * q > GEMM_LDS_DOUBLES;
        for (int attempt = (SWAP == 0 && P.fast && need_lq && (eig_ok || cholqr_ok || (x_in_lds && !lq_in_lds)) && p >= 2) ? 1 : 2; attempt <= 2 && !done; ++attempt) {
            bool ok = true;
            bool use_eig = false;
            bool cholqr = false;
            double gram_trace = 0.0;
            X = (x_in_lds || attempt == 1) ? S.ldsX : S.Xg;
            ldx = (x_in_lds || attempt == 1) ? 128 : p;
            if (attempt == 1) {
                // =========================== route G: L = chol(M M^T) ===========================
                wg_syrk(p, q, Mv, mkview(S.Ga, plain(1), plain(128)), inv_s0 * inv_s0, lds);
                PROF_MARK(7)
                // p = 128 with at most 64 vectors kept (the L->R steps of the benchmark sweep): eigen-decomposition of the Gram
                // matrix itself — tridiagonalisation, bisection, twisted factorisations (ttn_eig_kernels.h) — instead of
                // Cholesky + Jacobi on L; same outputs (sigs, perm, X = sigma_j u_j in LDS), same a-posteriori check below
                use_eig = eig_ok;
                if (use_eig) {
                    if (p > 64 && p < 128) {             // 64 < p < 128: the Gram matrix zero-padded to the 128-row solver (the padding adds
                        for (int e = tid; e < 128 * 128; e += TTN_WG)      // exact zero eigenvalues below the wanted ones)
                            if ((e & 127) >= p || (e >> 7) >= p) S.Ga[e] = 0.0;
                        __syncthreads();
                    }
                    const int r0 = (int)P.max_bond;
                    const int nev = (P.truncerr > 0.0 || (P.sv_out && step < P.sv_steps)) ? p : r0;
                    // trace of the Gram matrix = sum of ALL squared singular values: the polish branch below needs the weight of
                    // the part it drops (the solver only computes the kept eigenvalues)
                    { double t_ = 0.0; for (int i = tid; i < p; i += TTN_WG) t_ += S.Ga[i * 129]; gram_trace = unif64(wg_sum(t_, S.red)); }
                    ok = ((p == 64) ? wg_eig64(S.Ga, 128, S.Gb, r0, nev, S.sigs, lds, reinterpret_cast<int*>(S.Ts), S.Ts + 64)
                                    : wg_eig128(S.Ga, S.Gb, r0, nev, S.sigs, lds, reinterpret_cast<int*>(S.Ts), S.Ts + 64,
                                                (P.prof && P.prof_step == step) ? P.prof + (long long)P.tt.batch * 136 + (long long)b * 64 + 32 : nullptr)) == 0;
                    for (int j = tid; j < p; j += TTN_WG) { S.perm[j] = j; if (j >= nev) S.sigs[j] = 0.0; }      // (sigs / perm hold pmax entries)
                    if (tid == 0) S.scal[0] = P.jneg_mult * P.jneg_mult * (double)p * DBL_EPSILON * DBL_EPSILON * S.sigs[0] * S.sigs[0];
                    __syncthreads();
                    PROF_MARK(11)
                } else {
                wg_img_load(S.ldsX, 0, S.Ga, p, S.red);
                ok = wg_chol_lds128(p, S.ldsX, S.red, S.iflag, S.scal + 1) == 0;
                // with truncerr > 0 the rank rule reads the SMALL singular values too: need cond(M) <= kappa_max overall;
                // likewise when nothing will be truncated (p <= max_bond: every singular value is kept): the pivot ratio is a lower
                // bound of cond(M)^2, so a value above the limit means the a-posteriori test WILL fail — take the factor from
                // CholeskyQR2 (the rank-ramp steps of a sweep are such steps), or go to Householder now instead of after a wasted Jacobi
                // The pivot ratio is only a LOWER bound of cond(M)^2 (unpivoted: orders of magnitude low on the R->L ramp matrices), so a
                // step that may take CholeskyQR2 always does: its second pass measures the conditioning itself.
                if (ok && (P.truncerr > 0.0 || (long long)p <= P.max_bond)) {
                    if (cholqr_ok) { if (unif64(S.scal[1]) <= CHOLQR_PIVOT_MAX) cholqr = true; else ok = false; }
                    else if (unif64(S.scal[1]) > FAST_KAPPA_MAX * FAST_KAPPA_MAX) ok = false;
                }
                if (cholqr) {
                    for (int e = tid; e < p * p; e += TTN_WG) { const int i = e % p, j = e / p; S.T1[i + 128 * j] = S.ldsX[i + 128 * j]; }       // L1 (the GEMM below takes the image)
                    wg_trsm_lower_cols(p, q, S.ldsX, S.M, q, inv_s0, S.M2, q);                                                           // Q1 = L1^-1 M / s0
                    PROF_MARK(8)
                    const View Q1v = mkview(S.M2, plain(q), plain(1));
                    wg_syrk(p, q, Q1v, mkview(S.Cc, plain(1), plain(128)), 1.0, lds);
                    PROF_MARK(9)
                    // Q1 Q1^T = I + E, |E| ~ eps cond(M)^2: the route needs |E| << 1 (CHOLQR_ORTH_MAX), else Householder
                    const double dev = wg_img_load(S.ldsX, 0, S.Cc, p, S.red);
                    ok = dev <= CHOLQR_ORTH_MAX && wg_chol_lds128(p, S.ldsX, S.red, S.iflag, S.scal + 1) == 0;
                    if (ok) {
                        wg_img_load(S.ldsX, 64, S.T1, p, S.red);
                        wg_tril_mul_lds(p, S.ldsX);
                    }
                    PROF_MARK(10)
                }
                for (int e = tid; e < p * 128; e += TTN_WG) if ((e & 127) >= p) S.ldsX[e] = 0.0;      // zero row padding for the Jacobi
                __syncthreads();
                PROF_MARK(11)
                }
            } else {
                // =========================== route H: Householder LQ ===========================
                if (need_lq) {
                    wg_copy_scale(S.M2, S.M, (long long)p * q, inv_s0);
                    __syncthreads();
                    wg_lq_blocked(p, q, S.M2, q, S.Vb, S.Wb, nullptr, nullptr, lds, S.Ts, S.Ss, S.taus, S.red);
                }
                const double* Lsrc = need_lq ? S.M2 : S.M;               // row-major, ld = q
#ifndef TTN_NO_PRESORT
                // columns of L in order of decreasing norm (de Rijk): the rank-deficient L of the ramp steps converges in
                // fewer sweeps that way; any column order is fine for the caller (wg_svd_cols sorts by sigma anyway)
                if (need_lq && x_in_lds) {
                    const int lane_ = tid & 63, wave_ = tid >> 6;
                    for (int c = wave_; c < p; c += TTN_NWAVES) {
                        double a = 0.0;
                        for (int r_ = c + lane_; r_ < p; r_ += 64) { const double v = Lsrc[(long long)r_ * q + c]; a = fma(v, v, a); }
                        a = wave_sum(a);
                        if (lane_ == 0) S.sig[c] = a;
                    }
                    __syncthreads();
                    for (int c = tid; c < p; c += TTN_WG) {
                        const double sc = S.sig[c];
                        int pos = 0;
                        for (int j = 0; j < p; ++j) { const double sj = S.sig[j]; pos += (sj > sc) || (sj == sc && j < c); }
                        S.perm[c] = pos;                                    // column c of L goes to position pos
                    }
                    __syncthreads();
                }
                const bool presort = need_lq && x_in_lds;
#else
                const bool presort = false;
#endif
                for (int e = tid; e < p * ldx; e += TTN_WG) {
                    const int r_ = e % ldx, c = e / ldx;              // X[r + ldx*c] = L[r][c]; rows >= p are zero padding
                    const double v = (r_ < p) ? Lsrc[(long long)r_ * q + c] * (need_lq ? 1.0 : inv_s0) : 0.0;
                    const int cd = presort ? S.perm[c] : c;
                    X[(long long)cd * ldx + r_] = (need_lq && c > r_) ? 0.0 : v;
                }
                __syncthreads();
                PROF_MARK(2)
            }
            int nsw = 0;
            if (ok && !(attempt == 1 && use_eig)) {
                nsw = uni32(wg_svd_cols(P, S, p, X, ldx, x_in_lds));
                nsw_total += (nsw < 0 ? -nsw : nsw);
                if (attempt == 1) ok = nsw > 0;
                else if (nsw < 0 && tid == 0) ttn_set_status(&P.status[b], 1);
            }
            PROF_MARK(3)
            if (!ok) continue;
            FINE_MARK(20)
            const int r = wg_rank_rule(P, S, p, p, s0);
            FINE_MARK(21)
            if (SWAP != 0 && r > io.cap) {                                  // would not fit the slots: report, write nothing
                if (tid == 0) { ttn_set_status(&P.status[b], 2); *io.rank_out = -1; }
                __syncthreads();
                done = true;
                continue;
            }
            if (attempt == 1 && !cholqr) {
                // Gram route: the KEPT block must be well conditioned (error ~ eps*kappa^2, verified below); the discarded
                // singular values only matter to the rank rule, i.e. when truncerr > 0 (then all of them must qualify).
                const int rl = (P.truncerr > 0.0) ? p : r;
                if (__builtin_expect(!((S.sigs[rl - 1] * FAST_KAPPA_MAX >= S.sigs[0]) && (S.sigs[rl - 1] * S.sigs[rl - 1] > S.scal[0])), 0)) {
                    // Moderately ill-conditioned kept block and everything kept fits the LDS image: POLISH instead of starting over.
                    // The eigenvectors U of the Gram matrix span the dominant subspace up to an angle theta ~ eps kappa^2 / relgap; the
                    // r x q matrix B = U^T M is then nearly row-orthogonal, and a one-sided Jacobi on its rows (2-3 sweeps) delivers the
                    // singular values and right vectors of M restricted to that subspace to Jacobi accuracy; the left vectors follow as
                    // M v / sigma — one step of subspace iteration, so the product of the two outputs is M projected on the computed
                    // right vectors and the singular values are off by theta^2 / 2 only (<= 5e-11 at FAST_KAPPA_POLISH).
                    // That bound needs a GAP behind the kept block: the error of the product is theta * sigma_{r+1} / sigma_1, so the
                    // branch is taken only when what it drops is noise — dropped weight trace(G) - sum of the kept eigenvalues at most
                    // FAST_POLISH_DROP of the trace (sigma_{r+1} <= 1e-6 sigma_1: relgap = 1, theta <= eps kappa^2 = 1e-7, product
                    // error <= 1e-13).  The rank-deficient ramp step of a sweep that was just truncated on its other bond is such a
                    // step (its dropped part is rounding noise); a matrix with a smooth spectrum goes to the Householder route.
                    bool polish = use_eig && SWAP == 0 && !(P.fast & 16) && P.truncerr == 0.0 && q <= 128 && r <= 64 && r >= 2 &&
                                  (S.sigs[r - 1] * FAST_KAPPA_POLISH >= S.sigs[0]) && (S.sigs[r - 1] * S.sigs[r - 1] > S.scal[0]);
                    if (polish) {
                        double kept = 0.0;
                        for (int j = tid; j < r; j += TTN_WG) kept = fma(S.sigs[j], S.sigs[j], kept);
                        kept = unif64(wg_sum(kept, S.red));
                        polish = (gram_trace - kept) <= FAST_POLISH_DROP * gram_trace;
                    }
                    if (!polish) continue;
                    const double aneg0 = S.scal[0];
                    for (int e = tid; e < p * r; e += TTN_WG) {
                        const int row = e % p, j = e / p;
                        const double sj = S.sigs[j], xv = X[(long long)S.perm[j] * ldx + row];
                        S.Us[(long long)j * p + row] = ((sj > 0.0) && (sj * sj > aneg0)) ? xv / sj : 0.0;
                    }
                    __syncthreads();
                    wg_gemm_ra(r, q, p, mkview(S.Us, plain(p), plain(1)), Mv, mkview(S.M2, plain(q), plain(1)), inv_s0, lds);
                    for (int e = tid; e < r * 128; e += TTN_WG) { const int c = e & 127, j = e >> 7; S.ldsX[e] = (c < q) ? S.M2[(long long)j * q + c] : 0.0; }
                    __syncthreads();
                    const int nsp = uni32(wg_svd_cols(P, S, r, S.ldsX, 128, true, q));
                    nsw_total += (nsp < 0 ? -nsp : nsp);
                    if (nsp <= 0) continue;                                     // not converged: Householder route (M is intact)
                    const View Rfv = mkview(ck1, plain(n2), Idx{n2, 1, (long long)n2 * r});
                    const View Lo = wide ? Lfv : tview(Rfv);       // p x r
                    const View Ro = wide ? Rfv : tview(Lfv);       // r x q
                    const double sq0 = sqrt(s0), aneg = S.scal[0];
                    for (int e = tid; e < r * q; e += TTN_WG) {
                        const int col = e % q, j = e / q;
                        const double sj = S.sigs[j], xv = S.ldsX[S.perm[j] * 128 + col];
                        const bool keep = (sj > 0.0) && (sj * sj > aneg);
                        Ro.p[ix(Ro.r, j) + ix(Ro.c, col)] = keep ? xv * (sq0 / sqrt(sj)) : 0.0;          // sqrt(s0 sigma_j) v_j
                        S.M2[(long long)j * q + col] = keep ? xv * (sq0 / (sj * sqrt(sj))) : 0.0;
                    }
                    __syncthreads();
                    wg_gemm(p, r, q, Mv, mkview(S.M2, plain(1), plain(q)), Lo, inv_s0, 0.0, lds);          // M v_j sqrt(s0) / (s0 sqrt(sigma_j))
                    if (P.sv_out && step < P.sv_steps) {
                        double* so = P.sv_out + ((long long)b * P.sv_steps + step) * P.pmax;
                        for (int i = tid; i < P.pmax; i += TTN_WG) so[i] = (i < p) ? S.sigs[i] * s0 : -1.0;
                    }
                    if (P.prof && tid == 0) P.prof[(long long)b * 16 + 15] += 1;
                    if (tid == 0) *io.rank_out = r;
                    __syncthreads();
                    done = true;
                    route = 1;
                    PROF_MARK(5)
                    continue;
                }
            }
            if (attempt == 2 && P.sv_out && step < P.sv_steps) {
                double* so = P.sv_out + ((long long)b * P.sv_steps + step) * P.pmax;
                for (int i = tid; i < P.pmax; i += TTN_WG) so[i] = (i < p) ? S.sigs[i] * s0 : -1.0;
            }
            PROF_MARK(4)
            // ---- outputs.  left factor (p x r): x_j * sqrt(s0)/sqrt(sig_j) ; right factor (r x q):
            //      (x_j^T M / s0) * sqrt(s0) / (sig_j*sqrt(sig_j)).  Columns the Jacobi left alone as numerically
            //      zero (norm^2 <= aneg) are not singular vectors relative to their own size: written as exact zeros.
            const View Rfv = mkview(ck1, plain(n2), Idx{n2, 1, (long long)n2 * r});                 // (r x mc)
            const View Lo = wide ? Lfv : tview(Rfv);       // p x r
            const View Ro = wide ? Rfv : tview(Lfv);       // r x q
            const double sq0 = sqrt(s0);
            const double aneg = S.scal[0];
            if (p <= 128 && r <= 128) {
                wg_scale_image(X, ldx, (x_in_lds || attempt == 1) ? 1 : 0, S.sigs, S.perm, Lo, S.Us, p, r, s0, aneg, (SWAP == 0) ? 0 : (wide ? 1 : 2), S.Ts);
            } else {
            for (int e = tid; e < p * r; e += TTN_WG) {
                const int row = e % p, j = e / p;
                const double sj = S.sigs[j];
                const double xv = X[(long long)S.perm[j] * ldx + row];
                const bool keep = (sj > 0.0) && (sj * sj > aneg);
                // x_j = sigma_j * (singular vector of the short side), in units of s0.  Compress: sqrt(S) on both sides.
                // Swap: the LEFT core gets U, the right one S*Vt — the short side is the left one iff the matrix is wide.
                double lf, us;
                if (SWAP == 0) { lf = keep ? xv * (sq0 / sqrt(sj)) : 0.0; us = keep ? xv * (sq0 / (sj * sqrt(sj))) : 0.0; }
                else if (wide) { lf = keep ? xv / sj : 0.0; us = keep ? xv * (s0 / sj) : 0.0; }
                else { lf = keep ? xv * s0 : 0.0; us = keep ? xv / (sj * sj) : 0.0; }
                Lo.p[ix(Lo.r, row) + ix(Lo.c, j)] = lf;
                S.Us[(long long)j * p + row] = us;            // Us^T stored: (r x p) row-major
            }
            }
            __syncthreads();
            FINE_MARK(22)
            wg_gemm_ra(r, q, p, mkview(S.Us, plain(p), plain(1)), Mv, Ro, inv_s0, lds);
            FINE_MARK(23)
            if (attempt == 1) {
                wg_syrk(r, q, Ro, mkview(S.T2, plain(1), plain(128)), 1.0, lds);
                FINE_MARK(24)
                const double e2 = (r <= 128) ? wg_check_diag_tab(S.sigs, S.T2, 128, r, s0, S.Ts, S.red) : unif64(wg_check_diag(S, S.T2, 128, r, s0));
                FINE_MARK(25)
                if (!(e2 <= (cholqr ? CHOLQR_CHECK_TOL : FAST_CHECK_TOL))) continue;                     // redo with Householder (M is intact)
                if (P.sv_out && step < P.sv_steps) {
                    double* so = P.sv_out + ((long long)b * P.sv_steps + step) * P.pmax;
                    for (int i = tid; i < P.pmax; i += TTN_WG) so[i] = (i < p) ? S.sigs[i] * s0 : -1.0;
                }
            }
            if (tid == 0) *io.rank_out = r;
            __syncthreads();
            done = true;
            route = attempt;
            PROF_MARK(5)
        }
    }
    if (tid == 0) {
        P.sweep_stats[b] += nsw_total;
        if (P.prof && step < 120)
            P.prof[(long long)P.tt.batch * 16 + (long long)b * 120 + step] = ((long long)route << 48) | ((long long)p << 32) | (long long)nsw_total;
    }
    __syncthreads();
#undef PROF_MARK
#undef FINE_MARK
}

// tt_compress! form of the step: cores k, k+1 of train b of P.tt
__device__ __forceinline__ void wg_bond_step(const CompressArgs& P, int b, int k, int step, double* lds, bool virt = false /* core k+1 is A_{k+1} x_{k+1}, not yet written */) {
    const TTDev& T = P.tt;
    long long* rks = T.rks + (long long)b * (T.d + 1);
    // the ranks are read from memory this kernel also writes, so the compiler treats them (and every view, size and pointer
    // derived from them) as per-lane values: readfirstlane pins them to SGPRs — the bond step is full of calls, and every
    // VGPR that is live across a call costs a slot of the stack frame (HBM traffic, profiles/README.md "traffic by step")
    BondIO io;
    io.n1 = uni32(T.dims[k]); io.n2 = uni32(T.dims[k + 1]);
    io.Dl = uni32((int)rks[k]); io.rm = uni32((int)rks[k + 1]); io.Dr = uni32((int)rks[k + 2]);
    io.ck = T.data + (long long)b * T.stride + T.off[k];
    io.ck1 = T.data + (long long)b * T.stride + T.off[k + 1];
    io.rank_out = rks + k + 1;
    // tiny steps (short side <= 8): everything in LDS, no GEMM / LQ / image machinery.  Dispatched HERE, not inside the force-inlined
    // step: one more call site with live views in that function cost it 280 spilled VGPRs (-7 % on the whole benchmark, measured).
    {
        const int mr = io.n1 * io.Dl, mc = io.n2 * io.Dr;
        const int p = mr <= mc ? mr : mc, q = mr <= mc ? mc : mr;
        if (P.fast && !(P.fast & 1024) && p >= 2 && p <= SMALL_STEP_PMAX && q <= SMALL_STEP_QMAX && P.pmax >= p) {
            if (virt) wg_materialize_core(P, b, k + 1);
            SmallStepArgs Q;
            Q.jneg_mult = P.jneg_mult; Q.jtol_mult = P.jtol_mult; Q.truncerr = P.truncerr; Q.max_bond = P.max_bond; Q.rank_rule = P.rank_rule; Q.pmax = P.pmax;
            Q.sv_row = (P.sv_out && step < P.sv_steps) ? P.sv_out + ((long long)b * P.sv_steps + step) * P.pmax : nullptr;
            Q.status_b = P.status + b;
            // the step's LDS / scratch map (wg_bond_step_io): reduction scratch, scalars, flags; singular values and their order
            Q.red = lds + GEMM_LDS_TOTAL;
            Q.scal = Q.red + 32 + 2 * QR_NB * QR_NB + QR_NB;
            Q.iflag = reinterpret_cast<int*>(Q.scal + 8);
            double* scr = P.scratch + (long long)blockIdx.x * P.scratch_stride;
            const long long pq = (long long)P.pmax * P.qmax;
            double* sig = scr + 2 * pq + (long long)QR_NB * P.qmax + (long long)P.pmax * QR_NB + 2LL * P.pmax * P.pmax;
            Q.sigs = sig + P.pmax;
            Q.perm = reinterpret_cast<int*>(Q.sigs + P.pmax);
            const long long ts0 = P.prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
            const int nsw = wg_bond_small(Q, io.ck, io.ck1, io.n1, io.n2, io.Dl, io.rm, io.Dr, io.rank_out, lds);
            if (threadIdx.x == 0) {
                P.sweep_stats[b] += nsw;
                if (P.prof) {
                    if (P.prof_step < 0 || P.prof_step == step) P.prof[(long long)b * 16 + 3] += (long long)__builtin_amdgcn_s_memtime() - ts0;
                    if (step < 120) P.prof[(long long)P.tt.batch * 16 + (long long)b * 120 + step] = (2LL << 48) | ((long long)p << 32) | (long long)nsw;
                }
            }
            __syncthreads();
            return;
        }
    }
    wg_bond_step_io<0>(P, b, io, k, step, lds, virt);
}

__global__ void TTN_KERNEL_BOUNDS k_compress(CompressArgs P) {
    TTN_SETPRIO_BASE();
    extern __shared__ double lds[];
    const int d = P.tt.d;
    // ONE call site of the (force-inlined) bond step: as an out-of-line function it received its arguments in VGPRs, so
    // every size, view and pointer derived from them was a per-lane value and the step carried a 512-byte stack frame per
    // lane across its ~25 calls.  Inlined here everything uniform sits in SGPRs.
    const int per_sweep = 2 * (d - 1);
    const int nsteps = (P.k_single > 0) ? 1 : (P.k_single < 0) ? ((P.k_first <= P.k_last ? P.k_last - P.k_first : P.k_first - P.k_last) + 1)
                                                               : P.sweeps * per_sweep;
    int b = blockIdx.x;
    while (b < P.tt.batch) {
        if (threadIdx.x == 0) P.sweep_stats[b] = 0;        // (P.status is sticky: only failures are stored, the host clears on read)
        __syncthreads();
        // fused apply: the left core of the first bond is written out (unless it was imported), every core to its right stays
        // virtual until the front of the first L->R pass reaches it
        if (P.fused && !(P.k_single < 0 && P.fused_first_real)) wg_materialize_core(P, b, (P.k_single < 0) ? P.k_first : 0);
        for (int step = 0; step < nsteps; ++step) {
            int k;
            bool virt = false;
            if (P.k_single > 0) k = P.k_single - 1;                                  // _tt_bond_truncate!
            else if (P.k_single < 0) { k = (P.k_first <= P.k_last) ? P.k_first + step : P.k_first - step; virt = P.fused != 0; }   // ttn_sweep / ttn_apply_sweep
            else {                                                                   // tt_compress!: L->R then R->L per sweep
                const int i = step % per_sweep;
                k = (i < d - 1) ? i : per_sweep - 1 - i;
                virt = P.fused && step < d - 1;                                      // first L->R sweep of the fused op
            }
            const long long ts_ = P.prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
            wg_bond_step(P, b, k, step, lds, virt);
            if (P.prof && step < 120 && threadIdx.x == 0)            // bits 12..31 of the step word: kilo-cycles of the step
                P.prof[(long long)P.tt.batch * 16 + (long long)b * 120 + step] |= ((((long long)__builtin_amdgcn_s_memtime() - ts_) >> 10) & 0xFFFFF) << 12;
        }
        if (!P.next_train) break;
        // next train of this slot: one atomic per train, broadcast through LDS (the bond step ends with a barrier, so nobody still
        // reads the words behind the image)
        int* nb = reinterpret_cast<int*>(lds + GEMM_LDS_TOTAL);
        if (threadIdx.x == 0) *nb = (int)gridDim.x + atomicAdd(P.next_train, 1);
        __syncthreads();
        b = uni32(*nb);
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------
// Site-swap chains: hadamard_ttm (src/tt_operations.jl:398-422) and QTT reorder (src/qtt_tools.jl:733-775).
// Both are sequences of two-site swap SVDs (wg_bond_step_io<1>) on a chain of cores with one physical dimension n; the TTM
// Hadamard product additionally contracts neighbours site-wise (_ttm_contract!, :384-396), which shortens the chain.  One
// workgroup owns one train for the whole program; the host turns the reference's loops into a list of (type, slotA, slotB)
// ops on fixed storage slots, so a deleted core is simply a slot no later op names.
// -------------------------------------------------------------------------------------------------
struct ChainArgs {
    CompressArgs C;              // tolerance / rank rule / scratch / status (C.tt = the handle for kind 2)
    int kind;                    // 1: hadamard_ttm (slots in `arena`, x and y copied in, result copied to z); 2: reorder in place
    int n, nslots, nops, d;
    const int* ops;              // device [nops][3]: type (0 swap, 1 contract), slotA, slotB
    const int* final_slots;      // device [d] (kind 1): slots of the final chain, left to right
    double* arena;               // kind 1: per train nslots * slot_doubles
    long long arena_stride, slot_doubles;
    int cap;                     // kind 1: rank capacity of every slot
    long long* srk;              // per train [2*nslots + 2]: left ranks, right ranks, new-rank word
    long long srk_stride;
    TTDev x, y, z;               // kind 1
};

__global__ void TTN_KERNEL_BOUNDS k_swap_chain(ChainArgs Q) {
    extern __shared__ double lds[];
    const CompressArgs& P = Q.C;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = Q.n, ns = Q.nslots;
    long long* srl = Q.srk + (long long)b * Q.srk_stride;
    long long* srr = srl + ns;
    long long* rword = srr + ns;
    if (tid == 0) P.sweep_stats[b] = 0;
    __syncthreads();
    // ---- set-up ----
    if (Q.kind == 1) {
        const int d = Q.d;
        const long long* xr = Q.x.rks + (long long)b * (d + 1);
        const long long* yr = Q.y.rks + (long long)b * (d + 1);
        for (int k = 0; k < d; ++k) {
            const int rl = uni32((int)xr[k]), rr = uni32((int)xr[k + 1]);
            const double* src = Q.x.data + (long long)b * Q.x.stride + Q.x.off[k];
            double* dst = Q.arena + (long long)b * Q.arena_stride + (long long)k * Q.slot_doubles;
            for (int e = tid; e < n * rl * rr; e += TTN_WG) dst[e] = src[e];
            if (tid == 0) { srl[k] = rl; srr[k] = rr; }
        }
        for (int k = 0; k < d; ++k) {                          // slot d + k <- permutedims(y[d-1-k], (1,3,2))   (tt_operations.jl:411)
            const int ky = d - 1 - k;
            const int rl = uni32((int)yr[ky]), rr = uni32((int)yr[ky + 1]);
            const double* src = Q.y.data + (long long)b * Q.y.stride + Q.y.off[ky];
            double* dst = Q.arena + (long long)b * Q.arena_stride + (long long)(d + k) * Q.slot_doubles;
            for (int e = tid; e < n * rl * rr; e += TTN_WG) {
                const int s_ = e % n, t = e / n, a = t % rl, c = t / rl;
                dst[s_ + n * (c + (long long)rr * a)] = src[e];
            }
            if (tid == 0) { srl[d + k] = rr; srr[d + k] = rl; }
        }
    } else {
        const long long* rks = P.tt.rks + (long long)b * (P.tt.d + 1);
        for (int k = tid; k < ns; k += TTN_WG) { srl[k] = rks[k]; srr[k] = rks[k + 1]; }
    }
    __syncthreads();
    // ---- the program ----
    bool alive = true;
    for (int o = 0; o < Q.nops; ++o) {
        if (!alive) break;
        const int type = Q.ops[3 * o], sA = Q.ops[3 * o + 1], sB = Q.ops[3 * o + 2];
        double* cA = (Q.kind == 1) ? Q.arena + (long long)b * Q.arena_stride + (long long)sA * Q.slot_doubles
                                   : P.tt.data + (long long)b * P.tt.stride + P.tt.off[sA];
        double* cB = (Q.kind == 1) ? Q.arena + (long long)b * Q.arena_stride + (long long)sB * Q.slot_doubles
                                   : P.tt.data + (long long)b * P.tt.stride + P.tt.off[sB];
        const int rL = uni32((int)srl[sA]), rM = uni32((int)srr[sA]), rR = uni32((int)srr[sB]);
        if (type == 0) {
            BondIO io;
            io.ck = cA; io.ck1 = cB;
            io.n1 = n; io.n2 = n; io.Dl = rL; io.rm = rM; io.Dr = rR;
            io.rank_out = rword;
            io.cap = (Q.kind == 1) ? Q.cap : uni32((int)P.tt.cap[sB]);
            wg_bond_step_io<1>(P, b, io, sA, o, lds, false);
            const int r = uni32((int)*rword);
            if (r < 0) alive = false;
            else if (tid == 0) { srr[sA] = r; srl[sB] = r; }
            __syncthreads();
        } else {
            // Pi[s] = A[s] * B[s]   (rL x rM)(rM x rR) for every physical index s; the product replaces A's slot
            double* tmp = P.scratch + (long long)b * P.scratch_stride;
            for (int s_ = 0; s_ < n; ++s_) {
                const View Av = mkview(cA + s_, plain(n), plain((long long)n * rL));
                const View Bv = mkview(cB + s_, plain(n), plain((long long)n * rM));
                const View Cv = mkview(tmp + s_, plain(n), plain((long long)n * rL));
                wg_gemm(rL, rR, rM, Av, Bv, Cv, 1.0, 0.0, lds);
            }
            __syncthreads();
            for (int e = tid; e < n * rL * rR; e += TTN_WG) cA[e] = tmp[e];
            if (tid == 0) srr[sA] = rR;
            __syncthreads();
        }
    }
    // ---- results ----
    if (Q.kind == 1) {
        const int d = Q.d;
        long long* zr = Q.z.rks + (long long)b * (d + 1);
        if (alive) {
            for (int k = 0; k < d; ++k) {
                const int sl = Q.final_slots[k];
                const int rl = uni32((int)srl[sl]), rr = uni32((int)srr[sl]);
                if (rl > Q.z.cap[k] || rr > Q.z.cap[k + 1]) { if (tid == 0) ttn_set_status(&P.status[b], 2); alive = false; break; }
                const double* src = Q.arena + (long long)b * Q.arena_stride + (long long)sl * Q.slot_doubles;
                double* dst = Q.z.data + (long long)b * Q.z.stride + Q.z.off[k];
                for (int e = tid; e < n * rl * rr; e += TTN_WG) dst[e] = src[e];
                if (tid == 0) { zr[k] = rl; zr[k + 1] = rr; }
            }
        }
    } else if (alive) {
        long long* rks = P.tt.rks + (long long)b * (P.tt.d + 1);
        for (int k = tid; k < ns; k += TTN_WG) rks[k + 1] = srr[k];
    }
}

// -------------------------------------------------------------------------------------------------
// dot: transfer-matrix recurrence (src/tt_operations.jl:239-250), one workgroup per train pair.
//   T[al, (z,b)] = sum_be M[al,be] * B[z,be,b]        (ra x rb) * (rb x n*rb')
//   M'[a, b]     = sum_{(z,al)} A[z,al,a] * T[(z,al), b]
// -------------------------------------------------------------------------------------------------
struct DotArgs {
    TTDev a, b;
    double* scratch;            // per train: 2 * ramax*rbmax + nmax*ramax*rbmax
    long long scratch_stride;
    int ramax, rbmax, nmax;
    double* out;                // [batch] device
    long long* prof;            // TTN_PROF=1: s_memtime stamp after every site of train b at prof[16 * batch + 120 * b + k] (ttn_prof_steps)
};

// (the kernel: ttn_dot_kernels.h)

// -------------------------------------------------------------------------------------------------
// kernel unit-test hook for wg_gemm (tests/test_gpu_kernels.py): one workgroup, plain row-major operands
// (optionally viewed transposed so both LDS staging maps are exercised)
// -------------------------------------------------------------------------------------------------
__global__ void TTN_KERNEL_BOUNDS k_selftest_gemm(int m, int n, int k, double* A, double* B, double* C, double alpha,
                                                         double beta, int ta, int tb) {
    extern __shared__ double lds[];
    const View Av = (ta & 1) ? mkview(A, plain(1), plain(m)) : mkview(A, plain(k), plain(1));     // ta & 1: A stored k x m
    const View Bv = tb ? mkview(B, plain(1), plain(k)) : mkview(B, plain(n), plain(1));     // tb: B stored n x k
    if (ta & 2) wg_syrk(m, k, Av, mkview(C, plain(n), plain(1)), alpha, lds);               // ta & 2: C = alpha A A^T (n == m; B, beta unused)
    else if (ta & 4) wg_gemm_ra(m, n, k, Av, Bv, mkview(C, plain(n), plain(1)), alpha, lds);   // ta & 4: the register-A form (beta unused)
    else wg_gemm(m, n, k, Av, Bv, mkview(C, plain(n), plain(1)), alpha, beta, lds);
}

__global__ void TTN_KERNEL_BOUNDS k_bench_gemm(int m, int n, int k, double* A, double* B, double* C, int ta, int tb, int reps,
                                                      long long* cycles) {
    extern __shared__ double lds[];
    // every workgroup of the grid works on its own copy of the operands (the buffers hold gridDim.x of them back to back): a grid
    // of 2 x #CUs measures the GEMM with a second workgroup resident on the CU
    A += (long long)blockIdx.x * m * k; B += (long long)blockIdx.x * k * n; C += (long long)blockIdx.x * m * n;
    const View Av = (ta & 1) ? mkview(A, plain(1), plain(m)) : mkview(A, plain(k), plain(1));
    const View Bv = tb ? mkview(B, plain(1), plain(k)) : mkview(B, plain(n), plain(1));
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (ta & 2) wg_syrk(m, k, Av, mkview(C, plain(n), plain(1)), 1.0, lds);            // the Gram-product routine (n == m)
        else if (ta & 4) wg_gemm_ra(m, n, k, Av, Bv, mkview(C, plain(n), plain(1)), 1.0, lds);
        else wg_gemm(m, n, k, Av, Bv, mkview(C, plain(n), plain(1)), 1.0, 0.0, lds);
    }
    __syncthreads();
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// LDS building-block micro-benchmark (ttn_bench_lds): G = I*n + smooth symmetric perturbation, then Cholesky or Jacobi.
__global__ void TTN_KERNEL_BOUNDS k_bench_lds(int what, int n, int reps, long long* out) {
    extern __shared__ double lds[];
    double* red = lds + GEMM_LDS_TOTAL;
    double* scal = red + 32;
    int* iflag = (int*)(scal + 8);
    double* nrm2 = scal + 16;
    int sw = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        for (int e = threadIdx.x; e < 128 * 128; e += TTN_WG) {
            const int c = e >> 7, i = e & 127;
            lds[e] = (i < n && c < n) ? ((i == c) ? (double)n : 1.0 / (1.0 + (i > c ? i - c : c - i))) : 0.0;
        }
        __syncthreads();
        if (what == 1) wg_chol_lds128(n, lds, red, iflag, scal + 1);
        if (what == 2) sw = wg_jacobi_lds128(n, n, lds, nrm2, iflag, red, 1.0, 1.0, scal);
        __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = sw; }
}
