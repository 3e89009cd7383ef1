// ttn_ortho512.h — the right-to-left LQ sweep of orthogonalize (src/tt_tools.jl:528-536) over the tall QTT cores of rank <= 64, as
// 512-thread workgroups that live TWO TO A CU.
//
// The fused Cholesky-QR step of ttn_ortho_fused.h is ten short dependent phases; a third of it is the 64-pivot chain of the Cholesky
// factorisation, which runs on ONE wave while the other fifteen wait, and its LDS images (148 KB) leave no room for a second train on
// the CU.  Here the step needs 67 KB and eight waves, so a second workgroup's matrix products run under the first one's pivot chain:
//   * W never goes to LDS: wave w owns rows [16 w, 16 w + 16) of W (128 x 64, row = s * 64 + be) in its MFMA accumulators (four tiles).
//     The accumulator layout D[row = (lane >> 4) + 4 reg][col = lane & 15] is, register by register, BOTH operands of a product that
//     contracts over the tile's rows, so the wave forms its share of G = W^T W (and later of Q^T Q) from its own registers and adds
//     it into the LDS image with ds_add_f64;
//   * Q = W L^-T contracts over W's COLUMNS: the A fragments are the accumulator tiles transposed inside the wave by ds_bpermute
//     (the LDS crossbar, no LDS memory: 8 permutes per k-step);
//   * two 64 x 64 images with an XOR swizzle (both fragment patterns — 4 rows x 16 and 16 rows x 4 — free of bank conflicts) PING-PONG:
//     a step reads FL from one, accumulates G and factors it in place in the other, writes L^-1 over the dead FL, and its L IS the
//     next step's FL where it lies — no image is loaded or stored between sites (R still goes to global memory for the general route,
//     which takes over wherever this kernel stops: a site outside the class, or a step whose measured orthogonality is above the bar).
// The 1024-thread kernel runs the left sweep and the right sweep's leading ill-conditioned square sites first (mode 1), this kernel
// the tall sites, the 1024-thread kernel the rest and the centre core (mode 3); per-train state in OrthoArgs::state.
#pragma once
#include "ttn_ortho_kernels.h"

#define O5_WG 512
#define O5_POLISH_MAX 1.0e-5             // measured max |Q^T Q - I| up to which first-order second passes are taken (cond(W) <~ 2e5; each pass
                                         // squares the defect, the factor is then recomputed as the projection of W on the repaired Q)
#define O5_POLISH_TRUST 1.0e-9           // ... below which one pass is trusted without measuring again (it leaves O(64 E^2) <= 1e-16)
#define O5_IMG(k, i) ((k) * 64 + ((i) ^ ((((k) & 1) << 4) | ((((k) >> 1) & 3) << 2))))      // element (row i, column k) of a 64 x 64 image
#define O5_BUF 4096
#define O5_T16 (2 * O5_BUF)                 // inverse of the current diagonal block, [row * 17 + col]
#define O5_MISC (O5_T16 + 16 * 17 + 16)     // [0] flag (int), [1] dev bits, [2] dmax
#define O5_TAB (O5_MISC + 16)               // ints: per site 4 (rl, n, offX, offY) for sites 0..d, then yr is read from global
#define O5_LDS_BYTES(d) (sizeof(double) * O5_TAB + sizeof(int) * 4 * ((d) + 2))

__device__ __forceinline__ void o5_barrier() { __syncthreads(); }

// diagonal block jb: factor and invert in registers (as of_diag_block, on the swizzled image; Linv -> T16 and the Linv image)
__device__ __forceinline__ bool o5_diag_block(int jb, lds_f64* G, lds_f64* T16, lds_f64* Li, double dmin) {
    int lane = threadIdx.x & 63;
    // (opaque to the compiler: it hoisted the sixteen identity columns (li == jj ? 1 : 0) and the LDS addresses of this block out of the
    //  caller's loop over the diagonal blocks as loop invariants, spilled them, and reloaded one from scratch memory in every step — a
    //  memory latency in each link of the block's dependent chain: 9.1 k clk per block, measured)
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    double e[16], t[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) e[c] = G[O5_IMG(16 * jb + c, 16 * jb + li)];
    double dinv = 1.0;
    bool ok = true;
    // Factor and inverse in ONE loop.  Step jj of the factorisation finishes column jj of L; with it the elimination step jj of
    // U'^-1 (L = U' D, U'[i][k] = L[i][k] / L[k][k]) can run at once — its row jj is final, its multipliers are L[li][jj] rs_jj — so the
    // 15 - jj updates of the factor and the jj + 1 updates of the inverse are sixteen INDEPENDENT DPP FMAs per step (the two loops one
    // after the other were two dependent chains: 9.1 k clk per block, measured).  X = D^-1 U'^-1: one scaling at the end.
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const double d = of_readlane(e[jj], jj);
        ok = ok && (d > dmin);
        const double rs = fast_rsqrt2(d);
        const double l = e[jj] * rs;
        // (column jj of L is final: to the image now — with the columns of the inverse that are still the identity not yet in registers,
        //  17 doubles are live per step instead of 32; with 32 the compiler spilled, and the reload in every step was the block's time)
        if (lk == 0) G[O5_IMG(16 * jb + jj, 16 * jb + li)] = (jj <= li) ? l : 0.0;
        t[jj] = (li == jj) ? 1.0 : 0.0;
        dinv = (li == jj) ? rs : dinv;
        const double nl = -l;
        const double s_ = (li > jj) ? nl * rs : 0.0;                       // -L[li][jj] / L[jj][jj]
        OF_DPP_FENCE();
        DPP_FACTOR_GROUP(jj, e, l, nl);          // (one asm statement per group: csrc/ttn_ortho_dpp_gen.h)
        DPP_INVERSE_GROUP(jj, t, s_);
        OF_DPP_FENCE();
    }
    if (!ok) return false;
#pragma unroll
    for (int c = 0; c < 16; ++c) t[c] *= dinv;
    if (lk == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {                                                  // (L went to the image column by column, its strict upper triangle zeroed)
            T16[li * 17 + c] = (c <= li) ? t[c] : 0.0;
            Li[O5_IMG(16 * jb + c, 16 * jb + li)] = (c <= li) ? t[c] : 0.0;             // L^-1[row][col] as element (row, col)
        }
    }
    return true;
}

// the wave's share of (tiles)^T (tiles) added into the image: T[4] = this wave's 16 rows of a 128 x 64 matrix, nat column tiles
template <bool FULL>
__device__ __forceinline__ void o5_gram_add(const mfma_acc_t (&T)[4], int nat, lds_f64* G) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb <= ta; ++tb) {
            if (!FULL && ta >= nat) continue;
            mfma_acc_t g = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) g = __builtin_amdgcn_mfma_f64_16x16x4f64(T[ta][r], T[tb][r], g, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)         // element (row = 16 ta + lk + 4 reg, column = 16 tb + li)
                __hip_atomic_fetch_add(G + O5_IMG(16 * tb + li, 16 * ta + lk + 4 * reg), g[reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
}

// the wave's share of M = Qa^T Qb (64 x 64, all sixteen tiles) added into the image as element (row aq of Qa's columns, column aw of
// Qb's): both operands are the accumulator tiles themselves (a tile is a valid MFMA operand for a product contracting over its rows)
template <bool FULL>
__device__ __forceinline__ void o5_cross_add(const mfma_acc_t (&Qa)[4], const mfma_acc_t (&Qb)[4], int nat, lds_f64* M) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            if (!FULL && (ta >= nat || tb >= nat)) continue;
            mfma_acc_t g = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) g = __builtin_amdgcn_mfma_f64_16x16x4f64(Qa[ta][r], Qb[tb][r], g, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)         // g[reg] = M[row = 16 ta + lk + 4 reg][column = 16 tb + li]
                __hip_atomic_fetch_add(M + O5_IMG(16 * tb + li, 16 * ta + lk + 4 * reg), g[reg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
}

// out[c] = sum_{ta <= c} src[ta] X[c-block][ta-block]^T: the product of this wave's 16 rows (four accumulator tiles) with the transpose of
// a LOWER triangular 64 x 64 matrix X held in an image as element (row al', column al).  The A fragments are the source tiles transposed
// inside the wave by ds_bpermute: target lane (li = row, lk = k) of k-step u wants T[row][4 u + lk], which lives in lane
// ((row & 3) << 4) | (4 u + lk), register row >> 2.
template <bool FULL>
__device__ __forceinline__ void o5_apply(const mfma_acc_t (&src_)[4], mfma_acc_t (&out)[4], const lds_f64* Xi, int nat, bool active) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
    if (!active) return;
    const int rsel = li >> 2;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
        if (!FULL && ta >= nat) continue;
        double af[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int src = (((li & 3) << 4) | (4 * u + lk)) << 2;
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                union { double dd; int ii[2]; } in_, out_;
                in_.dd = src_[ta][r];
                out_.ii[0] = __builtin_amdgcn_ds_bpermute(src, in_.ii[0]);
                out_.ii[1] = __builtin_amdgcn_ds_bpermute(src, in_.ii[1]);
                v[r] = out_.dd;
            }
            af[u] = rsel == 0 ? v[0] : (rsel == 1 ? v[1] : (rsel == 2 ? v[2] : v[3]));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < ta || (!FULL && c >= nat)) continue;                    // X is lower triangular: column block c takes al-blocks ta <= c
#pragma unroll
            for (int u = 0; u < 4; ++u)
                out[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], Xi[O5_IMG(16 * ta + 4 * u + lk, 16 * c + li)], out[c], 0, 0, 0);
        }
    }
}

// the A fragments of an accumulator tile: af[u] (lane (li, lk)) = T[row = li][col = 4 u + lk] (see o5_apply)
__device__ __forceinline__ void o5_tile_frags(const mfma_acc_t& t_, double (&af)[4]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, rsel = li >> 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int src = (((li & 3) << 4) | (4 * u + lk)) << 2;
        double v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            union { double dd; int ii[2]; } in_, out_;
            in_.dd = t_[r];
            out_.ii[0] = __builtin_amdgcn_ds_bpermute(src, in_.ii[0]);
            out_.ii[1] = __builtin_amdgcn_ds_bpermute(src, in_.ii[1]);
            v[r] = out_.dd;
        }
        af[u] = rsel == 0 ? v[0] : (rsel == 1 ? v[1] : (rsel == 2 ? v[2] : v[3]));
    }
}

// Q = W L^-T by BLOCKED SUBSTITUTION: out[c] = (W_c - sum_{a < c} out[a] L[c][a]^T) X_cc^T with the inverses X_cc of the 16 x 16 diagonal
// blocks only.  The first version multiplied W by the explicit inverse of the whole L, assembled block-wise from the inverted
// diagonal blocks: that inverse satisfies X L = I only to eps cond(L)^2 (the error of a diagonal block's inverse is multiplied by the
// off-diagonal sums), so Q L^T = W held only to the level of the measured orthogonality defect — harmless below the acceptance bar,
// but a step repaired by second passes (defect up to 1e-5) kept that error in the TENSOR (1.4e-10 found on trains with cond 1e3
// sites).  Substitution uses the computed columns of Q themselves: Q L^T = W to eps cond(L_cc).  Same MFMA count (ten tile products
// per wave), four dependent stages, and the off-diagonal blocks of the inverse are not needed at all.
template <bool FULL>
__device__ __forceinline__ void o5_trsm_apply(const mfma_acc_t (&w_)[4], mfma_acc_t (&out)[4], const lds_f64* Lg, const lds_f64* Xd, int nat, bool active) {
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    double afo[3][4];                                                   // the A fragments of the finished tiles out[0 .. 2]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        out[c] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
        if (!active || (!FULL && c >= nat)) continue;
        mfma_acc_t acc = w_[c];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a >= c) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u)                                 // acc[rho][al'] -= out_a[rho][al] L[al'][al], al' in block c, al in block a
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-afo[a][u], Lg[O5_IMG(16 * a + 4 * u + lk, 16 * c + li)], acc, 0, 0, 0);
        }
        double af[4];
        o5_tile_frags(acc, af);
        mfma_acc_t o = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u)                                     // out_c[rho][al'] = sum_al acc[rho][al] X_cc[al'][al]
            o = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], Xd[O5_IMG(16 * c + 4 * u + lk, 16 * c + li)], o, 0, 0, 0);
        out[c] = o;
        if (c < 3) o5_tile_frags(o, afo[c]);
    }
}

// One site.  Returns false when the step is refused (bad pivot, or measured orthogonality above the bar): nothing in global memory
// has been written then.  cur / first: the image ping-pong (see the kernel).  FULL: rl = rr = ynext = 64.
// (out of line on purpose: inlined twice into the kernel's site loop the two instantiations cost 356 spilled VGPRs)
template <bool FULL>
__device__ __noinline__ int o5_step(double* lds, const double* Xj, double* Yj, const double* Rprev, double* Rn, int rl, int rr, int ynext,
                                    int cur, int first, long long* yr_j, long long* stamps) {
    lds = unip(lds); Xj = unip(Xj); Yj = unip(Yj); Rprev = unip(Rprev); Rn = unip(Rn); yr_j = unip(yr_j); stamps = unip(stamps);
    // TTN_PROF: clocks accumulated per phase (slots: 0 images / 1 carry + Gram / 2 dmax + padding / 3 Cholesky / 4 inverse / 5 apply /
    // 6 measured check / 7 second passes / 8 stores)
    long long t_prev = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define O5S(i) if (stamps) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) stamps[i] += now_ - t_prev; t_prev = now_; }
    rl = uni32(rl); rr = uni32(rr); ynext = uni32(ynext); cur = uni32(cur); first = uni32(first);
    lds_f64* buf = (lds_f64*)lds;
    lds_f64* T16 = (lds_f64*)lds + O5_T16;
    lds_f64* misc = (lds_f64*)lds + O5_MISC;
    lds_i32* flag = (lds_i32*)misc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int s_w = wave >> 2, tr = wave & 3;                               // this wave's rows of W: s = s_w, be = 16 tr + .
    const int nbt = FULL ? 4 : (ynext + 15) >> 4, nat = FULL ? 4 : (rl + 15) >> 4;
    if (FULL) { rl = 64; rr = 64; ynext = 64; }
    if (first) {                                                        // FL from global memory (zero padded), the other image zero
        for (int e = tid; e < 2 * O5_BUF; e += O5_WG) buf[e] = 0.0;
        __syncthreads();
        for (int e = tid; e < rr * ynext; e += O5_WG) { const int ga = e / ynext, be = e - ga * ynext; buf[O5_IMG(be, ga)] = Rprev[e]; }   // FL[ga][be] = element (ga, be)
    }
    if (tid == 0) { flag[0] = 0; ((__attribute__((address_space(3))) unsigned long long*)misc)[1] = 0ull; }
    __syncthreads();
    O5S(0)
    lds_f64* FLb = buf + cur * O5_BUF;
    lds_f64* Gb = buf + (cur ^ 1) * O5_BUF;
    // ---- P1: W_w[be][al] = sum_ga FL[ga][be] X_j[s, al, ga] ----
    mfma_acc_t w_[4];
#pragma unroll
    for (int tc = 0; tc < 4; ++tc) w_[tc] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
    if (tr < nbt) {
        typedef const __attribute__((address_space(1))) double gd;
        gd* xg = (gd*)Xj;
        if (FULL) {
            // X_j[s, al = 16 tc + li, ga = 4 t + lk] at s + 2 (al + 64 ga): every address is base + immediate, the loop unrolls
            // completely and the compiler keeps a window of loads in flight ahead of the MFMAs
            gd* xb = xg + s_w + 2 * (li + 64 * lk);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double a = FLb[O5_IMG(16 * tr + li, 4 * t + lk)];
#pragma unroll
                for (int tc = 0; tc < 4; ++tc)
                    w_[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[2 * (16 * tc + 64 * 4 * t)], w_[tc], 0, 0, 0);
            }
        } else {
            const int nt = (rr + 3) >> 2;
            for (int t = 0; t < nt; ++t) {
                const int ga = 4 * t + lk;
                const double a = FLb[O5_IMG(16 * tr + li, ga)];
                double bv[4];
#pragma unroll
                for (int tc = 0; tc < 4; ++tc) {                         // four loads in flight, masked (no per-tile branch)
                    const int al = 16 * tc + li;
                    const bool okk = al < rl && ga < rr;
                    bv[tc] = xg[okk ? s_w + 2 * (al + rl * ga) : 0];
                    bv[tc] = okk ? bv[tc] : 0.0;
                }
#pragma unroll
                for (int tc = 0; tc < 4; ++tc) w_[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[tc], w_[tc], 0, 0, 0);
            }
        }
        // ---- P2: this wave's share of G = W^T W ----
        o5_gram_add<FULL>(w_, nat, Gb);
    }
    __syncthreads();
    O5S(1)
    // ---- P3: dmax, padding, the FL image becomes the (zeroed) L^-1 image, blocked Cholesky ----
    if (wave == 0) {
        double dm = 0.0;
        for (int i = lane; i < rl; i += 64) dm = fmax(dm, Gb[O5_IMG(i, i)]);
        dm = wave_max(dm);
        if (lane == 0) misc[2] = dm;
    }
    // (the FL image is not zeroed here: of the inverse only the diagonal 16 x 16 blocks exist since the substitution form of the
    //  apply, and the wave that inverts a block writes its whole tile, zeros above the diagonal included)
    __syncthreads();
    const double dmax = unif64(misc[2]);
    const double dmin = 64.0 * DBL_EPSILON * dmax;
    bool good = dmax > 0.0;
    if (good) {
        for (int i = rl + tid; i < 16 * nat; i += O5_WG) Gb[O5_IMG(i, i)] = dmax;
        __syncthreads();
        lds_f64* Li = FLb;
        O5S(2)
        for (int jb = 0; jb < nat && good; ++jb) {
            if (wave == 0) { if (!o5_diag_block(jb, Gb, T16, Li, dmin) && lane == 0) flag[0] = 1; }
            O5S(9)
            __syncthreads();
            O5S(10)
            if (uni32(flag[0])) { good = false; break; }
            if (wave >= 1 && jb + wave < nat) {                         // panel: L[ib][jb] = G[ib][jb] X_jj^T
                const int ib = jb + wave;
                mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = 4 * t + lk;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Gb[O5_IMG(16 * jb + kk, 16 * ib + li)], T16[li * 17 + kk], acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Gb[O5_IMG(16 * jb + li, 16 * ib + lk + 4 * reg)] = acc[reg];
            }
            // (the blocks above the diagonal were never added to — only lower tiles are — so L's upper triangle is zero: it is the next FL)
            O5S(11)
            __syncthreads();
            O5S(12)
            {
                const int nrem = nat - jb - 1, ntile = nrem * (nrem + 1) / 2;
                for (int tile = wave; tile < ntile; tile += 8) {
                    int r_ = 0, base = 0;
                    while (base + r_ + 1 <= tile) { base += r_ + 1; ++r_; }
                    const int ib = jb + 1 + r_, kb = jb + 1 + (tile - base);
                    mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int kk = 4 * t + lk;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Gb[O5_IMG(16 * jb + kk, 16 * ib + li)], Gb[O5_IMG(16 * jb + kk, 16 * kb + li)], acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Gb[O5_IMG(16 * kb + li, 16 * ib + lk + 4 * reg)] -= acc[reg];
                }
            }
            __syncthreads();
        }
        O5S(3)
    }
    if (!good) return 0;                                            // refused: Rprev in global memory is intact, the state stays at site j
    O5S(4)
    // ---- P5: Q = W L^-T by blocked substitution (o5_trsm_apply: L's blocks from its image, the diagonal blocks' inverses from the other) ----
    mfma_acc_t q_[4];
    o5_trsm_apply<FULL>(w_, q_, Gb, FLb, nat, tr < nbt);
    O5S(5)
    // ---- P6: the orthogonality of Q, measured: max |Q^T Q - I| over the lower triangle (the image that held L^-1 takes Q^T Q) ----
    auto measure = [&]() -> double {
        __syncthreads();                                                // every read of the image (L^-1 / X2) is done
        for (int e = tid; e < O5_BUF; e += O5_WG) FLb[e] = 0.0;
        if (tid == 0) ((__attribute__((address_space(3))) unsigned long long*)misc)[1] = 0ull;
        __syncthreads();
        if (tr < nbt) o5_gram_add<FULL>(q_, nat, FLb);
        __syncthreads();
        double dev = 0.0;
        for (int e = tid; e < 64 * 64; e += O5_WG) {
            const int i = e & 63, k = e >> 6;
            if (i < rl && k < rl && i >= k) dev = fmax(dev, fabs(FLb[O5_IMG(k, i)] - ((i == k) ? 1.0 : 0.0)));
        }
        dev = wave_max(dev);
        if (lane == 0) atomicMax((unsigned long long*)misc + 1, (unsigned long long)__double_as_longlong(dev));
        __syncthreads();
        return unif64(__longlong_as_double((long long)((__attribute__((address_space(3))) unsigned long long*)misc)[1]));
    };
    double devmax = measure();
    O5S(6)
    for (int pass = 0; !(devmax <= ORTHO_FUSED_ACCEPT); ++pass) {
        if (!(devmax <= O5_POLISH_MAX) || pass == 3) return 0;
        const bool recheck = devmax > O5_POLISH_TRUST;
        // ---- moderately conditioned site (the first tall site behind the square ramp sites: cond ~ 150, eps cond^2 ~ 4e-12): one
        //      first-order pass of Cholesky-QR2.  Q^T Q = I + E = L2 L2^T with L2 = I + T + O(E^2), T = strict_lower(E) + diag(E) / 2:
        //      Q2 = Q (I - T)^T is the second pass to O(E^2) — no second factorisation, one more triangular product.  The image holds
        //      C = Q^T Q (lower part): it becomes X2 = I - T.  (The factor L is recomputed from the final Q after the passes.) ----
        for (int e = tid; e < 64 * 64; e += O5_WG) {
            const int i = e & 63, k = e >> 6;
            const double v = FLb[O5_IMG(k, i)];
            FLb[O5_IMG(k, i)] = (i == k) ? ((i < rl) ? 1.5 - 0.5 * v : 1.0) : ((i > k) ? -v : 0.0);
        }
        __syncthreads();
        mfma_acc_t q2_[4];
        o5_apply<FULL>(q_, q2_, FLb, nat, tr < nbt);
        // The factor that goes with the repaired Q: L <- L M, M = Q^T Q2 (a cross Gram of the tiles before and after the pass), i.e. the
        // projection (Q2^T (Q L^T))^T of what the substitution solved — NOT L (I + T): that product carries the O(E^2) of a first-order
        // pass into Q L^T = W (1e-10 for a defect of 1e-5); the projection only what lies outside the span of Q2 (rounding level).
        __syncthreads();                                                // every read of X2 is done
        for (int e = tid; e < O5_BUF; e += O5_WG) FLb[e] = 0.0;
        __syncthreads();
        if (tr < nbt) o5_cross_add<FULL>(q_, q2_, nat, FLb);
#pragma unroll
        for (int c = 0; c < 4; ++c) q_[c] = q2_[c];
        __syncthreads();
        {
            // L M, lower tiles only (the product is lower triangular up to rounding): ten tiles over the eight waves, results in registers
            // until every wave has read L
            mfma_acc_t lm[2];
            int tix[2] = {wave, wave + 8};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                lm[q] = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
                int ib = 0, base = 0;
                while (base + ib + 1 <= tix[q] && ib < 3) { base += ib + 1; ++ib; }
                const int jb = tix[q] - base;
                if (tix[q] < 10 && (FULL || ib < nat)) {
                    for (int kb = 0; kb <= ib; ++kb)
#pragma unroll
                        for (int t = 0; t < 4; ++t)                     // (L M)[16 ib + row][16 jb + col] += L[.][16 kb + 4 t + lk] M[16 kb + 4 t + lk][.]
                            lm[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gb[O5_IMG(16 * kb + 4 * t + lk, 16 * ib + li)],
                                                                         FLb[O5_IMG(16 * jb + li, 16 * kb + 4 * t + lk)], lm[q], 0, 0, 0);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int ib = 0, base = 0;
                while (base + ib + 1 <= tix[q] && ib < 3) { base += ib + 1; ++ib; }
                const int jb = tix[q] - base;
                if (tix[q] < 10 && (FULL || ib < nat)) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {                 // accumulator element (row = lk + 4 reg, column = li)
                        const int i = 16 * ib + lk + 4 * reg, jc = 16 * jb + li;
                        Gb[O5_IMG(jc, i)] = (i >= jc) ? lm[q][reg] : 0.0;
                    }
                }
            }
            __syncthreads();
        }
        if (!recheck) break;                                            // E <= 1e-9: the pass leaves O(64 E^2) <= 1e-16, not measured again
        devmax = measure();
    }
    O5S(7)
    // ---- P7: Y_j, R = L^T to global memory, the check image zeroed for the next G ----
    if (tr < nbt) {
        typedef __attribute__((address_space(1))) double gwd;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!FULL && c >= nat) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int be = 16 * tr + lk + 4 * reg, alp = 16 * c + li;
                if (be < ynext && alp < rl) *((gwd*)Yj + (s_w + 2 * (alp + rl * be))) = q_[c][reg];
            }
        }
    }
    for (int e = tid; e < rl * rl; e += O5_WG) { const int c = e / rl, i = e - c * rl; Rn[e] = (i <= c) ? (double)Gb[O5_IMG(i, c)] : 0.0; }     // R[i][c] = L[c][i]
    for (int e = tid; e < O5_BUF; e += O5_WG) FLb[e] = 0.0;
    if (tid == 0) *yr_j = rl;
    __syncthreads();
    O5S(8)
    return 1;                                                               // (the caller flips the images: this step's L is the next step's FL)
}

__global__ void __launch_bounds__(O5_WG, 4) k_ortho512(OrthoArgs P) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int b = blockIdx.x;
    const TTDev& X = P.x; const TTDev& Y = P.y;
    const int d = X.d, ic = P.center;
    const long long* xr = X.rks + (long long)b * (d + 1);
    long long* yr = Y.rks + (long long)b * (d + 1);
    double* scr = P.scratch + (long long)b * P.scratch_stride;
    double* Rc = scr + 2LL * P.mmax * P.rmax + 2LL * P.rmax * P.rmax;                        // (the layout of k_orthogonalize)
    double* Rd = Rc + (long long)P.rmax * P.rmax;
    int* st = P.state + 4 * b;
    lds_f64* buf = (lds_f64*)lds;
    lds_f64* T16 = (lds_f64*)lds + O5_T16;
    lds_f64* misc = (lds_f64*)lds + O5_MISC;
    lds_i32* flag = (lds_i32*)misc;
    long long* stamps = P.prof ? P.prof + 136LL * gridDim.x + 64LL * b + 16 : nullptr;
    lds_i32* tab = (lds_i32*)((lds_f64*)lds + O5_TAB);
    for (int k = tid; k <= d; k += O5_WG) {
        tab[4 * k + 0] = (int)xr[k];
        tab[4 * k + 1] = k < d ? X.dims[k] : 0;
        tab[4 * k + 2] = k < d ? (int)X.off[k] : 0;
        tab[4 * k + 3] = k < d ? (int)Y.off[k] : 0;
    }
    __syncthreads();
    int j = uni32(st[0]), whichL = uni32(st[1]);
    int cur = 0;                                                            // buf[cur]: FL; buf[cur ^ 1]: all zero at the start of a step
    bool first = true;
    double* Xbase = X.data + (long long)b * X.stride;
    double* Ybase = Y.data + (long long)b * Y.stride;
    while (j > ic && !(P.no_cholqr & 2)) {                                  // (TTN_ORTHO_CHOLQR = 0 / 1: no Cholesky-QR steps — everything is handed to the general route)
        const int rl = uni32(tab[4 * j]), rr = uni32(tab[4 * j + 4]), n = uni32(tab[4 * j + 1]);
        const int ynext = uni32((int)yr[j + 1]);
        if (!ortho512_eligible(n, rl, rr, ynext)) break;
        const double* Xj = Xbase + uni32(tab[4 * j + 2]);
        double* Yj = Ybase + uni32(tab[4 * j + 3]);
        const double* Rprev = whichL ? Rd : Rc;
        double* Rn = whichL ? Rc : Rd;
        const bool full = rl == 64 && rr == 64 && ynext == 64;
        if (first) cur = 0;
        const int ok = full ? o5_step<true>(lds, Xj, Yj, Rprev, Rn, rl, rr, ynext, cur, first ? 1 : 0, yr + j, stamps)
                            : o5_step<false>(lds, Xj, Yj, Rprev, Rn, rl, rr, ynext, cur, first ? 1 : 0, yr + j, stamps);
        first = false;
        if (!ok) break;
        cur ^= 1;
        whichL ^= 1;
        --j;
    }
    __syncthreads();
    int done = 0;
    if (j == ic && !(P.no_cholqr & 2)) {
        // ---- the centre core Y_i[s] = FR X_i[s] FL (src/tt_tools.jl:537-541) here as well: a third launch of 1024-thread workgroups
        //      for two small products costs 4 ms of workgroup dispatch on a batch of 1024 (measured) ----
        const int n = uni32(tab[4 * ic + 1]), rl = uni32(tab[4 * ic]), rr = uni32(tab[4 * ic + 4]);
        const int yl = uni32((int)yr[ic]), yn = uni32((int)yr[ic + 1]);
        const double* Xi = Xbase + uni32(tab[4 * ic + 2]);
        double* Yi = Ybase + uni32(tab[4 * ic + 3]);
        double* Tm = scr;
        double* Rb0 = scr + 2LL * P.mmax * P.rmax;
        double* Rb1 = Rb0 + (long long)P.rmax * P.rmax;
        const int wl = uni32(st[2]);
        const double* FRg = (ic > 0) ? (wl ? Rb1 : Rb0) : nullptr;           // FR[al][g] = FRg[al + yl g]; the 1 x 1 identity when ic = 0
        const double* FLg = (ic < d - 1) ? (whichL ? Rd : Rc) : nullptr;     // FL[ga][be] = FLg[be + yn ga]
        const int mm = yl * n;
        for (int e = tid; e < mm * rr; e += O5_WG) {                          // T[(al + yl s), ga] = sum_g FR[al][g] X[s, g, ga]
            const int row = e % mm, ga = e / mm, al = row % yl, s = row / yl;
            double a = 0.0;
            if (FRg) { for (int g = 0; g < rl; ++g) a = fma(FRg[al + yl * g], Xi[s + n * (g + rl * ga)], a); }
            else a = Xi[s + n * (al + rl * ga)];
            Tm[e] = a;
        }
        __syncthreads();
        for (int e = tid; e < mm * yn; e += O5_WG) {                          // Y_i[s, al, be] = sum_ga T[(al, s), ga] FL[ga][be]
            const int row = e % mm, be = e / mm, al = row % yl, s = row / yl;
            double a = 0.0;
            if (FLg) { for (int ga = 0; ga < rr; ++ga) a = fma(Tm[row + mm * ga], FLg[be + yn * ga], a); }
            else a = Tm[row + mm * be];
            Yi[s + n * (al + yl * be)] = a;
        }
        done = 1;
    }
    if (tid == 0) {
        st[3] = done;                                                       // 1: this train is finished; 0: the mode-3 launch takes it from site st[0]
        st[0] = j; st[1] = whichL;
        if (!done) { int* list = P.state + 4 * gridDim.x; list[1 + atomicAdd(list, 1)] = b; }   // (behind the per-train state: count and list of the trains the mode-3 launch takes)
    }
}
