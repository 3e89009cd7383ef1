// ttn_ortho_fused.h — one LQ step of orthogonalize's right-to-left sweep (src/tt_tools.jl:528-536) for QTT cores (n = 2) of rank <= 64,
// entirely on one workgroup's registers and LDS: Cholesky-QR with a measured orthogonality check.
//
//   W[(be, s), al] = sum_ga FL[ga, be] X_j[s, al, ga]                 the carry product (FL: the L factor of the site to the right)
//   G = W^T W,  G = L L^T,  Q = W L^-T,  Y_j[s, al', be] = Q[(be, s), al'],  FL_new = L
//
// The interior sites of a rank-64 train are 128 x 64 matrices W with cond(W) ~ 10 (measured on the benchmark's random trains; the
// square rank-ramp sites at the right end reach 1e8 and stay on the Householder route): one Cholesky-QR pass leaves
// max |Q^T Q - I| ~ 3e-15, and that number is MEASURED here for every step (a second Gram product) — above ORTHO_FUSED_ACCEPT the step
// returns 0 and the caller redoes it on the general route (Cholesky-QR2 / Householder, ttn_ortho_kernels.h).
// Round 2 ran such a site as a general GEMM + in-LDS Householder LQ + explicit Q: 64 reflectors, each applied twice, a barrier per
// reflector and pass — 446 k clk per site, 4.2 % of the fp64 peak counted on the Householder flops.  Here:
//   P1  carry: fragments of X_j straight from global memory (16-byte loads: the left rank index is contiguous and yields s = 0, 1
//       at once), FL from an LDS image; W goes from the accumulators to an XOR-swizzled LDS image (both its fragment read patterns —
//       4 rows x 16 columns for the Gram products, 16 rows x 4 columns for the right factor — are free of bank conflicts);
//   P2  G = W^T W by MFMA from the image (one 16 x 16 tile per wave, 32 k-steps);
//   P3  blocked Cholesky (block 16) in an LDS image: the diagonal block is factored AND inverted in the registers of wave 0 — every
//       16-lane row holds the block, row li in lane li, and the rank-1 updates take their operands through the DPP row broadcast of
//       v_fmac_f64 (no LDS round trip in the pivot chain: ~300 clk per pivot instead of ~680) —, panel and trailing update by MFMA;
//   P4  L^-1 block by block (three levels of 16 x 16 x 16 MFMA products, the inner sums handed on as accumulator registers);
//   P5  Q = W L^-T by MFMA (triangular: the column blocks are paired 0 + 3 / 1 + 2 so that every wave runs 20 k-steps), stored to
//       Y_j from the accumulators;  P6  max |Q^T Q - I| by MFMA from the image;  P7  R = L^T to global memory for the next site.
// Ten workgroup barriers per site, no global scratch.
#pragma once
#include "ttn_ortho_dpp_gen.h"
#include "ttn_common.h"
#include "ttn_dense_kernels.h"
#include "ttn_dot_kernels.h"

#define ORTHO_FUSED_ACCEPT 2.0e-13       // max |Q^T Q - I| up to which the single Cholesky-QR pass is the result (parity bar: 1e-12)
#define OF_AT(k, i) (80 * (k) + 4 * ((k) >> 1) + (i))                                       // k-major 64 x 64 image (as DOT_AT)
#define OF_W(rho, al) ((rho) * 64 + ((al) ^ ((((rho) & 1) << 4) | ((((rho) >> 1) & 3) << 2))))   // 128 x 64 image, XOR swizzle
#define OF_G(i, j) ((j) * 80 + (i))                                                         // column-major 64 x 64 image, pitch 80
#define OF_WIMG 0
#define OF_FIMG 8192                     // FL (P1), then L^-1 (P4, P5)
#define OF_GIMG (8192 + 5248)
#define OF_T16 (8192 + 5248 + 5120)      // inverse of the current diagonal block, [row * 17 + col]
#define OF_MISC (OF_T16 + 16 * 17 + 16)  // [0] flag (int), [1] dev bits (u64), [2] dmax
#define OF_LDS_DOUBLES (OF_MISC + 16)

template <int J>
__device__ __forceinline__ void of_fmac_bcast(double& a, double w, double s_) {     // a += (lane J of the caller's 16-lane row of w) * s
    // (the s_nop: gfx9 needs two wait states between a VALU write of a VGPR and a DPP read of it, and the compiler cannot see the DPP
    //  operand inside the asm — nor a copy it may place right in front of it)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(w), "v"(s_), "n"(J));
}
__device__ __forceinline__ double of_readlane(double v, int idx) {
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = __builtin_amdgcn_readlane(u.i[0], idx);
    r.i[1] = __builtin_amdgcn_readlane(u.i[1], idx);
    return r.d;
}
// gfx9: two wait states between a VALU write of a VGPR and a DPP read of it; the compiler does not see the DPP inside the asm above
#define OF_DPP_FENCE() asm volatile("s_nop 1" ::: "memory")

// Diagonal block jb of the Cholesky factorisation, by ONE wave: factor (L L^T = D) and inverse (X = L^-1), both lower triangular.
// Every 16-lane row of the wave holds the whole block: lane li = row li, e[c] = D[li][c] (the four rows compute the same).
// Returns false on a pivot <= dmin.  Writes L to the image (lower triangle) and X to T16[row * 17 + col] and to the L^-1 image.
__device__ __forceinline__ bool of_diag_block(int jb, lds_f64* G, lds_f64* T16, lds_f64* Li, double dmin) {
    int lane = threadIdx.x & 63;
    // (opaque to the compiler: it hoists the identity columns (li == jj ? 1 : 0) and the LDS addresses of this block out of the caller's
    //  loop over the diagonal blocks, spills them and reloads one from scratch memory in every step of the dependent chain below)
    asm volatile("" : "+v"(lane));
    const int li = lane & 15, lk = lane >> 4;
    double e[16], t[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) e[c] = G[OF_G(16 * jb + li, 16 * jb + c)];
    double dinv = 1.0;
    bool ok = true;
    // Factor and inverse in one loop: step jj of the factorisation finishes column jj of L, and with it the elimination step jj of
    // U'^-1 (L = U' D, U'[i][k] = L[i][k] / L[k][k]) can run — its row jj is final, its multipliers are -L[li][jj] rs_jj: 15 - jj updates
    // of the factor and jj + 1 of the inverse, independent of each other.  X = D^-1 U'^-1: one scaling at the end.  Column jj of L goes
    // to the image as soon as it is final and the columns of the inverse enter the registers when they stop being the identity:
    // 17 doubles live per step.
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const double d = of_readlane(e[jj], jj);
        ok = ok && (d > dmin);                                           // wave-uniform; no early exit: the loop must unroll (e[] stays in registers)
        const double rs = fast_rsqrt2(d);
        const double l = e[jj] * rs;                                     // L[li][jj] (li >= jj); sqrt(d) on the diagonal
        if (lk == 0 && jj <= li) G[OF_G(16 * jb + li, 16 * jb + jj)] = l;
        t[jj] = (li == jj) ? 1.0 : 0.0;
        dinv = (li == jj) ? rs : dinv;                                   // 1 / L[li][li]
        const double nl = -l;
        const double s_ = (li > jj) ? nl * rs : 0.0;                     // -L[li][jj] / L[jj][jj]
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
        for (int c = 0; c < 16; ++c) {
            T16[li * 17 + c] = (c <= li) ? t[c] : 0.0;
            Li[OF_AT(16 * jb + c, 16 * jb + li)] = (c <= li) ? t[c] : 0.0;          // L^-1[al' = row][al = col] at OF_AT(al, al')
        }
    }
    return true;
}

// max over the valid entries of |(image^T image)[a][b] - delta|, or the Gram matrix itself into G (transposed store: it is symmetric)
template <bool CHECK>
__device__ __forceinline__ double of_gram(const lds_f64* W, lds_f64* G, int nbt, int nat, int rl) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lk = lane >> 4;
    const int ta = wave & 3, tb = wave >> 2;
    double dev = 0.0;
    if (ta < nat && tb < nat) {
        mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
        for (int s = 0; s < 2; ++s)
            for (int t = 0; t < 4 * nbt; ++t) {
                const int rho = 64 * s + 4 * t + lk;
                const double a = W[OF_W(rho, 16 * ta + li)], b = W[OF_W(rho, 16 * tb + li)];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int al = 16 * ta + lk + 4 * reg, alp = 16 * tb + li;
            const bool valid = al < rl && alp < rl;
            if (CHECK) { if (valid) dev = fmax(dev, fabs(acc[reg] - ((al == alp) ? 1.0 : 0.0))); }
            else G[OF_G(alp, al)] = valid ? acc[reg] : ((al == alp) ? 1.0 : 0.0);
        }
    }
    return dev;
}

// Returns rl (= the new rank: full column rank) or 0 (bad pivot / orthogonality above the bar: nothing the caller relies on was changed
// except Yj and Rout, which the general route rewrites).  FLg: the previous site's R in the layout of wg_qr_explicit
// (FL[ga, be] = FLg[be + ynext * ga]); Rout receives R = L^T in the same layout.  All 16 waves, contains barriers.
__device__ __noinline__ int ortho_step_fused(const double* Xj, double* Yj, const double* FLg, double* Rout, int rl, int rr, int ynext, double* lds,
                                             long long* stamps) {
    Xj = unip(Xj); Yj = unip(Yj); FLg = unip(FLg); Rout = unip(Rout); lds = unip(lds);
    rl = uni32(rl); rr = uni32(rr); ynext = uni32(ynext);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    lds_f64* W = (lds_f64*)lds + OF_WIMG;
    lds_f64* F = (lds_f64*)lds + OF_FIMG;
    lds_f64* G = (lds_f64*)lds + OF_GIMG;
    lds_f64* T16 = (lds_f64*)lds + OF_T16;
    lds_f64* misc = (lds_f64*)lds + OF_MISC;
    lds_i32* flag = (lds_i32*)misc;
    const int nbt = (ynext + 15) >> 4, nat = (rl + 15) >> 4;            // tile rows per s of W, tile columns of W
    long long t_prev = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define OFS(i) if (stamps) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); if (tid == 0) stamps[i] += now_ - t_prev; t_prev = now_; }
    // ---- P0: FL image (zero padded), G := I, flags ----
    __syncthreads();
    for (int e = tid; e < 5248; e += TTN_WG) F[e] = 0.0;
    for (int e = tid; e < 5120; e += TTN_WG) G[e] = ((e % 80) == (e / 80)) ? 1.0 : 0.0;
    if (tid == 0) { flag[0] = 0; ((__attribute__((address_space(3))) unsigned long long*)misc)[1] = 0ull; }
    __syncthreads();
    for (int e = tid; e < rr * ynext; e += TTN_WG) { const int ga = e / ynext, be = e - ga * ynext; F[OF_AT(ga, be)] = FLg[e]; }
    __syncthreads();
    OFS(0)
    // ---- P1: W_s[be, al] = sum_ga FL[ga, be] X_j[s, al, ga]; wave (tr, tc) owns rows be = 16 tr + ., columns al = 16 tc + . ----
    {
        const int tr = wave & 3, tc = wave >> 2;
        if (tr < nbt && tc < nat) {
            mfma_acc_t w0 = (mfma_acc_t){0.0, 0.0, 0.0, 0.0}, w1 = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
            const int nt = (rr + 3) >> 2, aq = 16 * tc + li;
            for (int t = 0; t < nt; ++t) {
                const dot_f64x2 bv = dot_load2(Xj, aq, 4 * t + lk, rl, rr, rl);
                const double a = F[OF_AT(4 * t + lk, 16 * tr + li)];
                w0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv.x, w0, 0, 0, 0);
                w1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv.y, w1, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int be = 16 * tr + lk + 4 * reg;
                W[OF_W(be, aq)] = w0[reg];
                W[OF_W(64 + be, aq)] = w1[reg];
            }
        }
    }
    __syncthreads();
    OFS(1)
    // ---- P2: G = W^T W; the FL image is dead: zero it for L^-1 ----
    for (int e = tid; e < 5248; e += TTN_WG) F[e] = 0.0;
    of_gram<false>(W, G, nbt, nat, rl);
    __syncthreads();
    OFS(2)
    // ---- P3: blocked Cholesky of the leading 16 nat columns (the padding is the identity) ----
    if (wave == 0) {
        double dm = 0.0;
        for (int i = lane; i < rl; i += 64) dm = fmax(dm, G[OF_G(i, i)]);
        dm = wave_max(dm);
        if (lane == 0) misc[2] = dm;
    }
    __syncthreads();
    const double dmax = unif64(misc[2]);
    const double dmin = 64.0 * DBL_EPSILON * dmax;
    // the padding of the last block takes the scale of the matrix (a unit diagonal would fail the pivot test of a train whose carried
    // factor has grown to 1e26: the norm of a random rank-64 chain accumulates in FL)
    if (!(dmax > 0.0)) return 0;
    for (int i = rl + tid; i < 16 * nat; i += TTN_WG) G[OF_G(i, i)] = dmax;
    __syncthreads();
    for (int jb = 0; jb < nat; ++jb) {
        if (wave == 0) { if (!of_diag_block(jb, G, T16, F, dmin) && lane == 0) flag[0] = 1; }
        __syncthreads();
        if (uni32(flag[0])) return 0;
        // panel: L[i-block][jb] = G[i-block][jb] X_jj^T, one wave per block row
        if (wave >= 1 && jb + wave < nat) {
            const int ib = jb + wave;
            mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kk = 4 * t + lk;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G[OF_G(16 * ib + li, 16 * jb + kk)], T16[li * 17 + kk], acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) G[OF_G(16 * ib + lk + 4 * reg, 16 * jb + li)] = acc[reg];
        }
        __syncthreads();
        // trailing update: tiles (ib, kb), jb < kb <= ib < nat
        {
            const int nrem = nat - jb - 1, ntile = nrem * (nrem + 1) / 2;
            for (int tile = wave; tile < ntile; tile += TTN_NWAVES) {
                int r_ = 0, base = 0;
                while (base + r_ + 1 <= tile) { base += r_ + 1; ++r_; }
                const int ib = jb + 1 + r_, kb = jb + 1 + (tile - base);
                mfma_acc_t acc = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = 4 * t + lk;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(G[OF_G(16 * ib + li, 16 * jb + kk)], G[OF_G(16 * kb + li, 16 * jb + kk)], acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) G[OF_G(16 * ib + lk + 4 * reg, 16 * kb + li)] -= acc[reg];
            }
        }
        __syncthreads();
    }
    OFS(3)
    // ---- P4: off-diagonal blocks of X = L^-1, level by level: X_ij = -X_ii sum_{k = j}^{i - 1} L_ik X_kj ----
    for (int lev = 1; lev < nat; ++lev) {
        const int ib = lev + wave, jbk = wave;
        if (ib < nat) {
            mfma_acc_t s_ = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
            for (int kb = jbk; kb < ib; ++kb) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = 4 * t + lk;                            // S[row][col] += L[16 ib + row][16 kb + kk] X[16 kb + kk][16 jb + col]
                    s_ = __builtin_amdgcn_mfma_f64_16x16x4f64(G[OF_G(16 * ib + li, 16 * kb + kk)], F[OF_AT(16 * jbk + li, 16 * kb + kk)], s_, 0, 0, 0);
                }
            }
            mfma_acc_t x_ = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r)                                  // X_ij[row][col] = -sum_m X_ii[row][m] S[m][col]; S register r = rows m = 4 r + lk
                x_ = __builtin_amdgcn_mfma_f64_16x16x4f64(F[OF_AT(16 * ib + 4 * r + lk, 16 * ib + li)], s_[r], x_, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) F[OF_AT(16 * jbk + li, 16 * ib + lk + 4 * reg)] = -x_[reg];
        }
        __syncthreads();
    }
    OFS(4)
    // ---- P5: Q[rho][al'] = sum_{al <= al'} W[rho][al] X[al'][al]; wave (rb, ch): row block rb, column blocks {0, 3} / {1, 2} ----
    {
        const int rb = wave & 7, ch = wave >> 3;
        const int rho0 = 16 * (rb & 3) + 64 * (rb >> 2);
        const int c0 = ch ? 1 : 0, c1 = ch ? 2 : 3;
        const bool act = (rb & 3) < nbt;
        mfma_acc_t q0 = (mfma_acc_t){0.0, 0.0, 0.0, 0.0}, q1 = (mfma_acc_t){0.0, 0.0, 0.0, 0.0};
        if (act) {
            const int ntmax = 4 * min(nat, c1 + 1);
            for (int t = 0; t < ntmax; ++t) {
                const int al = 4 * t + lk;
                const double a = W[OF_W(rho0 + li, al)];
                if (t < 4 * (c0 + 1) && c0 < nat) q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, F[OF_AT(al, 16 * c0 + li)], q0, 0, 0, 0);
                if (c1 < nat) q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, F[OF_AT(al, 16 * c1 + li)], q1, 0, 0, 0);
            }
            // Y_j[s, al', be] at s + 2 (al' + rl be)
            const int s = rb >> 2;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int be = 16 * (rb & 3) + lk + 4 * reg;
                if (be < ynext) {
                    if (16 * c0 + li < rl) Yj[s + 2 * (16 * c0 + li + rl * be)] = q0[reg];
                    if (16 * c1 + li < rl) Yj[s + 2 * (16 * c1 + li + rl * be)] = q1[reg];
                }
            }
        }
        __syncthreads();                                                 // every fragment read of W is done: Q takes its place
        if (act) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rho = rho0 + lk + 4 * reg;
                if (c0 < nat) W[OF_W(rho, 16 * c0 + li)] = q0[reg];
                if (c1 < nat) W[OF_W(rho, 16 * c1 + li)] = q1[reg];
            }
        }
        __syncthreads();
    }
    OFS(5)
    // ---- P6: the orthogonality of Q, measured ----
    {
        double dev = of_gram<true>(W, G, nbt, nat, rl);
        dev = wave_max(dev);
        if (lane == 0) atomicMax((unsigned long long*)(lds + OF_MISC) + 1, (unsigned long long)__double_as_longlong(dev));
        __syncthreads();
        const double devmax = __longlong_as_double((long long)((__attribute__((address_space(3))) unsigned long long*)misc)[1]);
        if (!(unif64(devmax) <= ORTHO_FUSED_ACCEPT)) return 0;
    }
    OFS(6)
    // ---- P7: R = L^T for the next site: Rout[i + rl c] = L[c][i], i <= c ----
    for (int e = tid; e < rl * rl; e += TTN_WG) { const int c = e / rl, i = e - c * rl; Rout[e] = (i <= c) ? (double)G[OF_G(c, i)] : 0.0; }
    __syncthreads();
    OFS(7)
#undef OFS
    return rl;
}
