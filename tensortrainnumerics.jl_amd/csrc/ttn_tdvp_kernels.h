// ttn_tdvp_kernels.h — the local contractions of TDVP (src/solvers/tdvp.jl:29-43, :205-208) as chains of fp64 MFMA GEMMs, batched
// (one workgroup per system of the batch), real or complex.
//
//   _applyH1_lsr      HAC[α,s,β]        = FL[α,a,α'] AC[α',s',β'] M[a,s,b,s'] FR[β',b,β]                         (:29-31)
//   _applyH0          HC[α,β]           = FL[α,a,α'] C[α',β'] FR[β',a,β]                                         (:33-35)
//   _update_left_env  FLnext[α,a,β]     = FL[α',a',β'] A[β',s',β] M[a',s,a,s'] conj(A[α',s,α])                   (:37-39)
//   _update_right_env FRprev[α,a,β]     = A[α,s',α'] FR[α',a',β'] M[a,s,a',s'] conj(A[β,s,β'])                   (:41-43)
//   _applyH2_lsr      HAAC[α,s1,s2,β]   = FL[α,a,α'] AAC[α',s1',s2',β'] M1[a,s1,b,s1'] M2[b,s2,c,s2'] FR[β',c,β] (:205-208)
//
// All tensors are column-major in the reference's (l, s, r) / (a, s, b, s') layouts, exactly as `tdvp1sweep!` holds them
// (A_lsr = permutedims(ttv_vec, (2,1,3)), M_asbs = permutedims(tto_vec, (3,1,4,2)), :52-53).  Every contraction is the same
// three-tensor sandwich as the two-site DMRG operator: an environment product, the small operator contraction (inner dimension
// a*s, a GEMM with a two-level k index — no permuted copies), the other environment.  Intermediate layouts are chosen so that
// every GEMM operand is a strided View of the arrays as they lie.
// COMPLEX (real-time evolution, :166-172): ComplexF64 arrays are interleaved (re, im) pairs; the real and imaginary parts of an
// operand are strided Views of the same memory (element stride 2), and a complex product is four real GEMM calls
// (Cre = Are Bre - Aim Bim, Cim = Are Bim + Aim Bre; conj(A) flips the sign of Aim) — the MFMA GEMM itself is unchanged.
#pragma once
#include "ttn_dense_kernels.h"

struct TdvpArgs {
    int op;                      // 0 applyH1, 1 applyH0, 2 update_left_env, 3 update_right_env, 4 applyH2
    int cplx;                    // 0: Float64, 1: ComplexF64 (interleaved)
    int Dl, d, Dr, a, b, c, d2;  // ranks / physical dimensions / operator ranks (op-specific, see k_tdvp)
    const double *FL, *FR, *X, *M1, *M2;       // X: AC / C / A / AAC
    double* out;
    double* work;                // per system: two intermediates
    long long sFL, sFR, sX, sM1, sM2, sOut, sWork, w2off;     // strides between systems of the batch (in scalars of the element type; 0: shared)
};

// element-type aware View builders: `es` = doubles per element (1 or 2), `part` = 0 real / 1 imaginary
__device__ inline View tv(const double* base, int es, int part, Idx r, Idx c) {
    Idx r2 = r, c2 = c;
    r2.lo *= es; r2.hi *= es; c2.lo *= es; c2.hi *= es;
    return View{const_cast<double*>(base) + part, r2, c2};
}
// C = op(A) * B for real (es = 1) or complex (es = 2) operands given as element-index Views (strides in ELEMENTS); conjA: use conj(A)
__device__ void wg_zgemm(int m, int n, int k, const double* A, Idx Ar, Idx Ac, bool conjA, const double* B, Idx Br, Idx Bc,
                         double* C, Idx Cr, Idx Cc, int es, double* lds) {
    if (es == 1) {
        wg_gemm(m, n, k, tv(A, 1, 0, Ar, Ac), tv(B, 1, 0, Br, Bc), tv(C, 1, 0, Cr, Cc), 1.0, 0.0, lds);
        return;
    }
    const double sg = conjA ? -1.0 : 1.0;
    const View Are = tv(A, 2, 0, Ar, Ac), Aim = tv(A, 2, 1, Ar, Ac), Bre = tv(B, 2, 0, Br, Bc), Bim = tv(B, 2, 1, Br, Bc);
    const View Cre = tv(C, 2, 0, Cr, Cc), Cim = tv(C, 2, 1, Cr, Cc);
    wg_gemm(m, n, k, Are, Bre, Cre, 1.0, 0.0, lds);
    wg_gemm(m, n, k, Aim, Bim, Cre, -sg, 1.0, lds);
    wg_gemm(m, n, k, Are, Bim, Cim, 1.0, 0.0, lds);
    wg_gemm(m, n, k, Aim, Bre, Cim, sg, 1.0, lds);
}

#define TDVP_LDS_BYTES (sizeof(double) * GEMM_LDS_TOTAL)

__global__ void TTN_KERNEL_BOUNDS k_tdvp(TdvpArgs P) {
    extern __shared__ double lds[];
    const int t = blockIdx.x;
    const int es = P.cplx ? 2 : 1;
    const double* FL = P.FL ? P.FL + (long long)t * P.sFL * es : nullptr;
    const double* FR = P.FR ? P.FR + (long long)t * P.sFR * es : nullptr;
    const double* X = P.X + (long long)t * P.sX * es;
    const double* M1 = P.M1 ? P.M1 + (long long)t * P.sM1 * es : nullptr;
    const double* M2 = P.M2 ? P.M2 + (long long)t * P.sM2 * es : nullptr;
    double* out = P.out + (long long)t * P.sOut * es;
    double* W1 = P.work + (long long)t * P.sWork * es;
    double* W2 = W1 + P.w2off * es;
    const long long Dl = P.Dl, d = P.d, Dr = P.Dr, a = P.a, b = P.b;
    if (P.op == 0) {
        // T1[(α,a),(s',β')] = FL[(α,a),α'] AC[α',(s',β')]
        wg_zgemm((int)(Dl * a), (int)(d * Dr), (int)Dl, FL, plain(1), plain(Dl * a), false, X, plain(1), plain(Dl), W1, plain(1), plain(Dl * a), es, lds);
        // T2[(α,s),(β',b)] = sum_{(a,s')} T1[α,a,s',β'] M[a,s,b,s']:  rows (α,β'), k = a + A s', columns (s,b)
        wg_zgemm((int)(Dl * Dr), (int)(d * b), (int)(a * d), W1, Idx{(int)Dl, 1, Dl * a * d}, plain(Dl), false,
                 M1, Idx{(int)a, 1, a * d * b}, plain(a), W2, Idx{(int)Dl, 1, Dl * d}, Idx{(int)d, Dl, Dl * d * Dr}, es, lds);
        // HAC[(α,s),β] = sum_{(β',b)} T2[(α,s),(β',b)] FR[(β',b),β]
        wg_zgemm((int)(Dl * d), (int)Dr, (int)(Dr * b), W2, plain(1), plain(Dl * d), false, FR, plain(1), plain(Dr * b), out, plain(1), plain(Dl * d), es, lds);
    } else if (P.op == 1) {
        // T1[(α,a),β'] = FL[(α,a),α'] C[α',β'] ;  HC[α,β] = sum_{(β',a)} T1[α,a,β'] FR[β',a,β]
        wg_zgemm((int)(Dl * a), (int)Dr, (int)Dl, FL, plain(1), plain(Dl * a), false, X, plain(1), plain(Dl), W1, plain(1), plain(Dl * a), es, lds);
        wg_zgemm((int)Dl, (int)Dr, (int)(Dr * a), W1, plain(1), Idx{(int)Dr, Dl * a, Dl}, false, FR, plain(1), plain(Dr * a), out, plain(1), plain(Dl), es, lds);
    } else if (P.op == 2) {
        // FL (Dl, a, Dl), A (Dl, d, Dr), M (a, d, b, d) -> FLnext (Dr, b, Dr)
        // T1[(α',a'),(s',β)] = FL[(α',a'),β'] A[β',(s',β)]
        wg_zgemm((int)(Dl * a), (int)(d * Dr), (int)Dl, FL, plain(1), plain(Dl * a), false, X, plain(1), plain(Dl), W1, plain(1), plain(Dl * a), es, lds);
        // T2[(α',s),(a2,β)] = sum_{(a',s')} T1[α',a',s',β] M[a',s,a2,s']: rows (α',β), k = a' + A s', columns (s,a2)
        wg_zgemm((int)(Dl * Dr), (int)(d * b), (int)(a * d), W1, Idx{(int)Dl, 1, Dl * a * d}, plain(Dl), false,
                 M1, Idx{(int)a, 1, a * d * b}, plain(a), W2, Idx{(int)Dl, 1, Dl * d * b}, Idx{(int)d, Dl, Dl * d}, es, lds);
        // FLnext[α,(a2,β)] = sum_{(α',s)} conj(A[(α',s),α]) T2[(α',s),(a2,β)]
        wg_zgemm((int)Dr, (int)(b * Dr), (int)(Dl * d), X, plain(Dl * d), plain(1), true, W2, plain(1), plain(Dl * d), out, plain(1), plain(Dr), es, lds);
    } else if (P.op == 3) {
        // A (Dl, d, Dr), FR (Dr, a, Dr), M (b, d, a, d) -> FRprev (Dl, b, Dl)
        // T1[(α,s'),(a',β')] = A[(α,s'),α'] FR[α',(a',β')]
        wg_zgemm((int)(Dl * d), (int)(a * Dr), (int)Dr, X, plain(1), plain(Dl * d), false, FR, plain(1), plain(Dr), W1, plain(1), plain(Dl * d), es, lds);
        // T2[(α,a2),(s,β')] = sum_{(s',a')} T1[α,s',a',β'] M[a2,s,a',s']: rows (α,β'), k = s' + d a', columns (a2,s)
        wg_zgemm((int)(Dl * Dr), (int)(b * d), (int)(d * a), W1, Idx{(int)Dl, 1, Dl * d * a}, plain(Dl), false,
                 M1, Idx{(int)d, b * d * a, b * d}, plain(1), W2, Idx{(int)Dl, 1, Dl * b * d}, Idx{(int)b, Dl, Dl * b}, es, lds);
        // FRprev[(α,a2),β] = sum_{(s,β')} T2[(α,a2),(s,β')] conj(A[β,(s,β')])
        // (computed as its transpose: out^T[β,(α,a2)] = conj(A[β,(s,β')]) T2^T — the conjugate sits on the first operand of wg_zgemm)
        wg_zgemm((int)Dl, (int)(b * Dl), (int)(d * Dr), X, plain(1), plain(Dl), true, W2, plain(Dl * b), plain(1), out, plain(Dl * b), plain(1), es, lds);
    } else {
        // two-site: FL (Dl, a, Dl), AAC (Dl, d, d2, Dr), M1 (a, d, b, d), M2 (b, d2, c, d2), FR (Dr, c, Dr)
        const long long d2 = P.d2, c = P.c;
        // T1[(α,a),(s1',s2',β')] = FL AAC
        wg_zgemm((int)(Dl * a), (int)(d * d2 * Dr), (int)Dl, FL, plain(1), plain(Dl * a), false, X, plain(1), plain(Dl), W1, plain(1), plain(Dl * a), es, lds);
        // T2[(α,s1),b,(s2',β')] = sum_{(a,s1')} T1[α,a,s1',(s2',β')] M1[a,s1,b,s1']: rows (α,(s2',β')), k = a + A s1', columns (s1,b)
        wg_zgemm((int)(Dl * d2 * Dr), (int)(d * b), (int)(a * d), W1, Idx{(int)Dl, 1, Dl * a * d}, plain(Dl), false,
                 M1, Idx{(int)a, 1, a * d * b}, plain(a), W2, Idx{(int)Dl, 1, Dl * d * b}, plain(Dl), es, lds);
        // T3[(α,s1,s2),(β',c)] = sum_{(b,s2')} T2[(α,s1),b,s2',β'] M2[b,s2,c,s2']: rows ((α,s1),β'), k = b + B s2', columns (s2,c)
        wg_zgemm((int)(Dl * d * Dr), (int)(d2 * c), (int)(b * d2), W2, Idx{(int)(Dl * d), 1, Dl * d * b * d2}, plain(Dl * d), false,
                 M2, Idx{(int)b, 1, b * d2 * c}, plain(b), W1, Idx{(int)(Dl * d), 1, Dl * d * d2}, Idx{(int)d2, Dl * d, Dl * d * d2 * Dr}, es, lds);
        // HAAC[(α,s1,s2),β] = sum_{(β',c)} T3[(α,s1,s2),(β',c)] FR[(β',c),β]
        wg_zgemm((int)(Dl * d * d2), (int)Dr, (int)(Dr * c), W1, plain(1), plain(Dl * d * d2), false, FR, plain(1), plain(Dr * c), out, plain(1), plain(Dl * d * d2), es, lds);
    }
}
