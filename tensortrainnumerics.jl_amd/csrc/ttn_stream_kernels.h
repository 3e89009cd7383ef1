// ttn_stream_kernels.h — HBM-bound per-core kernels: apply (tto*ttv), hadamard, +, scalar*.
// Every kernel is launched on grid (tiles, d, batch): blockIdx.y = core, blockIdx.z = train.
// Ranks are read from device memory (they are data dependent after tt_compress!).
#pragma once
#include "ttn_common.h"

#define TTN_STREAM_TB 256
#define TTN_APPLY_K 4                     // output columns per thread in k_apply (n = 2); measured at B = 1024: K = 1 0.36, 2 0.53, 3 0.58, 4 0.635, 5 0.62, 6 0.54, 8 0.55, 12 0.51 of 8 TB/s
#define TTN_ADD_K 4                       // columns per thread in k_add (n = 2): 2 0.61, 4 0.65, 8 0.62
#define TTN_HAD_K 2                       // right indices per thread in k_hadamard (n = 2): 2 0.74, 4 0.69, 8 0.62

// ---------------------------------------------------------------------------------------------
// rank bookkeeping (Int64, bit-exact): tiny one-block kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_ranks_mul_op(TTDev y, TTODev A, TTDev x) {      // y.rks = A.rks .* x.rks  (tt_operations.jl:103)
    int b = blockIdx.x;
    for (int m = threadIdx.x; m <= x.d; m += blockDim.x)
        y.rks[(long long)b * (y.d + 1) + m] = A.rks[m] * x.rks[(long long)b * (x.d + 1) + m];
}
__global__ void k_ranks_mul(TTDev z, TTDev x, TTDev y) {          // hadamard: ranks multiply (tt_operations.jl:348)
    int b = blockIdx.x;
    for (int m = threadIdx.x; m <= x.d; m += blockDim.x)
        z.rks[(long long)b * (z.d + 1) + m] = x.rks[(long long)b * (x.d + 1) + m] * y.rks[(long long)b * (y.d + 1) + m];
}
__global__ void k_ranks_add(TTDev z, TTDev x, TTDev y) {          // +: ranks add, ends forced to 1 (tt_operations.jl:14-16)
    int b = blockIdx.x;
    for (int m = threadIdx.x; m <= x.d; m += blockDim.x) {
        long long r = x.rks[(long long)b * (x.d + 1) + m] + y.rks[(long long)b * (y.d + 1) + m];
        if (m == 0 || m == x.d) r = 1;
        z.rks[(long long)b * (z.d + 1) + m] = r;
    }
}
__global__ void k_ranks_copy(TTDev y, TTDev x) {
    int b = blockIdx.x;
    for (int m = threadIdx.x; m <= x.d; m += blockDim.x)
        y.rks[(long long)b * (y.d + 1) + m] = x.rks[(long long)b * (x.d + 1) + m];
}

// ---------------------------------------------------------------------------------------------
// apply:  Y_k[i, a' + Rl*v', a + Rr*v] = sum_j A_k[i,j,a',a] * X_k[j,v',v]
// (src/tt_operations.jl:101-111; operator index fastest in the combined rank index, from the
// reshape at :106).  The operator core (n*n*Rl*Rr doubles, 36 for the Laplacian) is staged in LDS.
// HBM-write bound.
// ---------------------------------------------------------------------------------------------
#define TTN_APPLY_LDS_DOUBLES 4096
#define TTN_APPLY_MAX_RL 8              // operator left ranks the LDS-transposed store path of k_apply handles
// One thread per INPUT fibre (v', v): reads the n doubles X_k[:, v', v] once and writes all Rl*Rr output fibres
// Y_k[:, a' + Rl*v', a + Rr*v].  For a fixed a the Rl fibres a' = 0..Rl-1 are contiguous (n*Rl doubles, 48 B for the
// Laplacian) and consecutive threads (consecutive v') continue the same run, so a wave writes Rr contiguous
// runs of 64*n*Rl doubles: fully coalesced stores, one integer division per thread, 16 B read per 144 B written.
// Dynamic LDS, sized by the host for the operator at hand (lds_a doubles for the operator core, then lds_rl * 64 double2 per wave for
// the store transpose): with the maximum sizes allocated statically (64 KB per 256-thread block) only two blocks fitted a CU —
// 8 waves, far too few to keep 3 GB of streaming stores in flight.
__global__ void __launch_bounds__(TTN_STREAM_TB) k_apply(TTODev A, TTDev x, TTDev y, int lds_a, int lds_rl) {
    extern __shared__ double apply_smem[];
    double* As = apply_smem;
    const int k = blockIdx.y, b = blockIdx.z;
    const int n = x.dims[k];
    const int Rl = (int)A.rks[k], Rr = (int)A.rks[k + 1];
    const long long* xr = x.rks + (long long)b * (x.d + 1);
    const int rl = (int)xr[k], rr = (int)xr[k + 1];
    const long long total = (long long)rl * rr;                       // input fibres
    const long long first = (long long)blockIdx.x * blockDim.x;
    const double* Ak = A.data + A.off[k];
    const int asz = n * n * Rl * Rr;
    const bool in_lds = asz <= lds_a;
    // (the output-row mapping below counts rows x column groups, the mappings after it input fibres: a block beyond the work of its mapping leaves)
    if ((n == 2 && in_lds) ? first >= (long long)Rl * rl * (((long long)Rr * rr + TTN_APPLY_K - 1) / TTN_APPLY_K) : first >= total) return;
    if (in_lds) {
        for (int e = threadIdx.x; e < asz; e += blockDim.x) As[e] = Ak[e];
        __syncthreads();
    }
    const double* Ap = in_lds ? As : Ak;
    const double* Xk = x.data + (long long)b * x.stride + x.off[k];
    double* Yk = y.data + (long long)b * y.stride + y.off[k];
    const long long P = (long long)Rl * rl;                           // left rank of Y
#ifndef TTN_APPLY_BY_INPUT_FIBRE
    // n = 2, operator core in LDS: one thread = one OUTPUT row p = a' + Rl v' and TTN_APPLY_K consecutive output columns c = a + Rr v
    // (the layout of k_hadamard: Y_k is the Kronecker product of the operator core and X_k, slice by slice).  For every j the lanes of
    // a wave write consecutive rows: 16 bytes per lane, coalesced, non-temporal; the index arithmetic is paid once per K fibres, the
    // column pair (a, v) advances by increment, the input fibre is reloaded only when v changes (every Rr columns; the Rl lanes that
    // share it hit the same line).  Replaces the wave-level LDS transpose of the input-fibre mapping (4.1 TB/s; this form: see §4.1).
    if (n == 2 && in_lds) {
        typedef double d2v_t __attribute__((ext_vector_type(2)));
        const unsigned int uP = (unsigned int)P, uQ = (unsigned int)((long long)Rr * rr), cgroups = (uQ + TTN_APPLY_K - 1) / TTN_APPLY_K, items = uP * cgroups;
        for (unsigned int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
            const unsigned int p = it % uP, c0 = (it / uP) * TTN_APPLY_K;
            const unsigned int al = p % (unsigned int)Rl, vl = p / (unsigned int)Rl;
            unsigned int ar = c0 % (unsigned int)Rr, vr = c0 / (unsigned int)Rr;
            d2v_t o[TTN_APPLY_K];
            d2v_t xv = (c0 < uQ) ? *reinterpret_cast<const d2v_t*>(Xk + 2 * ((long long)vl + (long long)rl * vr)) : (d2v_t){0.0, 0.0};
#pragma unroll
            for (int j = 0; j < TTN_APPLY_K; ++j) {
                const double* ap = As + 4 * ((long long)al + (long long)Rl * ar);      // A[i, j, a', a] at i + 2 j + 4 (a' + Rl a)
                o[j].x = fma(ap[2], xv.y, ap[0] * xv.x);
                o[j].y = fma(ap[3], xv.y, ap[1] * xv.x);
                if (++ar == (unsigned int)Rr) {
                    ar = 0; ++vr;
                    if (j + 1 < TTN_APPLY_K && c0 + j + 1 < uQ) xv = *reinterpret_cast<const d2v_t*>(Xk + 2 * ((long long)vl + (long long)rl * vr));
                }
            }
#pragma unroll
            for (int j = 0; j < TTN_APPLY_K; ++j)
                if (c0 + j < uQ) __builtin_nontemporal_store(o[j], reinterpret_cast<d2v_t*>(Yk + 2 * ((long long)p + (long long)uP * (c0 + j))));
        }
        return;
    }
#endif
#ifndef TTN_APPLY_NO_TRANSPOSE
    // n = 2 and rl a multiple of 64 (the interior cores): a wave's 64 input fibres are 64 consecutive left indices of ONE
    // right index, so for every operator right index `ar` its Rl*64 output fibres are one contiguous run of Rl*64*16 bytes.
    // Per-lane stores would hit that run 16 bytes at a stride of Rl*16 (Rl partial passes over every cache line); instead the
    // wave transposes the run through LDS and writes it with fully coalesced 16-byte-per-lane stores.
    double2* Tr = reinterpret_cast<double2*>(apply_smem + ((lds_a + 1) & ~1));
    if (n == 2 && (rl & 63) == 0 && Rl <= lds_rl && (total % blockDim.x) == 0) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        double2* tr = Tr + (long long)wv * lds_rl * 64;
        for (unsigned int e = (unsigned int)first + threadIdx.x; e < (unsigned int)total; e += gridDim.x * blockDim.x) {       // (32-bit index arithmetic)
            const int vl = (int)(e % (unsigned int)rl), vr = (int)(e / (unsigned int)rl);
            const double2 xv = *reinterpret_cast<const double2*>(Xk + 2 * (long long)e);
            const int vl0 = vl - lane;                                  // first left index of this wave (uniform)
            for (int ar = 0; ar < Rr; ++ar) {
                const double* ap = Ap + 4 * (long long)Rl * ar;
                for (int al = 0; al < Rl; ++al) {
                    double2 o;
                    o.x = fma(ap[4 * al + 2], xv.y, ap[4 * al + 0] * xv.x);
                    o.y = fma(ap[4 * al + 3], xv.y, ap[4 * al + 1] * xv.x);
                    tr[Rl * lane + al] = o;                             // position inside the run: al + Rl*(vl - vl0)
                }
                __builtin_amdgcn_wave_barrier();
                double2* yo = reinterpret_cast<double2*>(Yk + 2 * ((long long)Rl * vl0 + P * (ar + (long long)Rr * vr)));
                for (int t = 0; t < Rl; ++t) {
                    const double2 o = tr[64 * t + lane];
                    // non-temporal: y is written once and read by a later kernel; measured on C3, B = 256: 0.85 ms (3.8 TB/s)
                    // against 1.03 ms with plain stores and 0.95 ms for the per-lane strided stores below.  ONE 16-byte store per
                    // lane (a native 2-vector: the builtin on the two members of a double2 emits two 8-byte stores)
                    typedef double d2v_t __attribute__((ext_vector_type(2)));
                    d2v_t ov; ov.x = o.x; ov.y = o.y;
                    __builtin_nontemporal_store(ov, reinterpret_cast<d2v_t*>(&yo[64 * t + lane]));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
#endif
    for (unsigned int e = (unsigned int)first + threadIdx.x; e < (unsigned int)total; e += gridDim.x * blockDim.x) {
        const int vl = (int)(e % (unsigned int)rl), vr = (int)(e / (unsigned int)rl);
        const double* xs = Xk + (long long)n * e;
        if (n == 2) {
            const double x0 = xs[0], x1 = xs[1];
            for (int ar = 0; ar < Rr; ++ar) {
                double* yo = Yk + 2 * ((long long)Rl * vl + P * (ar + (long long)Rr * vr));
                const double* ap = Ap + 4 * (long long)Rl * ar;
                for (int al = 0; al < Rl; ++al) {
                    double2 o;
                    o.x = fma(ap[4 * al + 2], x1, ap[4 * al + 0] * x0);     // i=0: A[0,0]*x0 + A[0,1]*x1
                    o.y = fma(ap[4 * al + 3], x1, ap[4 * al + 1] * x0);     // i=1
                    *reinterpret_cast<double2*>(yo + 2 * al) = o;      // (non-temporal here: 3.5 ms — partial-line streaming stores)
                }
            }
        } else {
            for (int ar = 0; ar < Rr; ++ar)
                for (int al = 0; al < Rl; ++al) {
                    double* yo = Yk + (long long)n * ((al + (long long)Rl * vl) + P * (ar + (long long)Rr * vr));
                    const double* ap = Ap + (long long)n * n * (al + (long long)Rl * ar);
                    for (int i = 0; i < n; ++i) {
                        double acc = ap[i] * xs[0];
                        for (int j = 1; j < n; ++j) acc = fma(ap[i + n * j], xs[j], acc);
                        yo[i] = acc;
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// hadamard: Z_k[s, ay + ryl*ax, by + ryr*bx] = X_k[s,ax,bx] * Y_k[s,ay,by]
// (src/tt_operations.jl:343-361: kron per physical slice, y index fastest)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TTN_STREAM_TB) k_hadamard(TTDev x, TTDev y, TTDev z) {
    const int k = blockIdx.y, b = blockIdx.z;
    const int n = x.dims[k];
    const long long* xr = x.rks + (long long)b * (x.d + 1);
    const long long* yr = y.rks + (long long)b * (y.d + 1);
    const int rxl = (int)xr[k], rxr = (int)xr[k + 1], ryl = (int)yr[k], ryr = (int)yr[k + 1];
    const int P = rxl * ryl, Q = rxr * ryr;
    const long long total = (long long)P * Q;
    const double* Xk = x.data + (long long)b * x.stride + x.off[k];
    const double* Yk = y.data + (long long)b * y.stride + y.off[k];
    double* Zk = z.data + (long long)b * z.stride + z.off[k];
    // 32-bit index arithmetic (a core has fewer than 2^31 fibres: the host refuses larger ones): the 64-bit divisions of the first
    // version were most of the kernel's instructions, and the kernel was bound by them, not by HBM
    const unsigned int utotal = (unsigned int)total, uP = (unsigned int)P;
    if (n == 2) {
        // One thread = one left index p and TTN_HAD_K consecutive right indices q: the six integer divisions of an output fibre are
        // paid once per K fibres (the right pair (by, bx) advances by increment), the 2 K loads are in flight together, and for every
        // j the lanes of a wave still write consecutive p: 16 bytes per lane, 1 KB per wave, coalesced.  (One fibre per thread: 0.36
        // of 8 TB/s — the kernel was bound by its index arithmetic and by the latency of two dependent loads per 16 bytes written.)
        const unsigned int uQ = (unsigned int)Q, qgroups = (uQ + TTN_HAD_K - 1) / TTN_HAD_K, items = uP * qgroups;
        typedef double d2v_t __attribute__((ext_vector_type(2)));
        for (unsigned int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
            const unsigned int p = it % uP, qg = it / uP;
            const unsigned int ay = p % (unsigned int)ryl, ax = p / (unsigned int)ryl;
            const unsigned int q0 = qg * TTN_HAD_K;
            unsigned int by = q0 % (unsigned int)ryr, bx = q0 / (unsigned int)ryr;
            d2v_t xv[TTN_HAD_K], yv[TTN_HAD_K];
#pragma unroll
            for (int j = 0; j < TTN_HAD_K; ++j) {
                const bool in = q0 + j < uQ;
                xv[j] = in ? *reinterpret_cast<const d2v_t*>(Xk + 2 * (ax + (long long)rxl * bx)) : (d2v_t){0.0, 0.0};
                yv[j] = in ? *reinterpret_cast<const d2v_t*>(Yk + 2 * (ay + (long long)ryl * by)) : (d2v_t){0.0, 0.0};
                if (++by == (unsigned int)ryr) { by = 0; ++bx; }
            }
#pragma unroll
            for (int j = 0; j < TTN_HAD_K; ++j)
                if (q0 + j < uQ) __builtin_nontemporal_store(xv[j] * yv[j], reinterpret_cast<d2v_t*>(Zk + 2 * ((long long)p + (long long)uP * (q0 + j))));
        }
        return;
    }
    for (unsigned int e = blockIdx.x * blockDim.x + threadIdx.x; e < utotal; e += gridDim.x * blockDim.x) {
        const unsigned int p = e % uP, q = e / uP;
        const unsigned int ay = p % (unsigned int)ryl, ax = p / (unsigned int)ryl, by = q % (unsigned int)ryr, bx = q / (unsigned int)ryr;
        const double* xs = Xk + (long long)n * (ax + (long long)rxl * bx);
        const double* ys = Yk + (long long)n * (ay + (long long)ryl * by);
        double* zo = Zk + (long long)n * e;
        for (int s = 0; s < n; ++s) zo[s] = xs[s] * ys[s];
    }
}

// ---------------------------------------------------------------------------------------------
// +: block concat (src/tt_operations.jl:10-35).  first core [X Y], middle diag(X,Y), last [X;Y].
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TTN_STREAM_TB) k_add(TTDev x, TTDev y, TTDev z) {
    const int k = blockIdx.y, b = blockIdx.z, d = x.d;
    const int n = x.dims[k];
    const long long* xr = x.rks + (long long)b * (d + 1);
    const long long* yr = y.rks + (long long)b * (d + 1);
    const int rxl = (int)xr[k], rxr = (int)xr[k + 1], ryl = (int)yr[k], ryr = (int)yr[k + 1];
    const int zl = (k == 0) ? 1 : rxl + ryl, zr = (k == d - 1) ? 1 : rxr + ryr;
    const long long total = (long long)zl * zr;
    const double* Xk = x.data + (long long)b * x.stride + x.off[k];
    const double* Yk = y.data + (long long)b * y.stride + y.off[k];
    double* Zk = z.data + (long long)b * z.stride + z.off[k];
    const unsigned int utotal = (unsigned int)total;
    if (n == 2) {
        // one thread = one row a and TTN_ADD_K consecutive columns c of the output core: two integer divisions per K fibres, the loads of
        // the K source fibres in flight together, per j the lanes of a wave write consecutive rows (coalesced 16-byte stores, non-temporal)
        typedef double d2v_t __attribute__((ext_vector_type(2)));
        const unsigned int uzl = (unsigned int)zl, uzr = (unsigned int)zr, cgroups = (uzr + TTN_ADD_K - 1) / TTN_ADD_K, items = uzl * cgroups;
        for (unsigned int it = blockIdx.x * blockDim.x + threadIdx.x; it < items; it += gridDim.x * blockDim.x) {
            const int a = (int)(it % uzl), c0 = (int)(it / uzl) * TTN_ADD_K;
            d2v_t v[TTN_ADD_K];
#pragma unroll
            for (int j = 0; j < TTN_ADD_K; ++j) {
                const int c = c0 + j;
                const double* src = nullptr;
                if (c < zr) {
                    if (k == 0) src = (c < rxr) ? Xk + 2LL * ((long long)rxl * c) : Yk + 2LL * ((long long)ryl * (c - rxr));
                    else if (k == d - 1) src = (a < rxl) ? Xk + 2LL * a : Yk + 2LL * (a - rxl);
                    else if (a < rxl && c < rxr) src = Xk + 2LL * (a + (long long)rxl * c);
                    else if (a >= rxl && c >= rxr) src = Yk + 2LL * ((a - rxl) + (long long)ryl * (c - rxr));
                }
                v[j] = src ? *reinterpret_cast<const d2v_t*>(src) : (d2v_t){0.0, 0.0};
            }
#pragma unroll
            for (int j = 0; j < TTN_ADD_K; ++j)
                if (c0 + j < zr) __builtin_nontemporal_store(v[j], reinterpret_cast<d2v_t*>(Zk + 2LL * ((long long)a + (long long)zl * (c0 + j))));
        }
        return;
    }
    for (unsigned int e = blockIdx.x * blockDim.x + threadIdx.x; e < utotal; e += gridDim.x * blockDim.x) {
        const int a = (int)(e % (unsigned int)zl), c = (int)(e / (unsigned int)zl);
        const double* src = nullptr;
        // row block: first core has the single row shared by X and Y; column block likewise at the end
        const bool ax = (k == 0) ? true : (a < rxl);
        const bool cx = (k == d - 1) ? true : (c < rxr);
        const bool ay = (k == 0) ? true : (a >= rxl);
        const bool cy = (k == d - 1) ? true : (c >= rxr);
        if (k == 0) {
            src = (c < rxr) ? Xk + (long long)n * (0 + (long long)rxl * c) : Yk + (long long)n * (0 + (long long)ryl * (c - rxr));
        } else if (k == d - 1) {
            src = (a < rxl) ? Xk + (long long)n * (a + (long long)rxl * 0) : Yk + (long long)n * ((a - rxl) + (long long)ryl * 0);
        } else if (ax && cx) {
            src = Xk + (long long)n * (a + (long long)rxl * c);
        } else if (ay && cy) {
            src = Yk + (long long)n * ((a - rxl) + (long long)ryl * (c - rxr));
        }
        double* zo = Zk + (long long)n * e;
        if (n == 2) {
            double2 o = src ? *reinterpret_cast<const double2*>(src) : double2{0.0, 0.0};
            *reinterpret_cast<double2*>(zo) = o;
        } else
        for (int s = 0; s < n; ++s) zo[s] = src ? src[s] : 0.0;
    }
}

// y = f * x (or x, or zero) over t2 16-byte elements: four loads in flight per thread and trip, non-temporal stores (the output is
// read by a later kernel, not by this one).  One load -> one store per trip left the copy at 4.85 TB/s; k_hadamard's stores reach 5.5.
__device__ __forceinline__ void stream_scale_copy(const double* Xk, double* Yk, long long t2, double f, bool scale_it, bool zero) {
    typedef double d2v_t __attribute__((ext_vector_type(2)));
    const d2v_t* X2 = reinterpret_cast<const d2v_t*>(Xk);
    d2v_t* Y2 = reinterpret_cast<d2v_t*>(Yk);
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const double g = zero ? 0.0 : (scale_it ? f : 1.0);
    for (; e + 3 * stride < t2; e += 4 * stride) {
        d2v_t v0, v1, v2, v3;
        if (zero) { v0 = v1 = v2 = v3 = (d2v_t){0.0, 0.0}; }
        else { v0 = X2[e]; v1 = X2[e + stride]; v2 = X2[e + 2 * stride]; v3 = X2[e + 3 * stride]; }
        if (scale_it && !zero) { v0 *= g; v1 *= g; v2 *= g; v3 *= g; }
        __builtin_nontemporal_store(v0, &Y2[e]);
        __builtin_nontemporal_store(v1, &Y2[e + stride]);
        __builtin_nontemporal_store(v2, &Y2[e + 2 * stride]);
        __builtin_nontemporal_store(v3, &Y2[e + 3 * stride]);
    }
    for (; e < t2; e += stride) {
        d2v_t v = zero ? (d2v_t){0.0, 0.0} : X2[e];
        if (scale_it && !zero) v *= g;
        __builtin_nontemporal_store(v, &Y2[e]);
    }
}

// ---------------------------------------------------------------------------------------------
// scalar *: copy every core, scale core `which` by a (src/tt_operations.jl:256-266); zero==1 -> all-zero train
// ---------------------------------------------------------------------------------------------
// which_b (device, per train) overrides `which` when the trains of the batch carry different gauge flags
__global__ void __launch_bounds__(TTN_STREAM_TB) k_scale(TTDev x, TTDev y, double a, int which, int zero, const int* which_b) {
    const int k = blockIdx.y, b = blockIdx.z;
    if (which_b) which = which_b[b];
    const long long* xr = x.rks + (long long)b * (x.d + 1);
    const long long total = (long long)x.dims[k] * xr[k] * xr[k + 1];
    const double* Xk = x.data + (long long)b * x.stride + x.off[k];
    double* Yk = y.data + (long long)b * y.stride + y.off[k];
    const double f = (k == which) ? a : 1.0;
    // 16 bytes per lane and trip (every slot is 16-byte aligned), a scalar tail for an odd count
    const long long t2 = total >> 1;
    stream_scale_copy(Xk, Yk, t2, f, k == which, zero != 0);
    if ((total & 1) && blockIdx.x == 0 && threadIdx.x == 0) Yk[total - 1] = zero ? 0.0 : ((k == which) ? f * Xk[total - 1] : Xk[total - 1]);
}

// per-train scalar: y_b = a[b] * x_b (a on the device); a[b] == 0 writes the zero train
__global__ void __launch_bounds__(TTN_STREAM_TB) k_scale_batch(TTDev x, TTDev y, const double* a, int which, const int* which_b) {
    const int k = blockIdx.y, b = blockIdx.z;
    if (which_b) which = which_b[b];
    const long long* xr = x.rks + (long long)b * (x.d + 1);
    const long long total = (long long)x.dims[k] * xr[k] * xr[k + 1];
    const double* Xk = x.data + (long long)b * x.stride + x.off[k];
    double* Yk = y.data + (long long)b * y.stride + y.off[k];
    const double f = a[b];
    const long long t2 = total >> 1;
    stream_scale_copy(Xk, Yk, t2, f, k == which, f == 0.0);
    if ((total & 1) && blockIdx.x == 0 && threadIdx.x == 0) Yk[total - 1] = (f == 0.0) ? 0.0 : ((k == which) ? f * Xk[total - 1] : Xk[total - 1]);
}

// replicate train src over the whole batch (cores + ranks)
__global__ void __launch_bounds__(TTN_STREAM_TB) k_replicate(TTDev x, int src) {
    const int b = blockIdx.y;
    if (b == src) return;
    const double* s = x.data + (long long)src * x.stride;
    double* t = x.data + (long long)b * x.stride;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < x.stride; e += (long long)gridDim.x * blockDim.x) t[e] = s[e];
    if (blockIdx.x == 0)
        for (int m = threadIdx.x; m <= x.d; m += blockDim.x) x.rks[(long long)b * (x.d + 1) + m] = x.rks[(long long)src * (x.d + 1) + m];
}
