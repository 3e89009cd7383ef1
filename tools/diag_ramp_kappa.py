"""Diagnostic (not a test; CPU only): conditioning of the merged matrices of the rank-ramp bond steps of the benchmark sweep and how
many one-sided Jacobi sweeps the triangular factor needs under different preconditioning (which route those steps could take)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tt_oracle as O
import ttn_amd as T
from tests.helpers import to_oracle


def bench_train(d, r, seed):               # the benchmark's seeded generator
    return to_oracle(T.rand_tt((2,) * d, r, seed=seed))


def jacobi_sweeps(X, tol=1e-15, maxsw=30):
    """cyclic one-sided Jacobi on the columns of X (copy); returns sweeps until no rotation above tol"""
    X = X.copy(); n = X.shape[1]
    for sw in range(1, maxsw + 1):
        rot = 0
        for i in range(n - 1):
            for j in range(i + 1, n):
                a, b, g = X[:, i] @ X[:, i], X[:, j] @ X[:, j], X[:, i] @ X[:, j]
                if abs(g) <= tol * np.sqrt(a * b) or a == 0 or b == 0: continue
                rot += 1
                z = (b - a) / (2 * g); t = np.sign(z) / (abs(z) + np.sqrt(1 + z * z)) if z != 0 else 1.0
                c = 1 / np.sqrt(1 + t * t); s = c * t
                xi = X[:, i].copy(); X[:, i] = c * xi - s * X[:, j]; X[:, j] = s * xi + c * X[:, j]
        if rot == 0: return sw
    return maxsw


def analyse(M, tag):
    p, q = M.shape
    if p > q: M = M.T; p, q = q, p
    s = np.linalg.svd(M, compute_uv=False)
    rn = np.linalg.norm(M, axis=1)
    ss = np.linalg.svd(M / rn[:, None], compute_uv=False)
    # Gram + Cholesky, rows in order of decreasing norm
    order = np.argsort(-rn)
    G = (M @ M.T)[np.ix_(order, order)]
    try:
        L = np.linalg.cholesky(G)
        sl = np.sort(np.linalg.svd(L, compute_uv=False))[::-1]
        err = np.max(np.abs(sl - s) / s[0]); rel = np.max(np.abs(sl - s) / s)
    except np.linalg.LinAlgError:
        L = None; err = rel = float("nan")
    Lh = np.linalg.qr(M.T)[1].T                                    # Householder LQ
    co = np.argsort(-np.linalg.norm(Lh, axis=0))
    Lp = np.linalg.qr(M[order].T)[1].T                             # rows presorted, then LQ
    out = f"{tag}: {p}x{q} kappa {s[0]/s[-1]:.2e} row-scaled kappa {ss[0]/ss[-1]:.2e} | chol(G): sv err/s1 {err:.1e} rel {rel:.1e}"
    if p <= 64 and os.environ.get("SWEEPS", "1") == "1":
        out += f" | sweeps: LQ+colsort {jacobi_sweeps(Lh[:, co])} rowsort+LQ {jacobi_sweeps(Lp)}"
        if L is not None: out += f" rowsort+chol {jacobi_sweeps(L)}  L^T {jacobi_sweeps(np.ascontiguousarray(L.T))}"
    print(out, flush=True)


d, r = 30, 64
for seed in [int(a) for a in sys.argv[1:]] or [30, 31]:
    A = O.Delta(d); x = bench_train(d, r, seed)
    y = O.apply(A, x)
    for k in range(1, d):
        Ck, Ck1 = y.ttv_vec[k - 1], y.ttv_vec[k]
        M = np.einsum("sag,tgb->sabt", Ck, Ck1).reshape(Ck.shape[0] * Ck.shape[1], -1)
        if min(M.shape) <= 96 and min(M.shape) >= 16: analyse(M, f"seed {seed} L->R step {k-1}")
        O.tt_bond_truncate_(y, k, max_bond=r)
    for k in range(d - 1, 0, -1):
        Ck, Ck1 = y.ttv_vec[k - 1], y.ttv_vec[k]
        M = np.einsum("sag,tgb->sabt", Ck, Ck1).reshape(Ck.shape[0] * Ck.shape[1], -1)
        if min(M.shape) <= 64 and min(M.shape) >= 16 and (k >= d - 6 or k <= 6): analyse(M, f"seed {seed} R->L bond {k}")
        O.tt_bond_truncate_(y, k, max_bond=r)
