"""Shared test helpers: oracle <-> product conversions and golden-fixture loading."""
import os

import numpy as np

from oracle import tt_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tt_from_golden(g, prefix):
    dims = tuple(int(v) for v in g[f"{prefix}_dims"])
    rks = [int(v) for v in g[f"{prefix}_rks"]]
    ot = [int(v) for v in g[f"{prefix}_ot"]] if f"{prefix}_ot" in g else [0] * len(dims)
    cores = [np.array(g[f"{prefix}_core{k}"]) for k in range(len(dims))]
    return O.TTvector(len(dims), cores, dims, rks, ot)


def tto_from_golden(g, prefix):
    dims = tuple(int(v) for v in g[f"{prefix}_dims"])
    rks = [int(v) for v in g[f"{prefix}_rks"]]
    cores = [np.array(g[f"{prefix}_core{k}"]) for k in range(len(dims))]
    return O.TToperator(len(dims), cores, dims, rks, [0] * len(dims))


def to_product(x):
    """oracle TTvector/TToperator -> product (ttn_amd) TTvector/TToperator"""
    import ttn_amd as T
    if isinstance(x, O.TTvector):
        return T.TTvector(x.N, [np.asfortranarray(c) for c in x.ttv_vec], x.ttv_dims, x.ttv_rks, x.ttv_ot)
    return T.TToperator(x.N, [np.asfortranarray(c) for c in x.tto_vec], x.tto_dims, x.tto_rks, x.tto_ot)


def to_oracle(x):
    if hasattr(x, "ttv_vec"):
        return O.TTvector(x.N, [np.array(c) for c in x.ttv_vec], tuple(x.ttv_dims), list(x.ttv_rks), list(x.ttv_ot))
    return O.TToperator(x.N, [np.array(c) for c in x.tto_vec], tuple(x.tto_dims), list(x.tto_rks), list(x.tto_ot))


def tt_norm_stable(x):
    """||x|| without the cancellation of dot(x,x) on differences: orthogonalize to site 1 (oracle) and take
    the Frobenius norm of the centre core."""
    y = O.orthogonalize(x, i=1)
    return float(np.linalg.norm(y.ttv_vec[0]))


def tt_rel_diff(a, b):
    """||a - b|| / ||b|| for trains too long to densify (d=30).  The difference is formed as a TT
    (ranks add) and its norm taken after orthogonalization, so the result is accurate far below
    sqrt(eps) (aa - 2ab + bb would cancel catastrophically at ~1e-8)."""
    diff = O.sub(a, b)
    return tt_norm_stable(diff) / tt_norm_stable(b)


def sign_fix_compare(a, b):
    """tt_compress! output cores are unique up to one sign per bond (non-degenerate singular
    values): fix signs bond by bond and return the max relative core difference."""
    worst = 0.0
    carry = None
    for k in range(a.N):
        ca, cb = np.array(a.ttv_vec[k]), np.array(b.ttv_vec[k])
        if carry is not None:
            ca = ca * carry[None, :, None]
        if k < a.N - 1:
            # sign of each right-index slice
            sa = np.sign(np.einsum("iab,iab->b", ca, cb))
            sa[sa == 0] = 1.0
            ca = ca * sa[None, None, :]
            carry = sa
        worst = max(worst, float(np.max(np.abs(ca - cb)) / max(np.max(np.abs(cb)), 1e-300)))
    return worst


# ------------------------------------------------------------------------------------------------------------------
# Extended-precision arbiter for singular values (numpy.longdouble: 64-bit mantissa on x86, eps 1.1e-19).
# Used where two fp64 SVDs (LAPACK gesdd in the oracle, Householder/Gram + Jacobi on the device) of the same
# ill-conditioned matrix disagree at the 1e-12 sigma_1 level: neither can arbitrate, this can.
# ------------------------------------------------------------------------------------------------------------------
def _householder_r_ld(A):
    """R factor (n x n, upper triangular) of the m x n longdouble matrix A, m >= n, by Householder reflections."""
    A = np.array(A, dtype=np.longdouble)
    m, n = A.shape
    for j in range(n):
        x = A[j:, j]
        nx = np.sqrt(np.sum(x * x))
        if nx == 0:
            continue
        alpha = -nx if x[0] >= 0 else nx
        v = x.copy()
        v[0] -= alpha
        nv2 = np.sum(v * v)
        if nv2 == 0:
            continue
        A[j:, j:] -= np.outer(v, (2 / nv2) * (v @ A[j:, j:]))
    return np.triu(A[:n, :])


def ext_svdvals_rows(N, sweeps_max=60):
    """Singular values (descending, longdouble) of the p x q matrix N by one-sided Jacobi on its ROWS in longdouble:
    round-robin ordering, p/2 disjoint row pairs rotated at once (vectorised)."""
    X = np.array(N, dtype=np.longdouble)
    p = X.shape[0]
    if p % 2:
        X = np.vstack([X, np.zeros((1, X.shape[1]), dtype=np.longdouble)])
        p += 1
    eps = np.finfo(np.longdouble).eps
    idx = np.arange(p)
    for _ in range(sweeps_max):
        rotated = False
        order = idx.copy()
        for _round in range(p - 1):
            a, b = order[: p // 2], order[p // 2:][::-1]
            xa, xb = X[a], X[b]
            al, be, ga = np.sum(xa * xa, axis=1), np.sum(xb * xb, axis=1), np.sum(xa * xb, axis=1)
            act = np.abs(ga) > 16 * eps * np.sqrt(al * be)
            act &= (al > 0) & (be > 0)
            if np.any(act):
                rotated = True
                ga_s = np.where(act, ga, 1)
                zeta = (be - al) / (2 * ga_s)
                t = np.sign(zeta + (zeta == 0)) / (np.abs(zeta) + np.sqrt(1 + zeta * zeta))
                c = 1 / np.sqrt(1 + t * t)
                s = c * t
                c = np.where(act, c, 1)[:, None]
                s = np.where(act, s, 0)[:, None]
                X[a], X[b] = c * xa - s * xb, s * xa + c * xb
            order = np.concatenate([order[:1], order[-1:], order[1:-1]])
        if not rotated:
            break
    return np.sort(np.sqrt(np.sum(X * X, axis=1)))[::-1]


def ext_svdvals_product(A, B):
    """Singular values of A @ B (A: m x k with m >= k, B: k x q) to longdouble accuracy: the product is never rounded to fp64 —
    R = qr(A) in longdouble, then Jacobi on the k rows of R @ B."""
    A = np.array(A, dtype=np.longdouble)
    B = np.array(B, dtype=np.longdouble)
    if A.shape[0] < A.shape[1]:
        return ext_svdvals_rows(A @ B) if A.shape[0] <= B.shape[1] else ext_svdvals_rows((A @ B).T)
    R = _householder_r_ld(A)
    return ext_svdvals_rows(R @ B)
