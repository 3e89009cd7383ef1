"""Feasibility study (CPU, NumPy): could the 128x128 Jacobi SVD of the Gram route be replaced by
tridiagonalisation + bisection + inverse iteration WITHOUT reorthogonalisation?  Measures, on merged matrices of the
benchmark workload, the orthogonality of the kept eigenvectors and the a-posteriori quantity the Gram route checks."""
import sys
import numpy as np
import scipy.linalg as sla
sys.path.insert(0, ".")
from oracle import tt_oracle as O

d, r = 16, 64
rng = np.random.default_rng(30)
x = O.rand_tt((2,) * d, r, rng)
y = O.apply(O.Delta(d), x)
Ms = []
orig = O.svdtrunc
def spy(A, max_bond=2 ** 62, truncerr=0.0):
    if A.shape == (128, 384) or A.shape == (384, 128):
        Ms.append(A.copy() if A.shape[0] == 128 else A.T.copy())
    return orig(A, max_bond=max_bond, truncerr=truncerr)
O.svdtrunc = spy
O.tt_compress_(y, r)
print(len(Ms), "merged 128x384 matrices")

def inv_iter(dg, e, lam, iters=3):
    n = len(dg)
    ab = np.zeros((3, n))
    ab[0, 1:] = e; ab[2, :-1] = e
    rng = np.random.default_rng(1)
    z = rng.standard_normal(n)
    for _ in range(iters):
        ab[1] = dg - lam
        z = sla.solve_banded((1, 1), ab, z)
        z /= np.linalg.norm(z)
    return z

for M in Ms[:6]:
    s0 = np.max(np.abs(M))
    G = (M / s0) @ (M / s0).T
    sv = np.linalg.svd(M / s0, compute_uv=False)
    # Householder tridiagonalisation (LAPACK sytrd through hessenberg of a symmetric matrix)
    H, Q = sla.hessenberg(G, calc_q=True)
    dg, e = np.diag(H).copy(), np.diag(H, -1).copy()
    lam = sla.eigvalsh_tridiagonal(dg, e, lapack_driver="stebz")[::-1]          # bisection, descending
    k = 64
    # perturb the shift by a few ulps so that T - lam I is not exactly singular
    Z = np.stack([inv_iter(dg, e, l * (1 + 4e-16)) for l in lam[:k]], axis=1)
    U = Q @ Z
    orth = np.max(np.abs(U.T @ U - np.eye(k)))
    D = U.T @ G @ U
    chk = np.max(np.abs(D - np.diag(lam[:k])) / np.sqrt(np.outer(lam[:k], lam[:k])))
    relgap = np.min((lam[:k - 1] - lam[1:k]) / lam[:k - 1])
    print(f"kappa {sv[0] / sv[-1]:.1f}  sigma err {np.max(np.abs(np.sqrt(lam) - sv) / sv):.1e}  min rel gap(kept) {relgap:.1e}  |U'U-I| {orth:.1e}  check {chk:.1e}")
