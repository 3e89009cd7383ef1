"""Feasibility (CPU): eigenvectors of the tridiagonal by ONE twisted factorisation per eigenvalue (Fernando / dlar1v style,
no pivoting, no inverse-iteration loop, no reorthogonalisation), eigenvalues from bisection.  Same measurements as
eig_feasibility.py."""
import sys
import numpy as np
import scipy.linalg as sla
sys.path.insert(0, ".")
from oracle import tt_oracle as O

d, r = int(sys.argv[1]) if len(sys.argv) > 1 else 18, 64
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

def twisted(dg, e, lam):
    n = len(dg)
    piv = np.finfo(float).tiny * max(1.0, np.max(e * e))
    Dp = np.empty(n); Dm = np.empty(n)
    q = dg[0] - lam
    if abs(q) < piv: q = -piv
    Dp[0] = q
    for i in range(1, n):
        q = (dg[i] - lam) - e[i - 1] ** 2 / q
        if abs(q) < piv: q = -piv
        Dp[i] = q
    q = dg[n - 1] - lam
    if abs(q) < piv: q = -piv
    Dm[n - 1] = q
    for i in range(n - 2, -1, -1):
        q = (dg[i] - lam) - e[i] ** 2 / q
        if abs(q) < piv: q = -piv
        Dm[i] = q
    gam = Dp + Dm - (dg - lam)
    k = int(np.argmin(np.abs(gam)))
    z = np.zeros(n); z[k] = 1.0
    for i in range(k - 1, -1, -1):
        z[i] = -(e[i] / Dp[i]) * z[i + 1]
    for i in range(k, n - 1):
        z[i + 1] = -(e[i] / Dm[i + 1]) * z[i]
    return z / np.linalg.norm(z)

for M in Ms:
    s0 = np.max(np.abs(M))
    G = (M / s0) @ (M / s0).T
    sv = np.linalg.svd(M / s0, compute_uv=False)
    if sv[0] / sv[-1] > 128: continue
    H, Q = sla.hessenberg(G, calc_q=True)
    dg, e = np.diag(H).copy(), np.diag(H, -1).copy()
    lam = sla.eigvalsh_tridiagonal(dg, e, lapack_driver="stebz")[::-1]
    k = 64
    Z = np.stack([twisted(dg, e, l) for l in lam[:k]], axis=1)
    U = Q @ Z
    orth = np.max(np.abs(U.T @ U - np.eye(k)))
    D = U.T @ G @ U
    chk = np.max(np.abs(D - np.diag(lam[:k])) / np.sqrt(np.outer(lam[:k], lam[:k])))
    relgap = np.min((lam[:k - 1] - lam[1:k]) / lam[:k - 1])
    print(f"kappa {sv[0] / sv[-1]:.1f}  min rel gap(kept) {relgap:.1e}  |U'U-I| {orth:.1e}  check {chk:.1e}")
