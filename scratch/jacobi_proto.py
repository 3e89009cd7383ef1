import numpy as np, sys, scipy.linalg as sla
sys.path.insert(0,'.')
from oracle import tt_oracle as O
import ttn_amd as T
from tests.helpers import to_oracle

def capture(d=30, r=64, seed=30):
    x = to_oracle(T.rand_tt((2,)*d, r, seed=seed))
    y = O.apply(O.Delta(d), x)
    Ms = {}
    psi = y
    step = 0
    for k in range(1, psi.N):
        Ck, Ck1 = psi.ttv_vec[k-1], psi.ttv_vec[k]
        d1, Dl, _ = Ck.shape; d2, _, Dr = Ck1.shape
        M = np.einsum("sag,tgb->sabt", Ck, Ck1).reshape(d1*Dl, Dr*d2)
        if k in (12,): Ms['LR%d'%k] = M.copy()
        O.tt_bond_truncate_(psi, k, max_bond=r)
    for k in range(psi.N-1, 0, -1):
        Ck, Ck1 = psi.ttv_vec[k-1], psi.ttv_vec[k]
        d1, Dl, _ = Ck.shape; d2, _, Dr = Ck1.shape
        M = np.einsum("sag,tgb->sabt", Ck, Ck1).reshape(d1*Dl, Dr*d2)
        if k in (15,): Ms['RL%d'%k] = M.copy(); Ms['RLfac%d'%k] = (Ck.copy(), Ck1.copy())
        O.tt_bond_truncate_(psi, k, max_bond=r)
    return Ms

def rr_pairs(pe, rnd):
    prs = [(rnd, pe-1)]
    for kk in range(1, pe//2):
        i = (rnd+kk) % (pe-1); j = (rnd+pe-1-kk) % (pe-1)
        prs.append((min(i,j), max(i,j)))
    return prs

def jacobi(X, order='rr', derijk=False, tolm=1.0, maxsw=60, negm=1.0):
    X = X.copy(); m, p = X.shape
    eps = np.finfo(float).eps
    tol = tolm*np.sqrt(m)*eps
    amax = (X*X).sum(0).max(); aneg = negm**2*m*eps*eps*amax
    pe = p + (p & 1)
    for sw in range(maxsw):
        rot = 0
        for rnd in range(pe-1):
            prs = [(i,j) for (i,j) in rr_pairs(pe, rnd) if j < p]
            I = np.array([q[0] for q in prs]); J = np.array([q[1] for q in prs])
            xi, xj = X[:,I], X[:,J]
            a = (xi*xi).sum(0); b = (xj*xj).sum(0); g = (xi*xj).sum(0)
            if derijk:
                # swap so that the larger norm is in the lower index (i)
                sw_mask = b > a
            act = (a > aneg) & (b > aneg) & (np.abs(g) > tol*np.sqrt(a)*np.sqrt(b))
            rot += act.sum()
            gs = np.where(act, g, 1.0)
            zeta = (b-a)/(2*gs)
            t = np.copysign(1.0, zeta)/(np.abs(zeta)+np.sqrt(1+zeta*zeta))
            c = 1/np.sqrt(1+t*t); s = c*t
            c = np.where(act, c, 1.0); s = np.where(act, s, 0.0)
            ni = c*xi - s*xj; nj = s*xi + c*xj
            if derijk:
                a2 = (ni*ni).sum(0); b2 = (nj*nj).sum(0)
                swm = b2 > a2
                ni2 = np.where(swm, nj, ni); nj2 = np.where(swm, ni, nj)
                ni, nj = ni2, nj2
            X[:,I] = ni; X[:,J] = nj
        if rot == 0:
            return sw+1, X
    return -maxsw, X

Ms = capture()
for name in ('LR12','RL15'):
    M = Ms[name]; M = M/np.abs(M).max()
    p, q = M.shape
    print(name, M.shape, 'rank', np.linalg.matrix_rank(M))
    if q > p:
        Q, R = sla.qr(M.T, mode='economic'); L = R.T
        Qp, Rp, P = sla.qr(M.T, mode='economic', pivoting=True); Lp = Rp.T  # row-pivoted LQ of M
    else:
        L = M; Lp = None
    sref = np.linalg.svd(M, compute_uv=False)
    for label, X in (('L cols', L), ('L rows', L.T), ('Lpiv cols', Lp), ('Lpiv rows', None if Lp is None else Lp.T), ('M rows direct', M.T if q>p else None)):
        if X is None: continue
        for dr in (False, True):
            # initial sort by norm descending
            for presort in (False, True):
                Y = X
                if presort:
                    idx = np.argsort(-(Y*Y).sum(0)); Y = Y[:, idx]
                n, Z = jacobi(Y, derijk=dr)
                s = np.sort(np.sqrt((Z*Z).sum(0)))[::-1]
                err = np.max(np.abs(s[:len(sref)]-sref)/sref[0])
                print(f"  {label:14s} derijk={dr} presort={presort}: sweeps {n}  sv err {err:.1e}")
# factored approach for RL: QR(A_mat) LQ(B_mat) -> 64x64 core
Ck, Ck1 = Ms['RLfac15']
d1, Dl, rm = Ck.shape; d2, _, Dr = Ck1.shape
Am = Ck.transpose(0,1,2).reshape(d1*Dl, rm, order='C')  # rows (s, alpha)?? any row order is fine for sweeps count
Bm = Ck1.transpose(1,2,0).reshape(rm, Dr*d2)
Qa, Ra = sla.qr(Am, mode='economic'); Qb, Rb = sla.qr(Bm.T, mode='economic')
Cc = Ra @ Rb.T
print('factored core', Cc.shape)
for dr in (False, True):
    n, Z = jacobi(Cc/np.abs(Cc).max(), derijk=dr)
    print('  core cols derijk', dr, 'sweeps', n)
    n, Z = jacobi((Cc/np.abs(Cc).max()).T, derijk=dr)
    print('  core rows derijk', dr, 'sweeps', n)

print("---- ill-conditioned square case (R->L ramp: step 34, kappa ~1e8) ----")
import scipy.linalg as sla
def capture_rl(kwant, d=30, r=64, seed=30):
    x = to_oracle(T.rand_tt((2,)*d, r, seed=seed))
    psi = O.apply(O.Delta(d), x)
    for k in range(1, psi.N): O.tt_bond_truncate_(psi, k, max_bond=r)
    for k in range(psi.N-1, 0, -1):
        Ck, Ck1 = psi.ttv_vec[k-1], psi.ttv_vec[k]
        d1, Dl, _ = Ck.shape; d2, _, Dr = Ck1.shape
        M = np.einsum("sag,tgb->sabt", Ck, Ck1).reshape(d1*Dl, Dr*d2)
        if k == kwant: return M
        O.tt_bond_truncate_(psi, k, max_bond=r)
for kw in (6, 5):
    M = capture_rl(kw); M = M/np.abs(M).max()
    s = np.linalg.svd(M, compute_uv=False); print("k", kw, M.shape, "kappa %.2e" % (s[0]/s[-1]))
    n0,_ = jacobi(M); n1,_ = jacobi(M.T)
    Q,R = sla.qr(M.T, mode='economic'); L = R.T
    n2,_ = jacobi(L); n3,_ = jacobi(L.T)
    Qp,Rp,P = sla.qr(M.T, mode='economic', pivoting=True); Lp = Rp.T
    n4,_ = jacobi(Lp); n5,_ = jacobi(Lp.T)
    Q2,R2 = sla.qr(Lp, mode='economic')   # second QR: Lp = Q2 R2 ; Jacobi on R2^T columns
    n6,_ = jacobi(R2.T); n7,_ = jacobi(R2)
    print("  sweeps: M cols %d, M rows %d, L cols %d, L rows %d, Lpiv cols %d, Lpiv rows %d, R2^T cols %d, R2 cols %d" % (n0,n1,n2,n3,n4,n5,n6,n7))
