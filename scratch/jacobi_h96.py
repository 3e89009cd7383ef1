import numpy as np, sys, scipy.linalg as sla
sys.path.insert(0,'.')
exec(open('scratch/jacobi_proto.py').read().split("Ms = capture()")[0])
def capture_lr(kwant, d=30, r=64, seed=30):
    x = to_oracle(T.rand_tt((2,)*d, r, seed=seed))
    psi = O.apply(O.Delta(d), x)
    for k in range(1, psi.N):
        Ck, Ck1 = psi.ttv_vec[k-1], psi.ttv_vec[k]
        d1, Dl, _ = Ck.shape; d2, _, Dr = Ck1.shape
        M = np.einsum("sag,tgb->sabt", Ck, Ck1).reshape(d1*Dl, Dr*d2)
        if k == kwant: return M
        O.tt_bond_truncate_(psi, k, max_bond=r)
for kw in (25, 26, 24):
    M = capture_lr(kw); M = M/np.abs(M).max()
    if M.shape[0] > M.shape[1]: M = M.T
    s = np.linalg.svd(M, compute_uv=False); print("k", kw, M.shape, "kappa %.2e" % (s[0]/s[-1]), "s[63]/s[0] %.1e" % (s[min(63,len(s)-1)]/s[0]))
    Q,R = sla.qr(M.T, mode='economic'); L = R.T
    n2,_ = jacobi(L); n3,_ = jacobi(L.T)
    idx = np.argsort(-(L*L).sum(0)); n2s,_ = jacobi(L[:, idx])
    Qp,Rp,P = sla.qr(M.T, mode='economic', pivoting=True); Lp = Rp.T
    n4,_ = jacobi(Lp); n5,_ = jacobi(Lp.T)
    print("  sweeps: L cols %d, L cols presorted %d, L rows %d, Lpiv cols %d, Lpiv rows %d" % (n2,n2s,n3,n4,n5))
