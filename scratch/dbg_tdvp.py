import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import ttn_amd as T
from ttn_amd import tdvp as D
from oracle import tt_oracle as O
import torch; torch.cuda.is_available(); T.ensure_init(0)
torch, stream = D._dev()
rng = np.random.default_rng(1)
c = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
with torch.cuda.stream(stream):
    Dl, d1, d2, Dr, a, b, cc = 3, 2, 2, 4, 3, 2, 3
    AAC, FL, FR, M1, M2 = c(Dl, d1, d2, Dr), c(Dl, a, Dl), c(Dr, cc, Dr), c(a, d1, b, d1), c(b, d2, cc, d2)
    up = lambda x: D._up(x, np.complex128)
    got = D._down(D._d_applyH2(up(AAC), up(FL), up(FR), up(M1), up(M2)))
    ref = O.tdvp_applyH2_lsr(AAC, FL, FR, M1, M2)
    print("H2", np.max(np.abs(got - ref)))
    AC, FRb, M = c(Dl, d1, Dr), c(Dr, b, Dr), c(a, d1, b, d1)
    print("H1", np.max(np.abs(D._down(D._d_applyH1(up(AC), up(FL), up(FRb), up(M))) - O.tdvp_applyH1_lsr(AC, FL, FRb, M))))
    Cm, FRa = c(Dl, Dr), c(Dr, a, Dr)
    print("H0", np.max(np.abs(D._down(D._d_applyH0(up(Cm), up(FL), up(FRa))) - O.tdvp_applyH0(Cm, FL, FRa))))
    print("L", np.max(np.abs(D._down(D._d_left_env(up(AC), up(M), up(FL))) - O.tdvp_update_left_env(AC, M, FL))))
    Mr = c(a, d1, b, d1)
    print("R", np.max(np.abs(D._down(D._d_right_env(up(AC), up(Mr), up(FRb))) - O.tdvp_update_right_env(AC, Mr, FRb))))
    # exponentiate on a dense Hermitian matrix of size 64
    n = 64
    Hm = c(n, n); Hm = (Hm + Hm.conj().T) / 2
    x = c(n)
    Hd = torch.from_numpy(Hm).cuda()
    y = D.exponentiate(lambda v: Hd @ v, -0.05j, torch.from_numpy(x).cuda())
    import scipy.linalg as sla
    yref = sla.expm(-0.05j * Hm) @ x
    print("expm dev", np.max(np.abs(y.cpu().numpy() - yref)), "oracle", np.max(np.abs(O.tdvp_exponentiate(lambda v: Hm @ v, -0.05j, x) - yref)))
    y = D.exponentiate(lambda v: Hd @ v, -2.0j, torch.from_numpy(x).cuda())
    yref = sla.expm(-2.0j * Hm) @ x
    print("expm long dev", np.max(np.abs(y.cpu().numpy() - yref)), "oracle", np.max(np.abs(O.tdvp_exponentiate(lambda v: Hm @ v, -2.0j, x) - yref)))
