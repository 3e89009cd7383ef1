import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch; torch.cuda.is_available()
import ttn_amd as T
from ttn_amd import tdvp as D
from oracle import tt_oracle as O
from helpers import to_product
T.ensure_init(0)
torch, stream = D._dev()
d, r, dt = 5, 3, 0.05 + 0j
H = O._tdvp_complex_op(O.tto_scale(0.3, O.Delta(d)))
rng = np.random.default_rng(205)
x = O.rand_tt((2,) * d, r, rng); y = O.rand_tt((2,) * d, r, rng)
z = O.TTvector(d, [a + 1j * b for a, b in zip(x.ttv_vec, y.ttv_vec)], x.ttv_dims, list(x.ttv_rks), [0] * d)
psi = O.scale(1.0 / O.norm(z), O.orthogonalize(z))
with torch.cuda.stream(stream):
    S = D._State(to_product(psi), to_product(H), True)
    S.build_envs()
    A = [np.transpose(c, (1, 0, 2)) for c in psi.ttv_vec]
    M = [np.transpose(c, (2, 0, 3, 1)) for c in H.tto_vec]
    F = O._tdvp_envs(A, M, np.complex128)
    for k in range(d + 2):
        print("F", k, np.max(np.abs(D._down(S.F[k]) - F[k])))
    # first two-site step by hand
    N = d
    dth = dt / 2
    tm = D._real_or_complex_t(-1j * dth)
    AC = A[0].astype(complex)
    AAC = np.einsum("asg,gtb->astb", AC, A[1])
    AACd = torch.tensordot(S.A[1], S.A[0], dims=([2], [0])).contiguous()
    print("AAC", np.max(np.abs(D._down(AACd) - AAC)))
    E = O.tdvp_exponentiate(lambda x: O.tdvp_applyH2_lsr(x, F[0], F[3], M[0], M[1]), tm, AAC)
    Ed = D.exponentiate(lambda x: D._d_applyH2(x, S.F[0], S.F[3], S.M[0], S.M[1]), tm, AACd)
    print("exp", np.max(np.abs(D._down(Ed) - E)))
    Dl, d1, d2, Dr = E.shape
    U, s, Vt = O.svdtrunc(np.reshape(E, (Dl * d1, d2 * Dr), order="F"), truncerr=1e-12)
    U2, sd, V2h = torch.linalg.svd(Ed.reshape(Dr * d2, d1 * Dl), full_matrices=False)
    print("s", s, sd.tolist())
    rr = len(s)
    AL = np.reshape(U, (Dl, d1, rr), order="F")
    ALd = V2h[:rr].contiguous().reshape(rr, d1, Dl)
    # gauge-invariant: projector U U^H
    Ud = D._down(ALd).reshape(Dl * d1, rr, order="F")
    print("proj", np.max(np.abs(Ud @ Ud.conj().T - U @ U.conj().T)))
    ACn = np.reshape(s[:, None] * Vt, (rr, d2, Dr), order="F")
    ACd = (U2[:, :rr] * sd[:rr].to(Ed.dtype)[None, :]).contiguous().reshape(Dr, d2, rr)
    rec = np.einsum("asg,gtb->astb", AL, ACn)
    recd = np.einsum("asg,gtb->astb", D._down(ALd), D._down(ACd))
    print("recon", np.max(np.abs(rec - E)), np.max(np.abs(recd - E)))
