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
def dense(As):
    t = As[0]
    for a in As[1:]:
        t = np.tensordot(t, a, axes=([-1], [0]))
    return t
with torch.cuda.stream(stream):
    S = D._State(to_product(psi), to_product(H), True); S.build_envs()
    A = [np.transpose(c, (1, 0, 2)).astype(complex) for c in psi.ttv_vec]
    M = [np.transpose(c, (2, 0, 3, 1)) for c in H.tto_vec]
    F = O._tdvp_envs(A, M, np.complex128)
    N = d; dth = dt / 2
    tm, tp = D._real_or_complex_t(-1j * dth), D._real_or_complex_t(+1j * dth)
    AC = A[0]; ACd = S.A[0]
    for k in range(N - 1):
        AAC = np.einsum("asg,gtb->astb", AC, A[k + 1])
        AAC = O.tdvp_exponentiate(lambda x: O.tdvp_applyH2_lsr(x, F[k], F[k + 3], M[k], M[k + 1]), tm, AAC)
        Dl, d1, d2, Dr = AAC.shape
        U, s, Vt = O.svdtrunc(np.reshape(AAC, (Dl * d1, d2 * Dr), order="F"), truncerr=1e-12)
        A[k] = np.reshape(U, (Dl, d1, U.shape[1]), order="F")
        F[k + 1] = O.tdvp_update_left_env(A[k], M[k], F[k])
        AC = np.reshape(s[:, None] * Vt, (len(s), d2, Dr), order="F")
        AACd = torch.tensordot(S.A[k + 1], ACd, dims=([2], [0])).contiguous()
        AACd = D.exponentiate(lambda x: D._d_applyH2(x, S.F[k], S.F[k + 3], S.M[k], S.M[k + 1]), tm, AACd)
        print(k, "AAC after exp", np.max(np.abs(D._down(AACd) - AAC)))
        U2, sd, V2h = torch.linalg.svd(AACd.reshape(Dr * d2, d1 * Dl), full_matrices=False)
        rr = len(s)
        S.A[k] = V2h[:rr].contiguous().reshape(rr, d1, Dl)
        S.F[k + 1] = D._d_left_env(S.A[k], S.M[k], S.F[k])
        ACd = (U2[:, :rr] * sd[:rr].to(AACd.dtype)[None, :]).contiguous().reshape(Dr, d2, rr)
        if k < N - 2:
            AC = O.tdvp_exponentiate(lambda x: O.tdvp_applyH1_lsr(x, F[k + 1], F[k + 3], M[k + 1]), tp, AC)
            ACd = D.exponentiate(lambda x: D._d_applyH1(x, S.F[k + 1], S.F[k + 3], S.M[k + 1]), tp, ACd)
        full = dense(A[:k + 1] + [AC] + A[k + 2:])
        fulld = dense([D._down(a) for a in S.A[:k + 1]] + [D._down(ACd)] + [D._down(a) for a in S.A[k + 2:]])
        print(k, "state", np.linalg.norm(full - fulld) / np.linalg.norm(full))
    # re-run step 0 and check the device's left environment and back step in the device's own gauge
    S = D._State(to_product(psi), to_product(H), True); S.build_envs()
    A = [np.transpose(c, (1, 0, 2)).astype(complex) for c in psi.ttv_vec]
    F = O._tdvp_envs(A, M, np.complex128)
    k = 0
    AACd = torch.tensordot(S.A[1], S.A[0], dims=([2], [0])).contiguous()
    AACd = D.exponentiate(lambda x: D._d_applyH2(x, S.F[0], S.F[3], S.M[0], S.M[1]), tm, AACd)
    Dl, d1, d2, Dr = D._jshape(AACd)
    U2, sd, V2h = torch.linalg.svd(AACd.reshape(Dr * d2, d1 * Dl), full_matrices=False)
    rr = 2
    ALd = V2h[:rr].contiguous().reshape(rr, d1, Dl)
    FLd = D._d_left_env(ALd, S.M[0], S.F[0])
    AL_h = D._down(ALd)
    print("AL orthonormal", np.max(np.abs(AL_h.reshape(Dl * d1, rr, order="F").conj().T @ AL_h.reshape(Dl * d1, rr, order="F") - np.eye(rr))))
    FL_ref = O.tdvp_update_left_env(AL_h, M[0], F[0])
    print("left env dev vs oracle on the same AL", np.max(np.abs(D._down(FLd) - FL_ref)), D._jshape(FLd), FL_ref.shape, "M0", M[0].shape)
    ACd = (U2[:, :rr] * sd[:rr].to(AACd.dtype)[None, :]).contiguous().reshape(Dr, d2, rr)
    AC_h = D._down(ACd)
    e_ref = O.tdvp_exponentiate(lambda x: O.tdvp_applyH1_lsr(x, FL_ref, F[3], M[1]), tp, AC_h)
    e_dev = D.exponentiate(lambda x: D._d_applyH1(x, FLd, S.F[3], S.M[1]), tp, ACd)
    print("back step", np.max(np.abs(D._down(e_dev) - e_ref)))
    h_ref = O.tdvp_applyH1_lsr(AC_h, FL_ref, F[3], M[1]); h_dev = D._down(D._d_applyH1(ACd, FLd, S.F[3], S.M[1]))
    print("H1 apply", np.max(np.abs(h_dev - h_ref)), AC_h.shape, ACd.is_contiguous(), FLd.is_contiguous())
