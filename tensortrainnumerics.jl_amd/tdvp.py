"""Host mirror of the TDVP local contractions of the reference (src/solvers/tdvp.jl:29-43, :205-208) on the HIP path.

Same names, argument order and tensor layouts as the reference's helpers — sites in (l, s, r) layout, operator cores in
(a, s, b, s') layout (`_to_lsr`, `_mpo_to_asbs`, tdvp.jl:24-27) — real (Float64) or complex (ComplexF64) NumPy arrays, optionally
with one leading batch axis (every system of the batch is contracted by its own workgroup; an operator core without the batch
axis is shared).  Each call stages its arrays through the device (`ttn_tdvp_contract_f64`); device-resident chains bind the
device-pointer entry points `ttn_tdvp_apply_h1` ... of include/ttn.h directly.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _to_lsr(A):
    """permutedims(A, (2, 1, 3))  (tdvp.jl:24)"""
    return np.transpose(A, (1, 0, 2))


_to_slr = _to_lsr


def _mpo_to_asbs(M):
    """permutedims(M, (3, 1, 4, 2)): (s_out, s_in, a, b) -> (a, s_out, b, s_in)  (tdvp.jl:27)"""
    return np.transpose(M, (2, 0, 3, 1))


def _prep(arrs, nds):
    """Common dtype (float64 / complex128), column-major batches: returns (cplx, batch, [flat arrays or None], [shapes])."""
    cplx = any(a is not None and np.iscomplexobj(a) for a in arrs)
    dt = np.complex128 if cplx else np.float64
    batch = None
    out, shapes = [], []
    for a, nd in zip(arrs, nds):
        if a is None:
            out.append(None); shapes.append(None)
            continue
        a = np.asarray(a, dtype=dt)
        if a.ndim == nd + 1:
            if batch is not None and a.shape[0] != batch:
                raise AssertionError("batch sizes differ")
            batch = a.shape[0]
        elif a.ndim != nd:
            raise AssertionError(f"expected an array with {nd} (or {nd + 1}: leading batch) axes, got shape {a.shape}")
        out.append(a); shapes.append(a.shape[-nd:])
    return cplx, batch, out, shapes


def _flat(a, nd, batch, shared_ok=False):
    """Column-major tensors back to back."""
    if a is None:
        return None, False
    if a.ndim == nd:
        if batch is not None and not shared_ok:
            a = np.broadcast_to(a, (batch,) + a.shape)
        else:
            return np.ascontiguousarray(np.reshape(a, -1, order="F")), True
    return np.ascontiguousarray(np.stack([np.reshape(t, -1, order="F") for t in a])), False


def _contract(op, dims7, FL, FR, X, M1, M2, out_shape, nds):
    _lib.ensure_init()
    cplx, batch, (FL, FR, X, M1, M2), _ = _prep([FL, FR, X, M1, M2], nds)
    B = batch if batch is not None else 1
    fFL, _ = _flat(FL, nds[0], batch)
    fFR, _ = _flat(FR, nds[1], batch)
    fX, _ = _flat(X, nds[2], batch)
    fM1, sh1 = _flat(M1, nds[3], batch, shared_ok=True)
    fM2, sh2 = _flat(M2, nds[4], batch, shared_ok=True)
    m_shared = 1 if ((M1 is None or sh1) and (M2 is None or sh2)) else 0
    if not m_shared:                                         # mixed: expand whichever is shared
        if M1 is not None and sh1:
            fM1, _ = _flat(np.broadcast_to(M1, (B,) + M1.shape), nds[3], batch)
        if M2 is not None and sh2:
            fM2, _ = _flat(np.broadcast_to(M2, (B,) + M2.shape), nds[4], batch)
    n_out = int(np.prod(out_shape))
    out = np.zeros(B * n_out, dtype=np.complex128 if cplx else np.float64)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)      # noqa: E731
    d7 = (C.c_int64 * 7)(*[int(v) for v in dims7])
    _lib.check(_lib.lib().ttn_tdvp_contract_f64(op, 1 if cplx else 0, B, d7, ptr(fFL), ptr(fFR), ptr(fX), ptr(fM1), ptr(fM2), ptr(out), m_shared))
    res = np.stack([np.reshape(out[i * n_out:(i + 1) * n_out], out_shape, order="F") for i in range(B)])
    return res if batch is not None else res[0]


def _applyH1_lsr(AC, FL, FR, M):
    """HAC[α,s,β] = FL[α,a,α'] AC[α',s',β'] M[a,s,b,s'] FR[β',b,β]  (tdvp.jl:29-31)"""
    Dl, d, Dr = np.shape(AC)[-3:]
    a, b = np.shape(M)[-4], np.shape(M)[-2]
    assert np.shape(M)[-3] == d and np.shape(M)[-1] == d and np.shape(FL)[-3:] == (Dl, a, Dl) and np.shape(FR)[-3:] == (Dr, b, Dr)
    return _contract(0, (Dl, d, Dr, a, b, 1, 1), FL, FR, AC, M, None, (Dl, d, Dr), (3, 3, 3, 4, 4))


def _applyH0(C_, FL, FR):
    """HC[α,β] = FL[α,a,α'] C[α',β'] FR[β',a,β]  (tdvp.jl:33-35)"""
    Dl, Dr = np.shape(C_)[-2:]
    a = np.shape(FL)[-2]
    assert np.shape(FL)[-3:] == (Dl, a, Dl) and np.shape(FR)[-3:] == (Dr, a, Dr)
    return _contract(1, (Dl, 1, Dr, a, 1, 1, 1), FL, FR, C_, None, None, (Dl, Dr), (3, 3, 2, 4, 4))


def _update_left_env(A, M, FL):
    """FLnext[α,a,β] = FL[α',a',β'] A[β',s',β] M[a',s,a,s'] conj(A[α',s,α])  (tdvp.jl:37-39)"""
    Dl, d, Dr = np.shape(A)[-3:]
    a_in, a_out = np.shape(M)[-4], np.shape(M)[-2]
    assert np.shape(FL)[-3:] == (Dl, a_in, Dl)
    return _contract(2, (Dl, d, Dr, a_in, a_out, 1, 1), FL, None, A, M, None, (Dr, a_out, Dr), (3, 3, 3, 4, 4))


def _update_right_env(A, M, FR):
    """FRprev[α,a,β] = A[α,s',α'] FR[α',a',β'] M[a,s,a',s'] conj(A[β,s,β'])  (tdvp.jl:41-43)"""
    Dl, d, Dr = np.shape(A)[-3:]
    a_out, a_in = np.shape(M)[-4], np.shape(M)[-2]
    assert np.shape(FR)[-3:] == (Dr, a_in, Dr)
    return _contract(3, (Dl, d, Dr, a_in, a_out, 1, 1), None, FR, A, M, None, (Dl, a_out, Dl), (3, 3, 3, 4, 4))


def _applyH2_lsr(AAC, FL, FR, M1, M2):
    """HAAC[α,s1,s2,β] = FL[α,a,α'] AAC[α',s1',s2',β'] M1[a,s1,b,s1'] M2[b,s2,c,s2'] FR[β',c,β]  (tdvp.jl:205-208)"""
    Dl, d1, d2, Dr = np.shape(AAC)[-4:]
    a, b, c = np.shape(M1)[-4], np.shape(M1)[-2], np.shape(M2)[-2]
    assert np.shape(M2)[-4] == b and np.shape(FL)[-3:] == (Dl, a, Dl) and np.shape(FR)[-3:] == (Dr, c, Dr)
    return _contract(4, (Dl, d1, Dr, a, b, c, d2), FL, FR, AAC, M1, M2, (Dl, d1, d2, Dr), (3, 3, 4, 4, 4))


# ======================================================================================================================
# TDVP drivers on the device (src/solvers/tdvp.jl:45-203, :210-357): tdvp1sweep!, tdvp2sweep!, tdvp, tdvp2
#
# Everything a sweep touches stays in HBM: the site tensors, the operator cores and the environments are device arrays in the
# reference's layouts ((l, s, r), (a, s, b, s'), column-major); the five contractions are this library's kernels, called through the
# DEVICE-POINTER entry points of include/ttn.h; `exponentiate` is a Lanczos iteration around them whose vectors never leave the
# device (its 30 x 30 tridiagonal projection is exponentiated on the host, as KrylovKit does).  torch is plumbing here: it owns
# the device memory, runs on the library's own HIP stream (so kernels and tensor operations are one ordered queue), does the BLAS-1
# work on the Lanczos vectors and the dense QR / SVD of the local matrices on the device (hipSOLVER) — the library's own QR / SVD
# kernels are real-valued and work on TT handles; the local tensors of a real-time sweep are ComplexF64.  No CPU fallback: without
# a GPU and the HIP library these functions raise.
#
# A Julia array of shape (i, j, k) in column-major order is held as a CONTIGUOUS torch tensor of shape (k, j, i).
# ======================================================================================================================
_DEV = {}


def _dev():
    """(torch, library stream as a torch stream); raises when there is no GPU / no HIP library."""
    if not _DEV:
        import torch
        ok = torch.cuda.is_available()            # (asked before the library touches the device: afterwards this build of torch says no)
        _lib.ensure_init()
        if not ok:
            try:
                torch.cuda.init()
                ok = torch.cuda.device_count() > 0
            except Exception:
                ok = False
        if not ok:
            raise _lib.TTNError("TDVP drivers need the GPU (no CPU fallback)")
        h = C.c_void_p()
        _lib.check(_lib.lib().ttn_stream_handle(C.byref(h)))
        _DEV["torch"] = torch
        _DEV["stream"] = torch.cuda.ExternalStream(h.value)
    return _DEV["torch"], _DEV["stream"]


def _up(x, dt):
    """NumPy array in the reference's index order -> device array (column-major = reversed axes, contiguous)."""
    torch, _ = _dev()
    return torch.from_numpy(np.ascontiguousarray(np.transpose(np.asarray(x, dtype=dt)))).to("cuda")


def _down(t):
    return np.transpose(t.cpu().numpy())


def _jshape(t):
    return tuple(reversed(t.shape))


def _cplx(t):
    return 1 if t.is_complex() else 0


def _p(t):
    # the kernels read raw memory: a lazily conjugated view (torch.linalg.svd's Vh, x.conj()) or a strided one must be materialised first
    assert t.is_contiguous() and not t.is_conj(), "device array is a view"
    return C.c_void_p(t.data_ptr())


def _own(t):
    """materialise: conjugation resolved, contiguous memory"""
    return t.resolve_conj().contiguous()


def _d_applyH1(AC, FL, FR, M):
    torch, _ = _dev()
    Dl, d, Dr = _jshape(AC)
    a, _, b, _ = _jshape(M)
    out = torch.empty_like(AC)
    _lib.check(_lib.lib().ttn_tdvp_apply_h1(_cplx(AC), 1, Dl, d, Dr, a, b, _p(FL), _p(AC), _p(M), _p(FR), _p(out), 1))
    return out


def _d_applyH0(Cm, FL, FR):
    torch, _ = _dev()
    Dl, Dr = _jshape(Cm)
    a = _jshape(FL)[1]
    out = torch.empty_like(Cm)
    _lib.check(_lib.lib().ttn_tdvp_apply_h0(_cplx(Cm), 1, Dl, Dr, a, _p(FL), _p(Cm), _p(FR), _p(out)))
    return out


def _d_left_env(A, M, FL):
    torch, _ = _dev()
    Dl, d, Dr = _jshape(A)
    a_in, _, a_out, _ = _jshape(M)
    out = torch.empty((Dr, a_out, Dr), dtype=A.dtype, device=A.device)
    _lib.check(_lib.lib().ttn_tdvp_update_left_env(_cplx(A), 1, Dl, d, Dr, a_in, a_out, _p(A), _p(M), _p(FL), _p(out), 1))
    return out


def _d_right_env(A, M, FR):
    torch, _ = _dev()
    Dl, d, Dr = _jshape(A)
    a_out, _, a_in, _ = _jshape(M)
    out = torch.empty((Dl, a_out, Dl), dtype=A.dtype, device=A.device)
    _lib.check(_lib.lib().ttn_tdvp_update_right_env(_cplx(A), 1, Dl, d, Dr, a_out, a_in, _p(A), _p(M), _p(FR), _p(out), 1))
    return out


def _d_applyH2(AAC, FL, FR, M1, M2):
    torch, _ = _dev()
    Dl, d1, d2, Dr = _jshape(AAC)
    a, _, b, _ = _jshape(M1)
    c = _jshape(M2)[2]
    out = torch.empty_like(AAC)
    _lib.check(_lib.lib().ttn_tdvp_apply_h2(_cplx(AAC), 1, Dl, d1, d2, Dr, a, b, c, _p(FL), _p(AAC), _p(M1), _p(M2), _p(FR), _p(out), 1))
    return out


def _qr_j(Mt):
    """Thin QR of a Julia matrix (m x n, column-major) held as the contiguous torch tensor Mt of shape (n, m): returns (Qt, Rt), the
    column-major Q (m x r) and R (r x n) as torch tensors of shapes (r, m) and (n, r) — csrc/ttn_densefact_kernels.h (Householder,
    LAPACK's conventions), real or complex."""
    torch, _ = _dev()
    n, m = Mt.shape
    r = min(m, n)
    W = _own(Mt).clone()
    Qt = torch.empty((r, m), dtype=Mt.dtype, device=Mt.device)
    Rt = torch.empty((n, r), dtype=Mt.dtype, device=Mt.device)
    _lib.check(_lib.lib().ttn_dense_qr(_cplx(W), m, n, _p(W), _p(Qt), _p(Rt)))
    return Qt, Rt


def _svd_j(Mt):
    """Thin SVD X = U diag(s) Vt of a Julia matrix X (m x n) held as the torch tensor Mt of shape (n, m): returns (Ut, s, Vtt) with
    U (m x k) as a tensor of shape (k, m), Vt (k x n) as a tensor of shape (n, k), k = min(m, n), s on the HOST (descending) —
    one-sided Jacobi (csrc/ttn_densefact_kernels.h), on the matrix or on its conjugate transpose, whichever has fewer columns."""
    torch, _ = _dev()
    n, m = Mt.shape
    real_dt = torch.float64
    if m >= n:
        W = _own(Mt).clone()
        Ut = torch.empty((n, m), dtype=Mt.dtype, device=Mt.device)
        Vtt = torch.empty((n, n), dtype=Mt.dtype, device=Mt.device)
        sd = torch.empty((n,), dtype=real_dt, device=Mt.device)
        _lib.check(_lib.lib().ttn_dense_svd(_cplx(W), m, n, _p(W), _p(Ut), _p(sd), _p(Vtt)))
        return Ut, sd, Vtt
    W = _own(Mt.transpose(0, 1).conj()).clone()                   # X^H (n x m) as a tensor of shape (m, n)
    U2 = torch.empty((m, n), dtype=Mt.dtype, device=Mt.device)      # U' (n x m)
    V2 = torch.empty((m, m), dtype=Mt.dtype, device=Mt.device)      # V'h (m x m)
    sd = torch.empty((m,), dtype=real_dt, device=Mt.device)
    _lib.check(_lib.lib().ttn_dense_svd(_cplx(W), n, m, _p(W), _p(U2), _p(sd), _p(V2)))
    return _own(V2.transpose(0, 1).conj()), sd, _own(U2.transpose(0, 1).conj())      # U = V'h^H, Vt = U'^H


def exponentiate(Hfun, t, x, krylovdim: int = 30, tol: float = 1.0e-12, maxiter: int = 100):
    """y = exp(t H) x for a Hermitian H given as a function on device arrays of x's shape — KrylovKit.exponentiate(H, t, x;
    ishermitian = true) with its defaults (tdvp.jl:68, :88 ...): Lanczos with full reorthogonalisation up to `krylovdim` vectors, the
    tridiagonal projection exponentiated on the host, accepted when beta_m |e_m^T exp(tau T_m) e_1| <= tol, shorter time steps and
    restarts otherwise.  One host read (alpha_j, beta_j) per Lanczos step; the vectors stay on the device."""
    torch, _ = _dev()
    shape = x.shape
    if isinstance(t, complex) and not x.is_complex():
        x = x.to(torch.complex128)
    v = x.reshape(-1).clone()
    n = v.numel()
    remaining, total = t, abs(t)
    for _ in range(maxiter):
        nrm = float(torch.linalg.vector_norm(v))
        if nrm == 0.0 or remaining == 0:
            break
        m_max = min(krylovdim, n)
        V = torch.empty((m_max, n), dtype=v.dtype, device=v.device)
        V[0] = v / nrm
        alphas, betas = [], []
        m, happy = 0, False
        for j in range(m_max):
            w = Hfun(V[j].reshape(shape)).reshape(-1)
            a = torch.vdot(V[j], w).real
            w = w - a * V[j]
            if j > 0:
                w = w - betas[j - 1] * V[j - 1]
            for _r in range(2):                                               # full reorthogonalisation, twice
                w = w - (V[: j + 1].conj() @ w) @ V[: j + 1]
            ab = torch.stack([a, torch.linalg.vector_norm(w)]).tolist()       # the step's one host read
            alphas.append(ab[0])
            m = j + 1
            if ab[1] <= 1.0e-14 * max(1.0, abs(ab[0])) or m == n:
                happy = True
                break
            betas.append(ab[1])
            if j + 1 < m_max:
                V[j + 1] = w / ab[1]
        Tm = np.diag(alphas[:m]) + np.diag(betas[: m - 1], 1) + np.diag(betas[: m - 1], -1)
        lam, U = np.linalg.eigh(Tm)
        tau = remaining
        while True:
            y = U @ (np.exp(tau * lam) * U[0, :])
            err = 0.0 if (happy or len(betas) < m) else betas[m - 1] * abs(y[m - 1]) * nrm
            if happy or err <= tol * max(abs(tau) / total, 1.0e-3) or abs(tau) <= 1.0e-12 * total:
                break
            tau = tau / 2
        yd = torch.from_numpy(np.asarray(y, dtype=np.complex128 if v.is_complex() else np.float64)).to(v.device)
        if np.iscomplexobj(y) and not v.is_complex():
            raise AssertionError("complex time on a real vector")
        v = nrm * (yd @ V[:m])
        remaining = remaining - tau
        if abs(remaining) <= 1.0e-14 * total:
            break
    return v.reshape(shape)


def _real_or_complex_t(z):
    """a complex number without imaginary part becomes real  (tdvp.jl:22)"""
    z = complex(z)
    return z.real if z.imag == 0.0 else z


def _svd_rank(s, max_bond, truncerr):
    """the rule of _svdtrunc as the reference's tdvp.jl resolves it (src/tt_cross_interpolation.jl:149-166: relative tail norm)"""
    r = len(s)
    if truncerr > 0:
        nrm = float(np.linalg.norm(s))
        cum = 0.0
        for i in range(r, 0, -1):
            cum += float(s[i - 1]) ** 2
            if np.sqrt(cum) > truncerr * nrm:
                r = i
                break
    return min(r, int(max_bond))


class _State:
    """One train on the device: sites (l, s, r), operator cores (a, s, b, s'), environments F[0 .. N+1]."""

    def __init__(self, psi, H, dt_is_complex):
        torch, _ = _dev()
        cplx = dt_is_complex or any(np.iscomplexobj(c) for c in psi.ttv_vec) or any(np.iscomplexobj(c) for c in H.tto_vec)
        self.dt = np.complex128 if cplx else np.float64
        self.N = psi.N
        self.dims = tuple(psi.ttv_dims)
        self.A = [_up(np.transpose(np.asarray(c), (1, 0, 2)), self.dt) for c in psi.ttv_vec]           # permutedims(ttv_vec[k], (2, 1, 3))  (:52)
        self.M = [_up(np.transpose(np.asarray(c), (2, 0, 3, 1)), self.dt) for c in H.tto_vec]         # permutedims(tto_vec[k], (3, 1, 4, 2))  (:53)
        self.F = None

    def build_envs(self):
        torch, _ = _dev()
        N = self.N
        one = torch.ones((1, 1, 1), dtype=self.A[0].dtype, device="cuda")
        F = [None] * (N + 2)
        F[0], F[N + 1] = one, one.clone()
        for k in range(N - 1, -1, -1):
            F[k + 1] = _d_right_env(self.A[k], self.M[k], F[k + 2])                                       # (:58-60)
        self.F = F


def _sweep1(S: _State, dt, **kw):
    """tdvp1sweep! on a device state (tdvp.jl:64-145)."""
    torch, _ = _dev()
    N, A, M, F = S.N, S.A, S.M, S.F
    tm, tp = _real_or_complex_t(-1j * complex(dt)), _real_or_complex_t(+1j * complex(dt))
    AC = A[0]
    for k in range(N - 1):
        AC = exponentiate(lambda x: _d_applyH1(x, F[k], F[k + 2], M[k]), tm, AC, **kw)
        Dl, d, Dr = _jshape(AC)
        Qt, Rt = _qr_j(AC.reshape(Dr, d * Dl))                                               # Aqr = reshape(AC, Dl d, Dr)
        r = min(Dl * d, Dr)
        AL = Qt.reshape(r, d, Dl)                                                            # reshape(Qthin, Dl, d, r)
        A[k] = AL
        F[k + 1] = _d_left_env(AL, M[k], F[k])
        Cm = Rt                                                                              # C = Rthin (r x Dr), stored (Dr, r)
        Cm = exponentiate(lambda x: _d_applyH0(x, F[k + 1], F[k + 2]), tp, Cm, **kw)
        AC = _own(torch.matmul(A[k + 1], Cm))                                         # AC[α,s,β] = C[α,γ] A_{k+1}[γ,s,β]
    k = N - 1
    AC = exponentiate(lambda x: _d_applyH1(x, F[k], F[k + 2], M[k]), tm, AC, **kw)
    for k in range(N - 2, -1, -1):
        Dl, d, Dr = _jshape(AC)
        Qt, Rt = _qr_j(_own(AC.reshape(Dr * d, Dl).transpose(0, 1).conj()))                  # qr(A'), A = reshape(AC, Dl, d Dr)
        r = min(Dl, d * Dr)
        A_r = _own(Qt.transpose(0, 1).conj()).reshape(Dr, d, r)                              # reshape(Qthin', r, d, Dr)
        A[k + 1] = A_r
        F[k + 2] = _d_right_env(A_r, M[k + 1], F[k + 3])
        Lm = _own(Rt.transpose(0, 1).conj())                                                 # L = Rthin' (Dl x r), stored (r, Dl)
        Lm = exponentiate(lambda x: _d_applyH0(x, F[k + 1], F[k + 2]), tp, Lm, **kw)
        AC = _own(torch.tensordot(Lm, A[k], dims=([1], [0])))                         # AC[α,s,β] = A_k[α,s,γ] C[γ,β]
        AC = exponentiate(lambda x: _d_applyH1(x, F[k], F[k + 2], M[k]), tm, AC, **kw)
    A[0] = AC


def _sweep2(S: _State, dt, max_bond=2 ** 62, truncerr=0.0, **kw):
    """tdvp2sweep! on a device state (tdvp.jl:236-294)."""
    torch, _ = _dev()
    N, A, M, F = S.N, S.A, S.M, S.F
    dth = complex(dt) / 2
    tm, tp = _real_or_complex_t(-1j * dth), _real_or_complex_t(+1j * dth)
    AC = A[0]

    def split(AAC):
        Dl, d1, d2, Dr = _jshape(AAC)
        Ut, sd, Vtt = _svd_j(AAC.reshape(Dr * d2, d1 * Dl))                                 # X = reshape(AAC, Dl d1, d2 Dr) = U S Vt
        r = _svd_rank(sd.tolist(), max_bond, truncerr)
        return (Dl, d1, d2, Dr), r, Vtt[:, :r], sd[:r].to(AAC.dtype), Ut[:r, :]              # (Vt^T: (d2 Dr) x r; U^T: r x (Dl d1))

    for k in range(N - 1):
        AAC = _own(torch.tensordot(A[k + 1], AC, dims=([2], [0])))                    # AAC[α,s1,s2,β] = AC[α,s1,γ] A_{k+1}[γ,s2,β]
        AAC = exponentiate(lambda x: _d_applyH2(x, F[k], F[k + 3], M[k], M[k + 1]), tm, AAC, **kw)
        (Dl, d1, d2, Dr), r, U2, s, V2h = split(AAC)
        AL = _own(V2h).reshape(r, d1, Dl)                                             # reshape(U, Dl, d1, r): U = V2h^T
        A[k] = AL
        F[k + 1] = _d_left_env(AL, M[k], F[k])
        AC = _own(U2 * s[None, :]).reshape(Dr, d2, r)                               # reshape(S Vt, r, d2, Dr): Vt = U2^T
        if k < N - 2:
            AC = exponentiate(lambda x: _d_applyH1(x, F[k + 1], F[k + 3], M[k + 1]), tp, AC, **kw)
    for k in range(N - 2, -1, -1):
        AAC = _own(torch.tensordot(AC, A[k], dims=([2], [0])))                        # AAC[α,s1,s2,β] = A_k[α,s1,γ] AC[γ,s2,β]
        AAC = exponentiate(lambda x: _d_applyH2(x, F[k], F[k + 3], M[k], M[k + 1]), tm, AAC, **kw)
        (Dl, d1, d2, Dr), r, U2, s, V2h = split(AAC)
        AR = _own(U2).reshape(Dr, d2, r)                                              # reshape(Vt, r, d2, Dr)
        A[k + 1] = AR
        F[k + 2] = _d_right_env(AR, M[k + 1], F[k + 3])
        AC = _own(s[:, None] * V2h).reshape(r, d1, Dl)                              # reshape(U S, Dl, d1, r)
        if k > 0:
            AC = exponentiate(lambda x: _d_applyH1(x, F[k], F[k + 2], M[k]), tp, AC, **kw)
    A[0] = AC


def _state_to_host(S: _State, psi, force_real=False):
    """sites back to (s, l, r), ranks from the arrays, ttv_ot zeroed  (_sync_ranks_from_lsr!, tdvp.jl:8-18, :147-151)"""
    cores = []
    for k in range(S.N):
        c = np.transpose(_down(S.A[k]), (1, 0, 2))
        if force_real:
            assert float(np.max(np.abs(np.imag(c)))) <= 1e-12 * max(1.0, float(np.max(np.abs(c)))), "a real train picked up an imaginary part"
            c = np.real(c)
        cores.append(np.asfortranarray(c))
    psi.ttv_vec = cores
    psi.ttv_rks = [int(_jshape(S.A[k])[0]) for k in range(S.N)] + [int(_jshape(S.A[S.N - 1])[2])]
    psi.ttv_ot = [0] * S.N
    return psi


def _envs_in(S: _State, F):
    torch, _ = _dev()
    if F is None:
        S.build_envs()
    else:
        S.F = [f.to(S.A[0].dtype) if torch.is_tensor(f) else _up(f, S.dt) for f in F]         # F[i] = Tc.(F[i])  (:63-66)


def tdvp1sweep_(dt, psi, H, F=None, **kw):
    """tdvp1sweep!(dt, ψ, H, F = nothing; kwargs...) (tdvp.jl:45-152) on the device; mutates ψ (a host TTvector, real or complex cores)
    and returns (ψ, F) with F the list of N + 2 environments as DEVICE arrays (pass it back in to carry them; `envs_to_host` reads
    them in the reference's index order)."""
    torch, stream = _dev()
    with torch.cuda.stream(stream):
        S = _State(psi, H, isinstance(dt, complex))
        _envs_in(S, F)
        _sweep1(S, dt, **kw)
        return _state_to_host(S, psi), S.F


def tdvp2sweep_(dt, psi, H, F=None, max_bond=2 ** 62, truncerr=0.0, **kw):
    """tdvp2sweep!(dt, ψ, H, F = nothing; max_bond, truncerr, kwargs...) (tdvp.jl:210-301) on the device."""
    torch, stream = _dev()
    with torch.cuda.stream(stream):
        S = _State(psi, H, isinstance(dt, complex))
        _envs_in(S, F)
        _sweep2(S, dt, max_bond=max_bond, truncerr=truncerr, **kw)
        return _state_to_host(S, psi), S.F


def envs_to_host(F):
    return [_down(f) for f in F]


def _orthogonalize_state(S: _State):
    """orthogonalize(ψ) (centre 1: src/tt_tools.jl:528-541) on the device arrays of a state; returns the norm of ψ (the norm of
    the centre core of the orthogonalized train)."""
    torch, _ = _dev()
    A = S.A
    for k in range(S.N - 1, 0, -1):
        Dl, d, Dr = _jshape(A[k])
        Qt, Rt = _qr_j(_own(A[k].reshape(Dr * d, Dl).transpose(0, 1).conj()))                  # LQ of reshape(A_k, Dl, d Dr) = QR of its adjoint
        r = Qt.shape[0]
        A[k] = _own(Qt.transpose(0, 1).conj()).reshape(Dr, d, r)
        A[k - 1] = _own(torch.tensordot(_own(Rt.transpose(0, 1).conj()), A[k - 1], dims=([1], [0])))
    return float(torch.linalg.vector_norm(A[0]))


def _tt_dot(X, Y):
    """dot(x, y) of two trains given as device site lists (conjugates the first: src/tt_operations.jl:239-250)"""
    torch, _ = _dev()
    Mx = torch.ones((1, 1), dtype=torch.result_type(X[0], Y[0]), device="cuda")
    for Xr, Yr in zip(X, Y):
        Mx = torch.einsum("asx,bsy,xy->ab", Xr.conj().to(Mx.dtype), Yr.to(Mx.dtype), Mx)
    return complex(Mx[0, 0])


def _tt_apply(M, X):
    """H ψ as a site list (ranks multiply; src/tt_operations.jl:101-111)"""
    torch, _ = _dev()
    out = []
    for Mr, Xr in zip(M, X):                                                   # Mr (s', b, s, a), Xr (r, s', l)
        Y = torch.einsum("tbsa,rtl->rbsla", Mr, Xr.to(Mr.dtype))
        out.append(Y.reshape(Y.shape[0] * Y.shape[1], Y.shape[2], Y.shape[3] * Y.shape[4]).contiguous())
    return out


def _driver(sweep, H, u0, steps, normalize=True, return_error=False, sweeps=1, carry_env=True, verbose=False, imaginary_time=False, **kw):
    """tdvp / tdvp2 (tdvp.jl:154-203, :303-357) on the device: the train stays in HBM across all steps and sweeps."""
    torch, stream = _dev()
    from .tt import TTvector
    with torch.cuda.stream(stream):
        real_out = imaginary_time and not any(np.iscomplexobj(c) for c in u0.ttv_vec)
        S = _State(u0, H, True)                                   # dt_eff is complex in both time directions (:176, :326)
        _orthogonalize_state(S)
        prev = [a.clone() for a in S.A]
        for h in steps:
            prev = [a.clone() for a in S.A]
            dt_eff = (1j * h) if imaginary_time else complex(h)
            for _ in range(sweeps):
                if not carry_env or S.F is None:
                    S.build_envs()
                sweep(S, dt_eff, **kw)
            nrm = _orthogonalize_state(S)
            if normalize:
                S.A[0] = S.A[0] / nrm
            S.F = None
        psi = _state_to_host(S, TTvector(u0.N, [None] * u0.N, u0.ttv_dims, list(u0.ttv_rks), [0] * u0.N), force_real=real_out)
        psi.ttv_ot = [0] + [-1] * (psi.N - 1)                     # orthogonalize(ψ) closes every step (:185, :340): centre 1
        if not return_error:
            return psi
        h = steps[-1]
        c3 = -1.0 if imaginary_time else 1j                       # residual = (ψ - ψ_prev) / h - Hψ   or   + im Hψ  (:191-196)
        vecs = [S.A, prev, _tt_apply(S.M, S.A)]
        coef = [1.0 / h, -1.0 / h, c3]
        G = [[_tt_dot(vecs[i], vecs[j]) for j in range(3)] for i in range(3)]
        rr = sum(np.conj(coef[i]) * coef[j] * G[i][j] for i in range(3) for j in range(3))
        return psi, float(np.sqrt(max(rr.real, 0.0)) / np.sqrt(max(G[0][0].real, 0.0)))


def tdvp(H, u0, steps, **kw):
    """tdvp(H, u₀, steps; normalize, return_error, sweeps, carry_env, verbose, imaginary_time, kwargs...)  (tdvp.jl:154-203)"""
    return _driver(_sweep1, H, u0, steps, **kw)


def tdvp2(H, u0, steps, max_bond=2 ** 62, truncerr=0.0, **kw):
    """tdvp2(H, u₀, steps; ..., max_bond, truncerr, ...)  (tdvp.jl:303-357)"""
    return _driver(lambda S, dt, **k2: _sweep2(S, dt, max_bond=max_bond, truncerr=truncerr, **k2), H, u0, steps, **kw)
