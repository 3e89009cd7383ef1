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
