"""CPU oracle for the TT/QTT core-arithmetic hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy/SciPy-LAPACK *restatement* of the reference algorithms of
TensorTrainNumerics.jl v1.1.3 (Julia).  It exists so the HIP path can be checked;
it is never part of the shipped product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity pinning: Julia is not installed in the build container (nothing was denied;
the runtime is simply absent), so this oracle cannot be compared with the reference
run live.  It is pinned instead against every closed-form / known-answer test the
reference's own ``test/`` directory holds for this path (see ``tests/test_oracle_*``
and SURVEY.md §8c).  Quantities no reference test pins (values of ``tt_compress!``
on incompressible input, LAPACK sign/gauge choices, the ``truncerr>0`` rank rule
beyond two loose accuracy checks) are "parity unpinned by the reference" and are
pinned only against this restatement.

Conventions: a vector core is an ndarray of shape ``(n, r_left, r_right)`` indexed
``[i, a, b]`` exactly like the reference's ``Array{T,3}`` (physical index first); an
operator core is ``(n_out, n_in, R_left, R_right)``.  Site numbers in the public
functions are 1-based like the reference (``orthogonalize(x; i)``,
``_tt_bond_truncate!(psi, k)``).  All citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np
import scipy.linalg as sla

__all__ = [
    "TTvector", "TToperator", "r_and_d_to_rks", "zeros_tt", "zeros_tto", "rand_tt",
    "toeplitz_to_qtto", "Delta", "shift", "id_tto", "rand_tto", "qtt_sin", "qtt_cos", "qtt_exp",
    "qtt_polynom", "qtt_to_vector", "ttv_to_tensor", "tto_to_tensor", "qtto_to_matrix",
    "apply", "dot", "norm", "euclidean_distance", "hadamard", "add", "add_", "scale",
    "sub", "div", "orthogonalize", "svdtrunc", "svdtrunc_shadowed", "tt_bond_truncate_",
    "tt_compress_", "copy_tt", "rk4_method", "euler_method",
    "ttm_swap_", "ttm_contract_", "hadamard_ttm", "swap_adjacent_sites", "bubble_sort_swaps", "reorder_perm",
    "swap_sites_", "reorder", "swap_adjacent_sites_op", "reorder_op", "ttv_decomp",
    "als_linsolve", "tto_add", "tto_scale", "mals_linsolve", "sv_trunc",
]


# --------------------------------------------------------------------------------------
# Types — src/tt_tools.jl:23-29 (TTvector), :48-54 (TToperator)
# --------------------------------------------------------------------------------------
@dataclass
class TTvector:
    N: int
    ttv_vec: List[np.ndarray]
    ttv_dims: Tuple[int, ...]
    ttv_rks: List[int]
    ttv_ot: List[int]


@dataclass
class TToperator:
    N: int
    tto_vec: List[np.ndarray]
    tto_dims: Tuple[int, ...]
    tto_rks: List[int]
    tto_ot: List[int]


def copy_tt(x: TTvector) -> TTvector:
    """Base.copy — src/tt_tools.jl:172-178."""
    return TTvector(x.N, [c.copy() for c in x.ttv_vec], tuple(x.ttv_dims), list(x.ttv_rks), list(x.ttv_ot))


def _prod(xs) -> int:
    p = 1
    for v in xs:
        p *= int(v)
    return p


# --------------------------------------------------------------------------------------
# r_and_d_to_rks — src/tt_tools.jl:407-425 (loop over eachindex(dims) only; odd zero-dim
# branches pinned by test/test_tt_tools.jl:945-946)
# --------------------------------------------------------------------------------------
def r_and_d_to_rks(rks: Sequence[int], dims: Sequence[int], rmax: int = 1024) -> List[int]:
    new_rks = [1] * len(rks)
    for i in range(len(dims)):  # Julia i = i+1
        q = _prod(dims[i:])       # prod(dims[i:end])
        p = _prod(dims[:i])       # prod(dims[1:i-1])  (empty product = 1)
        if q > 0:
            if p > 0:
                new_rks[i] = min(int(rks[i]), p, q, rmax)
            else:
                new_rks[i] = min(int(rks[i]), q, rmax)
        else:
            if p > 0:
                new_rks[i] = min(int(rks[i]), p, rmax)
            else:
                new_rks[i] = min(int(rks[i]), rmax)
    return new_rks


# --------------------------------------------------------------------------------------
# zeros_tt / zeros_tto — src/tt_operators.jl:548-573, :601-616
# --------------------------------------------------------------------------------------
def zeros_tt(dims: Sequence[int], rks: Sequence[int], ot=None) -> TTvector:
    assert len(dims) + 1 == len(rks), "Dimensions and ranks are not compatible"
    d = len(dims)
    vec = [np.zeros((dims[i], rks[i], rks[i + 1])) for i in range(d)]
    return TTvector(d, vec, tuple(int(v) for v in dims), [int(r) for r in rks],
                    [0] * d if ot is None else [int(o) for o in ot])


def zeros_tt_ndr(n: int, d: int, r: int, r_and_d: bool = True) -> TTvector:
    """zeros_tt(n::Integer, d::Integer, r; r_and_d=true) — src/tt_operators.jl:560-569."""
    dims = (n,) * d
    if r_and_d:
        rks = r_and_d_to_rks([r] * (d + 1), dims)
    else:
        rks = [r] * (d + 1)
        rks[0], rks[-1] = 1, 1
    return zeros_tt(dims, rks)


def zeros_tto(dims: Sequence[int], rks: Sequence[int]) -> TToperator:
    assert len(dims) + 1 == len(rks), "Dimensions and ranks are not compatible"
    d = len(dims)
    vec = [np.zeros((dims[i], dims[i], rks[i], rks[i + 1])) for i in range(d)]
    return TToperator(d, vec, tuple(int(v) for v in dims), [int(r) for r in rks], [0] * d)


def zeros_tto_ndr(n: int, d: int, r: int) -> TToperator:
    """zeros_tto(n, d, r) — src/tt_operators.jl:611-616 (ranks capped with dims.^2, rmax=r)."""
    dims = (n,) * d
    rks = r_and_d_to_rks([r] * (d + 1), [v * v for v in dims], rmax=r)
    return zeros_tto(dims, rks)


# --------------------------------------------------------------------------------------
# rand_tt — src/tt_tools.jl:100-139.  The reference draws ``randn``; the oracle takes the
# standard-normal source as an argument so tests can hand identical bits to the HIP path.
# --------------------------------------------------------------------------------------
def rand_tt(dims: Sequence[int], rks, rng: np.random.Generator) -> TTvector:
    d = len(dims)
    if isinstance(rks, (int, np.integer)):
        rmax = int(rks)
        rks = r_and_d_to_rks([rmax] * (d + 1), dims, rmax=rmax)  # :134-139
    y = zeros_tt(dims, rks)
    for i in range(d):
        y.ttv_vec[i] = rng.standard_normal((dims[i], rks[i], rks[i + 1]))  # :122
    return y


def rand_tto(dims: Sequence[int], rmax: int, rng: np.random.Generator) -> TToperator:
    """rand_tto — src/tt_operators.jl:534-545."""
    d = len(dims)
    rks = [1] * (d + 1)
    vec = []
    for i in range(d):
        ri = min(_prod(dims[:i]), _prod(dims[i:]), rmax)
        rip = min(_prod(dims[: i + 1]), _prod(dims[i + 1:]), rmax)
        rks[i + 1] = rip
        vec.append(rng.standard_normal((dims[i], dims[i], ri, rip)))
    return TToperator(d, vec, tuple(dims), rks, [0] * d)


# --------------------------------------------------------------------------------------
# Operator constructors — src/tt_operators.jl:4-19 (toeplitz_to_qtto), :24 (shift),
# :283-285 (Δ), :519-532 (id_tto)
# --------------------------------------------------------------------------------------
def toeplitz_to_qtto(alpha: float, beta: float, gamma: float, d: int) -> TToperator:
    out = zeros_tto_ndr(2, d, 3)
    Id = np.eye(2)
    J = np.zeros((2, 2))
    J[0, 1] = 1.0
    for i in range(2):
        for j in range(2):
            out.tto_vec[0][i, j, 0, :] = [Id[i, j], J[j, i], J[i, j]]
            for k in range(1, d - 1):
                out.tto_vec[k][i, j, :, :] = [[Id[i, j], J[j, i], J[i, j]],
                                              [0.0, J[i, j], 0.0],
                                              [0.0, 0.0, J[j, i]]]
            out.tto_vec[d - 1][i, j, :, 0] = [alpha * Id[i, j] + beta * J[i, j] + gamma * J[j, i],
                                              gamma * J[i, j],
                                              beta * J[j, i]]
    return out


def Delta(d: int) -> TToperator:
    return toeplitz_to_qtto(2, -1, -1, d)


def shift(d: int) -> TToperator:
    return toeplitz_to_qtto(0, 1, 0, d)


def id_tto(d: int, n_dim: int = 2) -> TToperator:
    # note: cores are always 2x2 regardless of n_dim (src/tt_operators.jl:528-529)
    dims = (n_dim,) * d
    vec = [np.eye(2).reshape(2, 2, 1, 1).copy() for _ in range(d)]
    return TToperator(d, vec, dims, [1] * (d + 1), [0] * d)


# --------------------------------------------------------------------------------------
# Analytic QTT vectors — src/qtt_tools.jl:88-110 (polynom), :116-132 (cos), :138-154 (sin),
# :159-175 (exp)
# --------------------------------------------------------------------------------------
def qtt_sin(d: int, a: float = 0.0, b: float = 1.0, lam: float = 1.0) -> TTvector:
    out = zeros_tt_ndr(2, d, 2)
    h = (b - a) / (2 ** d - 1)
    t1 = a
    out.ttv_vec[0][0, 0, :] = [math.sin(lam * math.pi * t1), math.cos(lam * math.pi * t1)]
    t1 = a + h * 2 ** (d - 1)
    out.ttv_vec[0][1, 0, :] = [math.sin(lam * math.pi * t1), math.cos(lam * math.pi * t1)]
    for k in range(2, d):  # Julia k = 2:(d-1)
        out.ttv_vec[k - 1][0, :, :] = [[1.0, 0.0], [0.0, 1.0]]
        tk = h * 2 ** (d - k)
        c, s = math.cos(lam * math.pi * tk), math.sin(lam * math.pi * tk)
        out.ttv_vec[k - 1][1, :, :] = [[c, -s], [s, c]]
    out.ttv_vec[d - 1][0, 0, 0] = 1.0
    td = h
    out.ttv_vec[d - 1][1, :, 0] = [math.cos(lam * math.pi * td), math.sin(lam * math.pi * td)]
    return out


def qtt_cos(d: int, a: float = 0.0, b: float = 1.0, lam: float = 1.0) -> TTvector:
    out = zeros_tt_ndr(2, d, 2)
    h = (b - a) / (2 ** d - 1)
    t1 = a
    out.ttv_vec[0][0, 0, :] = [math.cos(lam * math.pi * t1), -math.sin(lam * math.pi * t1)]
    t1 = a + h * 2 ** (d - 1)
    out.ttv_vec[0][1, 0, :] = [math.cos(lam * math.pi * t1), -math.sin(lam * math.pi * t1)]
    for k in range(2, d):
        out.ttv_vec[k - 1][0, :, :] = [[1.0, 0.0], [0.0, 1.0]]
        tk = h * 2 ** (d - k)
        c, s = math.cos(lam * math.pi * tk), math.sin(lam * math.pi * tk)
        out.ttv_vec[k - 1][1, :, :] = [[c, -s], [s, c]]
    out.ttv_vec[d - 1][0, 0, 0] = 1.0
    td = h
    out.ttv_vec[d - 1][1, :, 0] = [math.cos(lam * math.pi * td), math.sin(lam * math.pi * td)]
    return out


def qtt_exp(d: int, a: float = 0.0, b: float = 1.0, alpha: float = 1.0, beta: float = 0.0) -> TTvector:
    out = zeros_tt_ndr(2, d, 1)
    h = (b - a) / (2 ** d - 1)
    out.ttv_vec[0][0, 0, 0] = math.exp(alpha * a + beta)
    out.ttv_vec[0][1, 0, 0] = math.exp(alpha * (a + h * 2 ** (d - 1)) + beta)
    for k in range(2, d):
        tk = h * 2 ** (d - k)
        out.ttv_vec[k - 1][0, 0, 0] = 1.0
        out.ttv_vec[k - 1][1, 0, 0] = math.exp(alpha * tk)
    out.ttv_vec[d - 1][0, 0, 0] = 1.0
    out.ttv_vec[d - 1][1, 0, 0] = math.exp(alpha * h)
    return out


def qtt_polynom(coef: Sequence[float], d: int, a: float = 0.0, b: float = 1.0) -> TTvector:
    p = len(coef)
    h = (b - a) / (2 ** d - 1)
    out = zeros_tt_ndr(2, d, p, r_and_d=False)

    def phi(x, s):
        return sum(coef[k] * x ** (k - s) * math.comb(k, s) for k in range(s, p))

    t1 = a
    out.ttv_vec[0][0, 0, :] = [phi(t1, k) for k in range(p)]
    t1 = a + h * 2 ** (d - 1)
    out.ttv_vec[0][1, 0, :] = [phi(t1, k) for k in range(p)]
    for k in range(2, d):
        tk = h * 2 ** (d - k)
        for j in range(p):
            out.ttv_vec[k - 1][0, j, j] = 1.0
            for i in range(p):
                # Julia: binomial(i, i-j) is 0 for i<j
                out.ttv_vec[k - 1][1, i, j] = (math.comb(i, i - j) * tk ** (i - j)) if i >= j else 0.0
    out.ttv_vec[d - 1][0, 0, 0] = 1.0
    td = h
    out.ttv_vec[d - 1][1, :, 0] = [td ** k for k in range(p)]
    return out


# --------------------------------------------------------------------------------------
# Densifiers used by the reference tests as their oracle
#   qtt_to_vector — src/qtt_tools.jl:57-71; ttv_to_tensor — src/tt_tools.jl:265-279;
#   tto_to_tensor — src/tt_tools.jl:374-392; qtto_to_matrix — src/qtt_tools.jl:180-188
# --------------------------------------------------------------------------------------
def qtt_to_vector(qtt: TTvector) -> np.ndarray:
    P = qtt.ttv_vec[0][:, 0, :]
    for k in range(1, qtt.N):
        G = qtt.ttv_vec[k]
        n_prev = P.shape[0]
        P_new = np.empty((2 * n_prev, G.shape[2]), dtype=P.dtype)
        P_new[0::2, :] = P @ G[0, :, :]
        P_new[1::2, :] = P @ G[1, :, :]
        P = P_new
    return P.reshape(-1, order="F")


def ttv_to_tensor(x: TTvector) -> np.ndarray:
    # full tensor T[t1,...,td] = prod_k X_k[t_k,:,:]
    cur = x.ttv_vec[0][:, 0, :]  # (n1, r1)
    shape = [x.ttv_dims[0]]
    for k in range(1, x.N):
        G = x.ttv_vec[k]
        cur = np.einsum("pa,iab->pib", cur, G).reshape(-1, G.shape[2])
        shape.append(x.ttv_dims[k])
    return cur[:, 0].reshape(shape)


def tto_to_tensor(A: TToperator) -> np.ndarray:
    d = A.N
    cur = A.tto_vec[0][:, :, 0, :]  # (n, n, R1)
    cur = cur.reshape(A.tto_dims[0], A.tto_dims[0], -1)
    # build as [x1..xk, y1..yk, R]
    T = cur[:, :, :]
    xs = [A.tto_dims[0]]
    out = T  # shape (x1, y1, R)
    for k in range(1, d):
        G = A.tto_vec[k]  # (n,n,Rl,Rr)
        out = np.tensordot(out, G, axes=([-1], [2]))  # (..., i, j, Rr)
        xs.append(A.tto_dims[k])
    # out has axes (x1,y1,x2,y2,...,xd,yd,Rd); drop last
    out = out[..., 0]
    perm = list(range(0, 2 * d, 2)) + list(range(1, 2 * d, 2))
    return np.transpose(out, perm)


def qtto_to_matrix(A: TToperator) -> np.ndarray:
    d = A.N
    T = tto_to_tensor(A)
    # tuple_to_index: site 1 is the most significant bit -> C-order flatten
    return T.reshape(2 ** d, 2 ** d)


# --------------------------------------------------------------------------------------
# tto * ttv — src/tt_operations.jl:101-111
# --------------------------------------------------------------------------------------
def apply(A: TToperator, v: TTvector) -> TTvector:
    assert tuple(A.tto_dims) == tuple(v.ttv_dims), "Incompatible dimensions"
    rks = [a * b for a, b in zip(A.tto_rks, v.ttv_rks)]
    y = zeros_tt(A.tto_dims, rks)
    for k in range(v.N):
        Ak, Xk = A.tto_vec[k], v.ttv_vec[k]
        # Y[i, a' + Rl*nu', a + Rr*nu] (operator index fastest, from the reshape at :106)
        T = np.einsum("ijab,jcd->icadb", Ak, Xk)
        y.ttv_vec[k] = T.reshape(Ak.shape[0], Xk.shape[1] * Ak.shape[2], Xk.shape[2] * Ak.shape[3])
    return y


# --------------------------------------------------------------------------------------
# dot / norm / distances — src/tt_operations.jl:239-250, :452-470
# --------------------------------------------------------------------------------------
def dot(A: TTvector, B: TTvector) -> float:
    assert tuple(A.ttv_dims) == tuple(B.ttv_dims), "TT dimensions are not compatible"
    M = np.ones((1, 1))
    for k in range(A.N):
        # M'[a,b] = sum_{z,al,be} conj(A[z,al,a]) * (B[z,be,b] * M[al,be])
        T = np.einsum("pq,zqb->zpb", M, B.ttv_vec[k])
        M = np.einsum("zpa,zpb->ab", np.conj(A.ttv_vec[k]), T)
    return M[0, 0]


def norm(a: TTvector) -> float:
    s = dot(a, a)
    v = float(np.real(s))
    v = 0.0 if v < 0 else v
    return math.sqrt(v)


def euclidean_distance(a: TTvector, b: TTvector) -> float:
    assert tuple(a.ttv_dims) == tuple(b.ttv_dims), "TT dimensions must match"
    return math.sqrt(max(float(np.real(dot(a, a) - 2 * np.real(dot(b, a)) + dot(b, b))), 0.0))


# --------------------------------------------------------------------------------------
# hadamard — src/tt_operations.jl:343-361
# --------------------------------------------------------------------------------------
def hadamard(x: TTvector, y: TTvector) -> TTvector:
    assert tuple(x.ttv_dims) == tuple(y.ttv_dims), "Incompatible TT dimensions"
    d = x.N
    rks = [x.ttv_rks[k] * y.ttv_rks[k] for k in range(d + 1)]
    vec = []
    for k in range(d):
        n = x.ttv_dims[k]
        core = np.zeros((n, rks[k], rks[k + 1]))
        for s in range(n):
            core[s, :, :] = np.kron(x.ttv_vec[k][s, :, :], y.ttv_vec[k][s, :, :])
        vec.append(core)
    return TTvector(d, vec, tuple(x.ttv_dims), rks, [0] * d)


# --------------------------------------------------------------------------------------
# + / add! / scalar * / - / /  — src/tt_operations.jl:10-66, :256-266, :283-295
# --------------------------------------------------------------------------------------
def add(x: TTvector, y: TTvector) -> TTvector:
    assert tuple(x.ttv_dims) == tuple(y.ttv_dims), "Incompatible dimensions"
    d = x.N
    rks = [a + b for a, b in zip(x.ttv_rks, y.ttv_rks)]
    rks[0] = 1
    rks[d] = 1
    dt = np.result_type(*[c.dtype for c in x.ttv_vec], *[c.dtype for c in y.ttv_vec])       # (complex trains: the TDVP drivers)
    vec = [np.zeros((x.ttv_dims[k], rks[k], rks[k + 1]), dtype=dt) for k in range(d)]
    rx = x.ttv_rks
    vec[0][:, :, : rx[1]] = x.ttv_vec[0]
    vec[0][:, :, rx[1]: rks[1]] = y.ttv_vec[0]
    for k in range(1, d - 1):
        vec[k][:, : rx[k], : rx[k + 1]] = x.ttv_vec[k]
        vec[k][:, rx[k]: rks[k], rx[k + 1]: rks[k + 1]] = y.ttv_vec[k]
    vec[d - 1][:, : rx[d - 1], :1] = x.ttv_vec[d - 1]
    vec[d - 1][:, rx[d - 1]: rks[d - 1], :1] = y.ttv_vec[d - 1]
    return TTvector(d, vec, tuple(x.ttv_dims), rks, [0] * d)


def add_(x: TTvector, y: TTvector) -> TTvector:
    z = add(x, y)
    x.ttv_vec, x.ttv_rks, x.ttv_ot = z.ttv_vec, z.ttv_rks, z.ttv_ot
    return x


def scale(a: float, A: TTvector) -> TTvector:
    if a == 0:
        return zeros_tt(A.ttv_dims, A.ttv_rks)
    i = next((k for k, o in enumerate(A.ttv_ot) if o == 0), 0)
    X = [c.copy() for c in A.ttv_vec]
    X[i] = a * X[i]
    return TTvector(A.N, X, tuple(A.ttv_dims), list(A.ttv_rks), list(A.ttv_ot))


def sub(A: TTvector, B: TTvector) -> TTvector:
    return add(scale(-1.0, B), A)


def div(A: TTvector, a: float) -> TTvector:
    return scale(1 / a, A)


# --------------------------------------------------------------------------------------
# orthogonalize — src/tt_tools.jl:511-543
# --------------------------------------------------------------------------------------
def _lq(T: np.ndarray):
    """Thin LQ via LAPACK (QR of the transpose): T = L @ Q."""
    q, r = sla.qr(T.T, mode="economic")
    return r.T, q.T


def orthogonalize(x: TTvector, i: int = 1) -> TTvector:
    d = x.N
    assert 1 <= i <= d, "Impossible orthogonalization"
    dims = x.ttv_dims
    y_rks = r_and_d_to_rks(x.ttv_rks, dims)
    y = zeros_tt(dims, y_rks)
    FR = np.ones((1, 1))
    for j in range(1, i):  # left sweep, sites 1..i-1
        y.ttv_ot[j - 1] = 1
        Xj = x.ttv_vec[j - 1]
        n = dims[j - 1]
        yr = y.ttv_rks[j - 1]
        T = np.einsum("ag,sgb->sab", FR, Xj)           # [s, alpha, beta]; row = alpha + yr*s
        Q, R = sla.qr(T.reshape(n * yr, Xj.shape[2]), mode="economic")
        rnew = Q.shape[1]
        y.ttv_rks[j] = rnew
        y.ttv_vec[j - 1] = Q.reshape(n, yr, rnew)
        FR = R[:rnew, :]
    FL = np.ones((1, 1))
    for j in range(d, i, -1):  # right sweep, sites d..i+1
        y.ttv_ot[j - 1] = -1
        Xj = x.ttv_vec[j - 1]
        n = dims[j - 1]
        yr = y.ttv_rks[j]
        T = np.einsum("sag,gb->asb", Xj, FL)           # [alpha, s, beta]; col = beta + yr*s
        L, Q = _lq(T.reshape(Xj.shape[1], n * yr))
        rnew = Q.shape[0]
        y.ttv_rks[j - 1] = rnew
        y.ttv_vec[j - 1] = Q.reshape(rnew, n, yr).transpose(1, 0, 2).copy()
        FL = L[:, :rnew]
    y.ttv_ot[i - 1] = 0
    Xi = x.ttv_vec[i - 1]
    core = np.zeros((dims[i - 1], y.ttv_rks[i - 1], y.ttv_rks[i]), dtype=np.result_type(Xi.dtype, FR.dtype, FL.dtype))
    for k in range(dims[i - 1]):
        core[k, :, :] = FR @ Xi[k, :, :] @ FL
    y.ttv_vec[i - 1] = core
    return y


# --------------------------------------------------------------------------------------
# _svdtrunc — the EFFECTIVE method is src/tt_cross_interpolation.jl:149-166 (relative
# tail-norm rule); src/tt_tools.jl:737-741 is shadowed for every Matrix argument.
# --------------------------------------------------------------------------------------
def svdtrunc(A: np.ndarray, max_bond: int = 2 ** 62, truncerr: float = 0.0):
    U, s, Vt = sla.svd(A, full_matrices=False, lapack_driver="gesdd")
    r = len(s)
    if truncerr > 0:
        nrm = float(np.linalg.norm(s))
        cum = 0.0
        for i in range(r, 0, -1):
            cum += float(abs(s[i - 1]) ** 2)
            if math.sqrt(cum) > truncerr * nrm:
                r = i
                break
    r = min(r, max_bond)
    return U[:, :r], s[:r], Vt[:r, :]


def svdtrunc_shadowed(A: np.ndarray, max_bond=None, truncerr: float = 0.0):
    """src/tt_tools.jl:737-741 — dead code for Matrix inputs; kept for documentation."""
    U, s, Vt = sla.svd(A, full_matrices=False, lapack_driver="gesdd")
    if max_bond is None:
        max_bond = max(A.shape)
    r = min(max_bond, int(np.count_nonzero(s >= truncerr)))
    return U[:, :r], s[:r], Vt[:r, :]


# --------------------------------------------------------------------------------------
# _tt_bond_truncate! / tt_compress! — src/tt_tools.jl:743-789
# --------------------------------------------------------------------------------------
def tt_bond_truncate_(psi: TTvector, k: int, max_bond: int = 2 ** 62, truncerr: float = 0.0,
                      faithful: bool = False, svals_out: list | None = None):
    """k is 1-based.  Mutates psi (cores k, k+1 and ttv_rks[k]); ttv_ot is untouched.

    ``faithful=True`` also performs the reference's trailing ``orthogonalize(psi; i=k)``
    (src/tt_tools.jl:769) and returns it; tt_compress! discards that value.
    """
    assert 1 <= k < psi.N, "k must be in 1:(N-1)"
    Ck, Ck1 = psi.ttv_vec[k - 1], psi.ttv_vec[k]
    d1, Dl, _ = Ck.shape
    d2, _, Dr = Ck1.shape
    # AAC[alpha,s1,s2,beta]; M[(alpha + Dl*s1), (s2 + d2*beta)]
    M = np.einsum("sag,tgb->sabt", Ck, Ck1).reshape(d1 * Dl, Dr * d2)
    U, s, Vt = svdtrunc(M, max_bond=max_bond, truncerr=truncerr)
    if svals_out is not None:
        svals_out.append(s.copy())
    ssq = np.sqrt(s)
    U = U * ssq[None, :]
    Vt = ssq[:, None] * Vt
    r = U.shape[1]
    psi.ttv_vec[k - 1] = U.reshape(d1, Dl, r).copy()
    psi.ttv_vec[k] = Vt.reshape(r, Dr, d2).transpose(2, 0, 1).copy()
    psi.ttv_rks[k] = r
    if faithful:
        return orthogonalize(psi, i=k)
    return None


def tt_compress_(psi: TTvector, max_bond: int, truncerr: float = 0.0, sweeps: int = 1,
                 faithful: bool = False, svals_out: list | None = None) -> TTvector:
    assert sweeps >= 1, "sweeps must be >= 1"
    for _ in range(sweeps):
        for k in range(1, psi.N):
            tt_bond_truncate_(psi, k, max_bond=max_bond, truncerr=truncerr, faithful=faithful, svals_out=svals_out)
        for k in range(psi.N - 1, 0, -1):
            tt_bond_truncate_(psi, k, max_bond=max_bond, truncerr=truncerr, faithful=faithful, svals_out=svals_out)
    return psi


# --------------------------------------------------------------------------------------
# hadamard_ttm — src/tt_operations.jl:363-422 (SVD with the relative criterion of the
# effective _svdtrunc; "Eq. (10) of arXiv:2410.19747" per the source comment)
# --------------------------------------------------------------------------------------
def ttm_swap_(cores: list, rks: list, j: int, tol: float = 0.0, rmax: int = 2 ** 62) -> int:
    """_ttm_swap!(cores, rks, j) (src/tt_operations.jl:365-382); j is 1-based."""
    A, B = cores[j - 1], cores[j]                   # (dA, rL, rM), (dB, rM, rR)
    dA, rL, _ = A.shape
    dB, _, rR = B.shape
    C = np.einsum("xma,yan->xymn", A, B)            # C[sA, sB, m, n]
    # permutedims(C, (3, 2, 1, 4)) -> (rL, dB, dA, rR), column-major reshape: rows (m, sB), cols (sA, n)
    mat = np.reshape(C.transpose(2, 1, 0, 3), (rL * dB, dA * rR), order="F")
    U, sv, Vt = svdtrunc(mat, max_bond=rmax, truncerr=tol)
    r = U.shape[1]
    cores[j - 1] = np.reshape(U, (rL, dB, r), order="F").transpose(1, 0, 2).copy()                       # (dB, rL, r)
    cores[j] = np.reshape(sv[:, None] * Vt, (r, dA, rR), order="F").transpose(1, 0, 2).copy()           # (dA, r, rR)
    rks[j] = r
    return r


def ttm_contract_(cores: list, rks: list, p: int) -> None:
    """_ttm_contract!(cores, rks, p) (src/tt_operations.jl:384-396); p is 1-based."""
    A, B = cores[p - 1], cores[p]
    Pi = np.einsum("sma,san->smn", A, B)
    cores[p - 1] = Pi
    del cores[p]
    del rks[p]


def hadamard_ttm(x: TTvector, y: TTvector, tol: float = 1.0e-14, rmax: int = 2 ** 62) -> TTvector:
    """src/tt_operations.jl:398-422."""
    assert tuple(x.ttv_dims) == tuple(y.ttv_dims), "Incompatible TT dimensions"
    d = x.N
    cores = [np.array(c, dtype=float) for c in x.ttv_vec]
    for k in range(1, d + 1):
        cores.append(np.transpose(y.ttv_vec[d - k], (0, 2, 1)).copy())
    rks = list(x.ttv_rks) + list(reversed(list(y.ttv_rks)))[1:]
    for it in range(1, d + 1):
        for j in range(d, d - it + 1, -1):          # j = d : -1 : (d - it + 2)
            ttm_swap_(cores, rks, j, tol=tol, rmax=rmax)
        ttm_contract_(cores, rks, d - it + 1)
    return TTvector(d, cores, tuple(x.ttv_dims), [int(r) for r in rks], [0] * d)


# --------------------------------------------------------------------------------------
# QTT reorder — src/qtt_tools.jl:660-775 (vector form)
# --------------------------------------------------------------------------------------
def swap_adjacent_sites(A: np.ndarray, B: np.ndarray, threshold: float = 0.0):
    """_swap_adjacent_sites (src/qtt_tools.jl:660-695)."""
    d1, rl, _ = A.shape
    d2, _, rr = B.shape
    C = np.einsum("xlm,ymr->xylr", A, B)            # C[s1, s2, l, r]
    # permutedims(C, (2, 3, 1, 4)) -> (s2, l, s1, r); reshape(d2*rl, d1*rr) column-major
    M = np.reshape(C.transpose(1, 2, 0, 3), (d2 * rl, d1 * rr), order="F")
    U, sv, Vt = sla.svd(M, full_matrices=False, lapack_driver="gesdd")
    if threshold > 0:
        r_new = max(1, int(np.count_nonzero(sv > threshold * sv[0])))
    else:
        r_new = len(sv)
    U, sv, Vt = U[:, :r_new], sv[:r_new], Vt[:r_new, :]
    new_A = np.reshape(U, (d2, rl, r_new), order="F").copy()
    new_B = np.reshape(sv[:, None] * Vt, (r_new, d1, rr), order="F").transpose(1, 0, 2).copy()
    return new_A, new_B


def bubble_sort_swaps(perm: Sequence[int]) -> List[int]:
    """_bubble_sort_swaps (src/qtt_tools.jl:705-718): 1-based adjacent swap positions."""
    p = list(perm)
    swaps = []
    n = len(p)
    for i in range(1, n + 1):
        for j in range(1, n - i + 1):
            if p[j - 1] > p[j]:
                p[j - 1], p[j] = p[j], p[j - 1]
                swaps.append(j)
    return swaps


def reorder_perm(n_dims: int, bits_per_dim: int, to_interleaved: bool) -> List[int]:
    """The target-position vector of reorder (src/qtt_tools.jl:740-757): perm[src] = tgt, 0-based values."""
    N = n_dims * bits_per_dim
    perm = [0] * N
    for dd in range(1, n_dims + 1):
        for b in range(bits_per_dim):
            if to_interleaved:
                src, tgt = (dd - 1) * bits_per_dim + b, b * n_dims + (dd - 1)
            else:
                src, tgt = b * n_dims + (dd - 1), (dd - 1) * bits_per_dim + b
            perm[src] = tgt
    return perm


def swap_sites_(x: TTvector, swaps: Sequence[int], threshold: float = 0.0) -> TTvector:
    """The swap loop of reorder (src/qtt_tools.jl:762-773) on a plain TTvector, in place."""
    for k in swaps:
        a, b = swap_adjacent_sites(x.ttv_vec[k - 1], x.ttv_vec[k], threshold=threshold)
        x.ttv_vec[k - 1], x.ttv_vec[k] = a, b
    x.ttv_rks = [1] + [int(c.shape[2]) for c in x.ttv_vec]
    x.ttv_ot = [0] * x.N
    return x


def reorder(x: TTvector, n_dims: int, bits_per_dim: int, to_interleaved: bool, threshold: float = 0.0) -> TTvector:
    """reorder(q, new_ordering; threshold) (src/qtt_tools.jl:733-775) for a QTT vector given as its TTvector plus metadata."""
    y = copy_tt(x)
    return swap_sites_(y, bubble_sort_swaps(reorder_perm(n_dims, bits_per_dim, to_interleaved)), threshold=threshold)


def swap_adjacent_sites_op(A: np.ndarray, B: np.ndarray, threshold: float = 0.0):
    """_swap_adjacent_sites_op (src/qtt_tools.jl:852-885): operator cores (phys, phys, r_left, r_right)."""
    d1, _, rl, _ = A.shape
    d2, _, _, rr = B.shape
    C = np.einsum("ijlm,pqmr->ijpqlr", A, B)        # C[i1, j1, i2, j2, l, r]
    # permutedims(C, (3, 4, 5, 1, 2, 6)) -> (i2, j2, l, i1, j1, r); column-major reshape
    M = np.reshape(C.transpose(2, 3, 4, 0, 1, 5), (d2 * d2 * rl, d1 * d1 * rr), order="F")
    U, sv, Vt = sla.svd(M, full_matrices=False, lapack_driver="gesdd")
    r_new = max(1, int(np.count_nonzero(sv > threshold * sv[0]))) if threshold > 0 else len(sv)
    new_A = np.reshape(U[:, :r_new], (d2, d2, rl, r_new), order="F").copy()
    SV = sv[:r_new, None] * Vt[:r_new, :]
    new_B = np.reshape(SV, (r_new, d1, d1, rr), order="F").transpose(1, 2, 0, 3).copy()
    return new_A, new_B


def reorder_op(A: TToperator, n_dims: int, bits_per_dim: int, to_interleaved: bool, threshold: float = 0.0) -> TToperator:
    """reorder(A::QTToperator, new_ordering; threshold) (src/qtt_tools.jl:894-932)."""
    cores = [np.array(c) for c in A.tto_vec]
    for k in bubble_sort_swaps(reorder_perm(n_dims, bits_per_dim, to_interleaved)):
        cores[k - 1], cores[k] = swap_adjacent_sites_op(cores[k - 1], cores[k], threshold=threshold)
    rks = [1] + [int(c.shape[3]) for c in cores]
    return TToperator(A.N, cores, tuple(2 for _ in range(A.N)), rks, [0] * A.N)


# --------------------------------------------------------------------------------------
# ttv_decomp — src/tt_tools.jl:186-252 (hierarchical SVD, absolute threshold)
# --------------------------------------------------------------------------------------
def ttv_decomp(tensor: np.ndarray, index: int = 1, tol: float = 1.0e-12) -> TTvector:
    """index is 1-based.  Left of the root the reference assigns ALL columns of u into a core sized by the truncated rank
    (:206-208), which throws a DimensionMismatch in Julia when a singular value falls below tol there; this restatement
    truncates u instead (the only consistent completion) — that corner is parity-unpinned."""
    tensor = np.asarray(tensor, dtype=float)
    dims = tuple(int(v) for v in tensor.shape)
    d = len(dims)
    ot = [-1] * d
    ot[index - 1] = 0
    for i in range(index, d):
        ot[i] = 1
    rks = [1] * (d + 1)
    vec: list = [None] * d
    cur = tensor
    for i in range(1, index):                                    # :199-211
        cur = np.reshape(cur, (rks[i - 1] * dims[i - 1], -1), order="F")
        u, sv, vt = sla.svd(cur, full_matrices=False, lapack_driver="gesdd")
        r = int(np.count_nonzero(sv >= tol))
        rks[i] = r
        core = np.zeros((dims[i - 1], rks[i - 1], r))
        for x in range(dims[i - 1]):
            core[x, :, :] = u[rks[i - 1] * x: rks[i - 1] * (x + 1), :r]
        vec[i - 1] = core
        cur = sv[:r, None] * vt[:r, :]
    for i in range(d, index, -1):                                # :214-233
        cur = np.reshape(cur, (-1, dims[i - 1] * rks[i]), order="F")
        u, sv, vt = sla.svd(cur, full_matrices=False, lapack_driver="gesdd")
        r = int(np.count_nonzero(sv >= tol))
        rks[i - 1] = r
        core = np.zeros((dims[i - 1], r, rks[i]))
        for x in range(dims[i - 1]):
            cols = [dims[i - 1] * be + x for be in range(rks[i])]
            core[x, :, :] = vt[:r, cols]
        vec[i - 1] = core
        cur = u[:, :r] * sv[None, :r]
    cur = np.reshape(cur, (dims[index - 1] * rks[index - 1], -1), order="F")   # :236-245
    core = np.zeros((dims[index - 1], rks[index - 1], rks[index]))
    for x in range(dims[index - 1]):
        core[x, :, :] = cur[rks[index - 1] * x: rks[index - 1] * (x + 1), :rks[index]]
    vec[index - 1] = core
    return TTvector(d, vec, dims, rks, ot)


# --------------------------------------------------------------------------------------
# TToperator + and scalar * (inputs of the ALS tests) — src/tt_operations.jl:71-96, :268-281
# --------------------------------------------------------------------------------------
def tto_add(A: TToperator, B: TToperator) -> TToperator:
    d = A.N
    rks = [a + b for a, b in zip(A.tto_rks, B.tto_rks)]
    rks[0] = rks[d] = 1
    vec = []
    for k in range(d):
        n = A.tto_dims[k]
        core = np.zeros((n, n, rks[k], rks[k + 1]))
        a, b = A.tto_vec[k], B.tto_vec[k]
        if k == 0:
            core[:, :, 0, :A.tto_rks[1]] = a[:, :, 0, :]
            core[:, :, 0, A.tto_rks[1]:] = b[:, :, 0, :]
        elif k == d - 1:
            core[:, :, :A.tto_rks[k], 0] = a[:, :, :, 0]
            core[:, :, A.tto_rks[k]:, 0] = b[:, :, :, 0]
        else:
            core[:, :, :A.tto_rks[k], :A.tto_rks[k + 1]] = a
            core[:, :, A.tto_rks[k]:, A.tto_rks[k + 1]:] = b
        vec.append(core)
    return TToperator(d, vec, tuple(A.tto_dims), rks, [0] * d)


def tto_scale(a: float, A: TToperator) -> TToperator:
    vec = [np.array(c) for c in A.tto_vec]
    vec[0] = a * vec[0]
    return TToperator(A.N, vec, tuple(A.tto_dims), list(A.tto_rks), list(A.tto_ot))


# --------------------------------------------------------------------------------------
# als_linsolve — src/solvers/als.jl:9-70 (environments, Ksolve), :102-135 (core moves), :161-222
# --------------------------------------------------------------------------------------
def _als_update_H(x, A, Hi):
    # Him[a, al, be] = conj(x)[j, al, ph] * Hi[z, ph, ch] * x[k, be, ch] * A[j, k, a, z]      (als.jl:23-26)
    return np.einsum("jap,zpc,kbc,jkyz->yab", x, Hi, x, A, optimize=True)


def _als_update_Hb(x, b, Hbi):
    # H_bim[al, be] = H_bi[ph, ch] * b[i, be, ch] * conj(x)[i, al, ph]                        (als.jl:42-45)
    return np.einsum("pc,ibc,iap->ab", Hbi, b, x, optimize=True)


def _als_update_G(x, A, Gi):
    # Gip[j, al, k, be, J] = conj(x)[l, ph, al] * (Gi[l, ph, m, ch, L] * x[m, ch, be]) * A[j, k, L, J]   (als.jl:47-50)
    return np.einsum("lpa,lpmcL,mcb,jkLJ->jakbJ", x, Gi, x, A, optimize=True)


def _als_update_Gb(x, b, Gbi):
    # G_bip[i, al, be] = b[i, ph, be] * G_bi[j, ch, ph] * conj(x)[j, ch, al]                   (als.jl:52-55)
    return np.einsum("ipb,jcp,jca->iab", b, Gbi, x, optimize=True)


def _als_ksolve(Gi, Gbi, Hi, Hbi):
    n, rim, ri = Gi.shape[0], Gi.shape[1], Hi.shape[1]
    N = n * rim * ri
    # K[(a,b,c),(d,e,f)] = Gi[a,b,d,e,z] * Hi[z,c,f], column-major multi-indices (als.jl:58-63)
    K6 = np.einsum("abdez,zcf->abcdef", Gi, Hi)
    K = np.reshape(K6, (N, N), order="F")
    Pb = np.einsum("iab,cb->iac", Gbi, Hbi)                      # (n, rim, ri)  (als.jl:68)
    V = np.linalg.solve(K, np.reshape(Pb, N, order="F"))         # K \ Pb[:]
    return np.reshape(V, (n, rim, ri), order="F")


def als_linsolve(A: TToperator, b: TTvector, tt_start: TTvector, sweep_count: int = 2) -> TTvector:
    """als_linsolve(A, b, tt_start; sweep_count) (src/solvers/als.jl:161-222), dense local solves (it_solver is ignored
    by the reference, :161,203)."""
    d = A.N
    x = orthogonalize(tt_start)                                  # :174
    dims = tuple(tt_start.ttv_dims)
    rks = list(tt_start.ttv_rks)                                 # fixed ranks (:177)
    G = [None] * d
    Gb = [None] * d
    G[0] = np.reshape(A.tto_vec[0][:, :, 0, :], (dims[0], 1, dims[0], 1, -1))        # :188
    Gb[0] = np.reshape(b.ttv_vec[0], (dims[0], 1, -1))                                # :189
    H = [None] * d
    Hb = [None] * d
    H[d - 1] = np.ones((1, 1, 1))
    Hb[d - 1] = np.ones((1, 1))
    for i in range(d - 1, 0, -1):                                # init_H / init_Hb (:9-21, :28-40)
        H[i - 1] = _als_update_H(x.ttv_vec[i], A.tto_vec[i], H[i])
        Hb[i - 1] = _als_update_Hb(x.ttv_vec[i], b.ttv_vec[i], Hb[i])
    nsweeps = 0
    while nsweeps < sweep_count:
        nsweeps += 1
        for i in range(d - 1):                                   # first half sweep (:199-207), i 0-based
            V = _als_ksolve(G[i], Gb[i], H[i], Hb[i])
            n, rim, ri = dims[i], rks[i], rks[i + 1]
            Q, R = np.linalg.qr(np.reshape(V, (n * rim, ri), order="F"))             # right_core_move (:122-135)
            x.ttv_vec[i] = np.reshape(Q[:, :ri], (n, rim, ri), order="F")
            x.ttv_ot[i] = -1
            x.ttv_vec[i + 1] = np.einsum("bz,azc->abc", R[:ri, :], x.ttv_vec[i + 1])
            x.ttv_ot[i + 1] = 0
            G[i + 1] = _als_update_G(x.ttv_vec[i], A.tto_vec[i + 1], G[i])
            Gb[i + 1] = _als_update_Gb(x.ttv_vec[i], b.ttv_vec[i + 1], Gb[i])
        if nsweeps == sweep_count:
            return x
        nsweeps += 1
        for i in range(d - 1, 0, -1):                            # second half sweep (:213-219)
            V = _als_ksolve(G[i], Gb[i], H[i], Hb[i])
            n, rim, ri = dims[i], rks[i], rks[i + 1]
            M = np.reshape(np.transpose(V, (0, 2, 1)), (n * ri, rim), order="F")     # left_core_move (:102-120)
            Q, R = np.linalg.qr(M)
            x.ttv_vec[i] = np.transpose(np.reshape(Q[:, :rim], (n, ri, rim), order="F"), (0, 2, 1)).copy()
            x.ttv_ot[i] = 1
            x.ttv_vec[i - 1] = np.einsum("abz,cz->abc", x.ttv_vec[i - 1], R[:rim, :])
            x.ttv_ot[i - 1] = 0
            H[i - 1] = _als_update_H(x.ttv_vec[i], A.tto_vec[i], H[i])
            Hb[i - 1] = _als_update_Hb(x.ttv_vec[i], b.ttv_vec[i], Hb[i])
    return x


# --------------------------------------------------------------------------------------
# mals_linsolve — src/solvers/mals.jl:10-168 (environments, two-site solve, SVD core moves), :240-312
# --------------------------------------------------------------------------------------
def sv_trunc(s: np.ndarray, tol: float) -> np.ndarray:
    """sv_trunc (src/solvers/mals.jl:42-56): drops the tail while its weight stays below tol * ||s||^2 — and keeps the value
    that crossed the line."""
    if tol == 0.0:
        return s
    d = len(s)
    i = 0
    weight = 0.0
    norm2 = float(np.sum(np.abs(s) ** 2))
    while i < d and weight < tol * norm2:
        weight += float(s[d - i - 1]) ** 2
        i += 1
    return s[: d - i + 1]


def mals_linsolve(A: TToperator, b: TTvector, tt_start: TTvector, tol: float = 1.0e-12, rmax: int | None = None) -> TTvector:
    """mals_linsolve(A, b, tt_start; tol, rmax) (src/solvers/mals.jl:240-312): one forward and one backward half sweep of
    two-site solves with dense local systems; ranks adapt through the truncated SVD of every local solution."""
    d = b.N
    dims = tuple(tt_start.ttv_dims)
    if rmax is None:
        rmax = int(round(math.sqrt(_prod(dims))))
    x = orthogonalize(tt_start)
    G = [None] * d
    Gb = [None] * d
    G[0] = np.reshape(A.tto_vec[0][:, :, 0, :], (dims[0], 1, dims[0], 1, -1))
    Gb[0] = np.reshape(b.ttv_vec[0], (dims[0], 1, -1))
    # H[i] (0-based i = 0..d-2) couples sites i, i+1: (R_{i+1}, n_{i+1}, r_{i+2}, n_{i+1}, r_{i+2})    (:15-40)
    H = [None] * (d - 1)
    Hb = [None] * (d - 1)
    H[d - 2] = np.reshape(np.transpose(A.tto_vec[d - 1], (2, 0, 1, 3)), (-1, dims[d - 1], 1, dims[d - 1], 1))
    Hb[d - 2] = np.reshape(np.transpose(b.ttv_vec[d - 1], (1, 0, 2)), (b.ttv_rks[d - 1], dims[d - 1], 1))
    for i in range(d - 2, 0, -1):
        # Him[a, i, al, l, be] = conj(x)[j, al, x] * (Hi[z, j, x, k, y] * x[k, be, y]) * A[i, l, a, z]      (:10-13), x = core i+1
        H[i - 1] = np.einsum("jax,zjxky,kby,ilwz->wialb", x.ttv_vec[i + 1], H[i], x.ttv_vec[i + 1], A.tto_vec[i], optimize=True)
        # Hbim[be, i, ch] = conj(x)[j, ch, a] * Hbi[ga, j, a] * b[i, be, ga]                                  (:60-66)
        Hb[i - 1] = np.einsum("jca,gja,ibg->bic", x.ttv_vec[i + 1], Hb[i], b.ttv_vec[i], optimize=True)

    def ksolve(i):
        Gi, Hi, Gbi, Hbi = G[i], H[i], Gb[i], Hb[i]
        kd = (Gi.shape[0], Gi.shape[1], Hi.shape[1], Hi.shape[2])
        N = kd[0] * kd[1] * kd[2] * kd[3]
        K = np.reshape(np.einsum("abefz,zcdgh->abcdefgh", Gi, Hi), (N, N), order="F")       # :148-157
        Pb = np.einsum("abz,zcd->abcd", Gbi, Hbi)                                            # :165
        # `Hermitian(K) \ b`: LAPACK's symmetric-indefinite solve on the upper triangle of K
        Ku = np.triu(K) + np.triu(K, 1).T
        V = sla.solve(Ku, np.reshape(Pb, N, order="F"), assume_a="sym")
        return np.reshape(V, kd, order="F")

    def split(V):
        M = np.reshape(V, (V.shape[0] * V.shape[1], -1), order="F")
        u, sv, vt = sla.svd(M, full_matrices=False, lapack_driver="gesdd")
        r = min(len(sv_trunc(sv, tol)), rmax)
        return u, sv, vt, r

    for i in range(d - 1):                                           # first half sweep (:268-283)
        V = ksolve(i)
        u, sv, vt, r = split(V)
        x.ttv_rks[i + 1] = r
        x.ttv_vec[i] = np.reshape(u[:, :r], (V.shape[0], V.shape[1], r), order="F")          # right_core_move_mals (:121-146)
        x.ttv_ot[i] = -1
        x.ttv_vec[i + 1] = np.transpose(np.reshape(sv[:r, None] * vt[:r, :], (r, V.shape[2], V.shape[3]), order="F"), (1, 0, 2)).copy()
        x.ttv_ot[i + 1] = 0
        G[i + 1] = _als_update_G(x.ttv_vec[i], A.tto_vec[i + 1], G[i])
        Gb[i + 1] = _als_update_Gb(x.ttv_vec[i], b.ttv_vec[i + 1], Gb[i])
    for i in range(d - 2, -1, -1):                                   # second half sweep (:286-305)
        V = ksolve(i)
        u, sv, vt, r = split(V)
        x.ttv_rks[i + 1] = r
        x.ttv_vec[i + 1] = np.transpose(np.reshape(vt[:r, :], (r, V.shape[2], V.shape[3]), order="F"), (1, 0, 2)).copy()   # left_core_move_mals (:94-119)
        x.ttv_vec[i] = np.reshape(u[:, :r] * sv[None, :r], (V.shape[0], V.shape[1], r), order="F")
        x.ttv_ot[i + 1] = 1
        x.ttv_ot[i] = 0
        if i > 0:
            H[i - 1] = np.einsum("jax,zjxky,kby,ilwz->wialb", x.ttv_vec[i + 1], H[i], x.ttv_vec[i + 1], A.tto_vec[i], optimize=True)
            Hb[i - 1] = np.einsum("jca,gja,ibg->bic", x.ttv_vec[i + 1], Hb[i], b.ttv_vec[i], optimize=True)
    return x


def cut_off_index(s: np.ndarray, tol: float, degen_tol: float = 1.0e-10) -> int:
    """cut_off_index(s, tol; degen_tol) (src/solvers/dmrg.jl:179-185): count(s > ||s|| tol), then extended over values
    `isapprox` (rtol = atol = degen_tol) to the last kept one.  Known answer: test/test_dmrg.jl:20-25."""
    s = np.asarray(s, dtype=np.float64)
    k = int(np.sum(s > np.linalg.norm(s) * tol))
    # Julia indexes s[k] here: k = 0 (all-zero spectrum) is a BoundsError in the reference; this restatement keeps k = 0
    while 0 < k < len(s) and abs(s[k - 1] - s[k]) <= max(degen_tol, degen_tol * max(abs(s[k - 1]), abs(s[k]))):
        k += 1
    return k


def dmrg_sweep_plan(sweep_schedule: Sequence[int], rmax_schedule: Sequence[int]):
    """The (rmax of every full sweep, rmax of the closing step) the while-loop of dmrg_linsolve walks through
    (src/solvers/dmrg.jl:421-443): sweep number s ends stage j when s == sweep_schedule[j]; the sweep that would end the last
    stage is replaced by the closing solve at site 1."""
    plan, n, j = [], 0, 0
    while True:
        n += 1
        if n == sweep_schedule[j]:
            j += 1
            if j >= len(sweep_schedule):
                return plan, int(rmax_schedule[-1])
        plan.append(int(rmax_schedule[j]))


KRYLOVDIM_DEFAULT = 30          # KrylovKit.KrylovDefaults.krylovdim (third party; compat 0.6.1 / 0.9 / 0.10)


def cg_solve(apply_K, rhs: np.ndarray, x0: np.ndarray, tol: float, maxiter: int):
    """Conjugate gradients as KrylovKit's `linsolve(f, b, x0; issymmetric = true, isposdef = true, tol, maxiter)` runs them (the
    call at src/solvers/dmrg.jl:170; KrylovKit is a third-party dependency, compat 0.6.1 / 0.9 / 0.10, not in the reference tree —
    this is its published CG recurrence): r = b - K x0; stop when ||r||_2 < tol (ABSOLUTE) or after maxiter iterations, returning
    the current iterate either way.  Returns (x, iterations)."""
    x = x0.copy()
    r = rhs - apply_K(x)
    rho = float(r @ r)
    p = r.copy()
    it = 0
    while not (math.sqrt(rho) < tol) and it < maxiter:
        q = apply_K(p)
        alpha = rho / float(p @ q)
        x += alpha * p
        r -= alpha * q
        rho_new = float(r @ r)
        beta = rho_new / rho
        rho = rho_new
        it += 1
        if math.sqrt(rho) < tol:
            break
        p = r + beta * p
    return x, it


def dmrg_linsolve(A: TToperator, b: TTvector, tt_start: TTvector, tol: float = 1.0e-12, sweep_schedule: Sequence[int] = (2,),
                  rmax_schedule: Sequence[int] | None = None, it_solver: bool = False, linsolv_maxiter: int = 200,
                  linsolv_tol: float | None = None, itslv_thresh: int = 10 ** 9, stats: dict | None = None) -> TTvector:
    """dmrg_linsolve(A, b, tt_start; N = 2, tol, sweep_schedule, rmax_schedule, it_solver, linsolv_maxiter, linsolv_tol,
    itslv_thresh) (src/solvers/dmrg.jl:388-472).  Local systems (Ksolve!, :92-177): dense (`K_full` + `K \\ Pb`, :57-62, :173-175)
    unless `it_solver` or the system has more than `itslv_thresh` unknowns; then conjugate gradients (cg_solve) on the symmetrised
    matrix-free operator 1/2 (K + K^T) of :99-168 started from V0 = the current two-site block (:311-316, :329-334).  Defaults here:
    dense everywhere (the reference's own defaults are it_solver = true, itslv_thresh = 256, linsolv_tol = max(sqrt(tol), 1e-8)).
    Environments G (R, r, r) / H (R, r, r) (dmrg.jl:27-35), merged operator / right-hand-side cores Amid / b_mid (:38-47, :89-96),
    core moves right_core_move! / left_core_move! (:187-232)."""
    d = b.N
    dims = tuple(tt_start.ttv_dims)
    if rmax_schedule is None:
        rmax_schedule = (math.isqrt(_prod(dims)),)
    plan, rmax_final = dmrg_sweep_plan(list(sweep_schedule), list(rmax_schedule))
    x = orthogonalize(tt_start)
    Av, bv = A.tto_vec, b.ttv_vec
    # window i (0-based) = sites i, i+1
    Amid = [np.reshape(np.einsum("aIJx,ijxb->aIiJjb", np.transpose(Av[i], (2, 0, 1, 3)), Av[i + 1]),
                       (A.tto_rks[i], dims[i] * dims[i + 1], dims[i] * dims[i + 1], A.tto_rks[i + 2]), order="F") for i in range(d - 1)]
    bmid = [np.reshape(np.einsum("aix,jxb->aijb", np.transpose(bv[i], (1, 0, 2)), bv[i + 1]),
                       (b.ttv_rks[i], dims[i] * dims[i + 1], b.ttv_rks[i + 2]), order="F") for i in range(d - 1)]
    G = [None] * (d - 1)
    Gb = [None] * (d - 1)
    H = [None] * (d - 1)
    Hb = [None] * (d - 1)
    G[0] = np.ones((1, 1, 1))
    Gb[0] = np.ones((1, 1))
    H[d - 2] = np.ones((1, 1, 1))
    Hb[d - 2] = np.ones((1, 1))

    def upd_H(i):          # H[i-1], Hb[i-1] from site i+1 (dmrg.jl:27-30, :80-83)
        xc = x.ttv_vec[i + 1]
        H[i - 1] = np.einsum("jap,zpc,kbc,jkwz->wab", xc, H[i], xc, Av[i + 1], optimize=True)
        Hb[i - 1] = np.einsum("pc,ibc,iap->ab", Hb[i], bv[i + 1], xc, optimize=True)

    def upd_G(i):          # G[i+1], Gb[i+1] from site i (dmrg.jl:32-35, :85-88)
        xc = x.ttv_vec[i]
        G[i + 1] = np.einsum("jpa,zpc,kcb,jkzw->wab", xc, G[i], xc, Av[i], optimize=True)
        Gb[i + 1] = np.einsum("pc,icb,ipa->ab", Gb[i], bv[i], xc, optimize=True)

    for i in range(d - 2, 0, -1):
        upd_H(i)

    if linsolv_tol is None:
        linsolv_tol = max(math.sqrt(tol), 1.0e-8)                      # dmrg.jl:394

    # start vector of the iterative local solve (V0_view): b_mid(tt_opt, 1, 2) at first (dmrg.jl:275), then what update_right
    # (:311-316) / update_left (:329-334) build from the moved factor V_move and the neighbouring core.  update_left reshapes
    # [alpha, J, i_k, gamma] with J (site i) running FASTER than i_k (site i-1): the two physical indices end up exchanged with
    # respect to the (site i-1 fast) convention of K_dims — restated as it is (it only is a start vector).
    v0 = {"V": None}

    def ksolve(i):
        Gi, Hi = G[i], H[i]
        kd = (Gi.shape[1], Amid[i].shape[1], Hi.shape[1])
        N = kd[0] * kd[1] * kd[2]
        Pb = np.einsum("ap,piq,cq->aic", Gb[i], bmid[i], Hb[i], optimize=True)
        if it_solver or N > itslv_thresh:
            Am = Amid[i]

            def apply_K(v):                                             # 1/2 (K + K^T) v, the intended contraction of dmrg.jl:165:
                V = np.reshape(v, kd, order="F")                        # Hrshp[a,b,c] = G[y,a,d] H[z,c,f] Amid[y,b,e,z] V[d,e,f] + transposes
                W = np.tensordot(Gi, V, axes=([2], [0]))                # [y, a, e, f]
                U = np.tensordot(Am, W, axes=([0, 2], [0, 2]))          # [b, z, a, f]
                t1 = np.tensordot(U, Hi, axes=([1, 3], [0, 2]))         # [b, a, c]
                W = np.tensordot(Gi, V, axes=([1], [0]))                # G[y,d,a] V[d,e,f] -> [y, a, e, f]
                U = np.tensordot(Am, W, axes=([0, 1], [0, 2]))          # Amid[y,e,b,z] -> [b, z, a, f]
                t2 = np.tensordot(U, Hi, axes=([1, 3], [0, 1]))         # H[z,f,c] -> [b, a, c]
                return 0.5 * np.reshape(np.transpose(t1 + t2, (1, 0, 2)), N, order="F")

            V0 = v0["V"]
            if V0 is None:                                              # b_mid(tt_opt, 1, 2)
                V0 = np.reshape(np.einsum("ajg,kgb->ajkb", np.transpose(x.ttv_vec[i], (1, 0, 2)), x.ttv_vec[i + 1]), kd, order="F")
            assert V0.shape == kd, (V0.shape, kd)
            # KrylovKit's algorithm selector turns `maxiter` into CG(maxiter = krylovdim * maxiter) for isposdef problems, with
            # krylovdim = KrylovDefaults.krylovdim = 30 (the convention src/solvers/euler.jl:29 spells out)
            v, iters = cg_solve(apply_K, np.reshape(Pb, N, order="F"), np.reshape(V0, N, order="F").copy(), linsolv_tol,
                                KRYLOVDIM_DEFAULT * linsolv_maxiter)
            if stats is not None:
                stats["cg_iterations"] = stats.get("cg_iterations", 0) + iters
                stats["cg_solves"] = stats.get("cg_solves", 0) + 1
            return np.reshape(v, kd, order="F")
        K = np.reshape(np.einsum("yad,zcf,ybez->abcdef", Gi, Hi, Amid[i], optimize=True), (N, N), order="F")
        Ku = np.triu(K) + np.triu(K, 1).T                              # Hermitian(K): the upper triangle
        V = sla.solve(Ku, np.reshape(Pb, N, order="F"), assume_a="sym")
        return np.reshape(V, kd, order="F")

    def right_move(V, i, rmax):
        rl, n1 = x.ttv_rks[i], dims[i]
        u, sv, vt = sla.svd(np.reshape(V, (rl * n1, -1), order="F"), full_matrices=False, lapack_driver="gesdd")
        r = min(cut_off_index(sv, tol), rmax)
        x.ttv_rks[i + 1] = r
        x.ttv_vec[i] = np.transpose(np.reshape(u[:, :r], (rl, n1, r), order="F"), (1, 0, 2)).copy()
        x.ttv_ot[i] = 1
        x.ttv_ot[i + 1] = 0
        return np.reshape(sv[:r, None] * vt[:r, :], (r, -1, V.shape[2]), order="F")          # V_move

    def left_move(V, i, rmax):       # the move at site j = i+1
        n2, rr = dims[i + 1], x.ttv_rks[i + 2]
        u, sv, vt = sla.svd(np.reshape(V, (-1, n2 * rr), order="F"), full_matrices=False, lapack_driver="gesdd")
        r = min(cut_off_index(sv, tol), rmax)
        x.ttv_rks[i + 1] = r
        x.ttv_vec[i + 1] = np.transpose(np.reshape(vt[:r, :], (r, n2, rr), order="F"), (1, 0, 2)).copy()
        x.ttv_ot[i + 1] = -1
        x.ttv_ot[i] = 0
        return np.reshape(u[:, :r] * sv[None, :r], (V.shape[0], -1, r), order="F")           # V_move

    for rmax in plan:
        for i in range(d - 2):                                        # first half sweep (:446-456)
            Vm = right_move(ksolve(i), i, rmax)                        # (r_{i+1}, n_{i+1}, r_{i+2})
            t = np.einsum("aJb,kbg->aJkg", Vm, x.ttv_vec[i + 2])       # update_right: [alpha, J, i_k, gamma]
            v0["V"] = np.reshape(t, (t.shape[0], -1, t.shape[3]), order="F")
            upd_G(i)
        for i in range(d - 2, 0, -1):                                 # second half sweep (:459-470)
            Vm = left_move(ksolve(i), i, rmax)                         # (r_i, n_i, r_{i+1})
            t = np.einsum("bJg,kab->aJkg", Vm, x.ttv_vec[i - 1])       # update_left: [alpha, J, i_k, gamma], J = site i FAST
            v0["V"] = np.reshape(t, (t.shape[0], -1, t.shape[3]), order="F")
            upd_H(i)
    Vm = left_move(ksolve(0), 0, rmax_final)                          # closing step (:426-441)
    x.ttv_vec[0] = np.transpose(np.reshape(Vm, (1, dims[0], -1), order="F"), (1, 0, 2)).copy()
    x.ttv_ot[0] = 0
    return x


# --------------------------------------------------------------------------------------
# TDVP local contractions — src/solvers/tdvp.jl:24-43, :205-208 (the @tensor index strings verbatim as einsum subscripts)
# --------------------------------------------------------------------------------------
def tdvp_to_lsr(A):
    """_to_lsr / _to_slr: permutedims(A, (2, 1, 3))  (tdvp.jl:24-25)"""
    return np.transpose(A, (1, 0, 2))


def tdvp_mpo_to_asbs(M):
    """_mpo_to_asbs: permutedims(M, (3, 1, 4, 2))  (tdvp.jl:27)"""
    return np.transpose(M, (2, 0, 3, 1))


def tdvp_dot3(X, Y):
    """_dot3 = LinearAlgebra.dot(vec(X), vec(Y)): conjugates the FIRST argument  (tdvp.jl:29)"""
    return np.vdot(np.reshape(X, -1, order="F"), np.reshape(Y, -1, order="F"))


def tdvp_applyH1_lsr(AC, FL, FR, M):
    """HAC[α,s,β] := FL[α,a,α′] * AC[α′,s′,β′] * M[a,s,b,s′] * FR[β′,b,β]  (tdvp.jl:29-31)"""
    return np.einsum("xay,ytq,asbt,qbz->xsz", FL, AC, M, FR, optimize=True)


def tdvp_applyH0(C, FL, FR):
    """HC[α,β] := FL[α,a,α′] * C[α′,β′] * FR[β′,a,β]  (tdvp.jl:33-35)"""
    return np.einsum("xay,yq,qaz->xz", FL, C, FR, optimize=True)


def tdvp_update_left_env(A, M, FL):
    """FLnext[α,a,β] := FL[α′,a′,β′] * A[β′,s′,β] * M[a′,s,a,s′] * conj(A[α′,s,α])  (tdvp.jl:37-39)"""
    return np.einsum("xpq,qtb,psat,xsu->uab", FL, A, M, np.conj(A), optimize=True)


def tdvp_update_right_env(A, M, FR):
    """FRprev[α,a,β] := A[α,s′,α′] * FR[α′,a′,β′] * M[a,s,a′,s′] * conj(A[β,s,β′])  (tdvp.jl:41-43)"""
    return np.einsum("xty,ypq,aspt,bsq->xab", A, FR, M, np.conj(A), optimize=True)


def tdvp_applyH2_lsr(AAC, FL, FR, M1, M2):
    """HAAC[α,s1,s2,β] := FL[α,a,α′] * AAC[α′,s1′,s2′,β′] * M1[a,s1,b,s1′] * M2[b,s2,c,s2′] * FR[β′,c,β]  (tdvp.jl:205-208)"""
    return np.einsum("xay,ytuq,asbt,bvcu,qcz->xsvz", FL, AAC, M1, M2, FR, optimize=True)


# --------------------------------------------------------------------------------------
# TDVP drivers — src/solvers/tdvp.jl:45-203 (tdvp1sweep!, tdvp), :210-357 (tdvp2sweep!, tdvp2)
#
# `exponentiate` is KrylovKit's (Project.toml compat "0.6.1, 0.9, 0.10"; the package is not in /root/reference): for a Hermitian
# operator its Lanczos variant builds an orthonormal Krylov basis V_m (krylovdim 30, full reorthogonalisation), exponentiates the
# tridiagonal projection, accepts when the a-posteriori estimate beta_m |e_m^T exp(t T_m) e_1| is below tol (1e-12) and otherwise
# advances by a shorter time and restarts (maxiter 100).  Restated here from that published algorithm; the reference's own tests
# (test/test_tdvp.jl) pin the drivers through known answers (H = 0 -> identity, an eigenmode of the heat operator to 1e-8, A = I/2
# -> zero residual), see tests/test_oracle_reference_pins.py.
# --------------------------------------------------------------------------------------
def tdvp_exponentiate(Hfun, t, x0, krylovdim: int = 30, tol: float = 1.0e-12, maxiter: int = 100):
    """y = exp(t H) x0 for Hermitian H given as a function on arrays of x0's shape; t real or complex."""
    shape = x0.shape
    x = np.reshape(x0, -1, order="F").astype(np.result_type(x0.dtype, type(t), np.float64))
    remaining = t
    total = abs(t)
    for _ in range(maxiter):
        nrm = np.linalg.norm(x)
        if nrm == 0.0 or remaining == 0:
            break
        n = x.size
        m_max = min(krylovdim, n)
        V = np.zeros((n, m_max), dtype=x.dtype if np.iscomplexobj(x) else np.result_type(x.dtype, Hfun(np.reshape(x, shape, order="F")).dtype))
        x = x.astype(V.dtype)
        V[:, 0] = x / nrm
        alphas, betas = [], []
        m = 0
        happy = False
        for j in range(m_max):
            w = np.reshape(Hfun(np.reshape(V[:, j], shape, order="F")), -1, order="F").astype(V.dtype)
            a = np.real(np.vdot(V[:, j], w))
            alphas.append(float(a))
            w = w - a * V[:, j]
            if j > 0:
                w = w - betas[j - 1] * V[:, j - 1]
            for _r in range(2):                                             # full reorthogonalisation, twice
                w = w - V[:, : j + 1] @ (V[:, : j + 1].conj().T @ w)
            b = float(np.linalg.norm(w))
            m = j + 1
            if b <= 1.0e-14 * max(1.0, abs(a)) or m == n:
                happy = True
                break
            betas.append(b)
            if j + 1 < m_max:
                V[:, j + 1] = w / b
        Tm = np.diag(alphas[:m]) + np.diag(betas[: m - 1], 1) + np.diag(betas[: m - 1], -1)
        lam, U = np.linalg.eigh(Tm)
        tau = remaining
        while True:
            y = U @ (np.exp(tau * lam) * U[0, :])
            err = 0.0 if happy else betas[m - 1] * abs(y[m - 1]) * nrm if len(betas) >= m else 0.0
            if happy or err <= tol * max(abs(tau) / total, 1.0e-3) or abs(tau) <= 1.0e-12 * total:
                break
            tau = tau / 2
        x = nrm * (V[:, :m] @ y)
        remaining = remaining - tau
        if abs(remaining) <= 1.0e-14 * total:
            break
    return np.reshape(x, shape, order="F")


def _tdvp_real_or_complex_t(z):
    """_real_or_complex_t: a complex number without imaginary part becomes real  (tdvp.jl:22)"""
    z = complex(z)
    return z.real if z.imag == 0.0 else z


def _tdvp_envs(A_lsr, M_asbs, Tc):
    """F[0] = F[N+1] = ones(1,1,1); F[k+1] = _update_right_env(A[k], M[k], F[k+2]) for k = N..1  (tdvp.jl:55-62)"""
    N = len(A_lsr)
    F = [None] * (N + 2)
    F[0] = np.ones((1, 1, 1), dtype=Tc)
    F[N + 1] = np.ones((1, 1, 1), dtype=Tc)
    for k in range(N - 1, -1, -1):
        F[k + 1] = tdvp_update_right_env(A_lsr[k], M_asbs[k], F[k + 2])
    return F


def _tdvp_sync(psi: TTvector, A_lsr) -> TTvector:
    """cores back to (s, l, r), ranks from the arrays, ttv_ot zeroed  (_sync_ranks_from_lsr!, tdvp.jl:8-18, :147-151)"""
    N = psi.N
    psi.ttv_vec = [np.transpose(A_lsr[k], (1, 0, 2)).copy() for k in range(N)]
    psi.ttv_rks = [A_lsr[k].shape[0] for k in range(N)] + [A_lsr[N - 1].shape[2]]
    psi.ttv_ot = [0] * N
    return psi


def tdvp1sweep_(dt, psi: TTvector, H: TToperator, F=None, **kw):
    """tdvp1sweep! (tdvp.jl:45-152): one left-to-right and one right-to-left pass of one-site TDVP.  Mutates and returns (psi, F)."""
    Tc = np.complex128 if (isinstance(dt, complex) or np.iscomplexobj(psi.ttv_vec[0])) else np.float64
    N = psi.N
    A = [np.transpose(psi.ttv_vec[k], (1, 0, 2)) for k in range(N)]
    M = [np.transpose(H.tto_vec[k], (2, 0, 3, 1)) for k in range(N)]
    F = _tdvp_envs(A, M, Tc) if F is None else [np.asarray(f, dtype=Tc) for f in F]
    AC = A[0].astype(Tc)
    tm = _tdvp_real_or_complex_t(-1j * complex(dt))
    tp = _tdvp_real_or_complex_t(+1j * complex(dt))
    for k in range(N - 1):
        AC = tdvp_exponentiate(lambda x: tdvp_applyH1_lsr(x, F[k], F[k + 2], M[k]), tm, AC, **kw)
        Dl, d, Dr = AC.shape
        Q, R = sla.qr(np.reshape(AC, (Dl * d, Dr), order="F"), mode="economic")
        r = min(Dl * d, Dr)
        AL = np.reshape(Q[:, :r], (Dl, d, r), order="F")
        A[k] = AL
        F[k + 1] = tdvp_update_left_env(AL, M[k], F[k])
        C = tdvp_exponentiate(lambda x: tdvp_applyH0(x, F[k + 1], F[k + 2]), tp, R[:r, :], **kw)
        AC = np.einsum("ag,gsb->asb", C, A[k + 1])
    k = N - 1
    AC = tdvp_exponentiate(lambda x: tdvp_applyH1_lsr(x, F[k], F[k + 2], M[k]), tm, AC, **kw)
    for k in range(N - 2, -1, -1):
        Dl, d, Dr = AC.shape
        Amat = np.reshape(AC, (Dl, d * Dr), order="F")
        Q, R = sla.qr(Amat.conj().T, mode="economic")
        r = min(Dl, d * Dr)
        L = R[:r, :].conj().T
        A_r = np.reshape(Q[:, :r].conj().T, (r, d, Dr), order="F")
        A[k + 1] = A_r
        F[k + 2] = tdvp_update_right_env(A_r, M[k + 1], F[k + 3])
        C = tdvp_exponentiate(lambda x: tdvp_applyH0(x, F[k + 1], F[k + 2]), tp, L, **kw)
        AC = np.einsum("asg,gb->asb", A[k], C)
        AC = tdvp_exponentiate(lambda x: tdvp_applyH1_lsr(x, F[k], F[k + 2], M[k]), tm, AC, **kw)
    A[0] = AC
    return _tdvp_sync(psi, A), F


def tdvp2sweep_(dt, psi: TTvector, H: TToperator, F=None, max_bond: int = 2 ** 62, truncerr: float = 0.0, **kw):
    """tdvp2sweep! (tdvp.jl:210-301): two-site TDVP with SVD truncation, half steps forward on the pairs, back on the sites."""
    Tc = np.complex128 if (isinstance(dt, complex) or np.iscomplexobj(psi.ttv_vec[0])) else np.float64
    N = psi.N
    dth = complex(dt) / 2
    A = [np.transpose(psi.ttv_vec[k], (1, 0, 2)) for k in range(N)]
    M = [np.transpose(H.tto_vec[k], (2, 0, 3, 1)) for k in range(N)]
    F = _tdvp_envs(A, M, Tc) if F is None else [np.asarray(f, dtype=Tc) for f in F]
    AC = A[0].astype(Tc)
    tm = _tdvp_real_or_complex_t(-1j * dth)
    tp = _tdvp_real_or_complex_t(+1j * dth)
    for k in range(N - 1):
        AAC = np.einsum("asg,gtb->astb", AC, A[k + 1])
        AAC = tdvp_exponentiate(lambda x: tdvp_applyH2_lsr(x, F[k], F[k + 3], M[k], M[k + 1]), tm, AAC, **kw)
        Dl, d1, d2, Dr = AAC.shape
        U, s, Vt = svdtrunc(np.reshape(AAC, (Dl * d1, d2 * Dr), order="F"), max_bond=max_bond, truncerr=truncerr)
        AL = np.reshape(U, (Dl, d1, U.shape[1]), order="F")
        A[k] = AL
        F[k + 1] = tdvp_update_left_env(AL, M[k], F[k])
        AC = np.reshape(s[:, None] * Vt, (len(s), d2, Dr), order="F")
        if k < N - 2:
            AC = tdvp_exponentiate(lambda x: tdvp_applyH1_lsr(x, F[k + 1], F[k + 3], M[k + 1]), tp, AC, **kw)
    for k in range(N - 2, -1, -1):
        AAC = np.einsum("asg,gtb->astb", A[k], AC)
        AAC = tdvp_exponentiate(lambda x: tdvp_applyH2_lsr(x, F[k], F[k + 3], M[k], M[k + 1]), tm, AAC, **kw)
        Dl, d1, d2, Dr = AAC.shape
        U, s, Vt = svdtrunc(np.reshape(AAC, (Dl * d1, d2 * Dr), order="F"), max_bond=max_bond, truncerr=truncerr)
        AR = np.reshape(Vt, (Vt.shape[0], d2, Dr), order="F")
        A[k + 1] = AR
        F[k + 2] = tdvp_update_right_env(AR, M[k + 1], F[k + 3])
        AC = np.reshape(U * s[None, :], (Dl, d1, len(s)), order="F")
        if k > 0:
            AC = tdvp_exponentiate(lambda x: tdvp_applyH1_lsr(x, F[k], F[k + 2], M[k]), tp, AC, **kw)
    A[0] = AC
    return _tdvp_sync(psi, A), F


def _tdvp_complex(x: TTvector) -> TTvector:
    return TTvector(x.N, [c.astype(np.complex128) for c in x.ttv_vec], tuple(x.ttv_dims), list(x.ttv_rks), list(x.ttv_ot))


def _tdvp_complex_op(H: TToperator) -> TToperator:
    return TToperator(H.N, [c.astype(np.complex128) for c in H.tto_vec], tuple(H.tto_dims), list(H.tto_rks), list(H.tto_ot))


def _tdvp_driver(sweep, H: TToperator, u0: TTvector, steps, normalize=True, return_error=False, sweeps=1, carry_env=True,
                 imaginary_time=False, **kw):
    """tdvp / tdvp2 (tdvp.jl:154-203, :303-357).  A real train stays real in imaginary time (the reference stores the complex local
    arrays back into a real TTvector: their imaginary parts are zero)."""
    psi = orthogonalize(u0)
    real_out = imaginary_time and not np.iscomplexobj(u0.ttv_vec[0])
    if not imaginary_time:
        psi = _tdvp_complex(psi)
    Hc = _tdvp_complex_op(H) if (not imaginary_time and not np.iscomplexobj(H.tto_vec[0])) else H
    psi_prev = psi
    F = None
    for h in steps:
        psi_prev_step = copy_tt(psi)
        dt_eff = (1j * h) if imaginary_time else complex(h)
        for _ in range(sweeps):
            psi, F = sweep(dt_eff, psi, Hc, F if carry_env else None, **kw)
            if real_out:
                assert max(float(np.max(np.abs(np.imag(c)))) for c in psi.ttv_vec) <= 1e-12 * max(1.0, max(float(np.max(np.abs(c))) for c in psi.ttv_vec))
                psi.ttv_vec = [np.real(c).copy() for c in psi.ttv_vec]
        if normalize:
            psi = scale(1 / norm(psi), psi)
        psi = orthogonalize(psi)
        F = None
        psi_prev = psi_prev_step
    if return_error:
        h = steps[-1]
        if imaginary_time:
            residual = sub(scale(1 / h, sub(psi, psi_prev)), apply(Hc, psi))
        else:
            residual = add(scale(1 / h, sub(psi, psi_prev)), scale(1j, apply(Hc, psi)))
        return psi, norm(residual) / norm(psi)
    return psi


def tdvp(H, u0, steps, **kw):
    return _tdvp_driver(tdvp1sweep_, H, u0, steps, **kw)


def tdvp2(H, u0, steps, max_bond: int = 2 ** 62, truncerr: float = 0.0, **kw):
    return _tdvp_driver(lambda dt, psi, Hc, F, **k2: tdvp2sweep_(dt, psi, Hc, F, max_bond=max_bond, truncerr=truncerr, **k2), H, u0, steps, **kw)


# --------------------------------------------------------------------------------------
# Explicit time steppers (callers of the hot path) — src/solvers/euler.jl:76-97, :193-209
# --------------------------------------------------------------------------------------
def rk4_method(A: TToperator, u0: TTvector, steps, max_bond: int, normalize: bool = True) -> TTvector:
    u = u0
    for h in steps:
        k1 = apply(A, u)
        k2 = apply(A, tt_compress_(add(u, scale(h / 2, k1)), max_bond))
        k3 = apply(A, tt_compress_(add(u, scale(h / 2, k2)), max_bond))
        k4 = apply(A, tt_compress_(add(u, scale(h, k3)), max_bond))
        incr = scale(h / 6, tt_compress_(add(add(add(k1, scale(2, k2)), scale(2, k3)), k4), max_bond))
        u_new = tt_compress_(add(u, incr), max_bond)
        if normalize:
            u_new = scale(1 / math.sqrt(dot(u_new, u_new)), u_new)
        u = u_new
    return u


def euler_method(A: TToperator, u0: TTvector, steps, normalize: bool = True) -> TTvector:
    sol = u0
    for h in steps:
        update = apply(A, sol)
        sol = orthogonalize(add(sol, scale(h, update)))
        if normalize:
            sol = scale(1 / math.sqrt(dot(sol, sol)), sol)
    return sol
