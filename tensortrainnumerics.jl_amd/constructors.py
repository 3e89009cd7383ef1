"""Host-side input generators for the hot path: closed-form operator /
vector cores the reference builds once on the host, plus the portable seeded generator the
benchmark uses for "rank-r random TTvector" inputs.

    zeros_tt / zeros_tto            src/tt_operators.jl:548-573, :601-616
    toeplitz_to_qtto, Δ, shift      src/tt_operators.jl:4-19, :24, :283-285
    id_tto                          src/tt_operators.jl:519-532
    qtt_sin / qtt_cos / qtt_exp     src/qtt_tools.jl:116-175
    rand_tt                         src/tt_tools.jl:100-139 (randn replaced by a portable stream)
    qtt_to_vector                   src/qtt_tools.jl:57-71 (densifier used by small tests)

These are rank-<=5 closed forms evaluated once per problem: host code, not GPU work.
"""
from __future__ import annotations

import math
from typing import List, Sequence

import numpy as np

from .tt import TToperator, TTvector, r_and_d_to_rks


def zeros_tt(dims: Sequence[int], rks: Sequence[int], ot=None) -> TTvector:
    assert len(dims) + 1 == len(rks), "Dimensions and ranks are not compatible"
    d = len(dims)
    vec = [np.zeros((int(dims[i]), int(rks[i]), int(rks[i + 1])), order="F") for i in range(d)]
    return TTvector(d, vec, tuple(dims), list(rks), [0] * d if ot is None else list(ot))


def zeros_tto(dims: Sequence[int], rks: Sequence[int]) -> TToperator:
    assert len(dims) + 1 == len(rks), "Dimensions and ranks are not compatible"
    d = len(dims)
    vec = [np.zeros((int(dims[i]), int(dims[i]), int(rks[i]), int(rks[i + 1])), order="F") for i in range(d)]
    return TToperator(d, vec, tuple(dims), list(rks), [0] * d)


def _qtt_ranks(d: int, r: int) -> List[int]:
    """zeros_tt(2, d, r): r_and_d_to_rks(r*ones(d+1), (2,...,2)) — src/tt_operators.jl:560-569."""
    return r_and_d_to_rks([r] * (d + 1), (2,) * d)


def toeplitz_to_qtto(alpha: float, beta: float, gamma: float, d: int) -> TToperator:
    """QTT cores of the Toeplitz matrix alpha*I + beta*(super-diagonal) + gamma*(sub-diagonal)."""
    rks = r_and_d_to_rks([3] * (d + 1), (4,) * d, rmax=3)          # zeros_tto(2, d, 3): dims.^2, rmax = r
    out = zeros_tto((2,) * d, rks)
    Id = np.eye(2)
    J = np.array([[0.0, 1.0], [0.0, 0.0]])
    first = np.stack([Id, J.T, J], axis=-1)                          # [i, j, :] = (I, J', J)
    out.tto_vec[0][:, :, 0, :] = first
    mid = np.zeros((2, 2, 3, 3))
    mid[:, :, 0, 0], mid[:, :, 0, 1], mid[:, :, 0, 2] = Id, J.T, J
    mid[:, :, 1, 1] = J
    mid[:, :, 2, 2] = J.T
    for k in range(1, d - 1):
        out.tto_vec[k][...] = mid
    last = np.stack([alpha * Id + beta * J + gamma * J.T, gamma * J, beta * J.T], axis=-1)
    out.tto_vec[d - 1][:, :, :, 0] = last
    return out


def Delta(d: int) -> TToperator:
    """Δ(d): Dirichlet–Dirichlet Laplacian tridiag(-1, 2, -1)."""
    return toeplitz_to_qtto(2, -1, -1, d)


def shift(d: int) -> TToperator:
    return toeplitz_to_qtto(0, 1, 0, d)


def id_tto(d: int, n_dim: int = 2) -> TToperator:
    vec = [np.asfortranarray(np.eye(2).reshape(2, 2, 1, 1)) for _ in range(d)]
    return TToperator(d, vec, (n_dim,) * d, [1] * (d + 1), [0] * d)


def _trig_train(d, a, b, lam, first_of):
    out = zeros_tt((2,) * d, _qtt_ranks(d, 2))
    h = (b - a) / (2 ** d - 1)
    w = lam * math.pi
    for row, t in ((0, a), (1, a + h * 2 ** (d - 1))):
        out.ttv_vec[0][row, 0, :] = first_of(w * t)
    for k in range(2, d):
        th = w * (h * 2 ** (d - k))
        c, s = math.cos(th), math.sin(th)
        out.ttv_vec[k - 1][0] = np.eye(2)
        out.ttv_vec[k - 1][1] = [[c, -s], [s, c]]
    out.ttv_vec[d - 1][0, 0, 0] = 1.0
    out.ttv_vec[d - 1][1, :, 0] = [math.cos(w * h), math.sin(w * h)]
    return out


def qtt_sin(d: int, a: float = 0.0, b: float = 1.0, lam: float = 1.0) -> TTvector:
    return _trig_train(d, a, b, lam, lambda t: [math.sin(t), math.cos(t)])


def qtt_cos(d: int, a: float = 0.0, b: float = 1.0, lam: float = 1.0) -> TTvector:
    return _trig_train(d, a, b, lam, lambda t: [math.cos(t), -math.sin(t)])


def qtt_exp(d: int, a: float = 0.0, b: float = 1.0, alpha: float = 1.0, beta: float = 0.0) -> TTvector:
    out = zeros_tt((2,) * d, _qtt_ranks(d, 1))
    h = (b - a) / (2 ** d - 1)
    out.ttv_vec[0][:, 0, 0] = [math.exp(alpha * a + beta), math.exp(alpha * (a + h * 2 ** (d - 1)) + beta)]
    for k in range(2, d):
        out.ttv_vec[k - 1][:, 0, 0] = [1.0, math.exp(alpha * (h * 2 ** (d - k)))]
    out.ttv_vec[d - 1][:, 0, 0] = [1.0, math.exp(alpha * h)]
    return out


def qtt_to_vector(qtt: TTvector) -> np.ndarray:
    """Dense 2^d vector (site 1 = most significant bit).  Test/debug helper only (exponential cost)."""
    P = qtt.ttv_vec[0][:, 0, :]
    for k in range(1, qtt.N):
        G = qtt.ttv_vec[k]
        P = np.stack([P @ G[0], P @ G[1]], axis=1).reshape(2 * P.shape[0], G.shape[2])
    return P[:, 0].copy() if P.shape[1] == 1 else P.reshape(-1)


# ---------------------------------------------------------------------------------------------
# portable seeded N(0,1) stream: splitmix64 -> 53-bit uniforms -> Box–Muller.
# Same bits from any language that implements these 64-bit integer steps and IEEE double math
# with correctly rounded log/sqrt and the same cos/sin; the benchmark only needs determinism.
# ---------------------------------------------------------------------------------------------
def _splitmix64(n: int, seed: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def portable_randn(n: int, seed: int) -> np.ndarray:
    m = (n + 1) // 2
    bits = _splitmix64(2 * m, seed)
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)   # (0,1)
    r = np.sqrt(-2.0 * np.log(u[0::2]))
    th = 2.0 * math.pi * u[1::2]
    out = np.empty(2 * m)
    out[0::2] = r * np.cos(th)
    out[1::2] = r * np.sin(th)
    return out[:n]


def rand_tt(dims: Sequence[int], rks, seed: int = 0) -> TTvector:
    """rand_tt(dims, rks) / rand_tt(dims, rmax) — src/tt_tools.jl:100-139, entries i.i.d. N(0,1)
    in column-major core order from the portable stream (core k uses seed*1000003 + k)."""
    d = len(dims)
    if isinstance(rks, (int, np.integer)):
        rmax = int(rks)
        rks = r_and_d_to_rks([rmax] * (d + 1), dims, rmax=rmax)
    y = zeros_tt(dims, rks)
    for k in range(d):
        shape = (int(dims[k]), int(rks[k]), int(rks[k + 1]))
        y.ttv_vec[k] = portable_randn(shape[0] * shape[1] * shape[2], seed * 1000003 + k).reshape(shape, order="F")
    return y
