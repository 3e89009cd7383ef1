"""Device-resident caller chains built from the hot-path ops (first step).

Mirrors of the explicit time steppers of the reference that are literally chains of
``A*u``, ``+``, scalar ``*``, ``tt_compress!`` / ``orthogonalize`` and ``dot``:

    euler_method(A, u0, steps; normalize)          src/solvers/euler.jl:76-97
    rk4_method(A, u0, steps, max_bond; normalize)   src/solvers/euler.jl:193-209

    krylov_linsolve(A, b, guess; max_bond, krylov_solver, ...)       src/solvers/euler.jl:34-74
    implicit_euler_method / crank_nicholson_method (tt_solver = "krylov")   src/solvers/euler.jl:98-190

They run on ``DeviceTT`` batches (every train of the batch is an independent initial condition / linear system),
never leave HBM between ops, and return a new ``DeviceTT``.  The ``return_error`` branches and the ALS/MALS/DMRG
solvers of the implicit steppers are not built.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Sequence

from . import _lib
from . import device as D
from .device import DeviceTT, DeviceTTO


def _ranks_of(x: DeviceTT) -> List[int]:
    """Per-bond maximum of the current ranks over the batch (host sync)."""
    return x.max_ranks()


def _apply(A: DeviceTTO, x: DeviceTT, xr: Sequence[int]):
    yr = [a * c for a, c in zip(A.rks, xr)]
    y = DeviceTT(x.dims, yr, x.batch)
    D.apply(A, x, y)
    return y, yr


def _axpy(x: DeviceTT, xr, a: float, y: DeviceTT, yr):
    """x + a*y as the reference evaluates it: scalar * first, then +."""
    ay = DeviceTT(y.dims, yr, y.batch)
    D.scale(a, y, ay)
    zr = [p + q for p, q in zip(xr, yr)]
    zr[0] = zr[-1] = 1
    z = DeviceTT(x.dims, zr, x.batch)
    D.add(x, ay, z)
    ay.free()
    return z, zr


def _compress(x: DeviceTT, xr, max_bond: int):
    """tt_compress!(x, max_bond) into a handle whose capacity also covers rank growth; returns (handle, rank bound)."""
    need, fin = D.compress_rank_bound(x.dims, xr, max_bond)
    if any(n > c for n, c in zip(need, x.cap)):
        big = DeviceTT(x.dims, need, x.batch)
        _lib.check(_lib.lib().ttn_tt_copy(big.h, x.h))
        x.free()
        x = big
    D.tt_compress_(x, max_bond)     # asynchronous: a failure code stays on the handle (or moves to the library word when the handle
    return x, fin                   # is freed) until the chain's one check per time step / iteration (D.status_all)


def _normalize(u: DeviceTT) -> None:
    nrm2 = D.dot(u, u)                                    # (1 / sqrt(dot(u, u))) * u   (euler.jl:83-85, :205-207)
    D.scale_batch([1.0 / math.sqrt(v) for v in nrm2], u, u)


def rk4_method(A: DeviceTTO, u0: DeviceTT, steps: Sequence[float], max_bond: int, normalize: bool = True) -> DeviceTT:
    """src/solvers/euler.jl:193-209 on a device-resident batch."""
    u, ur = u0, _ranks_of(u0)
    own = False
    for h in steps:
        k1, k1r = _apply(A, u, ur)
        t, tr = _axpy(u, ur, h / 2, k1, k1r)
        t, tr = _compress(t, tr, max_bond)
        k2, k2r = _apply(A, t, tr); t.free()
        t, tr = _axpy(u, ur, h / 2, k2, k2r)
        t, tr = _compress(t, tr, max_bond)
        k3, k3r = _apply(A, t, tr); t.free()
        t, tr = _axpy(u, ur, h, k3, k3r)
        t, tr = _compress(t, tr, max_bond)
        k4, k4r = _apply(A, t, tr); t.free()
        # k1 + 2k2 + 2k3 + k4, left to right like the reference
        s, sr = _axpy(k1, k1r, 2.0, k2, k2r)
        s2, s2r = _axpy(s, sr, 2.0, k3, k3r); s.free()
        s3, s3r = _axpy(s2, s2r, 1.0, k4, k4r); s2.free()
        for k in (k1, k2, k3, k4):
            k.free()
        s3, s3r = _compress(s3, s3r, max_bond)
        un, unr = _axpy(u, ur, h / 6, s3, s3r); s3.free()    # u + (h/6) * tt_compress!(...)
        un, unr = _compress(un, unr, max_bond)
        if normalize:
            _normalize(un)
        if own:
            u.free()
        u, ur, own = un, unr, True
        D.status_all()          # one check per time step: a non-converged SVD inside the step must not be committed silently
    return u


def euler_method(A: DeviceTTO, u0: DeviceTT, steps: Sequence[float], normalize: bool = True) -> DeviceTT:
    """src/solvers/euler.jl:76-97: solution = orthogonalize(solution + h * (A * solution)), optional normalisation."""
    u, ur = u0, _ranks_of(u0)
    own = False
    for h in steps:
        upd, updr = _apply(A, u, ur)
        t, tr = _axpy(u, ur, h, upd, updr); upd.free()
        un = DeviceTT(t.dims, tr, t.batch)
        D.orthogonalize(t, 1, un); t.free()
        unr = _ranks_of(un)
        if normalize:
            _normalize(un)
        if own:
            u.free()
        u, ur, own = un, unr, True
    return u


# ----------------------------------------------------------------------------------------------------------------------
# Krylov linear solves on handles: krylov_linsolve (src/solvers/euler.jl:34-74) and the implicit steppers
# that call it (implicit_euler_method :98-140, crank_nicholson_method :142-190) with tt_solver = "krylov".
#
# The reference hands `op`, `b`, `guess` to KrylovKit.linsolve (BiCGStab / GMRES / CG) — a third-party package that is not
# part of the reference tree (Project.toml compat 0.6.1, 0.9, 0.10), so its exact iteration (restart policy, breakdown
# handling) cannot be restated: what is mirrored here is the reference's OWN part — the operator
#     op = max_bond > 0 ? x -> tt_compress!(A*x, max_bond) : x -> A*x                              (euler.jl:55)
# the tolerance tol = max(atol, rtol*norm(b)) (:57), the solver selection (:56, :9-32) and the vector algebra of the
# VectorInterface extension (add = _round(beta*y + alpha*x) with _round = tt_compress!(., max_bond) during a bounded
# solve, orthogonalize otherwise; ext/...VectorInterfaceExt.jl:11-51) — around the textbook forms of the three Krylov
# methods.  Parity is therefore pinned where the reference's tests pin it: the solution against a dense solve
# (test/test_euler.jl:105-240: 1e-8 / 1e-7) and the rank bound; iterates are "parity unpinned".
# Every train of the batch is an independent system (same A, own right-hand side): scalars are per-train vectors.
# ----------------------------------------------------------------------------------------------------------------------
import numpy as np

from .tt import TToperator


def _tto_scale(a: float, A: TToperator) -> TToperator:
    """a * A for operators: scales the first core (src/tt_operations.jl:268-281)."""
    cores = [np.array(c, order="F") for c in A.tto_vec]
    cores[0] = a * cores[0]
    return TToperator(A.N, cores, A.tto_dims, list(A.tto_rks), list(A.tto_ot))


def _tto_add(A: TToperator, B: TToperator) -> TToperator:
    """A + B for operators: block concatenation of the cores (src/tt_operations.jl:71-96)."""
    assert A.tto_dims == B.tto_dims, "Incompatible dimensions"
    d = A.N
    cores, rks = [], [1]
    for k in range(d):
        a, b = A.tto_vec[k], B.tto_vec[k]
        n = a.shape[0]
        ral, rar, rbl, rbr = a.shape[2], a.shape[3], b.shape[2], b.shape[3]
        rl = 1 if k == 0 else ral + rbl
        rr = 1 if k == d - 1 else rar + rbr
        c = np.zeros((n, n, rl, rr), order="F")
        if d == 1:
            c[:] = a + b
        elif k == 0:
            c[:, :, 0, :rar] = a[:, :, 0, :]; c[:, :, 0, rar:] = b[:, :, 0, :]
        elif k == d - 1:
            c[:, :, :ral, 0] = a[:, :, :, 0]; c[:, :, ral:, 0] = b[:, :, :, 0]
        else:
            c[:, :, :ral, :rar] = a; c[:, :, ral:, rar:] = b
        cores.append(c)
        rks.append(rr)
    return TToperator(d, cores, A.tto_dims, rks, [0] * d)


class _Vec:
    """A DeviceTT batch together with the host-side rank bound the capacity arithmetic needs."""

    def __init__(self, h: DeviceTT, rks, own: bool = True):
        self.h, self.rks, self.own = h, list(rks), own

    def free(self):
        if self.own and self.h is not None:
            self.h.free()
            self.h = None


def _lin(a, x: _Vec, b, y: _Vec) -> _Vec:
    """a .* x + b .* y with per-train scalars (b*y + a*x as the extension writes it: scalar * first, then +)."""
    B = x.h.batch
    ax, by = DeviceTT(x.h.dims, x.rks, B), DeviceTT(y.h.dims, y.rks, B)
    D.scale_batch(np.broadcast_to(np.asarray(a, dtype=float), (B,)).copy(), x.h, ax)
    D.scale_batch(np.broadcast_to(np.asarray(b, dtype=float), (B,)).copy(), y.h, by)
    zr = [p + q for p, q in zip(x.rks, y.rks)]
    zr[0] = zr[-1] = 1
    z = DeviceTT(x.h.dims, zr, B)
    D.add(by, ax, z)
    ax.free(); by.free()
    return _Vec(z, zr)


def _copy(v: _Vec) -> _Vec:
    from . import _lib
    c = DeviceTT(v.h.dims, v.rks, v.h.batch)
    _lib.check(_lib.lib().ttn_tt_copy(c.h, v.h.h))
    return _Vec(c, v.rks)


def _scale(v: _Vec, coef) -> _Vec:
    """coef .* v (per-train scalars) into a fresh handle; same ranks, so nothing to round."""
    c = DeviceTT(v.h.dims, v.rks, v.h.batch)
    D.scale_batch(np.broadcast_to(np.asarray(coef, dtype=float), (v.h.batch,)).copy(), v.h, c)
    return _Vec(c, v.rks)


def _round(v: _Vec, round_rank: int) -> _Vec:
    """VectorInterface ext `_round`: tt_compress!(r, rk) during a bounded Krylov solve, orthogonalize(r) otherwise."""
    if round_rank > 0:
        h, fin = _compress(v.h, v.rks, round_rank)
        return _Vec(h, fin)
    o = DeviceTT(v.h.dims, v.rks, v.h.batch)
    D.orthogonalize(v.h, 1, o)
    v.free()
    return _Vec(o, _ranks_of(o))


def _vi_add(y: _Vec, x: _Vec, alpha, beta, round_rank: int) -> _Vec:
    """VectorInterface.add(y, x, alpha, beta) = _round(beta*y + alpha*x)."""
    return _round(_lin(alpha, x, beta, y), round_rank)


def _make_op(A: DeviceTTO, max_bond: int):
    def op(x: _Vec) -> _Vec:
        y, yr = _apply(A, x.h, x.rks)
        if max_bond > 0:
            y, yr = _compress(y, yr, max_bond)
        return _Vec(y, yr)
    return op


def _safe_div(num, den):
    num, den = np.asarray(num, dtype=float), np.asarray(den, dtype=float)
    return np.where(den != 0.0, num / np.where(den != 0.0, den, 1.0), 0.0)


def _bicgstab(op, b: _Vec, x: _Vec, tol, maxiter: int, rr: int) -> _Vec:
    """van der Vorst's BiCGStab, per-train scalars; trains that have converged keep iterating with zero updates."""
    Ax = op(x)
    r = _vi_add(b, Ax, -1.0, 1.0, rr); Ax.free()
    rhat = _Vec(r.h, r.rks, own=False)
    rhat_keep = r                                             # r is replaced below; keep the shadow residual alive
    r = _copy(rhat)
    B = b.h.batch
    rho = alpha = omega = np.ones(B)
    v = p = None
    for it in range(maxiter):
        if np.all(D.norm(r.h) <= tol):
            break
        rho_new = D.dot(rhat.h, r.h)
        if p is None:
            p = _copy(r)
        else:
            beta = _safe_div(rho_new, rho) * _safe_div(alpha, omega)
            t = _vi_add(p, v, -omega, 1.0, rr)                # p - omega v
            pn = _vi_add(r, t, beta, 1.0, rr); t.free(); p.free(); p = pn
        if v is not None:
            v.free()
        v = op(p)
        alpha = _safe_div(rho_new, D.dot(rhat.h, v.h))
        s = _vi_add(r, v, -alpha, 1.0, rr)
        t = op(s)
        tt_ = D.dot(t.h, t.h)
        omega = _safe_div(D.dot(t.h, s.h), tt_)
        xa = _vi_add(x, p, alpha, 1.0, rr); x.free()
        x = _vi_add(xa, s, omega, 1.0, rr); xa.free()
        rn = _vi_add(s, t, -omega, 1.0, rr); s.free(); t.free(); r.free(); r = rn
        rho = rho_new
        D.status_all()                                        # one status check per Krylov iteration
    for w in (r, p, v, rhat_keep):
        if w is not None:
            w.free()
    return x


def _cg(op, b: _Vec, x: _Vec, tol, maxiter: int, rr: int) -> _Vec:
    Ax = op(x)
    r = _vi_add(b, Ax, -1.0, 1.0, rr); Ax.free()
    p = _copy(r)
    rs = D.dot(r.h, r.h)
    for it in range(maxiter):
        if np.all(np.sqrt(np.maximum(rs, 0.0)) <= tol):
            break
        Ap = op(p)
        alpha = _safe_div(rs, D.dot(p.h, Ap.h))
        xn = _vi_add(x, p, alpha, 1.0, rr); x.free(); x = xn
        rn = _vi_add(r, Ap, -alpha, 1.0, rr); r.free(); Ap.free(); r = rn
        rs_new = D.dot(r.h, r.h)
        pn = _vi_add(r, p, _safe_div(rs_new, rs), 1.0, rr); p.free(); p = pn
        rs = rs_new
        D.status_all()                                        # one status check per Krylov iteration
    r.free(); p.free()
    return x


def _gmres(op, b: _Vec, x: _Vec, tol, krylovdim: int, maxiter: int, rr: int) -> _Vec:
    """Restarted GMRES(krylovdim), modified Gram-Schmidt, the small least-squares problems per train on the host."""
    B = b.h.batch
    for outer in range(maxiter):
        Ax = op(x)
        r = _vi_add(b, Ax, -1.0, 1.0, rr); Ax.free()
        beta = D.norm(r.h)
        if np.all(beta <= tol):
            r.free()
            break
        V = [_scale(r, _safe_div(1.0, beta))]; r.free()
        H = np.zeros((B, krylovdim + 1, krylovdim))
        m = 0
        for j in range(krylovdim):
            w = op(V[j])
            for i in range(j + 1):
                hij = D.dot(V[i].h, w.h)
                H[:, i, j] = hij
                wn = _vi_add(w, V[i], -hij, 1.0, rr); w.free(); w = wn
            hn = D.norm(w.h)
            H[:, j + 1, j] = hn
            m = j + 1
            if np.all(hn <= 1e-300) or j + 1 == krylovdim:
                w.free()
                break
            V.append(_scale(w, _safe_div(1.0, hn))); w.free()
        ycoef = np.zeros((B, m))
        for t in range(B):
            e1 = np.zeros(m + 1); e1[0] = beta[t]
            ycoef[t] = np.linalg.lstsq(H[t, : m + 1, :m], e1, rcond=None)[0]
        for j in range(m):
            xn = _vi_add(x, V[j], ycoef[:, j], 1.0, rr); x.free(); x = xn
        for v in V:
            v.free()
        D.status_all()                                        # one status check per restart cycle
    return x


def krylov_linsolve(A, b: DeviceTT, guess: DeviceTT, max_bond: int = 0, krylov_solver: str = "auto", krylovdim: int = 8,
                    maxiter: int = 20, rtol: float = 1.0e-8, atol: float = 1.0e-12, tol=None, issymmetric: bool = False,
                    ishermitian=None, isposdef: bool = False) -> DeviceTT:
    """src/solvers/euler.jl:34-74 on device-resident batches (A: TToperator or DeviceTTO; every train its own system)."""
    ishermitian = issymmetric if ishermitian is None else ishermitian
    dA = A if isinstance(A, DeviceTTO) else DeviceTTO(A)
    solver = "cg" if (krylov_solver == "auto" and isposdef and (issymmetric or ishermitian)) else krylov_solver   # :56
    if solver == "auto":
        solver = "bicgstab" if max_bond > 0 else "gmres"                                                             # :17
    if solver not in ("bicgstab", "gmres", "cg"):
        raise ValueError(f"Unknown Krylov solver: {krylov_solver}. Use :auto, :bicgstab, :cg, or :gmres.")           # :31
    bv = _Vec(b, _ranks_of(b), own=False)
    tol_value = np.maximum(atol, rtol * D.norm(b)) if tol is None else np.full(b.batch, float(tol))                   # :57
    x0 = _Vec(guess, _ranks_of(guess), own=False)
    x = _copy(x0)                                             # a working copy the iteration may free
    op = _make_op(dA, max_bond)
    if solver == "bicgstab":
        x = _bicgstab(op, bv, x, tol_value, maxiter, max_bond)
    elif solver == "cg":
        x = _cg(op, bv, x, tol_value, krylovdim * maxiter, max_bond)                                                 # :28
    else:
        x = _gmres(op, bv, x, tol_value, krylovdim, maxiter, max_bond)
    D.status_all()
    return x.h


def _implicit_stepper(A: TToperator, u0: DeviceTT, guess: DeviceTT, steps, normalize, tt_solver, max_bond, crank, kw) -> DeviceTT:
    if tt_solver != "krylov":
        if tt_solver in ("mals", "als", "dmrg"):
            raise NotImplementedError(f"tt_solver={tt_solver!r}: ALS/MALS/DMRG local solves are outside this backend (SURVEY §8 f1)")
        raise ValueError(f"Unknown TT solver: {tt_solver}")
    from .constructors import id_tto
    I = id_tto(A.N)
    sol, own = u0, False
    for h in steps:
        if crank:
            lhs = _tto_add(I, _tto_scale(-h / 2, A))                                  # I - (h/2) A      (:156)
            rhs_op = DeviceTTO(_tto_add(I, _tto_scale(h / 2, A)))                     # (I + (h/2) A) * solution
            rhs, _ = _apply(rhs_op, sol, _ranks_of(sol))
        else:
            lhs = _tto_add(I, _tto_scale(-h, A))                                      # M = I - h A      (:113)
            rhs = sol
        nxt = krylov_linsolve(lhs, rhs, guess, max_bond=max_bond, **kw)
        if crank:
            rhs.free()
        if normalize:
            D.scale_batch(1.0 / D.norm(nxt), nxt, nxt)                                 # next / norm(next)
        v = _round(_Vec(nxt, _ranks_of(nxt)), max_bond)                               # tt_compress!(next, max_bond) : orthogonalize(next)
        D.status_all()
        if own:
            sol.free()
        sol, own, guess = v.h, True, v.h
    return sol


def implicit_euler_method(A: TToperator, u0: DeviceTT, guess: DeviceTT, steps, normalize: bool = True, tt_solver: str = "krylov",
                          max_bond: int = 0, **kw) -> DeviceTT:
    """src/solvers/euler.jl:98-140 (tt_solver = "krylov"; the return_error branch is not built)."""
    return _implicit_stepper(A, u0, guess, steps, normalize, tt_solver, max_bond, False, kw)


def crank_nicholson_method(A: TToperator, u0: DeviceTT, guess: DeviceTT, steps, normalize: bool = True, tt_solver: str = "krylov",
                           max_bond: int = 0, **kw) -> DeviceTT:
    """src/solvers/euler.jl:142-190 (tt_solver = "krylov"; the return_error branch is not built)."""
    return _implicit_stepper(A, u0, guess, steps, normalize, tt_solver, max_bond, True, kw)


# ---------------------------------------------------------------------------------------------------------------------
# als_linsolve (src/solvers/als.jl:161-222) — csrc/ttn_als_kernels.h
# ---------------------------------------------------------------------------------------------------------------------
def als_linsolve_(A: DeviceTTO, b: DeviceTT, x0: DeviceTT, x: DeviceTT, sweep_count: int = 2) -> DeviceTT:
    """x_b = als_linsolve(A, b_b, x0_b; sweep_count) for every train of the batch; x keeps x0's ranks."""
    assert sweep_count >= 1, "sweep_count must be >= 1"
    _lib.check(_lib.lib().ttn_als_linsolve(A.h, b.h, x0.h, x.h, int(sweep_count)))
    return x


def als_linsolve(A: TToperator, b: TTvector, tt_start: TTvector, sweep_count: int = 2) -> TTvector:
    """Host-level form for one right-hand side (upload, solve on the device, download)."""
    dA = DeviceTTO(A)
    db, dx0 = DeviceTT.from_host(b), DeviceTT.from_host(tt_start)
    dx = DeviceTT(tt_start.ttv_dims, tt_start.ttv_rks)
    als_linsolve_(dA, db, dx0, dx, sweep_count)
    D.compress_status(dx)
    return dx.download(0)


# ---------------------------------------------------------------------------------------------------------------------
# mals_linsolve (src/solvers/mals.jl:240-312) — csrc/ttn_als_kernels.h (k_mals_linsolve)
# ---------------------------------------------------------------------------------------------------------------------
def mals_linsolve_(A: DeviceTTO, b: DeviceTT, x0: DeviceTT, x: DeviceTT, tol: float = 1.0e-12, rmax: int = 2 ** 30) -> DeviceTT:
    """x_b = mals_linsolve(A, b_b, x0_b; tol, rmax) for every train of the batch; x's capacity bounds the adapted ranks."""
    _lib.check(_lib.lib().ttn_mals_linsolve(A.h, b.h, x0.h, x.h, float(tol), int(min(rmax, 2 ** 30))))
    return x


def mals_capacity(dims, start_rks, rmax: int, limit: int = 2048):
    """Rank capacity for the result handle: min(rmax, prod(dims[:k]), prod(dims[k:])) like the reference's buffers
    (mals.jl:258, :23), at least the start ranks, lowered uniformly until every two-site system fits the device limit."""
    d = len(dims)
    full = [1] + [min(int(rmax), int(math.prod(dims[:k])), int(math.prod(dims[k:]))) for k in range(1, d)] + [1]
    cut = max(full)
    while True:
        cap = [max(min(f, cut), int(s)) for f, s in zip(full, start_rks)]
        worst = max(dims[i] * cap[i] * dims[i + 1] * cap[i + 2] for i in range(d - 1)) if d > 1 else 1
        if worst <= limit or cut <= 1:
            return cap
        cut -= 1


def mals_linsolve(A: TToperator, b: TTvector, tt_start: TTvector, tol: float = 1.0e-12, rmax: int | None = None) -> TTvector:
    """Host-level form for one right-hand side.  rmax defaults to round(sqrt(prod(dims))) like the reference (mals.jl:244)."""
    if rmax is None:
        rmax = int(round(math.sqrt(math.prod(tt_start.ttv_dims))))
    dA = DeviceTTO(A)
    db, dx0 = DeviceTT.from_host(b), DeviceTT.from_host(tt_start)
    dx = DeviceTT(tt_start.ttv_dims, mals_capacity(tt_start.ttv_dims, tt_start.ttv_rks, rmax))
    mals_linsolve_(dA, db, dx0, dx, tol, rmax)
    D.compress_status(dx)
    dx.max_ranks()
    return dx.download(0)


# ---------------------------------------------------------------------------------------------------------------------
# dmrg_linsolve, N = 2 (src/solvers/dmrg.jl:388-472) — the same persistent two-site kernel in its DMRG mode
# ---------------------------------------------------------------------------------------------------------------------
def dmrg_linsolve_(A: DeviceTTO, b: DeviceTT, x0: DeviceTT, x: DeviceTT, tol: float = 1.0e-12, sweep_schedule: Sequence[int] = (2,),
                   rmax_schedule: Sequence[int] | None = None, it_solver: bool = False, linsolv_maxiter: int = 200,
                   linsolv_tol: float | None = None, itslv_thresh: int = 2048) -> DeviceTT:
    """x_b = dmrg_linsolve(A, b_b, x0_b; N = 2, tol, sweep_schedule, rmax_schedule, it_solver, linsolv_maxiter, linsolv_tol,
    itslv_thresh) for every train of the batch; x's capacity bounds the adapted ranks.  Local systems: dense LU (the reference's
    it_solver = false branch, dmrg.jl:173-175) unless `it_solver` or the system has more than `itslv_thresh` unknowns — then
    matrix-free conjugate gradients (dmrg.jl:99-171).  The defaults here (it_solver False, itslv_thresh 2048) solve to rounding
    wherever the dense path reaches; the reference's own defaults are it_solver = True, itslv_thresh = 256."""
    if rmax_schedule is None:
        rmax_schedule = (math.isqrt(math.prod(x0.dims)),)                   # dmrg.jl:391
    if linsolv_tol is None:
        linsolv_tol = max(math.sqrt(tol), 1.0e-8)                           # dmrg.jl:394
    ss = [int(v) for v in sweep_schedule]
    rs = [int(min(v, 2 ** 30)) for v in rmax_schedule]
    if len(rs) < len(ss):
        raise _lib.TTNError("dmrg_linsolve: rmax_schedule is shorter than sweep_schedule")      # BoundsError in the reference
    n = len(ss)
    arr = (C.c_int64 * max(n, 1))
    _lib.check(_lib.lib().ttn_dmrg_linsolve_it(A.h, b.h, x0.h, x.h, float(tol), n, arr(*ss) if n else None, arr(*rs[:n]) if n else None,
                                               1 if it_solver else 0, int(linsolv_maxiter), float(linsolv_tol), int(itslv_thresh)))
    return x


def dmrg_cg_iterations(batch: int):
    """Total conjugate-gradient iterations per train of the last two-site solve (0 when every local system was solved densely)."""
    out = (C.c_int64 * batch)()
    _lib.check(_lib.lib().ttn_dmrg_cg_iterations(batch, out))
    return [int(v) for v in out]


def dmrg_capacity(dims, start_rks, rmax: int, dense_only: bool = False):
    """Rank capacity of the result handle of dmrg_linsolve: the reference's buffer bounds min(rmax, prod(dims[:k]), prod(dims[k:]))
    (dmrg.jl:411), at least the start ranks, clamped to what the SVD core moves take (n_i * rank <= 256: ranks saturate there
    instead of growing to rmax; only START ranks beyond it are refused).  dense_only: lowered until every two-site system fits
    the dense solver (2048 unknowns) like mals_capacity."""
    if dense_only:
        return mals_capacity(dims, start_rks, rmax)
    d = len(dims)
    cap = [1]
    for k in range(1, d):
        lim = 256 // max(int(dims[k - 1]), int(dims[k]))        # the device bound: n * rank <= 256 on both cores that share bond k
        if int(start_rks[k]) > lim:
            raise _lib.TTNError("dmrg_linsolve: a start rank with n_k * rank above 256 is not supported (ranks up to 128 for n = 2)")
        # the reference's buffer bound, CLAMPED to the device bound (so the default rmax_schedule = isqrt(prod(dims)) — 4096 for the
        # 24-site C5 problem — works: the ranks then saturate at 128 for n = 2, which is the documented rank limit of this backend)
        cap.append(max(min(int(rmax), int(math.prod(dims[:k])), int(math.prod(dims[k:])), lim), int(start_rks[k])))
    return cap + [1]


def dmrg_linsolve(A: TToperator, b: TTvector, tt_start: TTvector, tol: float = 1.0e-12, sweep_schedule: Sequence[int] = (2,),
                  rmax_schedule: Sequence[int] | None = None, N: int = 2, it_solver: bool = False, linsolv_maxiter: int = 200,
                  linsolv_tol: float | None = None, itslv_thresh: int = 2048) -> TTvector:
    """Host-level form for one right-hand side (same keywords as src/solvers/dmrg.jl:388-396; see dmrg_linsolve_ for the two
    defaults that differ)."""
    if N != 2:
        raise _lib.TTNError("dmrg_linsolve: only the two-site scheme N = 2 is offered (single-site: als_linsolve)")
    if rmax_schedule is None:
        rmax_schedule = (math.isqrt(math.prod(tt_start.ttv_dims)),)
    dA = DeviceTTO(A)
    db, dx0 = DeviceTT.from_host(b), DeviceTT.from_host(tt_start)
    rtop = max(int(v) for v in rmax_schedule)
    cap = dmrg_capacity(tt_start.ttv_dims, tt_start.ttv_rks, rtop)
    if not it_solver and max(tt_start.ttv_dims[i] * cap[i] * tt_start.ttv_dims[i + 1] * cap[i + 2] for i in range(len(cap) - 2)) <= 2048:
        cap = mals_capacity(tt_start.ttv_dims, tt_start.ttv_rks, rtop)
    dx = DeviceTT(tt_start.ttv_dims, cap)
    dmrg_linsolve_(dA, db, dx0, dx, tol, sweep_schedule, rmax_schedule, it_solver, linsolv_maxiter, linsolv_tol, itslv_thresh)
    D.compress_status(dx)
    dx.max_ranks()
    return dx.download(0)
