"""Device-resident caller chains built from the hot-path ops (first step).

Mirrors of the explicit time steppers of the reference that are literally chains of
``A*u``, ``+``, scalar ``*``, ``tt_compress!`` / ``orthogonalize`` and ``dot``:

    euler_method(A, u0, steps; normalize)          src/solvers/euler.jl:76-97
    rk4_method(A, u0, steps, max_bond; normalize)   src/solvers/euler.jl:193-209

They run on ``DeviceTT`` batches (every train of the batch is an independent initial condition), never
leave HBM between ops, and return a new ``DeviceTT``.  The ``return_error`` branches and the implicit
steppers (which need ALS/DMRG/Krylov linear solves) are not built.
"""
from __future__ import annotations

import math
from typing import List, Sequence

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
        import ctypes as C
        from . import _lib
        _lib.check(_lib.lib().ttn_tt_copy(big.h, x.h))
        x.free()
        x = big
    D.tt_compress_(x, max_bond)
    return x, fin


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
