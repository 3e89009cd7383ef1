"""Device-resident batches of trains (the handle half of include/ttn.h).

``DeviceTT`` wraps a ``ttn_tt`` handle: ``batch`` independent TT vectors with common dims and a
per-bond rank capacity, resident in HBM.  ``DeviceTTO`` wraps one TT operator.  Chains such as
``tt_compress!(A*x, r)`` then never cross PCIe.  All ops are asynchronous on the
library's HIP stream.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib
from .tt import TToperator, TTvector, _f, _i64, _ptrs


class DeviceTTO:
    def __init__(self, A: TToperator):
        _lib.ensure_init()
        self.dims = tuple(A.tto_dims)
        self.rks = list(A.tto_rks)
        self.N = A.N
        cores = [_f(c) for c in A.tto_vec]
        h = C.c_void_p()
        _lib.check(_lib.lib().ttn_tto_create(A.N, _i64(A.tto_dims), _i64(A.tto_rks), _ptrs(cores), C.byref(h)))
        self.h = h

    def free(self):
        if self.h:
            _lib.lib().ttn_tto_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceTT:
    def __init__(self, dims: Sequence[int], cap_rks: Sequence[int], batch: int = 1):
        _lib.ensure_init()
        self.dims = tuple(int(v) for v in dims)
        self.cap = [int(r) for r in cap_rks]
        self.N = len(self.dims)
        self.batch = int(batch)
        h = C.c_void_p()
        _lib.check(_lib.lib().ttn_tt_create(self.N, _i64(self.dims), _i64(self.cap), self.batch, C.byref(h)))
        self.h = h

    @classmethod
    def from_host(cls, x: TTvector, batch: int = 1, cap_rks: Sequence[int] | None = None) -> "DeviceTT":
        """Upload x as train 0 and replicate it over the batch."""
        t = cls(x.ttv_dims, cap_rks if cap_rks is not None else x.ttv_rks, batch)
        t.upload(0, x)
        if batch > 1:
            t.replicate(0)
        return t

    def upload(self, b: int, x: TTvector) -> None:
        cores = [_f(c) for c in x.ttv_vec]
        _lib.check(_lib.lib().ttn_tt_upload(self.h, int(b), _ptrs(cores), _i64(x.ttv_rks), _i64(x.ttv_ot)))

    def replicate(self, src: int = 0) -> None:
        _lib.check(_lib.lib().ttn_tt_replicate(self.h, int(src)))

    def ranks(self, b: int = 0):
        rks = (C.c_int64 * (self.N + 1))()
        ot = (C.c_int64 * self.N)()
        _lib.check(_lib.lib().ttn_tt_ranks(self.h, int(b), rks, ot))
        return [int(v) for v in rks], [int(v) for v in ot]

    def max_ranks(self):
        """Per-bond maximum of the current ranks over the batch (synchronises; also tightens the library's host-side
        rank bounds)."""
        out = (C.c_int64 * (self.N + 1))()
        _lib.check(_lib.lib().ttn_tt_max_ranks(self.h, out))
        return [int(v) for v in out]

    def download(self, b: int = 0) -> TTvector:
        rks, ot = self.ranks(b)
        cores = [np.zeros((self.dims[k], rks[k], rks[k + 1]), order="F") for k in range(self.N)]
        _lib.check(_lib.lib().ttn_tt_download(self.h, int(b), _ptrs(cores)))
        return TTvector(self.N, cores, self.dims, rks, ot)

    def free(self):
        if self.h:
            _lib.lib().ttn_tt_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    # parity instrumentation
    def capture_singular_values(self, on: bool = True) -> None:
        _lib.check(_lib.lib().ttn_sv_capture(self.h, 1 if on else 0))

    def singular_values(self, b: int, step: int, cap: int = 8192) -> np.ndarray:
        out = (C.c_double * cap)()
        n = C.c_int64(0)
        _lib.check(_lib.lib().ttn_sv_get(self.h, int(b), int(step), out, cap, C.byref(n)))
        return np.array(out[: n.value])


def apply(A: DeviceTTO, x: DeviceTT, y: DeviceTT) -> DeviceTT:
    _lib.check(_lib.lib().ttn_apply(A.h, x.h, y.h))
    return y


def compress_rank_bound(dims, rks, max_bond: int, sweeps: int = 1, k: int = 0):
    """(need, final) rank bounds of tt_compress! / _tt_bond_truncate! — see ttn_compress_rank_bound in include/ttn.h."""
    d = len(dims)
    need = (C.c_int64 * (d + 1))()
    fin = (C.c_int64 * (d + 1))()
    _lib.check(_lib.lib().ttn_compress_rank_bound(d, _i64(dims), _i64(rks), int(min(max_bond, 2 ** 62)), int(sweeps), int(k), need, fin))
    return [int(v) for v in need], [int(v) for v in fin]


def tt_compress_(psi: DeviceTT, max_bond: int, truncerr: float = 0.0, sweeps: int = 1) -> DeviceTT:
    assert sweeps >= 1, "sweeps must be >= 1"
    _lib.check(_lib.lib().ttn_compress(psi.h, int(min(max_bond, 2 ** 62)), float(truncerr), int(sweeps)))
    return psi


def apply_compress(A: DeviceTTO, x: DeviceTT, y: DeviceTT, max_bond: int, truncerr: float = 0.0, sweeps: int = 1) -> DeviceTT:
    _lib.check(_lib.lib().ttn_apply_compress(A.h, x.h, y.h, int(max_bond), float(truncerr), int(sweeps)))
    return y


def compress_status(psi: DeviceTT) -> List[int]:
    """Raises if any Jacobi SVD failed to converge; returns total Jacobi sweeps per train."""
    out = (C.c_int64 * psi.batch)()
    _lib.check(_lib.lib().ttn_compress_status(psi.h, out))
    return [int(v) for v in out]


def status_all() -> None:
    """Raises if any live handle, or a handle freed since the last query, carries a failure code (ttn_status_all): the ONE
    check (one stream sync) a chain of asynchronous ops needs per time step / iteration."""
    _lib.check(_lib.lib().ttn_status_all())


def dot(a: DeviceTT, b: DeviceTT) -> np.ndarray:
    out = (C.c_double * a.batch)()
    _lib.check(_lib.lib().ttn_dot(a.h, b.h, out))
    return np.array(out[:])


def norm(a: DeviceTT) -> np.ndarray:
    out = (C.c_double * a.batch)()
    _lib.check(_lib.lib().ttn_norm(a.h, out))
    return np.array(out[:])


def hadamard(x: DeviceTT, y: DeviceTT, z: DeviceTT) -> DeviceTT:
    _lib.check(_lib.lib().ttn_hadamard(x.h, y.h, z.h))
    return z


def add(x: DeviceTT, y: DeviceTT, z: DeviceTT) -> DeviceTT:
    _lib.check(_lib.lib().ttn_add(x.h, y.h, z.h))
    return z


def scale(a: float, x: DeviceTT, y: DeviceTT) -> DeviceTT:
    _lib.check(_lib.lib().ttn_scale(float(a), x.h, y.h))
    return y


def scale_batch(a, x: DeviceTT, y: DeviceTT) -> DeviceTT:
    """y_b = a[b] * x_b (one scalar per train)."""
    arr = (C.c_double * x.batch)(*[float(v) for v in a])
    _lib.check(_lib.lib().ttn_scale_batch(arr, x.h, y.h))
    return y


def orthogonalize(x: DeviceTT, i: int, y: DeviceTT) -> DeviceTT:
    _lib.check(_lib.lib().ttn_orthogonalize(x.h, int(i), y.h))
    return y


def last_launch_ms() -> float:
    """HIP-event time of the kernel of the last dot / norm / orthogonalize call alone (ttn_last_launch_ms)."""
    ms = C.c_float(0.0)
    _lib.check(_lib.lib().ttn_last_launch_ms(C.byref(ms)))
    return float(ms.value)


def sync() -> None:
    _lib.check(_lib.lib().ttn_sync())


class StreamTimer:
    """HIP-event timer on the library stream (the stream the kernels are launched on)."""

    def __enter__(self):
        _lib.check(_lib.lib().ttn_timer_begin())
        self.ms = None
        return self

    def __exit__(self, *exc):
        ms = C.c_float(0.0)
        _lib.check(_lib.lib().ttn_timer_end(C.byref(ms)))
        self.ms = float(ms.value)
        return False


def event_record(slot: int) -> None:
    _lib.check(_lib.lib().ttn_event_record(int(slot)))


def event_elapsed_ms(a: int, b: int) -> float:
    ms = C.c_float(0.0)
    _lib.check(_lib.lib().ttn_event_elapsed(int(a), int(b), C.byref(ms)))
    return float(ms.value)
