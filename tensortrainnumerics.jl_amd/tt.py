"""Host-side mirror of the reference's L1 interface for the hot path, on top of the C ABI.

Julia is not available in this image, so the host side that would be ``julia/TTNBackend.jl``
(shown in INTEGRATION.md) is mirrored here in Python with the reference's names, argument
meaning and error behaviour:

    TTvector / TToperator                    src/tt_tools.jl:23-29, :48-54
    A * v, A(v)                              src/tt_operations.jl:101-111, :151-157
    dot, norm, euclidean_distance            src/tt_operations.jl:239-250, :452-470
    hadamard (⊕)                             src/tt_operations.jl:343-363
    +, add!, scalar *, -, /                  src/tt_operations.jl:10-66, :256-295
    orthogonalize(x; i=1)                    src/tt_tools.jl:511-543
    _tt_bond_truncate!, tt_compress!         src/tt_tools.jl:743-789   (``!`` -> trailing ``_``)
    r_and_d_to_rks                           src/tt_tools.jl:407-425

Every arithmetic function calls libttn_hip.so; nothing here computes on the CPU.
Site numbers (``i``, ``k``) are 1-based like the reference.  Cores are numpy arrays of shape
``(n, r_left, r_right)`` (operator: ``(n, n, R_left, R_right)``) and are handed to the ABI in
column-major order, exactly the reference's memory layout.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib

log = logging.getLogger("TensorTrainNumerics")


def _i64(seq) -> "C.Array":
    seq = [int(v) for v in seq]
    return (C.c_int64 * len(seq))(*seq)


def _f(core: np.ndarray) -> np.ndarray:
    return np.asfortranarray(core, dtype=np.float64)


def _ptrs(arrs: Sequence[np.ndarray]):
    return (C.POINTER(C.c_double) * len(arrs))(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in arrs])


def _empty_cores(dims, rks) -> List[np.ndarray]:
    return [np.zeros((int(dims[k]), int(rks[k]), int(rks[k + 1])), order="F") for k in range(len(dims))]


class TTvector:
    """mutable struct TTvector{T,M} — src/tt_tools.jl:23-29 (fields keep the reference's names)."""

    def __init__(self, N: int, ttv_vec: List[np.ndarray], ttv_dims: Tuple[int, ...], ttv_rks: List[int], ttv_ot: List[int]):
        self.N = int(N)
        self.ttv_vec = list(ttv_vec)
        self.ttv_dims = tuple(int(v) for v in ttv_dims)
        self.ttv_rks = [int(r) for r in ttv_rks]
        self.ttv_ot = [int(o) for o in ttv_ot]

    def copy(self) -> "TTvector":
        return TTvector(self.N, [c.copy(order="F") for c in self.ttv_vec], self.ttv_dims, list(self.ttv_rks), list(self.ttv_ot))

    # arithmetic operators of the reference
    def __add__(self, other: "TTvector") -> "TTvector":
        return add(self, other)

    def __sub__(self, other: "TTvector") -> "TTvector":
        return sub(self, other)

    def __rmul__(self, a) -> "TTvector":
        return scale(a, self)

    def __mul__(self, a) -> "TTvector":
        return scale(a, self)

    def __truediv__(self, a) -> "TTvector":
        return div(self, a)


class TToperator:
    """struct TToperator{T,M} — src/tt_tools.jl:48-54."""

    def __init__(self, N: int, tto_vec: List[np.ndarray], tto_dims: Tuple[int, ...], tto_rks: List[int], tto_ot: List[int]):
        self.N = int(N)
        self.tto_vec = list(tto_vec)
        self.tto_dims = tuple(int(v) for v in tto_dims)
        self.tto_rks = [int(r) for r in tto_rks]
        self.tto_ot = [int(o) for o in tto_ot]

    def __mul__(self, v):
        if isinstance(v, TTvector):
            return apply(self, v)
        return NotImplemented

    def __call__(self, v: TTvector, *_):       # (A::TToperator)(x), src/tt_operations.jl:151-157
        return apply(self, v)


# ---------------------------------------------------------------------------------------------
def r_and_d_to_rks(rks: Sequence[int], dims: Sequence[int], rmax: int = 1024) -> List[int]:
    out = (C.c_int64 * len(rks))()
    _lib.check(_lib.lib().ttn_r_and_d_to_rks(len(dims), _i64(dims), len(rks), _i64(rks), int(rmax), out))
    return [int(v) for v in out]


def apply(A: TToperator, v: TTvector) -> TTvector:
    """*(A::TToperator, v::TTvector) — src/tt_operations.jl:101-111."""
    assert tuple(A.tto_dims) == tuple(v.ttv_dims), "Incompatible dimensions"
    d = v.N
    yr = [a * b for a, b in zip(A.tto_rks, v.ttv_rks)]
    Y = _empty_cores(v.ttv_dims, yr)
    Ac = [_f(c) for c in A.tto_vec]
    Xc = [_f(c) for c in v.ttv_vec]
    _lib.check(_lib.lib().ttn_apply_f64(d, _i64(v.ttv_dims), _ptrs(Ac), _i64(A.tto_rks), _ptrs(Xc), _i64(v.ttv_rks), _ptrs(Y)))
    return TTvector(d, Y, v.ttv_dims, yr, [0] * d)


def apply_compress(A: TToperator, v: TTvector, max_bond: int, truncerr: float = 0.0, sweeps: int = 1) -> TTvector:
    """tt_compress!(A * v, max_bond; truncerr, sweeps) — the operator krylov_linsolve and the time steppers iterate
    (src/solvers/euler.jl:55) — as ONE stateless call (ttn_apply_compress_f64): A * v is never materialised, neither in HBM nor over
    PCIe.  Same result as tt_compress_(apply(A, v), max_bond)."""
    assert tuple(A.tto_dims) == tuple(v.ttv_dims), "Incompatible dimensions"
    assert sweeps >= 1, "sweeps must be >= 1"
    d = v.N
    max_bond = int(min(max_bond, 2 ** 62))
    L = _lib.lib()
    cap = (C.c_int64 * (d + 1))()
    _lib.check(L.ttn_apply_compress_rank_bound(d, _i64(v.ttv_dims), _i64(A.tto_rks), _i64(v.ttv_rks), max_bond, int(sweeps), cap))
    bufs = [np.zeros(v.ttv_dims[j] * int(cap[j]) * int(cap[j + 1])) for j in range(d)]
    rks = (C.c_int64 * (d + 1))()
    Ac = [_f(c) for c in A.tto_vec]
    Xc = [_f(c) for c in v.ttv_vec]
    _lib.check(L.ttn_apply_compress_f64(d, _i64(v.ttv_dims), _ptrs(Ac), _i64(A.tto_rks), _ptrs(Xc), _i64(v.ttv_rks), _ptrs(bufs), rks,
                                        max_bond, float(truncerr), int(sweeps)))
    rk = [int(r) for r in rks]
    cores = [np.reshape(bufs[j][: v.ttv_dims[j] * rk[j] * rk[j + 1]], (v.ttv_dims[j], rk[j], rk[j + 1]), order="F").copy(order="F") for j in range(d)]
    return TTvector(d, cores, v.ttv_dims, rk, [0] * d)


def dot(A: TTvector, B: TTvector) -> float:
    """dot(A, B) — src/tt_operations.jl:239-250."""
    assert tuple(A.ttv_dims) == tuple(B.ttv_dims), "TT dimensions are not compatible"
    out = C.c_double(0.0)
    Ac = [_f(c) for c in A.ttv_vec]
    Bc = [_f(c) for c in B.ttv_vec]
    _lib.check(_lib.lib().ttn_dot_f64(A.N, _i64(A.ttv_dims), _ptrs(Ac), _i64(A.ttv_rks), _ptrs(Bc), _i64(B.ttv_rks), C.byref(out)))
    return float(out.value)


def norm(a: TTvector) -> float:
    """norm(a) — src/tt_operations.jl:465-470."""
    v = dot(a, a)
    v = 0.0 if v < 0 else v
    return math.sqrt(v)


def euclidean_distance(a: TTvector, b: TTvector) -> float:
    """src/tt_operations.jl:452-455."""
    assert tuple(a.ttv_dims) == tuple(b.ttv_dims), "TT dimensions must match"
    return math.sqrt(max(dot(a, a) - 2 * dot(b, a) + dot(b, b), 0.0))


def hadamard(x: TTvector, y: TTvector) -> TTvector:
    """hadamard(x, y) / ⊕ — src/tt_operations.jl:343-363."""
    assert tuple(x.ttv_dims) == tuple(y.ttv_dims), "Incompatible TT dimensions"
    d = x.N
    zr = [a * b for a, b in zip(x.ttv_rks, y.ttv_rks)]
    Z = _empty_cores(x.ttv_dims, zr)
    Xc = [_f(c) for c in x.ttv_vec]
    Yc = [_f(c) for c in y.ttv_vec]
    _lib.check(_lib.lib().ttn_hadamard_f64(d, _i64(x.ttv_dims), _ptrs(Xc), _i64(x.ttv_rks), _ptrs(Yc), _i64(y.ttv_rks), _ptrs(Z)))
    return TTvector(d, Z, x.ttv_dims, zr, [0] * d)


def add(x: TTvector, y: TTvector) -> TTvector:
    """+(x, y) — src/tt_operations.jl:10-35."""
    assert tuple(x.ttv_dims) == tuple(y.ttv_dims), "Incompatible dimensions"
    d = x.N
    zr = [a + b for a, b in zip(x.ttv_rks, y.ttv_rks)]
    zr[0] = 1
    zr[d] = 1
    Z = _empty_cores(x.ttv_dims, zr)
    Xc = [_f(c) for c in x.ttv_vec]
    Yc = [_f(c) for c in y.ttv_vec]
    _lib.check(_lib.lib().ttn_add_f64(d, _i64(x.ttv_dims), _ptrs(Xc), _i64(x.ttv_rks), _ptrs(Yc), _i64(y.ttv_rks), _ptrs(Z)))
    return TTvector(d, Z, x.ttv_dims, zr, [0] * d)


def add_(x: TTvector, y: TTvector) -> TTvector:
    """add!(x, y) — src/tt_operations.jl:37-66: rebinds x's fields and returns x."""
    z = add(x, y)
    x.ttv_vec, x.ttv_rks, x.ttv_ot = z.ttv_vec, z.ttv_rks, z.ttv_ot
    return x


def scale(a: float, A: TTvector) -> TTvector:
    """*(a::Number, A::TTvector) — src/tt_operations.jl:256-266."""
    d = A.N
    Y = _empty_cores(A.ttv_dims, A.ttv_rks)
    Xc = [_f(c) for c in A.ttv_vec]
    yot = (C.c_int64 * d)()
    _lib.check(_lib.lib().ttn_scale_f64(d, _i64(A.ttv_dims), float(a), _ptrs(Xc), _i64(A.ttv_rks), _i64(A.ttv_ot), _ptrs(Y), yot))
    return TTvector(d, Y, A.ttv_dims, list(A.ttv_rks), [int(v) for v in yot])


def sub(A: TTvector, B: TTvector) -> TTvector:
    """-(A, B) = (-1.0)*B + A — src/tt_operations.jl:285-287."""
    return add(scale(-1.0, B), A)


def div(A: TTvector, a: float) -> TTvector:
    """/(A, a) = (1/a)*A — src/tt_operations.jl:293-295."""
    return scale(1 / a, A)


def orthogonalize(x_tt: TTvector, i: int = 1) -> TTvector:
    """orthogonalize(x_tt; i=1) — src/tt_tools.jl:511-543 (non-mutating)."""
    d = x_tt.N
    assert 1 <= i <= d, "Impossible orthogonalization"
    Y = _empty_cores(x_tt.ttv_dims, x_tt.ttv_rks)      # max-size buffers: output ranks never exceed the input's
    Xc = [_f(c) for c in x_tt.ttv_vec]
    yr = (C.c_int64 * (d + 1))()
    yot = (C.c_int64 * d)()
    _lib.check(_lib.lib().ttn_orthogonalize_f64(d, _i64(x_tt.ttv_dims), _ptrs(Xc), _i64(x_tt.ttv_rks), int(i), _ptrs(Y), yr, yot))
    yr = [int(v) for v in yr]
    cores = _rewrap(Y, x_tt.ttv_dims, yr)
    return TTvector(d, cores, x_tt.ttv_dims, yr, [int(v) for v in yot])


def _rewrap(bufs: List[np.ndarray], dims, rks) -> List[np.ndarray]:
    """Max-size buffers hold the compact cores of the NEW ranks at their start; re-wrap them as
    exact-size arrays so that size(core) == (n, r_l, r_r) as the reference's tests require."""
    out = []
    for k, buf in enumerate(bufs):
        n, rl, rr = int(dims[k]), int(rks[k]), int(rks[k + 1])
        flat = buf.reshape(-1, order="F")[: n * rl * rr]
        out.append(np.array(flat.reshape((n, rl, rr), order="F"), order="F"))
    return out


def _tt_bond_truncate_(psi: TTvector, k: int, max_bond: int = 2 ** 62, truncerr: float = 0.0) -> TTvector:
    """_tt_bond_truncate!(psi, k; max_bond, truncerr) — src/tt_tools.jl:743-770.  Mutates cores k, k+1
    and ttv_rks[k+1]; ttv_ot untouched.  Returns orthogonalize(psi; i=k) like the reference (:769)."""
    assert 1 <= k < psi.N, "k must be in 1:(N-1)"
    _compress_call(psi, k, max_bond, truncerr, 1)
    return orthogonalize(psi, i=k)


def tt_compress_(psi: TTvector, max_bond: int, truncerr: float = 0.0, sweeps: int = 1, verbose: bool = False) -> TTvector:
    """tt_compress!(psi, max_bond; truncerr=0.0, sweeps=1, verbose=false) — src/tt_tools.jl:772-789.
    Returns the SAME object.  The per-bond `orthogonalize` whose value the reference discards
    (:769, :779, :785) is not computed."""
    assert sweeps >= 1, "sweeps must be >= 1"
    if verbose:
        # the reference logs per sweep (tt_tools.jl:775-783); the whole call is one kernel here
        for sw in range(1, sweeps + 1):
            log.info("TT compress: sweep %d (L→R)", sw)
            log.info("TT compress: sweep %d (R→L)", sw)
    _compress_call(psi, 0, max_bond, truncerr, sweeps)
    return psi


def _compress_call(psi: TTvector, k: int, max_bond: int, truncerr: float, sweeps: int) -> None:
    d = psi.N
    max_bond = int(min(max_bond, 2 ** 62))
    L = _lib.lib()
    # a rank-deficient bond can grow up to min(n r_left, n r_right, max_bond) (the reference keeps
    # min(length(s), max_bond) singular values): size the in/out buffers for that
    need = (C.c_int64 * (d + 1))()
    _lib.check(L.ttn_compress_rank_bound(d, _i64(psi.ttv_dims), _i64(psi.ttv_rks), max_bond, int(sweeps), int(k), need, None))
    bufs = []
    for j in range(d):
        buf = np.zeros(psi.ttv_dims[j] * int(need[j]) * int(need[j + 1]))
        buf[: psi.ttv_vec[j].size] = _f(psi.ttv_vec[j]).reshape(-1, order="F")
        bufs.append(buf)
    rks = _i64(psi.ttv_rks)
    if k > 0:
        rc = L.ttn_bond_truncate_f64(d, _i64(psi.ttv_dims), _ptrs(bufs), rks, int(k), max_bond, float(truncerr))
    else:
        rc = L.ttn_compress_f64(d, _i64(psi.ttv_dims), _ptrs(bufs), rks, max_bond, float(truncerr), int(sweeps))
    _lib.check(rc)
    new_rks = [int(v) for v in rks]
    cores = _rewrap(bufs, psi.ttv_dims, new_rks)
    for j in range(d):
        psi.ttv_vec[j] = cores[j]      # in-place slot assignment like the reference (:764-767)
    psi.ttv_rks[:] = new_rks
