"""Site-swap chains on the device: hadamard_ttm and QTT reorder.

Both are sequences of the two-site swap SVD (csrc/ttn_dense_kernels.h: wg_bond_step_io<1>, k_swap_chain) — the same Jacobi
SVD step as tt_compress!, with the physical indices of the two cores exchanged and factors U, S*Vt.  The index work (bubble
sort network of reorder, the op list of hadamard_ttm) is integer work and bit-exact with the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

from . import _lib
from .device import DeviceTT, compress_status
from .tt import TToperator, TTvector, _i64


def bubble_sort_swaps(perm: Sequence[int]) -> List[int]:
    """_bubble_sort_swaps (src/qtt_tools.jl:705-718): 1-based positions k of the adjacent swaps (k, k+1) that sort `perm`."""
    p = list(perm)
    out: List[int] = []
    for done in range(len(p)):
        for j in range(len(p) - 1 - done):
            if p[j] > p[j + 1]:
                p[j], p[j + 1] = p[j + 1], p[j]
                out.append(j + 1)
    return out


def reorder_perm(n_dims: int, bits_per_dim: int, new_ordering: str) -> List[int]:
    """Target position of every site (src/qtt_tools.jl:740-757): serial site (dim, bit) = dim*bits + bit, interleaved site
    = bit*n_dims + dim (0-based)."""
    assert new_ordering in ("interleaved", "serial"), "ordering must be :interleaved or :serial"
    perm = [0] * (n_dims * bits_per_dim)
    for dim in range(n_dims):
        for bit in range(bits_per_dim):
            ser, il = dim * bits_per_dim + bit, bit * n_dims + dim
            if new_ordering == "interleaved":
                perm[ser] = il
            else:
                perm[il] = ser
    return perm


# ---- device level ---------------------------------------------------------------------------------------------------
def swap_sites_(x: DeviceTT, swaps: Sequence[int], threshold: float = 0.0) -> DeviceTT:
    """In-place adjacent site swaps (1-based positions) on every train of the handle (src/qtt_tools.jl:762-769)."""
    sw = [int(k) for k in swaps]
    _lib.check(_lib.lib().ttn_swap_sites(x.h, len(sw), _i64(sw) if sw else None, float(threshold)))
    return x


def hadamard_ttm_(x: DeviceTT, y: DeviceTT, z: DeviceTT, tol: float = 1.0e-14, rmax: int = 2 ** 62, work_cap: int = 0) -> DeviceTT:
    """z = hadamard_ttm(x, y; tol, rmax) (src/tt_operations.jl:398-422) for every train pair of the two handles."""
    if work_cap <= 0:
        work_cap = 256 // int(x.dims[0])
    _lib.check(_lib.lib().ttn_hadamard_ttm(x.h, y.h, z.h, float(tol), int(min(rmax, 2 ** 62)), int(work_cap)))
    return z


# ---- host level (one train: upload, run, download) ----------------------------------------------------------------------
def _swap_capacity(d: int, cap: int) -> List[int]:
    """Uniform rank capacity: a swap SVD keeps min(n r_left, n r_right) directions (all of them when threshold == 0), which
    can exceed the minimal-TT bound prod(dims[:k]) of the bond — the chain is not in minimal form in between."""
    return [1] + [cap] * (d - 1) + [1]


def hadamard_ttm(x: TTvector, y: TTvector, tol: float = 1.0e-14, rmax: int = 2 ** 62) -> TTvector:
    assert tuple(x.ttv_dims) == tuple(y.ttv_dims), "Incompatible TT dimensions"
    n = int(x.ttv_dims[0])
    cap = 256 // n
    dx, dy = DeviceTT.from_host(x), DeviceTT.from_host(y)
    dz = DeviceTT(x.ttv_dims, _swap_capacity(x.N, cap))
    hadamard_ttm_(dx, dy, dz, tol, rmax, cap)
    compress_status(dz)
    dz.max_ranks()
    return dz.download(0)


def reorder(x: TTvector, n_dims: int, bits_per_dim: int, new_ordering: str, threshold: float = 0.0) -> TTvector:
    """reorder(q, new_ordering; threshold) (src/qtt_tools.jl:733-775) for the TTvector of a QTTvector with the given
    metadata (currently in the OTHER ordering).  Returns the reordered TTvector."""
    assert x.N == n_dims * bits_per_dim
    n = int(x.ttv_dims[0])
    dx = DeviceTT.from_host(x, cap_rks=_swap_capacity(x.N, 256 // n))
    swap_sites_(dx, bubble_sort_swaps(reorder_perm(n_dims, bits_per_dim, new_ordering)), threshold)
    compress_status(dx)
    dx.max_ranks()
    return dx.download(0)


def reorder_op(A: TToperator, n_dims: int, bits_per_dim: int, new_ordering: str, threshold: float = 0.0) -> TToperator:
    """reorder(A::QTToperator, new_ordering; threshold) (src/qtt_tools.jl:852-932).  An operator core (i, j, a, b) IS a vector
    core with the physical index i + n*j in memory, and _swap_adjacent_sites_op is _swap_adjacent_sites on that index: the
    operator rides the same kernel as a train with physical dimension n^2."""
    import numpy as np
    assert A.N == n_dims * bits_per_dim
    n = int(A.tto_dims[0])
    dims = tuple(n * n for _ in range(A.N))
    vec = TTvector(A.N, [np.reshape(np.asfortranarray(c), (n * n, c.shape[2], c.shape[3]), order="F") for c in A.tto_vec],
                   dims, list(A.tto_rks), [0] * A.N)
    dx = DeviceTT.from_host(vec, cap_rks=_swap_capacity(A.N, 256 // (n * n)))
    swap_sites_(dx, bubble_sort_swaps(reorder_perm(n_dims, bits_per_dim, new_ordering)), threshold)
    compress_status(dx)
    dx.max_ranks()
    out = dx.download(0)
    cores = [np.reshape(np.asfortranarray(c), (n, n, c.shape[1], c.shape[2]), order="F") for c in out.ttv_vec]
    return TToperator(A.N, cores, tuple(A.tto_dims), list(out.ttv_rks), [0] * A.N)


# ---- ttv_decomp ---------------------------------------------------------------------------------------------------------
def ttv_decomp_(z: DeviceTT, tensors, index: int = 1, tol: float = 1.0e-12) -> DeviceTT:
    """z_b = ttv_decomp(tensors[b]; index, tol) (src/tt_tools.jl:186-252) for a batch of dense tensors of shape z.dims."""
    import numpy as np
    arr = np.asarray(tensors, dtype=np.float64)
    assert arr.shape == (z.batch,) + tuple(z.dims), "tensors must have shape (batch, *dims)"
    flat = np.ascontiguousarray(np.stack([np.ravel(arr[b], order="F") for b in range(z.batch)]))
    _lib.check(_lib.lib().ttn_ttv_decomp(z.h, flat.ctypes.data_as(C.c_void_p), int(index), float(tol)))
    return z


def ttv_decomp(tensor, index: int = 1, tol: float = 1.0e-12, rank_cap: int = 1024) -> TTvector:
    """Host-level form for one tensor: capacity = the exact-rank bounds min(prod(dims[:k]), prod(dims[k:]), rank_cap)."""
    import numpy as np
    t = np.asarray(tensor, dtype=np.float64)
    dims = tuple(int(v) for v in t.shape)
    d = len(dims)
    cap = [1] * (d + 1)
    for k in range(1, d):
        left = int(np.prod(dims[:k], dtype=object))
        right = int(np.prod(dims[k:], dtype=object))
        cap[k] = max(1, min(left, right, rank_cap))
    z = DeviceTT(dims, cap)
    ttv_decomp_(z, t[None, ...], index, tol)
    compress_status(z)
    z.max_ranks()
    return z.download(0)
