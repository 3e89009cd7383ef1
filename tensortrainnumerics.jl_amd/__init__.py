"""MI355X-native TT/QTT core-arithmetic backend for TensorTrainNumerics.jl's hot path.

Host-side mirror of the reference interface (tt.py), input generators (constructors.py),
device-resident batched handles (device.py) and the ctypes binding of the C ABI (_lib.py).
The arithmetic lives in csrc/*.h, csrc/ttn_api.hip -> libttn_hip.so (hand-written HIP, gfx950).
"""
from . import _lib, constructors, device, pipeline, qtt, shard, solvers, tdvp, tt
from ._lib import TTNError, build, ensure_init, finalize
from .constructors import (Delta, id_tto, portable_randn, qtt_cos, qtt_exp, qtt_sin, qtt_to_vector, rand_tt, shift,
                           toeplitz_to_qtto, zeros_tt, zeros_tto)
from .device import DeviceTT, DeviceTTO, StreamTimer
from .qtt import bubble_sort_swaps, hadamard_ttm, reorder, reorder_op, reorder_perm, ttv_decomp
from .tt import (TToperator, TTvector, _tt_bond_truncate_, add, add_, apply, apply_compress, div, dot, euclidean_distance, hadamard, norm,
                 orthogonalize, r_and_d_to_rks, scale, sub, tt_compress_)

__all__ = [n for n in dir() if not n.startswith("__")]
