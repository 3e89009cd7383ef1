"""ctypes binding of libttn_hip.so (the C ABI declared in include/ttn.h).

This is what a Julia ``ccall`` would bind.  There is NO CPU fallback:
if the shared library is missing or cannot be loaded the import fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTN_LIB") or os.path.join(_HERE, "libttn_hip.so")   # TTN_LIB: experiment builds
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

# error codes of include/ttn.h
TTN_OK = 0
TTN_ERR_DIMS = -1
TTN_ERR_BOND_INDEX = -2
TTN_ERR_SWEEPS = -3
TTN_ERR_CENTER = -4
TTN_ERR_CAPACITY = -5
TTN_ERR_ARG = -6
TTN_ERR_NOT_INIT = -7
TTN_ERR_UNSUPPORTED = -8
TTN_ERR_NO_CONVERGENCE = -9
TTN_ERR_SINGULAR = -10

i64 = C.c_int64
p_i64 = C.POINTER(C.c_int64)
p_f64 = C.POINTER(C.c_double)
pp_f64 = C.POINTER(C.POINTER(C.c_double))
handle = C.c_void_p
p_handle = C.POINTER(C.c_void_p)

# name -> (restype, argtypes): every symbol include/ttn.h declares
SIGNATURES = {
    "ttn_init": (C.c_int, [C.c_int]),
    "ttn_finalize": (C.c_int, []),
    "ttn_version": (C.c_char_p, []),
    "ttn_last_error_string": (C.c_char_p, []),
    "ttn_sync": (C.c_int, []),
    "ttn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ttn_tt_create": (C.c_int, [i64, p_i64, p_i64, i64, p_handle]),
    "ttn_tt_free": (C.c_int, [handle]),
    "ttn_tt_upload": (C.c_int, [handle, i64, pp_f64, p_i64, p_i64]),
    "ttn_tt_replicate": (C.c_int, [handle, i64]),
    "ttn_tt_ranks": (C.c_int, [handle, i64, p_i64, p_i64]),
    "ttn_tt_max_ranks": (C.c_int, [handle, p_i64]),
    "ttn_tt_download": (C.c_int, [handle, i64, pp_f64]),
    "ttn_tt_batch": (C.c_int, [handle, p_i64]),
    "ttn_tt_copy": (C.c_int, [handle, handle]),
    "ttn_tto_create": (C.c_int, [i64, p_i64, p_i64, pp_f64, p_handle]),
    "ttn_tto_free": (C.c_int, [handle]),
    "ttn_apply": (C.c_int, [handle, handle, handle]),
    "ttn_compress": (C.c_int, [handle, i64, C.c_double, i64]),
    "ttn_compress_status": (C.c_int, [handle, p_i64]),
    "ttn_status_all": (C.c_int, []),
    "ttn_compress_rank_bound": (C.c_int, [i64, p_i64, p_i64, i64, i64, i64, p_i64, p_i64]),
    "ttn_bond_truncate": (C.c_int, [handle, i64, i64, C.c_double]),
    "ttn_apply_compress": (C.c_int, [handle, handle, handle, i64, C.c_double, i64]),
    "ttn_sweep": (C.c_int, [handle, i64, i64, i64, C.c_double]),
    "ttn_apply_begin": (C.c_int, [handle, handle, handle]),
    "ttn_apply_sweep": (C.c_int, [handle, handle, handle, i64, i64, i64, C.c_double, C.c_int]),
    "ttn_stream_handle": (C.c_int, [C.POINTER(C.c_void_p)]),
    "ttn_tt_core_extent": (C.c_int, [handle, i64, p_i64, p_i64, p_i64]),
    "ttn_tt_core_export": (C.c_int, [handle, i64, C.c_void_p, C.c_void_p]),
    "ttn_tt_core_import": (C.c_int, [handle, i64, C.c_void_p, C.c_void_p, i64, i64]),
    "ttn_dot": (C.c_int, [handle, handle, p_f64]),
    "ttn_norm": (C.c_int, [handle, p_f64]),
    "ttn_last_launch_ms": (C.c_int, [C.POINTER(C.c_float)]),
    "ttn_debug_ortho_state": (C.c_int, [i64, p_i64]),
    "ttn_hadamard": (C.c_int, [handle, handle, handle]),
    "ttn_hadamard_ttm": (C.c_int, [handle, handle, handle, C.c_double, i64, i64]),
    "ttn_swap_sites": (C.c_int, [handle, i64, p_i64, C.c_double]),
    "ttn_ttv_decomp": (C.c_int, [handle, C.c_void_p, i64, C.c_double]),
    "ttn_als_linsolve": (C.c_int, [handle, handle, handle, handle, i64]),
    "ttn_mals_linsolve": (C.c_int, [handle, handle, handle, handle, C.c_double, i64]),
    "ttn_dmrg_linsolve": (C.c_int, [handle, handle, handle, handle, C.c_double, i64, p_i64, p_i64]),
    "ttn_dmrg_linsolve_it": (C.c_int, [handle, handle, handle, handle, C.c_double, i64, p_i64, p_i64, C.c_int, i64, C.c_double, i64]),
    "ttn_dmrg_cg_iterations": (C.c_int, [i64, p_i64]),
    "ttn_tdvp_apply_h1": (C.c_int, [C.c_int, i64, i64, i64, i64, i64, i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ttn_tdvp_apply_h0": (C.c_int, [C.c_int, i64, i64, i64, i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ttn_tdvp_update_left_env": (C.c_int, [C.c_int, i64, i64, i64, i64, i64, i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ttn_tdvp_update_right_env": (C.c_int, [C.c_int, i64, i64, i64, i64, i64, i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ttn_tdvp_apply_h2": (C.c_int, [C.c_int, i64, i64, i64, i64, i64, i64, i64, i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ttn_dense_qr": (C.c_int, [C.c_int, i64, i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ttn_dense_svd": (C.c_int, [C.c_int, i64, i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ttn_tdvp_contract_f64": (C.c_int, [C.c_int, C.c_int, i64, p_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ttn_selftest_eig128": (C.c_int, [C.c_void_p, i64, i64, i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ttn_add": (C.c_int, [handle, handle, handle]),
    "ttn_scale": (C.c_int, [C.c_double, handle, handle]),
    "ttn_scale_batch": (C.c_int, [p_f64, handle, handle]),
    "ttn_orthogonalize": (C.c_int, [handle, i64, handle]),
    "ttn_sv_capture": (C.c_int, [handle, C.c_int]),
    "ttn_sv_get": (C.c_int, [handle, i64, i64, p_f64, i64, p_i64]),
    "ttn_timer_begin": (C.c_int, []),
    "ttn_timer_end": (C.c_int, [C.POINTER(C.c_float)]),
    "ttn_bench_lds": (C.c_int, [C.c_int, i64, i64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ttn_bench_gemm": (C.c_int, [i64, i64, i64, C.c_int, C.c_int, i64, C.POINTER(C.c_int64)]),
    "ttn_selftest_gemm": (C.c_int, [i64, i64, i64, p_f64, p_f64, p_f64, C.c_double, C.c_double, C.c_int, C.c_int]),
    "ttn_prof_get": (C.c_int, [i64, p_i64]),
    "ttn_prof_steps": (C.c_int, [i64, p_i64]),
    "ttn_prof_fine": (C.c_int, [i64, p_i64]),
    "ttn_event_record": (C.c_int, [i64]),
    "ttn_event_elapsed": (C.c_int, [i64, i64, C.POINTER(C.c_float)]),
    "ttn_apply_f64": (C.c_int, [i64, p_i64, pp_f64, p_i64, pp_f64, p_i64, pp_f64]),
    "ttn_dot_f64": (C.c_int, [i64, p_i64, pp_f64, p_i64, pp_f64, p_i64, p_f64]),
    "ttn_hadamard_f64": (C.c_int, [i64, p_i64, pp_f64, p_i64, pp_f64, p_i64, pp_f64]),
    "ttn_add_f64": (C.c_int, [i64, p_i64, pp_f64, p_i64, pp_f64, p_i64, pp_f64]),
    "ttn_scale_f64": (C.c_int, [i64, p_i64, C.c_double, pp_f64, p_i64, p_i64, pp_f64, p_i64]),
    "ttn_orthogonalize_f64": (C.c_int, [i64, p_i64, pp_f64, p_i64, i64, pp_f64, p_i64, p_i64]),
    "ttn_compress_f64": (C.c_int, [i64, p_i64, pp_f64, p_i64, i64, C.c_double, i64]),
    "ttn_apply_compress_rank_bound": (C.c_int, [i64, p_i64, p_i64, p_i64, i64, i64, p_i64]),
    "ttn_apply_compress_f64": (C.c_int, [i64, p_i64, pp_f64, p_i64, pp_f64, p_i64, pp_f64, p_i64, i64, C.c_double, i64]),
    "ttn_bond_truncate_f64": (C.c_int, [i64, p_i64, pp_f64, p_i64, i64, i64, C.c_double]),
    "ttn_r_and_d_to_rks": (C.c_int, [i64, p_i64, i64, p_i64, i64, p_i64]),
}

_lib = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/ttn_api.hip for gfx950 into libttn_hip.so (in-tree).  hipcc cross-compiles
    without a GPU."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    srcs.append(os.path.join(INCLUDE, "ttn.h"))
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(s) for s in srcs)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
           "-o", LIB_PATH, os.path.join(CSRC, "ttn_api.hip"), os.path.join(CSRC, "ttn_wg512.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib() -> C.CDLL:
    """Load the shared library (once).  Fails loudly if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)       # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error() -> str:
    return lib().ttn_last_error_string().decode()


class TTNError(RuntimeError):
    pass


# messages of the reference's @assert sites (src/tt_operations.jl:11,102,240,344; src/tt_tools.jl:513,744,773)
_ASSERT_CODES = {TTN_ERR_DIMS, TTN_ERR_BOND_INDEX, TTN_ERR_SWEEPS, TTN_ERR_CENTER}


def check(rc: int) -> None:
    """Map a return code to the exception type the reference throws."""
    if rc == 0:
        return
    msg = last_error()
    if rc in _ASSERT_CODES:
        raise AssertionError(msg)
    raise TTNError(f"ttn error {rc}: {msg}")


_initialised = False


def ensure_init(device: int | None = None) -> None:
    global _initialised
    if _initialised and device is None:
        return
    if device is None:
        device = int(os.environ.get("TTN_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    check(lib().ttn_init(device))
    _initialised = True


def finalize() -> None:
    global _initialised
    if _lib is not None:
        _lib.ttn_finalize()
    _initialised = False
