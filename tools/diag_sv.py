"""Diagnostic: per-bond singular values GPU vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_oracle
d, r, seed = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (12, 16, 2)
T.ensure_init(0)
x = T.rand_tt((2,) * d, r, seed=seed)
A = T.Delta(d)
dA, dx = T.DeviceTTO(A), T.DeviceTT.from_host(x)
dy = T.DeviceTT(x.ttv_dims, [a * b for a, b in zip(A.tto_rks, x.ttv_rks)])
dy.capture_singular_values(True)
T.device.apply_compress(dA, dx, dy, r, 0.0, 1)
sw = T.device.compress_status(dy)
sv = []
ref = O.tt_compress_(O.apply(O.Delta(d), to_oracle(x)), r, svals_out=sv)
for i, s_ref in enumerate(sv):
    s = dy.singular_values(0, i)
    err = np.max(np.abs(s[:len(s_ref)] - s_ref)) / s_ref[0]
    print(i, len(s), "relerr %.2e" % err, "" if err < 1e-10 else "  <<<<<< " + str(s[:4]) + str(s_ref[:4]))
    if err > 1e-10:
        break

if os.environ.get("TTN_PROF"):
    import ctypes as C
    st = (C.c_int64 * 120)()
    T._lib.check(T._lib.lib().ttn_prof_steps(0, st))
    import struct
    print("dbg:", [struct.unpack("d", struct.pack("q", st[100 + i]))[0] for i in range(6)])
    print("per step:", " ".join(f"{v >> 32}:{v & 0xffffffff}" for v in st[:2 * (d - 1)]))
