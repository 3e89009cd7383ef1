"""Diagnostic: phase counters of single bond steps of the C3 sweep (unfused: y = A x is materialised first).
   TTN_PROF=1 python tools/diag_step.py k [k2 ...]   (1-based bonds of the L->R half sweep)"""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
import ttn_amd as T
from ttn_amd import device as D

ks = [int(v) for v in sys.argv[1:]] or [6]
d, r = 30, 64
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x0 = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT.from_host(x0)
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x0.ttv_rks)])
names = ["merge", "scale", "LQ/chol", "jacobi", "sort", "split", "F:rest", "gramG", "F:grams", "F:chols", "F:jacobi", "G:eig"]
L = T._lib.lib()
for k in sorted(ks):
    D.apply(dA, dx, dy)
    if k > 1:
        T._lib.check(L.ttn_sweep(dy.h, 1, k - 1, r, 0.0))
    T._lib.check(L.ttn_sweep(dy.h, k, k, r, 0.0))
    D.sync()
    out = (C.c_int64 * 16)()
    T._lib.check(L.ttn_prof_get(0, out))
    st = (C.c_int64 * 120)()
    T._lib.check(L.ttn_prof_steps(0, st))
    v = st[0]
    print(f"bond {k}: {'FGHD'[(v >> 48) & 3]}{(v >> 32) & 0xffff}:{v & 0xfff}@{(v >> 12) & 0xfffff} kclk ", {n: int(t) // 1000 for n, t in zip(names, out) if t})
