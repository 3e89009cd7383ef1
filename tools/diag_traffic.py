"""Diagnostic: one G128 (L->R) and one F128 (R->L) bond step as their own k_compress launches (ttn_sweep), so that
rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE attribute HBM bytes to a single step (dispatch order: see the prints)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttn_amd as T
from ttn_amd import device as D
B, d, r = 256, 30, 64
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT.from_host(x, batch=B)
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x.ttv_rks)], batch=B)
D.apply(dA, dx, dy)
L = T._lib.lib()
def sweep(a, b):
    T._lib.check(L.ttn_sweep(dy.h, a, b, r, 0.0)); D.sync()
order = [(1, 12, "L->R bonds 1..12"), (13, 13, "ONE G128 step (bond 13)"), (14, 29, "L->R bonds 14..29"),
         (29, 16, "R->L bonds 29..16"), (15, 15, "ONE F128 step (bond 15)"), (14, 1, "R->L bonds 14..1")]
for a, b, what in order:
    sweep(a, b)
    print("k_compress launch:", what)
print(dy.ranks(0)[0])
