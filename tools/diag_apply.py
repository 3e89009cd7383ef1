"""Diagnostic: k_apply alone on C3 trains (B = 256), HBM bytes / time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttn_amd as T
from ttn_amd import device as D
B, d, r = 256, 30, 64
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT.from_host(x, batch=B)
ycap = [a * c for a, c in zip(A.tto_rks, x.ttv_rks)]
dy = T.DeviceTT((2,) * d, ycap, batch=B)
for _ in range(3): D.apply(dA, dx, dy)
D.sync()
n = 20
t0 = time.perf_counter()
for _ in range(n): D.apply(dA, dx, dy)
D.sync()
t = (time.perf_counter() - t0) / n
bx = 8 * sum(2 * a * b for a, b in zip(x.ttv_rks[:-1], x.ttv_rks[1:]))
by = 8 * sum(2 * a * b for a, b in zip(ycap[:-1], ycap[1:]))
print(f"apply B={B}: {1e3*t:.3f} ms  -> {(bx+by)*B/t/1e12:.2f} TB/s algorithmic (x {bx/1e6:.2f} MB + y {by/1e6:.2f} MB per train)")
