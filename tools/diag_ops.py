"""Diagnostic: time the secondary ops on C3-sized trains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttn_amd as T
from ttn_amd import device as D
d, r, B = 30, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 8
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT.from_host(x, batch=B)
ycap = [a * c for a, c in zip(A.tto_rks, x.ttv_rks)]
dy = T.DeviceTT((2,) * d, ycap, batch=B)
def timeit(name, fn, n=3):
    fn(); D.sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    D.sync()
    print(f"{name:28s} {1e3*(time.perf_counter()-t0)/n:9.3f} ms  (batch {B})")
timeit("apply", lambda: D.apply(dA, dx, dy))
timeit("dot(x,x)", lambda: D.dot(dx, dx))
timeit("dot(y,y) ranks 192", lambda: D.dot(dy, dy))
dz = T.DeviceTT((2,) * d, x.ttv_rks, batch=B)
timeit("orthogonalize(x) i=1", lambda: D.orthogonalize(dx, 1, dz))
dw = T.DeviceTT((2,) * d, ycap, batch=B)
timeit("orthogonalize(y) i=1 r192", lambda: D.orthogonalize(dy, 1, dw), n=1)
ds = T.DeviceTT((2,) * d, [min(2 * a, 10**9) if 0 < i < d else 1 for i, a in enumerate(x.ttv_rks)], batch=B)
timeit("add(x,x)", lambda: D.add(dx, dx, ds))
dh = T.DeviceTT((2,) * 12, [1] + [64] * 11 + [1], batch=B)
h1 = T.DeviceTT.from_host(T.rand_tt((2,) * 12, 8, seed=1), batch=B)
timeit("hadamard r8*r8 d=12", lambda: D.hadamard(h1, h1, dh))
timeit("compress(y)", lambda: (D.apply(dA, dx, dy), D.tt_compress_(dy, r)), n=2)
