"""Diagnostic: apply+round at ranks above 64 (merged short side p > 128: global-memory Jacobi fallback)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ttn_amd as T
from ttn_amd import device as D
T.ensure_init(0)
for d, r, B in ((20, 64, 4), (20, 96, 4), (20, 128, 4)):
    A = T.Delta(d); dA = T.DeviceTTO(A)
    x = T.rand_tt((2,) * d, r, seed=5)
    dx = T.DeviceTT.from_host(x, batch=B)
    dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x.ttv_rks)], batch=B)
    for it in range(2):
        D.sync(); t0 = time.perf_counter()
        D.apply_compress(dA, dx, dy, r); D.sync()
        t1 = time.perf_counter()
    sw = D.compress_status(dy)
    print(f"d={d} r={r} B={B}: {1e3*(t1-t0):8.1f} ms  jacobi sweeps {sw[0]}  -> {d/(t1-t0):.0f} cores/s per train-slot")
