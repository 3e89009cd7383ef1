"""Diagnostic: apply+round on a COMPRESSIBLE C3-shaped input (qtt_sin + 1e-3 * random rank-62), routes taken and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
from ttn_amd import device as D
from oracle import tt_oracle as O
from tests.helpers import to_oracle, tt_rel_diff
d, r = 30, 64
T.ensure_init(0)
noise = T.rand_tt((2,) * d, 62, seed=31)
x = T.qtt_sin(d, lam=3.0) + 1e-3 * noise
print("x ranks", x.ttv_rks)
A = T.Delta(d); dA = T.DeviceTTO(A)
dx = T.DeviceTT.from_host(x)
yr = [a * c for a, c in zip(A.tto_rks, x.ttv_rks)]
need, _ = D.compress_rank_bound((2,) * d, yr, r)
print('y ranks', yr); print('need   ', need)
dy = T.DeviceTT((2,) * d, need)
for it in range(2):
    D.apply(dA, dx, dy); D.sync()
    t0 = time.perf_counter(); D.tt_compress_(dy, r); D.sync(); t1 = time.perf_counter()
    sw = D.compress_status(dy)
    print(f"compress {1e3*(t1-t0):.1f} ms sweeps {sw}")
got = dy.download()
t0 = time.perf_counter()
ref = O.tt_compress_(O.apply(O.Delta(d), to_oracle(x)), r)
print("oracle %.1f ms" % (1e3 * (time.perf_counter() - t0)), "ranks equal", got.ttv_rks == ref.ttv_rks, "rel diff %.2e" % tt_rel_diff(to_oracle(got), ref))
if os.environ.get("TTN_PROF"):
    import ctypes as C
    st = (C.c_int64 * 120)()
    T._lib.check(T._lib.lib().ttn_prof_steps(0, st))
    print("per step:", " ".join(f"{'FGHD'[(v >> 48) & 3]}{(v >> 32) & 0xffff}:{v & 0xffffffff}" for v in st[:2 * (d - 1)]))
