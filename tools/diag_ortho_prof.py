"""Diagnostic (not a test): per-step s_memtime stamps of k_orthogonalize (TTN_PROF=1), right-to-left sweep (centre 1).
   python tools/diag_ortho_prof.py [batch] [rank]"""
import ctypes as C
import os
import sys

os.environ["TTN_PROF"] = "1"
sys.path.insert(0, ".")
import numpy as np
import ttn_amd as T
from ttn_amd import device as D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
r = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = 30
T.ensure_init(0)
x0 = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT((2,) * d, x0.ttv_rks, batch=B)
for b in range(B):
    dx.upload(b, T.rand_tt((2,) * d, r, seed=30 + b))
dy = T.DeviceTT((2,) * d, x0.ttv_rks, batch=B)
for _ in range(3):
    D.orthogonalize(dx, 1, dy)
D.sync()
print("kernel ms", D.last_launch_ms())
L = T._lib.lib()
for b in sorted({0, B - 1}):
    out = (C.c_int64 * 120)()
    T._lib.check(L.ttn_prof_steps(b, out))
    st = np.array(out[:d - 1], dtype=np.int64)
    print(f"train {b}: clk per step (sites d .. 2):", np.diff(st).tolist(), "total", int(st[-1] - st[0]))
    print(f"train {b}: last 1024-thread launch: start -> centre -> first product -> end (clk):", [int(out[101 + i] - out[100 + i]) for i in range(3)])
    ph = (C.c_int64 * 64)()
    T._lib.check(L.ttn_prof_fine(b, ph))
    print("   Cholesky-QR steps (general route), accumulated clk: Gram / load+chol+copy / trsm / Gram check+load / R out:", list(ph[:5]))
    print("   fused steps, accumulated clk: P0 FL image / P1 carry / P2 Gram / P3 Cholesky / P4 inverse / P5 apply / P6 check / P7 R out:", list(ph[8:16]))
    print("   k_ortho512 steps, accumulated clk: images / carry+Gram / dmax+padding / Cholesky / inverse / apply / check / second passes / stores:", list(ph[16:25]), "sum", sum(ph[16:25]))
    print("      inside Cholesky: diag block / barrier / panel + X row / barrier (then trailing + barrier = the rest):", list(ph[25:29]))
if os.environ.get("TTN_ORTHO512") == "1":
    import ctypes as C2
    fn = T._lib.lib().ttn_debug_ortho_state
    for b in sorted({0, B - 1}):
        o = (C2.c_int64 * 4)()
        fn(b, o)
        print(f"train {b}: three-launch state [next site, right buffer, left buffer, sites taken by k_ortho512] = {list(o)}")
    notdone = []
    for b in range(B):
        o = (C2.c_int64 * 4)()
        fn(b, o)
        if o[3] != 1:
            notdone.append((b, list(o)))
    print(f"trains k_ortho512 did not finish: {len(notdone)} of {B}", notdone[:8])
