"""Diagnostic (not a test): per-site s_memtime stamps of k_dot_fused (TTN_PROF=1).   python tools/diag_dot_prof.py [batch] [rank]"""
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
dx = T.DeviceTT((2,) * d, T.rand_tt((2,) * d, r, seed=30).ttv_rks, batch=B)
for b in range(B):
    dx.upload(b, T.rand_tt((2,) * d, r, seed=30 + b))
for _ in range(3):
    D.dot(dx, dx)
print("kernel ms", D.last_launch_ms())
L = T._lib.lib()
for b in sorted({0, B // 2, B - 1}):
    out = (C.c_int64 * 120)()
    T._lib.check(L.ttn_prof_steps(b, out))
    st = np.array(out[:d], dtype=np.int64)
    dt = np.diff(st)
    print(f"train {b}: total ticks (100 MHz) {st[-1] - st[0]}, per site (ticks):", dt.tolist())
    ph = (C.c_int64 * 64)()
    T._lib.check(L.ttn_prof_fine(b, ph))
    nfull = int(np.sum(dt > 30000))
    print("   full sites:", nfull, "; per site clk at: end of product 1 / end of product 2 (before the last adds) / adds issued / after barrier")
    for slot, w in enumerate((0, 1, 2, 3, 4, 8, 12, 15)):
        print("   wave", w, [int(ph[8 * slot + i] / max(nfull, 1)) for i in range(4)])
