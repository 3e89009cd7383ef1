"""Diagnostic: conditioning of the kept block of every bond step of the C3 sweep (sigma_1 / sigma_r) and the numerical rank.
   python tools/diag_kappa.py [d] [rank]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
d = int(sys.argv[1]) if len(sys.argv) > 1 else 30
r = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x = T.rand_tt((2,) * d, r, seed=30)
dx = T.DeviceTT.from_host(x)
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x.ttv_rks)])
dy.capture_singular_values(True)
T.device.apply_compress(dA, dx, dy, r, 0.0, 1)
T.device.compress_status(dy)
for i in range(2 * (d - 1)):
    s = dy.singular_values(0, i)
    s = s[s >= 0]
    k = min(len(s), r)
    nz = int(np.sum(s > 1e-13 * s[0]))
    print(f"step {i:2d}: p={len(s):3d} kept={k:3d} numerical rank {nz:3d}  s1/s_kept={s[0] / max(s[k - 1], 1e-300):9.2e}  s1/s_numrank={s[0] / s[nz - 1]:9.2e}")
