import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
d, r = 30, 64
T.ensure_init(0)
x = T.rand_tt((2,) * d, r, seed=30)
A = T.Delta(d)
dA, dx = T.DeviceTTO(A), T.DeviceTT.from_host(x)
dy = T.DeviceTT(x.ttv_dims, [a * b for a, b in zip(A.tto_rks, x.ttv_rks)])
dy.capture_singular_values(True)
T.device.apply_compress(dA, dx, dy, r, 0.0, 1)
T.device.compress_status(dy)
for i in range(2 * (d - 1)):
    s = dy.singular_values(0, i)
    nz = s[s > 1e-13 * s[0]]
    print(i, len(s), "kappa(all) %.3g  kappa(nonzero %d) %.3g" % (s[0] / max(s[-1], 1e-300), len(nz), s[0] / nz[-1]))
