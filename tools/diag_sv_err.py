"""Diagnostic: worst absolute singular-value error per bond step relative to sigma_1, device vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_oracle
d, r = int(sys.argv[1]), int(sys.argv[2])
x = T.rand_tt((2,) * d, r, seed=4)
A = T.Delta(d)
dA, dx = T.DeviceTTO(A), T.DeviceTT.from_host(x)
dy = T.DeviceTT(x.ttv_dims, [a * b for a, b in zip(A.tto_rks, x.ttv_rks)])
dy.capture_singular_values(True)
T.device.apply_compress(dA, dx, dy, r)
sv = []
ref = O.tt_compress_(O.apply(O.Delta(d), to_oracle(x)), r, svals_out=sv)
worst = []
for i, s_ref in enumerate(sv):
    s = dy.singular_values(0, i)[: len(s_ref)]
    worst.append((float(np.max(np.abs(s - s_ref)) / s_ref[0]), i, len(s_ref), float(s_ref[0] / max(s_ref[-1], 1e-300))))
worst.sort(reverse=True)
print("worst |ds|/s1 (value, step, n, cond):", [(f"{w:.1e}", i, n, f"{c:.1e}") for w, i, n, c in worst[:5]])
