"""Diagnostic: per-bond-step relative error of the kept singular values of the fused apply+round against the CPU oracle.
   python tools/diag_sv_error.py [d] [rank] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_oracle, tt_rel_diff
d = int(sys.argv[1]) if len(sys.argv) > 1 else 30
r = int(sys.argv[2]) if len(sys.argv) > 2 else 64
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 30
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
x = T.rand_tt((2,) * d, r, seed=seed)
dx = T.DeviceTT.from_host(x)
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, x.ttv_rks)])
dy.capture_singular_values(True)
T.device.apply_compress(dA, dx, dy, r, 0.0, 1)
T.device.compress_status(dy)
sv = []
ref = O.tt_compress_(O.apply(O.Delta(d), to_oracle(x)), r, svals_out=sv)
worst = 0.0
for i, s_ref in enumerate(sv):
    s = dy.singular_values(0, i)[: len(s_ref)]
    big = s_ref > 1e-12 * s_ref[0]
    err = float(np.max(np.abs(s[big] - s_ref[big]) / s_ref[big]))
    worst = max(worst, err)
    if err > 1e-12 or i == 24:
        print(f"step {i}: p={len(s_ref)} max rel err of kept singular values {err:.2e} (s1/s_min_kept {s_ref[0] / s_ref[big][-1]:.1e})")
print(f"worst {worst:.2e}; tensor rel diff {tt_rel_diff(to_oracle(dy.download()), ref):.2e}")
