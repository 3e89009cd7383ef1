"""Diagnostic: per bond step, the deviation of the captured singular values from the oracle's for some bench seeds (both builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_oracle, tt_rel_diff
seeds = [int(s) for s in sys.argv[1:]] or [492, 534, 30]
d, r = 30, 64
T.ensure_init(0)
A = T.Delta(d); dA = T.DeviceTTO(A)
xs = [T.rand_tt((2,) * d, r, seed=s) for s in seeds]
dx = T.DeviceTT((2,) * d, xs[0].ttv_rks, batch=len(seeds))
for b, x in enumerate(xs): dx.upload(b, x)
dy = T.DeviceTT((2,) * d, [a * c for a, c in zip(A.tto_rks, xs[0].ttv_rks)], batch=len(seeds))
dy.capture_singular_values(True)
T.device.apply_compress(dA, dx, dy, r, 0.0, 1); T.device.compress_status(dy)
for b, sd in enumerate(seeds):
    sv = []
    ref = O.tt_compress_(O.apply(O.Delta(d), to_oracle(xs[b])), r, svals_out=sv)
    got = dy.download(b)
    rows = []
    for i, s_ref in enumerate(sv):
        s = dy.singular_values(b, i)[: len(s_ref)]
        dev = np.abs(s - s_ref) / s_ref[0]
        j = int(np.argmax(dev))
        rows.append((dev[j], i, j, s_ref[j] / s_ref[0], s_ref[0] / s_ref[min(len(s_ref), r) - 1]))
    rows.sort(reverse=True)
    print(f"seed {sd}: tensor rel diff {tt_rel_diff(to_oracle(got), ref):.2e}; worst steps (abs dev/sigma1, step, index, s/sigma1, kappa_kept):",
          " ".join(f"({a:.1e},{i},{j},{q:.1e},{k:.1e})" for a, i, j, q, k in rows[:5]))
