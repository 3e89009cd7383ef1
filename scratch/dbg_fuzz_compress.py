import sys, os; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from helpers import to_oracle, to_product, tt_rel_diff
T.ensure_init(0)
rng = np.random.default_rng(2024)
bad = 0
for trial in range(120):
    d = int(rng.integers(2, 13))
    n = 2 if trial % 4 else 3
    dims = (n,) * d
    rks = [1] + [int(rng.integers(2, 41)) for _ in range(d - 1)] + [1]
    x = O.rand_tt(dims, rks, rng)
    mb = int(rng.integers(1, 30))
    te = [0.0, 0.0, 1e-8, 1e-4][trial % 4]
    sw = 1 if trial % 7 else 2
    ref = O.tt_compress_(O.copy_tt(x), mb, truncerr=te, sweeps=sw)
    got = T.tt_compress_(to_product(O.copy_tt(x)), mb, truncerr=te, sweeps=sw)
    ok_r = list(got.ttv_rks) == ref.ttv_rks
    err = tt_rel_diff(to_oracle(got), ref) if ok_r else float("nan")
    if not ok_r or not (err < 1e-9):
        bad += 1; print("compress", trial, d, n, rks, mb, te, sw, list(got.ttv_rks), ref.ttv_rks, err)
    # fused apply + compress with a random operator
    A = O.rand_tto(dims, int(rng.integers(1, 4)), rng)
    ref2 = O.tt_compress_(O.apply(A, x), mb, truncerr=te, sweeps=sw)
    got2 = T.apply_compress(to_product(A), to_product(x), mb, truncerr=te, sweeps=sw)
    ok2 = list(got2.ttv_rks) == ref2.ttv_rks
    err2 = tt_rel_diff(to_oracle(got2), ref2) if ok2 else float("nan")
    if not ok2 or not (err2 < 1e-9):
        bad += 1; print("apply_compress", trial, d, n, rks, mb, te, sw, list(got2.ttv_rks), ref2.ttv_rks, err2)
print("bad", bad)
