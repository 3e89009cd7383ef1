"""Diagnostic: randomized parity sweep of apply_compress against the oracle (dims, ranks, operator ranks, max_bond, truncerr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_oracle, tt_rel_diff
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"          # larger ranks: merged short sides up to 256, wide matrices
bad = 0
for it in range(N):
    d = int(rng.integers(4, 9)) if BIG else int(rng.integers(2, 11))
    dims = tuple(int(v) for v in rng.integers(2, 5, size=d))
    xr = int(rng.integers(8, 44)) if BIG else int(rng.integers(1, 20))
    oprks = [1] + [int(v) for v in rng.integers(1, 4, size=d - 1)] + [1]
    A = T.TToperator(d, [np.asfortranarray(rng.standard_normal((dims[k], dims[k], oprks[k], oprks[k + 1]))) for k in range(d)], dims, oprks, [0] * d)
    x = T.rand_tt(dims, xr, seed=int(rng.integers(1, 10 ** 6)))
    mb = int(rng.integers(8, 120)) if BIG else int(rng.integers(1, 24))
    te = float(rng.choice([0.0, 0.0, 1e-10, 1e-6, 1e-3]))
    try:
        ref = O.tt_compress_(O.apply(to_oracle(A), to_oracle(x)), mb, truncerr=te)
        cap = [a * c for a, c in zip(A.tto_rks, x.ttv_rks)]
        need, _ = T.device.compress_rank_bound(dims, cap, mb)
        dy = T.DeviceTT(dims, [max(a, b) for a, b in zip(cap, need)])
        T.device.apply_compress(T.DeviceTTO(A), T.DeviceTT.from_host(x), dy, mb, te)
        T.device.compress_status(dy)
        got = dy.download()
        ok = got.ttv_rks == ref.ttv_rks
        err = tt_rel_diff(to_oracle(got), ref) if ok else float("nan")
        if not ok or not (err <= 1e-9):
            bad += 1
            print("MISMATCH", it, dims, xr, oprks, mb, te, got.ttv_rks, ref.ttv_rks, err)
    except Exception as e:
        bad += 1
        print("ERROR", it, dims, xr, oprks, mb, te, repr(e)[:200])
print(f"fuzz: {N} cases, {bad} bad")
