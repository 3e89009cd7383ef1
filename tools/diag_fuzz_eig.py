"""Diagnostic: targeted parity fuzz of the steps that take the eigensolver routes (n = 2, rank-64 trains: 128-row Gram steps,
64 x 64 route-F cores) against the oracle: operators, seeds, max_bond, truncerr.   python tools/diag_fuzz_eig.py [N] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ttn_amd as T
from oracle import tt_oracle as O
from tests.helpers import to_oracle, to_product, tt_rel_diff
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
T.ensure_init(0)
bad = 0
worst = 0.0
for it in range(N):
    d = int(rng.integers(12, 17))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        A = O.Delta(d)
    elif kind == 1:
        A = O.tto_add(O.Delta(d), O.tto_scale(float(rng.uniform(0.1, 3.0)), O.shift(d)))
    else:
        A = O.rand_tto((2,) * d, int(rng.integers(2, 4)), rng)
    xr = int(rng.choice([64, 64, 64, 48, 40]))
    x = O.rand_tt((2,) * d, xr, rng)
    mb = int(rng.choice([64, 64, 64, 50, 33, 20]))
    te = float(rng.choice([0.0, 0.0, 0.0, 1e-12, 1e-8, 1e-4]))
    ref = O.tt_compress_(O.apply(A, x), mb, truncerr=te)
    cap = [a * c for a, c in zip(A.tto_rks, x.ttv_rks)]
    need, _ = T.device.compress_rank_bound((2,) * d, cap, mb)
    dy = T.DeviceTT((2,) * d, [max(a, b) for a, b in zip(cap, need)])
    T.device.apply_compress(T.DeviceTTO(to_product(A)), T.DeviceTT.from_host(to_product(x)), dy, mb, te)
    T.device.compress_status(dy)
    got = dy.download()
    ok = list(got.ttv_rks) == list(ref.ttv_rks)
    err = tt_rel_diff(to_oracle(got), ref) if ok else float("nan")
    worst = max(worst, err) if ok else worst
    flag = "" if (ok and err <= 1e-9) else "   <-- MISMATCH"
    bad += bool(flag)
    print(f"{it:3d} d={d} op={kind} xr={xr} max_bond={mb} truncerr={te:g}: ranks {'ok' if ok else str(got.ttv_rks) + ' vs ' + str(ref.ttv_rks)} rel diff {err:.1e}{flag}")
print(f"{N} cases, {bad} mismatches, worst rel diff {worst:.1e}")
